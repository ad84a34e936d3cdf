"""Expected decision of the recorded phasing vote (tests/golden/votes/vote_16Mb_diploid.npz), derived from the ORACLE.

The `packed` vote in that file was recorded on a GPU box by tools/vote_dump.py (the product's vote kernels on a 16 Mb
diploid contig: 4 generated pieces, seeds 700..703, k21 + k31 tables).  Its `losers` used to be what the product's own
single-threaded host code decided at the time — product against product.  This script regenerates the same contig on
the CPU (the generator is seeded and host-only), runs the oracle's whole phasing pass on it (mark_hete_lqseqs,
phase_reads_by_lqseqs, Louvain: main.rs:916-1015, louvain.rs:59-356) and stores ITS removed reads (trace invalid_ids of
pass 0) as `losers`.  No GPU needed:  python tests/golden/make_vote_fixture.py"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from nextpolish2_amd import Opts  # noqa: E402
from nextpolish2_amd.synth import Synth, concat_pileups  # noqa: E402
from oracle.np2_oracle import Oracle  # noqa: E402

path = os.path.join(HERE, "votes", "vote_16Mb_diploid.npz")
z = np.load(path)
L, NP = 16_000_000, 4
parts = [Synth(L // NP, depth=30, seed=700 + i, diploid=True) for i in range(NP)]
pu = concat_pileups([p.pileup for p in parts], "ctg")
assert pu.n_reads == int(z["n_reads"][0]), "the generator no longer reproduces the recorded contig"
yaks = [Synth.yak_assembly(parts, k) for k in (21, 31)]
o = Oracle(yaks)
o.set_trace(True)
t = time.time()
o.polish(pu, Opts(iter_count=2))
losers = np.unique(o.trace(0, "invalid_ids")).astype(np.uint32)
print(f"oracle: {time.time() - t:.1f} s, {len(losers)} reads removed by the phasing pass; "
      f"equal to the decision stored before: {np.array_equal(losers, z['losers'])}")
np.savez_compressed(path, packed=z["packed"], n_reads=z["n_reads"], losers=losers,
                    losers_source=np.array(["oracle (tests/golden/make_vote_fixture.py): trace invalid_ids of pass 0"]))
