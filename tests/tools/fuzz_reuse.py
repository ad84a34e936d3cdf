"""Randomised parity run with ONE context polishing many different contigs back to back (buffer reuse, epochs, deferred
output in flight across contigs) — run on a GPU box: python tests/tools/fuzz_reuse.py <seed> <batches>"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.synth import Synth
from oracle.np2_oracle import Oracle
from test_oracle import yak_from_seqs

rng = np.random.default_rng(int(sys.argv[1])); n_batch = int(sys.argv[2]); bad = 0; n = 0
for b in range(n_batch):
    syn = []
    for _ in range(int(rng.integers(3, 7))):
        L = int(rng.choice([1500, 4000, 12000, 30000, 60000]))
        syn.append(Synth(L, depth=int(rng.choice([5, 20, 40])), seed=int(rng.integers(1, 1 << 30)), diploid=bool(rng.integers(0, 2)),
                         read_err_rate=float(rng.choice([0.002, 0.01])), read_len_mean=min(4000.0, L / 2), read_len_sd=min(600.0, L / 12),
                         read_len_min=min(1000, L // 4)))
    haps = []
    for s in syn:
        haps += [s.hap1.decode()] + ([s.hap2.decode()] if s.diploid else [])
    ks = [21] if rng.integers(0, 2) else [21, 31]
    yaks = [yak_from_seqs(haps, k, count=int(rng.choice([9, 50]))) for k in ks]
    orc, pol = Oracle(yaks), Polisher(yaks)
    resident = [pol.upload(s.pileup) for s in syn]
    pending = None  # (index, expected) of the deferred fetch in flight
    for step in range(3 * len(syn)):
        i = int(rng.integers(0, len(syn)))
        o = Opts(iter_count=int(rng.choice([1, 2, 3])), min_kmer_count=int(rng.choice([2, 5])), model=str(rng.choice(["ref", "len"])),
                 use_all_reads=bool(rng.integers(0, 2)))
        eb, ep = orc.polish(syn[i].pileup, o)
        mode = int(rng.integers(0, 3))
        n += 1
        if mode == 0:
            gb, gp = pol.polish(syn[i].pileup, o)
            ok = np.array_equal(eb, gb) and np.array_equal(ep, gp)
        elif mode == 1:
            gb, gp = pol.polish_resident(resident[i], o)
            ok = np.array_equal(eb, gb) and np.array_equal(ep, gp)
        else:
            _, span = pol.polish_resident(resident[i], o, want_pos=False, defer_output=True)
            ok = span == (int(ep[0]), int(ep[-1]))
            if pending is not None:
                ok = ok and np.array_equal(pol.fetch_end(), pending)
            pol.fetch_begin()
            pending = eb
        if not ok:
            bad += 1
            print("MISMATCH batch", b, "step", step, "contig", i, "mode", mode, vars(o))
    if pending is not None and not np.array_equal(pol.fetch_end(), pending):
        bad += 1
        print("MISMATCH batch", b, "final deferred fetch")
print("reuse polishes", n, "bad", bad)
