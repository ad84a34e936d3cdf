"""Generates tests/golden/shards/shard_votes.npz ON A GPU BOX (gpurun): what the two shards of a small diploid contig hand to
the exchange steps of nextpolish2_amd.dist.polish_sharded — their votes of the phasing pass, their final pieces —
plus the oracle's result for the whole contig.  tests/test_shard_cpu.py replays the exchange (gloo, world 2) from it
without a GPU.  usage: python tests/golden/make_shard_fixture.py <out.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nextpolish2_amd import Opts, Polisher  # noqa: E402
from nextpolish2_amd.api import ShardRun, shard_plan, vote_decide  # noqa: E402
from nextpolish2_amd.synth import Synth  # noqa: E402
from oracle.np2_oracle import Oracle  # noqa: E402

s = Synth(120000, seed=881, diploid=True, read_len_mean=7000.0, read_len_sd=1000.0)
yaks = [s.yak(21)]
pol = Polisher(yaks)
plans = shard_plan(s.pileup, 2, 20000)
runs = [ShardRun(pol.clone(), s.pileup, pl, Opts(), 1024) for pl in plans]
votes = [r.vote() for r in runs]
losers = vote_decide(votes, s.pileup.n_reads, Opts())
for r in runs:
    r.apply(losers)
pieces = [r.final() for r in runs]
o = Oracle(yaks)
o.set_trace(True)
ob, op = o.polish(s.pileup, Opts())
assert np.array_equal(o.trace(0, "invalid_ids"), losers)
out = {"n_reads": np.array([s.pileup.n_reads]), "losers": losers, "oracle_bases": ob, "oracle_pos": op,
       "plans": np.array([[getattr(pl, f) for f, _ in pl._fields_] for pl in plans], dtype=np.uint32)}
for k in range(2):
    out[f"vote{k}"] = np.frombuffer(votes[k].to_bytes(), dtype=np.uint8)
    out[f"piece{k}_bases"] = np.asarray(pieces[k][0])
    out[f"piece{k}_pos"] = np.asarray(pieces[k][1])
np.savez_compressed(sys.argv[1], **out)
print("wrote", sys.argv[1], {k: (v.shape, v.dtype) for k, v in out.items()})
