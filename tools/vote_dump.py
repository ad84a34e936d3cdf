"""Dump the phasing vote of a diploid contig (one shard = the whole contig) for host-side experiments without a GPU:
   python tools/vote_dump.py [L] -> gpurun_out/vote_<L>.npz (packed Vote + n_reads) and the losers' checksum."""
import os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.api import ShardRun, shard_plan, vote_decide
from nextpolish2_amd.synth import Synth, concat_pileups
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 16_000_000
NP = 4
parts = [Synth(L // NP, depth=30, seed=700 + i, diploid=True) for i in range(NP)]
pu = concat_pileups([p.pileup for p in parts], "ctg")
yaks = [Synth.yak_assembly(parts, k) for k in (21, 31)]
pol = Polisher(yaks)
plans = shard_plan(pu, 1, 65536)
run = ShardRun(pol, pu, plans[0], Opts(), 1024)
v = run.vote()
t = time.time(); losers = vote_decide([v], pu.n_reads); dt = time.time() - t
print(f"L {pu.L} reads {pu.n_reads} pairs {len(v.pair_key)} vote readers {len(v.read_id)}; np2_vote_decide {dt*1e3:.1f} ms, losers {len(losers)} crc {zlib.crc32(losers.tobytes()):08x}")
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(f"gpurun_out/vote_{L}.npz", packed=v.pack(), n_reads=np.array([pu.n_reads]), losers=losers)
print("saved", os.path.getsize(f"gpurun_out/vote_{L}.npz") / 1e6, "MB")
