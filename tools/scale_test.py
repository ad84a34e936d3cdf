"""Large-contig check: polish a synthetic contig of the given size on the GPU, verify the truth is recovered.
usage: python tools/scale_test.py <L> [diploid]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.synth import Synth

L = int(sys.argv[1]); dip = len(sys.argv) > 2 and sys.argv[2] == "1"
t = time.time(); s = Synth(L, depth=30, seed=5, diploid=dip); print(f"gen {time.time()-t:.1f}s reads {s.pileup.n_reads} cols {s.pileup.n_columns()}", flush=True)
t = time.time(); yaks = [s.yak(21)] + ([s.yak(31)] if dip else []); print(f"yak {time.time()-t:.1f}s words {[len(y.words) for y in yaks]}", flush=True)
t = time.time(); p = Polisher(yaks); print(f"ctx {time.time()-t:.1f}s", flush=True)
t = time.time(); c = p.upload(s.pileup); print(f"upload {time.time()-t:.2f}s", flush=True)
p.set_timing(True)  # per-stage HIP-event timers (adds event packets: the wall time below is a few % pessimistic)
for i in range(3):
    t = time.time(); b, span = p.polish_resident(c, Opts(), want_pos=False); dt = time.time() - t
    print(f"polish {dt*1e3:.1f} ms -> {L/dt/1e6:.0f} Mbp/s span {span} timings {dict((k, round(v,2)) for k,v in p.timings().items())}", flush=True)
print("equals hap1:", b.tobytes() == s.hap1, "len", len(b), len(s.hap1))
