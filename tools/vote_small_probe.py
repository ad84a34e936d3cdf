import os, sys, time
sys.path.insert(0, os.getcwd())
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.synth import Synth
s = Synth(1531933, depth=30, seed=104, diploid=True, name="chr4")
yaks = [s.yak(21), s.yak(31)]
pol = Polisher(yaks)
c = pol.upload(s.pileup)
pol.set_timing(True)
for i in range(3):
    t = time.time(); b, span = pol.polish_resident(c, Opts(), want_pos=False); dt = time.time() - t
    tm = pol.timings()
    print(f"polish {dt*1e3:.2f} ms; wall_louvain {tm.get('wall_louvain', 0.0):.2f} ms", flush=True)
