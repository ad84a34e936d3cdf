import sys, time, threading, gc
sys.path.insert(0, "/root/repo")
import numpy as np
from bench import YEAST, make_assembly, Groups
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.synth import Synth
syn = make_assembly(YEAST, 30, 1, True)
yaks = [Synth.yak_assembly(syn, k) for k in (21, 31)]
pol = Polisher(yaks)
contigs = [pol.upload(s.pileup) for s in syn]
g = Groups(pol, contigs, YEAST, int(sys.argv[1]) if len(sys.argv) > 1 else 4)
opts = Opts()
g.run(opts, 3)
gc.collect(); gc.disable()
# per-group free-running loops with detailed timers
res = {}
def loop(k, steps=30):
    bp = g.bps[k]; cs = [contigs[i] for i in g.members[k]]
    t_call = t_tot = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        a = time.perf_counter()
        out = bp.polish(cs, opts)
        b = time.perf_counter()
        t_call += bp.last_call_ms
        t_tot += (b - a) * 1e3
    res[k] = ((time.perf_counter() - t0) * 1e3 / steps, t_tot / steps, t_call / steps)
ths = [threading.Thread(target=loop, args=(k,)) for k in range(len(g.bps))]
t0 = time.perf_counter()
[t.start() for t in ths]; [t.join() for t in ths]
print("wall per step %.3f ms" % ((time.perf_counter() - t0) * 1e3 / 30))
for k in range(len(g.bps)):
    print("group %d: iteration %.3f ms, polish() %.3f ms, C call %.3f ms" % ((k,) + res[k]))
