"""Where does a batch step's wall time go outside the flushes?  Times the raw C call, the Python wrapper and the loop."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import YEAST, make_assembly
from nextpolish2_amd import BatchPolisher, Opts, Polisher
from nextpolish2_amd.api import lib
from nextpolish2_amd.synth import Synth
syn = make_assembly(YEAST, 30, 1, True)
yaks = [Synth.yak_assembly(syn, k) for k in (21, 31)]
pol = Polisher(yaks)
contigs = [pol.upload(s.pileup) for s in syn]
bp = BatchPolisher(pol, len(contigs))
for _ in range(3):
    bp.polish(contigs, Opts())
n = len(contigs)
o = Opts().c()
hs = (C.c_void_p * n)(*[c._h for c in contigs])
for mode in ("raw C call, results freed", "python wrapper"):
    ts = []
    for rep in range(12):
        t0 = time.perf_counter()
        if mode.startswith("raw"):
            ob = (C.c_void_p * n)(); on = (C.c_uint64 * n)(); rcs = (C.c_int * n)(); span = (C.c_uint32 * (2 * n))()
            lib().np2_batch_polish(bp._h, hs, n, C.byref(o), ob, None, on, span, rcs)
            t1 = time.perf_counter()
            for i in range(n):
                lib().np2_free(C.c_void_p(ob[i]))
        else:
            out = bp.polish(contigs, Opts())
            t1 = time.perf_counter()
        fl = bp.flush_log()
        ts.append((t1 - t0) * 1e3)
        if rep == 11:
            print(mode, "last call %.2f ms; flush totals host %.2f issue %.2f wait %.2f (sum %.2f)" % (
                ts[-1], sum(f[0] for f in fl), sum(f[1] for f in fl), sum(f[2] for f in fl), sum(sum(f) for f in fl)))
    print(mode, "median %.2f ms min %.2f max %.2f" % (np.median(ts), min(ts), max(ts)), flush=True)
# per-flush breakdown of the last call: (host ms before the flush, issue ms, wait ms)
fl = bp.flush_log()
for i, f in enumerate(fl):
    print("flush %2d  host %.3f  issue %.3f  wait %.3f" % (i, f[0], f[1], f[2]))
