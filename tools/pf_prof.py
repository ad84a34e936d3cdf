#!/usr/bin/env python3
"""Phase timers of the fused pass-front kernel (k_pf_tile) on one diploid contig: python tools/pf_prof.py [L] [depth]
(NP2_PF_PROF makes the host print mean / max shader clocks per phase and tile after every pass; plain context)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NP2_PF_PROF"] = "1"
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.synth import Synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1500000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 30
s = Synth(L, depth=depth, seed=77, diploid=True)
yaks = [s.yak(21), s.yak(31)]
g = Polisher(yaks)
c = g.upload(s.pileup)
for i in range(2):
    t0 = time.time()
    g.polish_resident(c, Opts())
    print(f"polish {i}: {1e3 * (time.time() - t0):.2f} ms", file=sys.stderr)
