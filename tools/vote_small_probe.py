"""Host side of the phasing vote on a yeast-chromosome-sized diploid contig (run on a GPU box): NP2_PHASE_PROFILE marks
+ the stage clocks.  python tools/vote_small_probe.py [L]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NP2_PHASE_PROFILE"] = "1"
from nextpolish2_amd.synth import Synth
from nextpolish2_amd import Polisher
from nextpolish2_amd._types import Opts
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1531933
s = Synth(L, depth=30, seed=4, diploid=True)
pol = Polisher([s.yak(21), s.yak(31)])
c = pol.upload(s.pileup)
pol.set_timing(True)
for rep in range(4):
    print(f"--- polish {rep}", file=sys.stderr, flush=True)
    pol.polish_resident(c, Opts())
    tm = {k: round(v, 2) for k, v in pol.timings().items() if k.startswith("wall_")}
    print(tm, file=sys.stderr, flush=True)
