import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.dist import polish_sharded_local
from nextpolish2_amd.synth import Synth
from oracle.np2_oracle import Oracle
for L, dip, ns in [(400000, True, 2), (400000, True, 3), (600000, False, 4), (300000, True, 2)]:
    s = Synth(L, seed=900 + ns, diploid=dip)
    yaks = [s.yak(21), s.yak(31)] if dip else [s.yak(21)]
    pol = Polisher(yaks)
    b0, p0 = pol.polish(s.pileup, Opts())
    for halo in (65536, 16384):
        b1, p1 = polish_sharded_local(pol, s.pileup, Opts(), n_shards=ns, halo=halo)
        print(L, dip, ns, halo, "sharded == whole:", np.array_equal(b0, b1) and np.array_equal(p0, p1), len(b0), len(b1), flush=True)
    ob, op = Oracle(yaks).polish(s.pileup, Opts())
    print("   whole == oracle:", np.array_equal(ob, b0) and np.array_equal(op, p0))
