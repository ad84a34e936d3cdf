import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.dist import polish_sharded_local
from nextpolish2_amd.synth import Synth
s = Synth(3000000, seed=901, diploid=True)
yaks = [s.yak(21), s.yak(31)]
pol = Polisher(yaks)
b0, p0 = pol.polish(s.pileup, Opts())
for ns in (1, 2):
    for rep in range(2):
        t = time.time()
        b1, p1 = polish_sharded_local(pol, s.pileup, Opts(), n_shards=ns, halo=65536)
        print(ns, "sharded == whole:", np.array_equal(b0, b1) and np.array_equal(p0, p1), f"{1e3*(time.time()-t):.1f} ms", flush=True)
