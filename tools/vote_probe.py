import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["NP2_PHASE_PROFILE"] = "1"
from bench import make_assembly
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.synth import Synth
syn = make_assembly([1531933], 30, 4, True)
yaks = [Synth.yak_assembly(syn, k) for k in (21, 31)]
pol = Polisher(yaks)
c = pol.upload(syn[0].pileup)
for _ in range(4):
    t0 = time.perf_counter()
    pol.polish_resident(c, Opts(), want_pos=False)
    print("polish %.2f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
