"""Where a cold command-line run spends its time (GPU box): python tools/cli_profile.py"""
import os, sys, tempfile, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import YEAST, make_assembly
from nextpolish2_amd import Opts, Polisher, io as np2io
from nextpolish2_amd.bamio import write_bam_raw
from nextpolish2_amd.synth import Synth
syn = make_assembly(YEAST, 30, 1, True)
yaks = [Synth.yak_assembly(syn, k) for k in (21, 31)]
td = tempfile.mkdtemp()
bam = td + "/a.bam"
write_bam_raw(bam, [(s.pileup.name, s.pileup.L) for s in syn], [s.bam_records(i) for i, s in enumerate(syn)])
t0 = time.perf_counter()
pol = Polisher(yaks)
print("context + tables %.1f ms" % ((time.perf_counter() - t0) * 1e3))
b = np2io.Bam(bam)
for rep in range(2):
    tot_f = tot_p = 0.0
    for s in syn:
        t0 = time.perf_counter()
        c = np2io.contig_from_bam(pol, b, s.pileup.name, s.pileup.ref.tobytes())
        t1 = time.perf_counter()
        pol.polish_resident(c, Opts(), want_pos=False)
        t2 = time.perf_counter()
        c.free()
        tot_f += t1 - t0
        tot_p += t2 - t1
        if rep == 0:
            print("%-6s L=%8d front %.1f ms polish %.1f ms" % (s.pileup.name, s.pileup.L, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    print("pass %d: front %.1f ms, polish %.1f ms" % (rep, tot_f * 1e3, tot_p * 1e3))
