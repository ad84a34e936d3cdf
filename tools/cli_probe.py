"""Whole-assembly command-line timing for different -t (run on a GPU box): python tools/cli_probe.py [scale]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import YEAST, make_assembly
from nextpolish2_amd import cli, io as np2io
from nextpolish2_amd.bamio import write_bam_raw
from nextpolish2_amd.synth import Synth
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
syn = make_assembly([max(20000, int(l * scale)) for l in YEAST], 30, 1, True)
yaks = [Synth.yak_assembly(syn, k) for k in (21, 31)]
td = tempfile.mkdtemp()
bam, fa = td + "/a.bam", td + "/a.fa"
write_bam_raw(bam, [(s.pileup.name, s.pileup.L) for s in syn], [s.bam_records(i) for i, s in enumerate(syn)])
with open(fa, "wb") as f:
    for s in syn:
        f.write(b">%s\n%s\n" % (s.pileup.name.encode(), s.pileup.ref.tobytes()))
yk = []
for y in yaks:
    yk.append(td + f"/k{y.k}.yak")
    np2io.write_yak(yk[-1], y)
for t in (4, 4, 2, 2, 1):
    t0 = time.perf_counter()
    cli.main([bam, fa] + yk + ["-o", td + f"/o{t}_{time.time()}.fa", "-t", str(t), "-L", "20000"])
    print(f"-t {t}: {time.perf_counter() - t0:.3f} s", flush=True)
