"""The command line on the yeast-sized assembly with read extraction on the device and 4 front-end threads, profile
switches on: where a contig's front end goes when several run side by side (run on a GPU box)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NP2_INFLATE", "gpu")
os.environ.setdefault("NP2_CLI_FRONT", "4")
os.environ.setdefault("NP2_CLI_WORKERS", "2")
from bench import YEAST, make_assembly
from nextpolish2_amd import cli, io as np2io
from nextpolish2_amd.bamio import write_bam_raw
from nextpolish2_amd.synth import Synth
syn = make_assembly(list(YEAST), 30, 1, True)
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
for rep in range(4):
    if rep == 3:
        os.environ["NP2_CLI_PROFILE"] = os.environ["NP2_IO_PROFILE"] = os.environ["NP2_ALLOC_PROFILE"] = "1"
    t0 = time.perf_counter()
    cli.main([bam, fa] + yk + ["-o", td + f"/o{rep}.fa", "-t", "2", "-L", "20000"])
    print(f"run {rep}: {time.perf_counter() - t0:.3f} s", flush=True)
