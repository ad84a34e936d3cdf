"""Whole-assembly command-line timing for different front-end / polish thread splits and both read-extraction paths
(run on a GPU box): python tools/cli_split_probe.py"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
outs = {}
for mode in ("libdeflate", "gpu"):
    for front, workers in ((1, 1), (2, 2), (3, 2), (4, 2), (4, 3), (6, 2)):
        os.environ["NP2_INFLATE"], os.environ["NP2_CLI_FRONT"], os.environ["NP2_CLI_WORKERS"] = mode, str(front), str(workers)
        best = 1e9
        for rep in range(4):
            o = td + f"/o_{mode}_{front}_{workers}_{rep}.fa"
            t0 = time.perf_counter()
            stderr, sys.stderr = sys.stderr, open(os.devnull, "w")
            try:
                cli.main([bam, fa] + yk + ["-o", o, "-t", "2", "-L", "20000"])
            finally:
                sys.stderr = stderr
            best = min(best, time.perf_counter() - t0)
        outs[(mode, front, workers)] = open(o, "rb").read()
        print(f"NP2_INFLATE={mode} front ends {front} polish contexts {workers}: best of 4 {best * 1e3:.1f} ms = {sum(YEAST) / best / 1e6:.0f} Mbp/s", flush=True)
print("all outputs identical:", len(set(outs.values())) == 1)
