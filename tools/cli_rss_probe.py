"""Peak RSS and wall time of the command line as a FRESH process on the yeast-sized assembly's files (run on a GPU box):
the files are written by this script, nextpolish2_amd.cli runs in a subprocess of its own (python -m), with read
extraction on the host pool and on the device.  python tools/cli_rss_probe.py"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import YEAST, make_assembly
from nextpolish2_amd import io as np2io
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
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mode in ("libdeflate", "gpu", "libdeflate", "gpu"):
    out = td + f"/o_{mode}.fa"
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", bam, fa] + yk + ["-o", out, "-t", "2", "-L", "20000"], capture_output=True, text=True,
                       env=dict(os.environ, PYTHONPATH=root, NP2_INFLATE=mode))
    info = [l for l in r.stderr.splitlines() if "Real time" in l]
    print(f"NP2_INFLATE={mode}: rc {r.returncode}; {info[-1] if info else r.stderr[-300:]}", flush=True)
