"""BASELINE configs[4] as one of its 8 GPUs sees it, at full size: rank 0's share of a CHM13-sized assembly (~380 Mb in
whole chromosomes), 30x, -r, k21 + k31 tables padded to 10^9 words each.  python tools/t2t_share_probe.py [scale]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_t2t_share import run_share
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
run_share(scale, int(1e9), log=lambda s: print(s, flush=True))
print("ok: truth recovered on every contig, rerun identical", flush=True)
