"""The inflate kernel alone on synthetic BAMs of several sizes: python tools/inflate_only.py [L ...] (run through gpurun)."""
import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nextpolish2_amd import Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import write_bam_raw
from nextpolish2_amd.synth import Synth
Ls = [int(x) for x in sys.argv[1:]] or [4641652]
td = tempfile.mkdtemp()
pol = None
for L in Ls:
    s = Synth(L, depth=30, seed=5)
    write_bam_raw(os.path.join(td, "m.bam"), [("ctgA", L)], [s.bam_records(0)], level=6)
    if pol is None:
        pol = Polisher([s.yak(21)])
    data = np.fromfile(os.path.join(td, "m.bam"), dtype=np.uint8)
    for i in range(3):
        out, ms = np2io.bgzf_inflate_device(pol, data)
    nblk = int(np.count_nonzero((data[:-3] == 31) & (data[1:-2] == 139) & (data[2:-1] == 8) & (data[3:] == 4)))
    print(f"L {L}: ~{nblk} blocks, {len(data) / 1e6:.1f} MB -> {len(out) / 1e6:.1f} MB, kernel {ms:.2f} ms = {len(out) / ms / 1e6:.1f} GB/s inflated", flush=True)
