#!/usr/bin/env python3
"""Throughput of the BGZF inflate kernel and of a contig's front end through the device against the host pool:
    python tools/inflate_probe.py [L] [depth]          (run through gpurun; taskset -c 0-1 ... for a rank's share of a node)
Writes a synthetic BAM of one contig (no qualities: QUAL = 0xFF), inflates the whole file on the device
(np2_bgzf_inflate_device: kernel time by HIP events), then builds the resident pileup from the BAM with NP2_INFLATE=gpu and
with the host pool, three times each in fresh processes (the switch is read once per process)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nextpolish2_amd import Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import write_bam_raw
from nextpolish2_amd.synth import Synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4641652
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 30
td = tempfile.mkdtemp()
s = Synth(L, depth=depth, seed=5)
t = time.time()
write_bam_raw(os.path.join(td, "m.bam"), [("ctgA", L)], [s.bam_records(0)], level=6)
print(f"BAM written in {time.time() - t:.1f} s: {os.path.getsize(os.path.join(td, 'm.bam')) / 1e6:.1f} MB", flush=True)
open(os.path.join(td, "ref.txt"), "wb").write(s.pileup.ref.tobytes())
np2io.write_yak(os.path.join(td, "k21.yak"), s.yak(21))
pol = Polisher([s.yak(21)])
data = np.fromfile(os.path.join(td, "m.bam"), dtype=np.uint8)
for i in range(3):
    t = time.time()
    out, ms = np2io.bgzf_inflate_device(pol, data)
    print(f"inflate kernel: {len(data) / 1e6:.1f} MB -> {len(out) / 1e6:.1f} MB in {ms:.2f} ms = {len(out) / ms / 1e6:.1f} GB/s inflated "
          f"({len(data) / ms / 1e6:.1f} GB/s of file); call {1e3 * (time.time() - t):.1f} ms", flush=True)
import zlib
code = ("import sys, time, numpy as np; sys.path.insert(0, %r)\n"
        "from nextpolish2_amd import io as np2io\n"
        "pol = np2io.polisher_from_yak_files([sys.argv[3]])\n"
        "ref = open(sys.argv[2], 'rb').read()\n"
        "bam = np2io.Bam(sys.argv[1])\n"
        "for i in range(4):\n"
        "    t = time.time()\n"
        "    c = np2io.contig_from_bam(pol, bam, 'ctgA', ref, np2io.FrontOpts())\n"
        "    print('contig_from_bam %%d: %%.2f ms' %% (i, 1e3 * (time.time() - t)), flush=True)\n"
        "    c.free()\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for mode in ("gpu", "libdeflate"):
    for pre in ([], ["taskset", "-c", "0-1"]):
        r = subprocess.run(pre + [sys.executable, "-c", code, os.path.join(td, "m.bam"), os.path.join(td, "ref.txt"), os.path.join(td, "k21.yak")],
                           capture_output=True, text=True, env=dict(os.environ, NP2_INFLATE=mode, NP2_IO_PROFILE="1" if mode == "gpu" else ""))
        print(f"--- NP2_INFLATE={mode} {' '.join(pre)}")
        print(r.stdout.strip())
        if r.returncode:
            print(r.stderr[-1500:])
        else:
            print("\n".join(l for l in r.stderr.splitlines() if "fetch_records_gpu" in l)[-600:])
