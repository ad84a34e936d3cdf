"""End-to-end probe: BAM -> resident pileup -> polish for one synthetic contig; NP2_IO_PROFILE=1 prints the front-end split."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextpolish2_amd import Polisher, Opts
from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import pileup_to_records, write_bam
from nextpolish2_amd.synth import Synth
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 750000
s = Synth(L, depth=30, seed=3, diploid=True)
t = time.time(); recs = pileup_to_records(s.pileup, decorate=False); print(f"records {len(recs)} in {time.time()-t:.1f}s (python)", flush=True)
d = tempfile.mkdtemp()
t = time.time(); write_bam(d + "/a.bam", [("ctg", s.pileup.L)], recs); print(f"bam {os.path.getsize(d + '/a.bam')} bytes in {time.time()-t:.1f}s", flush=True)
pol = Polisher([s.yak(21), s.yak(31)])
bam = np2io.Bam(d + "/a.bam")
ref = s.pileup.ref.tobytes()
for i in range(4):
    t = time.time(); c = np2io.contig_from_bam(pol, bam, "ctg", ref); dt = time.time() - t
    t2 = time.time(); b, _ = pol.polish_resident(c, Opts(), want_pos=False); dp = time.time() - t2
    print(f"from_bam {dt*1e3:.2f} ms ({L/dt/1e6:.1f} Mbp/s)  polish {dp*1e3:.2f} ms  -> end to end {L/(dt+dp)/1e6:.1f} Mbp/s", flush=True)
    c.free()
print("polished == truth:", b.tobytes() == s.hap1)
