"""Times the input side on a synthetic contig: records -> GPU columnariser -> resident pileup (np2_contig_from_records)
and BAM file -> resident pileup (np2_contig_from_bam).  usage: python tools/bench_frontend.py [L]"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextpolish2_amd import Polisher, Opts
from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import pileup_to_records, records_to_arrays, write_bam
from nextpolish2_amd.synth import Synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
s = Synth(L, depth=30, seed=3)
t = time.time(); recs = pileup_to_records(s.pileup, decorate=False); print(f"records {len(recs)} in {time.time()-t:.1f}s (python)")
arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
cols = int(s.pileup.reads["n_cols"][1:].sum())
pol = Polisher([s.yak(21)])
ref = s.pileup.ref.tobytes()
for i in range(3):
    t = time.time(); c = np2io.contig_from_records(pol, ref, arr, cig, seq4); dt = time.time() - t
    tm = pol.timings()
    print(f"from_records wall {dt*1e3:.2f} ms; k_columnarise {tm.get('columnarise', 0):.3f} ms -> {cols/ (tm.get('columnarise',1e9)*1e-3)/1e9:.2f} Gcol/s, {L/dt/1e6:.0f} Mbp/s wall")
    c.free()
d = tempfile.mkdtemp()
write_bam(d + "/a.bam", [("ctg", s.pileup.L)], recs)
print("bam bytes", os.path.getsize(d + "/a.bam"))
bam = np2io.Bam(d + "/a.bam")
for i in range(3):
    t = time.time(); c = np2io.contig_from_bam(pol, bam, "ctg", ref); dt = time.time() - t
    print(f"from_bam wall {dt*1e3:.1f} ms -> {L/dt/1e6:.1f} Mbp/s (BGZF inflate on host threads + parse)")
    b, _ = pol.polish_resident(c, Opts(), want_pos=False)
    c.free()
print("polished == truth:", b.tobytes() == s.hap1)
