"""Input side at E. coli size (10.6 k records, 138 M columns): BAM -> np2_contig_from_bam, with the GPU columnariser's
own time.  BAM written from the generator's records (no per-column Python).  usage: python tools/bench_frontend_big.py [L]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextpolish2_amd import Polisher, Opts
from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import write_bam_raw
from nextpolish2_amd.synth import Synth

L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4600000
s = Synth(L, depth=30, seed=3, name="ctg")
cols = int(s.pileup.reads["n_cols"][1:].sum())
d = tempfile.mkdtemp()
write_bam_raw(d + "/a.bam", [("ctg", s.pileup.L)], [s.bam_records(0)])
print("records", s.pileup.n_reads - 1, "columns", cols, "bam bytes", os.path.getsize(d + "/a.bam"))
pol = Polisher([s.yak(21)])
pol.set_timing(True)
ref = s.pileup.ref.tobytes()
bam = np2io.Bam(d + "/a.bam")
for i in range(5):
    t = time.time(); c = np2io.contig_from_bam(pol, bam, "ctg", ref); dt = time.time() - t
    tm = pol.timings()
    if i == 4:
        print({k: round(v, 3) for k, v in tm.items()})
    col_ms = tm.get("columnarise", 0.0)
    print(f"from_bam wall {dt*1e3:.1f} ms -> {L/dt/1e6:.1f} Mbp/s; k_columnarise {col_ms:.3f} ms"
          + (f" = {cols/(col_ms*1e-3)/1e9:.1f} Gcolumns/s" if col_ms else ""))
    b, _ = pol.polish_resident(c, Opts(), want_pos=False)
    c.free()
print("polished == truth:", b.tobytes() == s.hap1)
