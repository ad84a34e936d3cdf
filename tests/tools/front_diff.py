import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from nextpolish2_amd import Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import pileup_to_records, records_to_arrays
from nextpolish2_amd.synth import Synth
from oracle import np2_oracle as orc
from test_frontend_cpu import nib_streams

s = Synth(60000, depth=25, seed=51, read_len_mean=7000.0, read_len_sd=1200.0)
recs = pileup_to_records(s.pileup, rng=np.random.default_rng(51), decorate=False)
arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
exp = orc.front_end(s.pileup.ref.tobytes(), arr, cig, asc, asc_off, np2io.FrontOpts())
pol = Polisher([s.yak(21)])
c = np2io.contig_from_records(pol, s.pileup.ref.tobytes(), arr, cig, seq4, np2io.FrontOpts())
got = np2io.export_contig(pol, c, s.pileup.ref)
print("n_reads", got.n_reads, exp.n_reads)
for f in ("aln_t_s", "aln_t_e", "n_cols", "flags"):
    a, b = got.reads[f], exp.reads[f]
    n = min(len(a), len(b))
    d = np.nonzero(a[:n] != b[:n])[0]
    print(f, "ndiff", len(d), "first", (int(d[0]), int(a[d[0]]), int(b[d[0]])) if len(d) else None)
ga, ea = nib_streams(got), nib_streams(exp)
bad = 0
for i, (x, y) in enumerate(zip(ga, ea)):
    if x != y:
        bad += 1
        if bad <= 5:
            xa, ya = np.frombuffer(x, np.uint8), np.frombuffer(y, np.uint8)
            m = min(len(xa), len(ya))
            dd = np.nonzero(xa[:m] != ya[:m])[0]
            print("read", i, "len", len(xa), len(ya), "ndiff", len(dd), "first", dd[:5], [hex(v) for v in xa[dd[:5]]], [hex(v) for v in ya[dd[:5]]], "tail", xa[-3:], ya[-3:])
print("bad streams", bad)
