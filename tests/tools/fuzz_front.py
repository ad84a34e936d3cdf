"""Randomised parity run of the input side (run on a GPU box): python tests/tools/fuzz_front.py <seed> <cases>"""
# BAM-record front end (decorated records: clips, strands, supplementary, mapq) + resident polish
import sys, os, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd.api import Np2Error
from nextpolish2_amd.bamio import pileup_to_records, records_to_arrays
from nextpolish2_amd.synth import Synth
from oracle import np2_oracle as orc
from test_frontend_cpu import same_pileup
rng = np.random.default_rng(int(sys.argv[1])); n = int(sys.argv[2]); bad = 0
for case in range(n):
    L = int(rng.choice([3000, 10000, 40000])); depth = int(rng.choice([5, 20, 50])); dip = bool(rng.integers(0, 2))
    seed = int(rng.integers(1, 1 << 30)); rl = float(rng.choice([1500, 4000]))
    s = Synth(L, depth=depth, seed=seed, diploid=dip, read_err_rate=float(rng.choice([0.002, 0.02])), read_len_mean=min(rl, L / 2), read_len_sd=rl / 6, read_len_min=min(1000, L // 4))
    recs = pileup_to_records(s.pileup, rng=np.random.default_rng(seed), decorate=True)
    fo = np2io.FrontOpts(use_supplementary=bool(rng.integers(0, 2)), min_map_qual=int(rng.choice([0, 1, 30])), min_read_len=int(rng.choice([500, 1000, 2000])),
                         min_map_len=int(rng.choice([200, 500, 1500])), min_map_fra=float(rng.choice([0.2, 0.5, 0.9])), max_clip_len=int(rng.choice([0, 10, 100, 100000])))
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    ref = s.pileup.ref.tobytes()
    yaks = [s.yak(21)]
    try:
        exp = orc.front_end(ref, arr, cig, asc, asc_off, fo); oerr = None
    except Exception as e:
        oerr = str(e)[:60]
    pol = Polisher(yaks)
    try:
        c = np2io.contig_from_records(pol, ref, arr, cig, seq4, fo); gerr = None
    except Np2Error as e:
        gerr = str(e)[:60]
    if (oerr is None) != (gerr is None):
        bad += 1; print("ERR-MISMATCH", L, depth, dip, seed, vars(fo), oerr, gerr); continue
    if oerr: print("ok(err)", case); continue
    got = np2io.export_contig(pol, c, s.pileup.ref)
    if not same_pileup(got, exp):
        bad += 1; print("PILEUP MISMATCH", L, depth, dip, seed, vars(fo)); continue
    try:
        ob, op = orc.Oracle(yaks).polish(exp, Opts()); oe = None
    except Exception as e:
        oe = e
    try:
        gb, gp = pol.polish_resident(c, Opts()); ge = None
    except Np2Error as e:
        ge = e
    if (oe is None) != (ge is None) or (oe is None and not (np.array_equal(ob, gb) and np.array_equal(op, gp))):
        bad += 1; print("POLISH MISMATCH", L, depth, dip, seed, vars(fo), oe, ge)
print("front cases", n, "bad", bad)
