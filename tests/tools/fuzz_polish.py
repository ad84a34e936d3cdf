"""Randomised parity run, HIP path vs oracle (run on a GPU box): python tests/tools/fuzz_polish.py <seed> <cases> [heavy|wide]

`heavy` draws larger contigs, deeper pileups and more divergent haplotypes (slower: the oracle dominates); `wide` draws
from a wider space of small cases: very short contigs, depth 1-2, up to four k-mer tables (k 17/21/27/31), iter_count up
to 4, min_kmer_count 1-20, max_indel_len 0-50, haplotypes from identical to 2 % apart."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.api import Np2Error
from nextpolish2_amd.synth import Synth
from oracle.np2_oracle import Oracle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_case = int(sys.argv[2]) if len(sys.argv) > 2 else 40
heavy = len(sys.argv) > 3 and sys.argv[3] == "heavy"
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
bad = 0
t0 = time.time()
for case in range(n_case):
    L = int(rng.choice([200000, 400000, 1000000] if heavy else [1500, 3000, 8000, 20000, 50000, 120000]))
    depth = int(rng.choice([2, 30, 120, 250] if heavy else [3, 8, 15, 30, 60, 100]))
    snp = float(rng.choice([0.005, 0.02, 0.05])) if heavy else 0.005
    hind = float(rng.choice([0.002, 0.01])) if heavy else 0.002
    dip = bool(rng.integers(0, 2))
    rerr = float(rng.choice([0.0005, 0.002, 0.01, 0.03]))
    aerr = float(rng.choice([1e-4, 1e-3, 5e-3]))
    rl = float(rng.choice([1200, 3000, 8000]))
    seed = int(rng.integers(1, 1 << 30))
    ks = [21] if rng.integers(0, 2) else [21, 31]
    o = Opts(min_kmer_count=int(rng.choice([2, 5, 8])), iter_count=int(rng.choice([1, 2, 3])), model=str(rng.choice(["ref", "len"])),
             use_all_reads=bool(rng.integers(0, 2)), max_indel_len=int(rng.choice([5, 20])))
    if wide:
        L = int(rng.choice([300, 700, 1500, 4000, 12000, 40000]))
        depth = int(rng.choice([1, 2, 3, 5, 12, 30, 80]))
        snp = float(rng.choice([0.0, 0.001, 0.005, 0.02]))
        hind = float(rng.choice([0.0, 0.002, 0.01]))
        rl = float(rng.choice([400, 1200, 3000, 8000]))
        ks = sorted(int(k) for k in rng.choice([17, 21, 27, 31], size=int(rng.integers(1, 5)), replace=False))
        o = Opts(min_kmer_count=int(rng.choice([1, 2, 5, 8, 20])), iter_count=int(rng.choice([1, 2, 3, 4])), model=str(rng.choice(["ref", "len"])),
                 use_all_reads=bool(rng.integers(0, 2)), max_indel_len=int(rng.choice([0, 1, 5, 20, 50])))
    try:
        s = Synth(L, depth=depth, seed=seed, diploid=dip, snp_rate=snp, hap_indel_rate=hind, read_err_rate=rerr, asm_err_rate=aerr, read_len_mean=min(rl, L / 2), read_len_sd=min(rl / 6, L / 12), read_len_min=min(1000 if not wide else 200, L // 4))
    except Exception as e:
        print("synth failed", e); continue
    yaks = [s.yak(k) for k in ks]
    desc = dict(L=L, depth=depth, dip=dip, snp=snp, hind=hind, rerr=rerr, aerr=aerr, rl=rl, seed=seed, ks=ks, o=vars(o), reads=s.pileup.n_reads)
    try:
        ob, op = Oracle(yaks).polish(s.pileup, o); oerr = None
    except Exception as e:
        ob = op = None; oerr = str(e)[:80]
    try:
        gb, gp = Polisher(yaks).polish(s.pileup, o); gerr = None
    except Np2Error as e:
        gb = gp = None; gerr = str(e)[:80]
    if (oerr is None) != (gerr is None):
        bad += 1; print("ERROR-MISMATCH", desc, "oracle:", oerr, "hip:", gerr)
    elif oerr is None and not (np.array_equal(ob, gb) and np.array_equal(op, gp)):
        bad += 1; print("MISMATCH", desc, len(ob), len(gb))
    else:
        print("ok", case, L, depth, dip, rerr, ks, "err" if oerr else len(ob), flush=True)
print("cases", n_case, "bad", bad, "time %.1f" % (time.time() - t0))
