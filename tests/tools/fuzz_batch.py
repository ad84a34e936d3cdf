"""Randomised parity run of the batch driver (run on a GPU box): python tests/tools/fuzz_batch.py <seed> <cases>

Every case polishes a random mix of 2-12 contigs (lengths 300 .. 300 k, depths 1 .. 80, haploid and diploid ones side by
side, some without a single low-quality region) through one np2_batch_t, two waves in a row (the second reuses the slot
contexts' buffers), and compares every contig with the oracle: the same sequence and positions, or an error on both
sides.  The contigs of a case share the k-mer tables (built over all of them) and the options, like an assembly does."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.api import BatchPolisher, Np2Error
from nextpolish2_amd.synth import Synth
from oracle.np2_oracle import Oracle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_case = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
t0 = time.time()
for case in range(n_case):
    n = int(rng.integers(2, 13))
    ks = sorted(int(k) for k in rng.choice([17, 21, 27, 31], size=int(rng.integers(1, 4)), replace=False))
    o = Opts(min_kmer_count=int(rng.choice([2, 5, 8])), iter_count=int(rng.choice([1, 2, 2, 3])), model=str(rng.choice(["ref", "len"])),
             use_all_reads=bool(rng.integers(0, 4) == 0), max_indel_len=int(rng.choice([5, 20])))
    syn, desc = [], []
    for i in range(n):
        L = int(rng.choice([300, 1500, 6000, 20000, 60000, 150000, 300000]))
        depth = int(rng.choice([1, 3, 8, 20, 30, 80]))
        dip = bool(rng.integers(0, 2))
        rerr = float(rng.choice([0.0, 0.0005, 0.002, 0.01]))
        rl = float(rng.choice([1200, 4000, 12000]))
        seed = int(rng.integers(1, 1 << 30))
        syn.append(Synth(L, depth=depth, seed=seed, diploid=dip, read_err_rate=rerr, asm_err_rate=float(rng.choice([0.0, 1e-4, 2e-3])),
                         read_len_mean=min(rl, L / 2), read_len_sd=min(rl / 6, L / 12), read_len_min=min(500, L // 4), name=f"c{i}"))
        desc.append((L, depth, dip, rerr, seed))
    yaks = [Synth.yak_assembly(syn, k) for k in ks]
    want = []
    orc = Oracle(yaks)
    for s in syn:
        try:
            want.append(orc.polish(s.pileup, o))
        except Exception as e:
            want.append(str(e)[:60])
    pol = Polisher(yaks)
    bp = BatchPolisher(pol, int(rng.integers(max(2, n // 2), n + 1)) if rng.integers(0, 2) else n)
    cs = [pol.upload(s.pileup) for s in syn]
    ok = True
    for wave in range(2):
        order = list(rng.permutation(n)) if wave else list(range(n))
        # (a contig that makes the reference panic fails its slot only; the call reports it: polish it on its own to see)
        got = {}
        for lo in range(0, n, bp.n_slots):
            part = order[lo:lo + bp.n_slots]
            try:
                res = bp.polish([cs[i] for i in part], o, want_pos=True)
                for i, r in zip(part, res):
                    got[i] = r
            except Np2Error:
                for i in part:  # one by one through the batch: which one fails?
                    try:
                        got[i] = bp.polish([cs[i]], o, want_pos=True)[0]
                    except Np2Error as e:
                        got[i] = str(e)[:60]
        for i in range(n):
            w, g = want[i], got[i]
            if isinstance(w, str) != isinstance(g, str):
                ok = False; print("ERROR-MISMATCH", case, wave, i, desc[i], ks, vars(o), "oracle:", w if isinstance(w, str) else "ok", "hip:", g if isinstance(g, str) else "ok")
            elif not isinstance(w, str) and not (np.array_equal(w[0], g[0]) and np.array_equal(w[1], g[1])):
                ok = False; print("MISMATCH", case, wave, i, desc[i], ks, vars(o), len(w[0]), len(g[0]))
    for c in cs:
        c.free()
    bp.close(); pol.close()
    bad += 0 if ok else 1
    print("ok" if ok else "BAD", case, n, ks, o.iter_count, sum(d[0] for d in desc), flush=True)
print("batch cases", n_case, "bad", bad, "time %.1f" % (time.time() - t0))
