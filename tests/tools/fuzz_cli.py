"""Fuzz of the command line's pipeline (run on a GPU box): seeded random assemblies (2-9 contigs of 4-60 kb, haploid and
diploid, some shorter than -L: passed through) from files — htslib-style BAM, FASTA, one k-mer dump — through
nextpolish2_amd.cli.main with the polish workers batching (16 slots, -t 2 or 4, read extraction on the device or the host
pool) and without (one contig per turn, -t 1): identical output files; one polished contig per case against the oracle
(front end + polish over the same records).
   python tests/tools/fuzz_cli.py <seed> <cases>"""
import io, os, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "tools"))
import numpy as np
from nextpolish2_amd import Opts, cli
from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import pileup_to_records, records_to_arrays
from nextpolish2_amd.synth import Synth
from oracle import np2_oracle as orc
from fuzz_bam import write_bam_straddling
from test_oracle import yak_from_seqs

rng = np.random.default_rng(int(sys.argv[1])); n_case = int(sys.argv[2]); bad = 0
td = tempfile.mkdtemp()
t0 = time.time()
for case in range(n_case):
    n_ctg = int(rng.integers(2, 10))
    syn, recs, refs, haps = [], [], [], []
    for tid in range(n_ctg):
        L = int(rng.choice([4000, 9000, 20000, 60000]))
        seed = int(rng.integers(1, 1 << 30))
        s = Synth(L, depth=int(rng.choice([8, 20, 30])), seed=seed, diploid=bool(rng.integers(0, 2)), read_len_mean=min(4000.0, L / 2), read_len_sd=600.0,
                  read_len_min=min(1000, L // 4), name=f"c{tid}")
        syn.append(s); refs.append((f"c{tid}", s.pileup.L))
        haps += [s.hap1.decode()] + ([s.hap2.decode()] if s.diploid else [])
        recs += pileup_to_records(s.pileup, tid=tid, rng=np.random.default_rng(seed), decorate=bool(rng.integers(0, 2)))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    bam, fa, yk = os.path.join(td, f"a{case}.bam"), os.path.join(td, f"a{case}.fa"), os.path.join(td, f"a{case}.yak")
    write_bam_straddling(bam, refs, recs, int(rng.choice([4096, 20000, 0xff00])), int(rng.integers(1, 10)))
    with open(fa, "w") as f:
        for i, s in enumerate(syn):
            f.write(f">c{i}\n{s.pileup.ref.tobytes().decode()}\n")
    y21 = yak_from_seqs(haps, 21)
    np2io.write_yak(yk, y21)
    min_len = int(rng.choice([1000, 8000]))  # (contigs shorter than that pass through)
    extra = (["-u"] if rng.random() < 0.3 else []) + (["--out_pos"] if rng.random() < 0.15 else [])
    outs = []
    for batch, t, mode in (("16", str(rng.choice([2, 4])), str(rng.choice(["gpu", "libdeflate"]))), ("1", "1", "libdeflate")):
        os.environ["NP2_CLI_BATCH"], os.environ["NP2_INFLATE"] = batch, mode
        o = os.path.join(td, f"o{case}_{batch}.fa")
        err, sys.stderr = sys.stderr, io.StringIO()
        try:
            rc = cli.main([bam, fa, yk, "-o", o, "-t", t, "-L", str(min_len)] + extra)
        except SystemExit as e:
            rc = "exit: " + str(e)[:60]
        except Exception as e:
            rc = "error: " + str(e)[:60]
        finally:
            sys.stderr = err
        outs.append((rc, open(o, "rb").read() if os.path.exists(o) else b""))
        if os.path.exists(o):
            os.remove(o)
    if outs[0] != outs[1]:
        bad += 1; print("MISMATCH batched vs one by one", sys.argv[1], case, outs[0][0], outs[1][0], len(outs[0][1]), len(outs[1][1]), flush=True)
    elif outs[0][0] == 0 and "--out_pos" not in extra:
        tid = int(rng.integers(0, n_ctg))
        if syn[tid].pileup.L >= min_len:
            arr, cig, seq4, asc, asc_off = records_to_arrays([r for r in recs if r["tid"] == tid])
            try:
                b, p = orc.Oracle([y21]).polish(orc.front_end(syn[tid].pileup.ref.tobytes(), arr, cig, asc, asc_off, np2io.FrontOpts()), Opts())
                want = b">c%d start:%d end:%d\n%s\n" % (tid, p[0], p[-1], b.tobytes().upper() if "-u" in extra else b.tobytes())
                if want not in outs[0][1]:
                    bad += 1; print("MISMATCH vs oracle", sys.argv[1], case, tid, flush=True)
            except Exception as e:
                bad += 1; print("MISMATCH oracle fails where the command line does not", sys.argv[1], case, tid, str(e)[:80], flush=True)
    for f in (bam, bam + ".bai", fa, yk):
        os.remove(f)
print(f"cli cases {n_case} bad {bad} time {time.time() - t0:.1f}")
