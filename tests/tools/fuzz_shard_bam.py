"""Fuzz of the shards' input side (run on a GPU box): a contig read straight from a BAM file in 2-4 reference intervals
(np2_shard_bam_*: every shard fetches only the records of its zone, the file offsets number the reads contig-wide), read
extraction on the DEVICE and on the host pool, the stitched polish compared with the whole contig's (np2_contig_from_bam +
np2_polish_resident) — files written htslib-style (records straddle blocks), a second reference before or behind.
   python tests/tools/fuzz_shard_bam.py <seed> <cases>"""
import os, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "tools"))
import numpy as np
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import pileup_to_records
from nextpolish2_amd.dist import polish_sharded_bam_local
from nextpolish2_amd.synth import Synth
from fuzz_bam import write_bam_straddling

rng = np.random.default_rng(int(sys.argv[1])); n_case = int(sys.argv[2]); bad = 0
td = tempfile.mkdtemp()
t0 = time.time()
for case in range(n_case):
    L = int(rng.choice([60000, 100000, 160000]))
    seed = int(rng.integers(1, 1 << 30))
    s = Synth(L, depth=int(rng.choice([10, 25])), seed=seed, diploid=bool(rng.integers(0, 2)), read_len_mean=3000.0, read_len_sd=500.0, name="long1")
    other = Synth(int(rng.choice([3000, 20000])), depth=8, seed=seed + 1, name="other")
    first = bool(rng.integers(0, 2))  # the long contig before or behind the other reference
    refs = [("long1", s.pileup.L), ("other", other.pileup.L)] if first else [("other", other.pileup.L), ("long1", s.pileup.L)]
    recs = pileup_to_records(s.pileup, tid=0 if first else 1, rng=np.random.default_rng(seed), decorate=True) + \
        pileup_to_records(other.pileup, tid=1 if first else 0, rng=np.random.default_rng(seed + 2), decorate=True)
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    path = os.path.join(td, f"s{case}.bam")
    write_bam_straddling(path, refs, recs, int(rng.choice([300, 4096, 20000, 0xff00])), int(rng.integers(0, 10)))
    yaks = [s.yak(21)]
    pol = Polisher(yaks)
    ref = s.pileup.ref.tobytes()
    n_shards = int(rng.integers(2, 5)); halo = int(rng.choice([12000, 20000]))
    res = {}
    for mode in ("gpu", "libdeflate"):
        os.environ["NP2_INFLATE"] = mode
        try:
            whole = np2io.contig_from_bam(pol, np2io.Bam(path), "long1", ref)
            b0, p0 = pol.polish_resident(whole, Opts())
            whole.free()
            b1, p1 = polish_sharded_bam_local(pol, path, "long1", ref, n_shards, Opts(), halo=halo)
            res[mode] = (b0.tobytes(), p0.tobytes(), b1.tobytes(), p1.tobytes())
        except Exception as e:
            res[mode] = "error: " + str(e)[:80]
    a, b = res["gpu"], res["libdeflate"]
    if isinstance(a, str) or isinstance(b, str):
        if a != b:
            bad += 1; print("ERR-MISMATCH", sys.argv[1], case, a if isinstance(a, str) else "ok", b if isinstance(b, str) else "ok", flush=True)
    elif not (a[0] == a[2] and a[1] == a[3] and a == b):
        bad += 1; print("MISMATCH", sys.argv[1], case, "device whole == shards:", a[0] == a[2] and a[1] == a[3], "host whole == shards:", b[0] == b[2] and b[1] == b[3],
                        "device == host:", a == b, n_shards, halo, L, flush=True)
    os.remove(path); os.remove(path + ".bai")
print(f"shard-bam cases {n_case} bad {bad} time {time.time() - t0:.1f}")
