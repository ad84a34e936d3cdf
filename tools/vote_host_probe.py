"""Host side of the phasing vote on a synthetic read graph (no GPU needed): python tools/vote_host_probe.py [reads] [partners]
Two haplotypes, reads alternating between them in start order, every read paired with its next `partners` reads:
agreeing counts inside a haplotype, disagreeing ones across (with some noise), a few reads flagged bad.  Prints the wall
time of np2_vote_decide (NP2_PHASE_PROFILE=1 for its parts) and a checksum of the losers — equal before and after any
change of the host code."""
import os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextpolish2_amd.api import Vote, vote_decide
R = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(7)
hap = rng.integers(0, 2, R).astype(np.int8)
a = np.repeat(np.arange(1, R, dtype=np.uint64), P)
b = a + np.tile(np.arange(1, P + 1, dtype=np.uint64), R - 1)
keep = b < R
a, b = a[keep], b[keep]
same_hap = hap[a.astype(np.int64)] == hap[b.astype(np.int64)]
n = rng.integers(3, 20, a.shape[0]).astype(np.uint32)
noise = rng.random(a.shape[0]) < 0.02
agree = np.where(same_hap ^ noise, n, rng.integers(0, 3, a.shape[0]).astype(np.uint32))
neg = np.where(same_hap ^ noise, rng.integers(0, 2, a.shape[0]).astype(np.uint32), n)
pair_key = (a << np.uint64(32)) | b
pair_cnt = (agree & 0xFFFF) | (neg << 16)
read_id = np.arange(1, R, dtype=np.uint32)
flags = np.full(R - 1, 1 | 2, dtype=np.uint8)
flags[rng.random(R - 1) < 0.01] |= 4
ref_w = np.where(hap[1:] == 0, rng.integers(1, 10, R - 1), -rng.integers(1, 10, R - 1)).astype(np.int32)
first_pos = (np.arange(1, R) * 400).astype(np.uint32)
v = Vote(pair_key=pair_key, pair_cnt=pair_cnt.astype(np.uint32), read_id=read_id, first_pos=first_pos, ref_w=ref_w, flags=flags)
print(f"reads {R}, pairs {len(pair_key)}", flush=True)
for rep in range(3):
    t = time.time()
    losers = vote_decide([v], R)
    print(f"np2_vote_decide {1e3 * (time.time() - t):.1f} ms, losers {len(losers)}, crc {zlib.crc32(losers.tobytes()):08x}", flush=True)
