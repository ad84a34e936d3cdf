"""Replay a dumped phasing vote (tools/vote_dump.py) on the host, tiled `copies` times along the contig (read ids and
positions shifted) to reach chromosome scale: python tools/vote_replay.py gpurun_out/vote_16000000.npz [copies]"""
import os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextpolish2_amd.api import Vote, vote_decide
z = np.load(sys.argv[1])
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 1
v0 = Vote.unpack(z["packed"])
R0 = int(z["n_reads"][0])
if copies == 1:
    v, R = v0, R0
else:
    span = int(v0.first_pos.max()) + 100000
    ks, cs, ids, fp, rw, fl = [], [], [], [], [], []
    for c in range(copies):
        sh = np.uint64(c * (R0 - 1))
        ks.append(v0.pair_key + ((sh << np.uint64(32)) | sh))
        cs.append(v0.pair_cnt)
        ids.append(v0.read_id + np.uint32(c * (R0 - 1)))
        fp.append(v0.first_pos + np.uint32(c * span))
        rw.append(v0.ref_w)
        fl.append(v0.flags)
    v = Vote(pair_key=np.concatenate(ks), pair_cnt=np.concatenate(cs), read_id=np.concatenate(ids), first_pos=np.concatenate(fp),
             ref_w=np.concatenate(rw), flags=np.concatenate(fl))
    R = 1 + copies * (R0 - 1)
print(f"reads {R}, pairs {len(v.pair_key)}", flush=True)
for rep in range(3):
    t = time.time()
    losers = vote_decide([v], R)
    print(f"np2_vote_decide {1e3 * (time.time() - t):.1f} ms, losers {len(losers)}, crc {zlib.crc32(losers.tobytes()):08x}", flush=True)
if copies == 1:
    print("equals the dumped decision:", np.array_equal(losers, z["losers"]))
