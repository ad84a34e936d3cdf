"""How fast are k-mer probes against an HBM-resident yak table of human scale, and would an LDS cache of "hot buckets"
help?  Fabricates a yak v2 table of N words (uniform over the 1024 file buckets, as the invertible hash makes them),
then times np2_lookup_hashes for (a) uniformly random present keys, (b) a small hot set probed over and over (what
the candidates of one region do: 30 reads share most k-mers), (c) absent keys.  Run under rocprofv3 --kernel-trace --stats
to get the k_lookup durations.  usage: python tools/yak_probe.py [n_words] [n_probes]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextpolish2_amd import Polisher
from nextpolish2_amd._types import Yak

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 28
P = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1 << 25
rng = np.random.default_rng(1)
t = time.time()
per = N // 1024
off = (np.arange(1025, dtype=np.uint64) * per)
# distinct keys inside a bucket: a random permutation-free construction (sorted distinct 44-bit values)
keys = rng.integers(0, 1 << 44, size=per * 1024, dtype=np.uint64)
keys = (keys.reshape(1024, per) | (np.arange(per, dtype=np.uint64) << np.uint64(44))[None, :]).reshape(-1)  # distinct per bucket
words = (keys << np.uint64(10)) | rng.integers(5, 1000, size=per * 1024, dtype=np.uint64)
print(f"table: {len(words)/1e6:.0f} M words ({len(words)*8/1e9:.1f} GB) fabricated in {time.time()-t:.1f}s", flush=True)
t = time.time()
pol = Polisher([Yak(21, words, off)])
print(f"HBM table built in {time.time()-t:.1f}s", flush=True)
bucket = rng.integers(0, 1024, size=P, dtype=np.uint64)
idx = rng.integers(0, per, size=P, dtype=np.uint64)
present = ((words[bucket * np.uint64(per) + idx] >> np.uint64(10)) << np.uint64(10)) | bucket
hot = present[: 1 << 14][rng.integers(0, 1 << 14, size=P)]
absent = present ^ np.uint64(1 << 40)
for name, h in (("uniform present", present), ("hot 16k set", hot), ("absent", absent)):
    for rep in range(2):
        t = time.time(); c = pol.lookup_hashes(0, h); dt = time.time() - t
    ok = (c > 0).mean()
    print(f"{name:16s}: {P/1e6:.0f} M probes, wall {dt*1e3:.1f} ms incl. transfers, hit fraction {ok:.3f}", flush=True)
