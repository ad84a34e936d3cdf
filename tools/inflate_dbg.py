"""Debug helper: the blocks of tests/test_gpu_inflate.py::test_inflate_kernel_equals_zlib one by one (run through gpurun)."""
import os, sys, zlib
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from nextpolish2_amd import Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd.synth import Synth
from test_gpu_inflate import bgzf_block
rng = np.random.default_rng(5)
pol = Polisher([Synth(2000, seed=3).yak(21)])
cases = []
def add(name, data, **kw):
    cases.append((name, bytes(data), kw))
add("empty", b"")
add("A", b"A")
add("stored", bytes(rng.integers(0, 256, 65280, dtype=np.uint8)), level=0)
add("rand", bytes(rng.integers(0, 256, 40000, dtype=np.uint8)), level=6)
for n in (1, 2, 3, 257, 258, 259, 4095, 4096, 4097, 32767, 32768, 32769, 65280):
    add("run%d" % n, b"\xff" * n)
for lvl in (1, 4, 6, 9):
    for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        n = int(rng.integers(1, 65281))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            d = ((1 << rng.integers(0, 4, n)) << 4 | (1 << rng.integers(0, 4, n))).astype(np.uint8)
        elif kind == 1:
            d = np.frombuffer((b"GATTACA-%d-" % lvl) * (n // 8 + 2), dtype=np.uint8)[:n].copy()
            d[rng.integers(0, n, n // 50 + 1)] = rng.integers(0, 256, n // 50 + 1)
        elif kind == 2:
            base = rng.integers(0, 256, 32768, dtype=np.uint8)
            d = np.concatenate([base, base])[: max(n, 40000)][:65280]
        else:
            d = np.repeat(rng.integers(0, 42, n // 40 + 1, dtype=np.uint8), 40)[:n]
        add("l%d_s%d_k%d_n%d" % (lvl, strat, kind, len(d)), d.tobytes(), level=lvl, strategy=strat, memlevel=1 if (lvl + kind) % 2 else 8)
for name, data, kw in cases:
    try:
        got, ms = np2io.bgzf_inflate_device(pol, bgzf_block(data, **kw))
    except Exception as e:
        print(name, "ERROR", str(e)[:120])
        continue
    g = got.tobytes()
    if g == data:
        continue
    k = next((i for i in range(min(len(g), len(data))) if g[i] != data[i]), -1)
    nbad = sum(1 for i in range(min(len(g), len(data))) if g[i] != data[i])
    print(name, "MISMATCH first at", k, "of", len(data), "(%d bytes differ)" % nbad, "got", g[max(0, k - 4):k + 8], "want", data[max(0, k - 4):k + 8])
print("done", len(cases))
# ... and all of them in one call
allb = b"".join(bgzf_block(d, **kw) for _, d, kw in cases)
got, ms = np2io.bgzf_inflate_device(pol, allb)
g = got.tobytes()
o = 0
for name, data, kw in cases:
    seg = g[o:o + len(data)]
    if seg != data:
        k = next((i for i in range(len(data)) if seg[i] != data[i]), -1)
        nbad = sum(1 for i in range(len(data)) if seg[i] != data[i])
        print("in one call:", name, "out_off", o, "(mod 64: %d)" % (o % 64), "MISMATCH first at", k, "of", len(data), "(%d differ)" % nbad, "got", seg[max(0, k - 4):k + 8], "want", data[max(0, k - 4):k + 8])
    o += len(data)
print("one call done")
