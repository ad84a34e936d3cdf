"""Fuzz of the BGZF inflate kernel (csrc/np2_inflate.hip) against zlib: seeded random payloads of different statistics
(nucleotide nibbles, qualities in runs, text with repeats, far repeats, random bytes, mixtures), every compression level
and strategy zlib has, window sizes down to 512 bytes (memLevel 1), payload lengths 0 .. 65280, many blocks per call.
   python tests/tools/fuzz_inflate.py <seed> <calls> [blocks per call]        (run on a GPU box)"""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nextpolish2_amd import Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd.synth import Synth

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
per_call = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rng = np.random.default_rng(seed)
pol = Polisher([Synth(2000, seed=3).yak(21)])
STRATS = (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED)


def block(data, level, strategy, memlevel, wbits):
    co = zlib.compressobj(level, zlib.DEFLATED, -wbits, memlevel, strategy)
    comp = co.compress(data) + co.flush()
    if len(comp) + 26 > 65536:
        return None
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25)
    return hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def payload():
    kind = int(rng.integers(0, 8))
    n = int(rng.integers(0, 65281)) if rng.random() < 0.8 else int(rng.choice([0, 1, 2, 3, 255, 256, 257, 258, 259, 260, 8191, 8192, 8193, 32767, 32768, 32769, 65279, 65280]))
    if n == 0:
        return b""
    if kind == 0:    # packed nucleotides (a BAM's SEQ)
        d = ((1 << rng.integers(0, 4, n)) << 4 | (1 << rng.integers(0, 4, n))).astype(np.uint8)
    elif kind == 1:  # qualities: a few values in runs of random length
        runs = rng.integers(1, int(rng.integers(2, 400)), n // 2 + 1)
        d = np.repeat(rng.integers(0, int(rng.integers(2, 94)), len(runs), dtype=np.uint8), runs)[:n]
        if len(d) < n:
            d = np.concatenate([d, np.zeros(n - len(d), np.uint8)])
    elif kind == 2:  # text with repeats and a few random edits
        unit = bytes(rng.integers(32, 127, int(rng.integers(1, 300)), dtype=np.uint8))
        d = np.frombuffer(unit * (n // len(unit) + 2), dtype=np.uint8)[:n].copy()
        k = int(rng.integers(0, n // 20 + 2))
        d[rng.integers(0, n, k)] = rng.integers(0, 256, k)
    elif kind == 3:  # a repeat at a far distance (up to the whole window)
        dist = int(rng.integers(1, 32769))
        base = rng.integers(0, 256, dist, dtype=np.uint8)
        d = np.frombuffer(base.tobytes() * (n // dist + 2), dtype=np.uint8)[:n].copy()
    elif kind == 4:  # random bytes (literals only, or stored)
        d = rng.integers(0, 256, n, dtype=np.uint8)
    elif kind == 5:  # one byte
        d = np.full(n, int(rng.integers(0, 256)), np.uint8)
    elif kind == 6:  # a small alphabet, no structure (short codes, many symbols per word)
        d = rng.integers(0, int(rng.integers(2, 17)), n, dtype=np.uint8)
    else:            # pieces of the above glued together
        parts, left = [], n
        while left > 0:
            m = min(left, int(rng.integers(1, 20000)))
            q = int(rng.integers(0, 3))
            parts.append(rng.integers(0, 256, m, dtype=np.uint8) if q == 0 else
                         np.full(m, int(rng.integers(0, 256)), np.uint8) if q == 1 else
                         np.frombuffer(bytes(rng.integers(65, 70, int(rng.integers(1, 40)), dtype=np.uint8)) * (m + 1), dtype=np.uint8)[:m])
            left -= m
        d = np.concatenate(parts)
    return d.tobytes()


bad = n_blocks = n_bytes = 0
t0 = time.time()
for call in range(n_calls):
    blocks, want = [], []
    while len(blocks) < per_call:
        data = payload()
        b = block(data, int(rng.integers(0, 10)), STRATS[int(rng.integers(0, 5))], int(rng.integers(1, 10)), int(rng.integers(9, 16)))
        if b is None:  # (incompressible at this level: stored blocks of 65280 bytes fit, deflated random bytes may not)
            b = block(data, 0, zlib.Z_DEFAULT_STRATEGY, 8, 15)
        blocks.append(b)
        want.append(data)
    exp = b"".join(want)
    try:
        got, _ = np2io.bgzf_inflate_device(pol, b"".join(blocks))
        ok = got.tobytes() == exp
    except Exception as e:
        ok = False
        print("EXCEPTION", seed, call, e, flush=True)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "call", call, flush=True)
    n_blocks += len(blocks)
    n_bytes += len(exp)
print(f"inflate calls {n_calls} blocks {n_blocks} bytes {n_bytes} bad {bad} time {time.time() - t0:.1f}")
