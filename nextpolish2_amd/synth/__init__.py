"""Synthetic pileup / yak generator (host-only; SURVEY.md §8(d) recipe). Binding of np2_synth.cpp."""
import ctypes as C
import os
import subprocess

import numpy as np

from .._types import READ_DTYPE, Pileup, Yak, np2_read_t

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class np2s_params_t(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("L", C.c_uint32), ("depth", C.c_uint32), ("diploid", C.c_uint32),
        ("read_len_min", C.c_uint32), ("snp_rate", C.c_double), ("hap_indel_rate", C.c_double),
        ("asm_err_rate", C.c_double), ("read_err_rate", C.c_double), ("read_len_mean", C.c_double),
        ("read_len_sd", C.c_double),
    ]


def build():
    src = os.path.join(_HERE, "np2_synth.cpp")
    out = os.path.join(_HERE, "libnp2_synth.so")
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", out, src])
    return out


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libnp2_synth.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.np2s_generate.restype = C.c_void_p
        L.np2s_generate.argtypes = [C.POINTER(np2s_params_t)]
        L.np2s_free.argtypes = [C.c_void_p]
        for f in ("np2s_ref",):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.np2s_hap.restype = C.c_void_p
        L.np2s_hap.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
        L.np2s_reads.restype = C.c_void_p
        L.np2s_reads.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.np2s_nibbles.restype = C.c_void_p
        L.np2s_nibbles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.np2s_yak_build.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_uint64, C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
        L.np2s_yak_build_multi.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_double, C.c_uint64,
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
        L.np2s_yak_build_multi_mt.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_double, C.c_uint64, C.c_uint32,
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
        L.np2s_bam_records.restype = C.c_uint32
        L.np2s_bam_records.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.np2s_bam_records_at.restype = C.c_uint32
        L.np2s_bam_records_at.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.np2s_free_buf.argtypes = [C.c_void_p]
        L.np2s_pack_alignment.restype = C.c_uint64
        L.np2s_pack_alignment.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
        _LIB = L
    return _LIB


def _copy(ptr, nbytes, dtype):
    if nbytes == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return np.frombuffer(bytes(buf), dtype=dtype).copy()


class Synth:
    """One synthetic contig: assembly + packed HiFi pileup + truth haplotypes; builds yak tables."""

    def __init__(self, L, depth=30, seed=1, diploid=False, snp_rate=0.005, hap_indel_rate=0.002,
                 asm_err_rate=1e-4, read_err_rate=0.002, read_len_mean=13000.0, read_len_sd=2000.0,
                 read_len_min=1000, name="ctg"):
        p = np2s_params_t(seed, L, depth, 1 if diploid else 0, read_len_min, snp_rate, hap_indel_rate, asm_err_rate,
                          read_err_rate, read_len_mean, read_len_sd)
        self._h = lib().np2s_generate(C.byref(p))
        self.diploid = diploid
        n32, n64 = C.c_uint32(), C.c_uint64()
        ptr = lib().np2s_ref(self._h, C.byref(n32))
        ref = _copy(ptr, n32.value, np.uint8)
        ptr = lib().np2s_reads(self._h, C.byref(n32))
        reads = _copy(ptr, n32.value * C.sizeof(np2_read_t), READ_DTYPE)
        ptr = lib().np2s_nibbles(self._h, C.byref(n64))
        nib = _copy(ptr, n64.value, np.uint8)
        self.pileup = Pileup(ref, reads, nib, name=name)
        ptr = lib().np2s_hap(self._h, 0, C.byref(n32))
        self.hap1 = _copy(ptr, n32.value, np.uint8).tobytes()
        ptr = lib().np2s_hap(self._h, 1, C.byref(n32))
        self.hap2 = _copy(ptr, n32.value, np.uint8).tobytes()

    def bam_records(self, tid=0, pos0=0, name0=0):
        """The reads as encoded BAM alignment records (see np2s_bam_records): -> (blob bytes, record offsets [n + 1],
        positions [n], reference lengths [n]) for bamio.write_bam_raw.  pos0 / name0: position and record-name offset of a
        generated piece inside a contig laid out of several (concat_pileups)."""
        b, o, p, r = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        n = lib().np2s_bam_records_at(self._h, tid, pos0, name0, C.byref(b), C.byref(o), C.byref(p), C.byref(r))
        off = _copy(o.value, (n + 1) * 8, np.uint64)
        out = (bytes((C.c_uint8 * int(off[n])).from_address(b.value)) if n else b"", off,
               _copy(p.value, n * 4, np.int32), _copy(r.value, n * 4, np.uint32))
        for x in (b, o, p, r):
            lib().np2s_free_buf(x)
        return out

    def yak(self, k, coverage=60.0, read_len=150, seed=7):
        """yak table of the true haplotype(s): lambda = cov*(rl-k+1)/rl/ploidy."""
        lam = coverage * (read_len - k + 1) / read_len / (2 if self.diploid else 1)
        w, n, o = C.c_void_p(), C.c_uint64(), C.c_void_p()
        rc = lib().np2s_yak_build(self._h, k, lam, seed, C.byref(w), C.byref(n), C.byref(o))
        if rc != 0:
            raise ValueError("k must be in [2, 32)")
        words = _copy(w.value, n.value * 8, np.uint64)
        off = _copy(o.value, 1025 * 8, np.uint64)
        return Yak(k, words, off)

    @staticmethod
    def yak_assembly(synths, k, coverage=60.0, read_len=150, seed=7, threads=0):
        """One yak table over every contig of a synthetic assembly (what `yak count` on the short reads gives).
        threads > 1: built on that many host threads (chromosome-scale inputs; the same recipe, other random draws)."""
        dip = synths[0].diploid
        lam = coverage * (read_len - k + 1) / read_len / (2 if dip else 1)
        hs = (C.c_void_p * len(synths))(*[s._h for s in synths])
        w, n, o = C.c_void_p(), C.c_uint64(), C.c_void_p()
        if threads > 1:
            rc = lib().np2s_yak_build_multi_mt(hs, len(synths), k, lam, seed, threads, C.byref(w), C.byref(n), C.byref(o))
        else:
            rc = lib().np2s_yak_build_multi(hs, len(synths), k, lam, seed, C.byref(w), C.byref(n), C.byref(o))
        if rc != 0:
            raise ValueError("k must be in [2, 32)")
        return Yak(k, _copy(w.value, n.value * 8, np.uint64), _copy(o.value, 1025 * 8, np.uint64))

    def close(self):
        if self._h:
            lib().np2s_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def concat_pileups(parts, name="ctg"):
    """One contig out of several generated pieces laid end to end (reads never span a joint; coverage tapers to the
    contig's own base there).  Lets a chromosome-sized contig be generated on many host threads."""
    L = sum(p.L for p in parts)
    ref = np.concatenate([p.ref for p in parts])
    # slot 0: the whole contig aligned to itself
    codes = np.full(256, 4, dtype=np.uint8)
    for ch, c in ((b"Aa", 0), (b"Cc", 1), (b"Gg", 2), (b"TtUu", 3), (b"Nn", 5), (b"Mm", 6)):
        for x in ch:
            codes[x] = c
    cod = codes[ref]
    if L & 1:
        cod = np.concatenate([cod, np.array([15], dtype=np.uint8)])
        packed = (cod[0::2] << 4) | cod[1::2]
    else:
        packed = np.concatenate([(cod[0::2] << 4) | cod[1::2], np.array([0xFF], dtype=np.uint8)])
    slot0 = ((len(packed) + 1 + 15) // 16) * 16
    chunks = [packed, np.zeros(slot0 - len(packed), dtype=np.uint8)]
    reads = [np.array([(0, L - 1, 0, L, 0)], dtype=READ_DTYPE)]
    off, pos0 = slot0, 0
    for p in parts:
        r = p.reads[1:].copy()
        first = int(p.reads["nib_off"][1]) if len(r) else 0
        r["aln_t_s"] += pos0
        r["aln_t_e"] += pos0
        r["nib_off"] = r["nib_off"] - first + off
        reads.append(r)
        body = p.nibbles[first:]
        pad = (-len(body)) % 16
        chunks += [body, np.zeros(pad, dtype=np.uint8)]
        off += len(body) + pad
        pos0 += p.L
    chunks.append(np.zeros(64, dtype=np.uint8))
    return Pileup(ref, np.concatenate(reads), np.concatenate(chunks), name=name)


def pack_alignment(t_aln, q_aln, aln_t_s):
    """AlignSeq::new (src/main.rs:279-312) for explicit gapped strings -> (bytes, aln_t_e, n_cols)."""
    assert len(t_aln) == len(q_aln)
    n = len(t_aln)
    dst = np.zeros(((n + 1) >> 1) + 1, dtype=np.uint8)
    te = C.c_uint32()
    lib().np2s_pack_alignment(t_aln.encode() if isinstance(t_aln, str) else t_aln,
                              q_aln.encode() if isinstance(q_aln, str) else q_aln, n, aln_t_s, dst.ctypes.data, C.byref(te))
    return dst, te.value, n


def pileup_from_alignments(ref, alns, name="ctg"):
    """Build a Pileup from the contig string and [(aln_t_s, t_aln, q_aln)] gapped alignments.

    reads[0] is the contig aligned to itself (src/main.rs:1732-1739)."""
    if isinstance(ref, str):
        ref = ref.encode()
    items = [(0, ref.decode(), ref.decode())] + [(s, t, q) for (s, t, q) in alns]
    reads = np.zeros(len(items), dtype=READ_DTYPE)
    chunks, off = [], 0
    for i, (s, t, q) in enumerate(items):
        b, te, n = pack_alignment(t, q, s)
        reads[i] = (s, te, off, n, 0)
        pad = (-len(b)) % 16
        chunks.append(b)
        chunks.append(np.zeros(pad, dtype=np.uint8))
        off += len(b) + pad
    chunks.append(np.zeros(64, dtype=np.uint8))
    return Pileup(np.frombuffer(ref, dtype=np.uint8), reads, np.concatenate(chunks), name=name)
