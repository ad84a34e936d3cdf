"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package nextpolish2_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from nextpolish2_amd._types import Opts, np2_opts_t, np2_read_t, np2_yak_t, yaks_array

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libnp2_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.np2o_ctx_create.restype = C.c_void_p
        L.np2o_ctx_create.argtypes = [C.POINTER(np2_yak_t), C.c_int]
        L.np2o_ctx_destroy.argtypes = [C.c_void_p]
        L.np2o_swiss_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
        L.np2o_set_yak_files.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int]
        L.np2o_ctx_clone.restype = C.c_void_p
        L.np2o_ctx_clone.argtypes = [C.c_void_p, C.c_uint16]
        L.np2o_last_error.restype = C.c_char_p
        L.np2o_last_error.argtypes = [C.c_void_p]
        L.np2o_set_trace.argtypes = [C.c_void_p, C.c_int]
        L.np2o_polish_contig.argtypes = [
            C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(np2_opts_t),
            C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
        ]
        L.np2o_free.argtypes = [C.c_void_p]
        L.np2o_trace_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.np2o_last_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.np2o_score_strings.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint16, C.c_void_p]
        L.np2o_lookup_hashes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint16, C.c_void_p]
        L.np2o_yak_hash64.restype = C.c_uint64
        L.np2o_yak_hash64.argtypes = [C.c_uint64, C.c_uint32]
        L.np2o_phase_communities.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
            C.c_void_p, C.POINTER(C.c_uint64),
        ]
        _LIB = L
    return _LIB


TRACE_DTYPES = {
    "graph.off": np.uint32, "graph.bases": np.uint16, "graph.delta": np.uint16, "graph.count": np.uint32,
    "lq.start": np.uint32, "lq.end": np.uint32, "invalid_ids": np.uint32,
}
for _t in ("cand", "seed", "hete", "rech0", "rech1", "rech2"):
    TRACE_DTYPES.update({
        f"{_t}.start": np.uint32, f"{_t}.end": np.uint32, f"{_t}.lable": np.uint8, f"{_t}.sudo_off": np.uint32,
        f"{_t}.sudo": np.uint8, f"{_t}.cand_off": np.uint32, f"{_t}.order": np.uint32, f"{_t}.kscore": np.uint16,
        f"{_t}.kmer": np.uint64, f"{_t}.seq_off": np.uint32, f"{_t}.seq": np.uint8,
    })
for _t in ("cns_raw", "cns_succ", "cns_rech0", "cns_rech1", "cns_rech2"):
    TRACE_DTYPES.update({f"{_t}.pos": np.uint32, f"{_t}.base": np.uint8})


class Unsupported(RuntimeError):
    """An input both the product and the oracle refuse (NP2_E_UNSUPPORTED)."""


class RefPanic(RuntimeError):
    pass


class Oracle:
    """CPU restatement of the reference hot path (oracle/np2_oracle.cpp)."""

    def __init__(self, yaks):
        self._yaks = list(yaks)  # keep numpy buffers alive
        arr = yaks_array(self._yaks)
        self._h = lib().np2o_ctx_create(arr, len(self._yaks))
        if not self._h:
            raise ValueError("oracle: unsupported yak table (k must be < 32)")

    def set_yak_files(self, paths):
        """Variant (i) of the CPU baseline: every scoring phase re-streams these .yak dumps like the reference does
        (kmer.rs:132-170); results are unchanged.  None switches back to the in-memory tables only."""
        n = len(self._yaks)
        arr = (C.c_char_p * n)(*[(p.encode() if p else None) for p in (paths or [None] * n)])
        if lib().np2o_set_yak_files(self._h, arr, n) != 0:
            raise ValueError("one path per yak table")

    def clone(self, min_kmer_count=5):
        """A further oracle over the SAME in-memory k-mer tables (one per host thread of the CPU baseline)."""
        o = Oracle.__new__(Oracle)
        o._yaks = self._yaks
        o._parent = self
        o._h = lib().np2o_ctx_clone(self._h, min_kmer_count)
        return o

    def close(self):
        if self._h:
            lib().np2o_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_trace(self, on=True):
        lib().np2o_set_trace(self._h, 1 if on else 0)

    def polish(self, pileup, opts=None):
        opts = opts or Opts()
        o = opts.c()
        ob, op, on = C.c_void_p(), C.c_void_p(), C.c_uint64()
        rc = lib().np2o_polish_contig(
            self._h, pileup.ref.ctypes.data, pileup.L, pileup.reads.ctypes.data, pileup.n_reads,
            pileup.nibbles.ctypes.data, C.byref(o), C.byref(ob), C.byref(op), C.byref(on),
        )
        if rc != 0:
            msg = lib().np2o_last_error(self._h).decode()
            if rc == -5:
                raise RefPanic(msg)
            if rc == -4:
                raise Unsupported(msg)
            raise RuntimeError(f"oracle rc={rc}: {msg}")
        n = on.value
        bases = np.ctypeslib.as_array(C.cast(ob, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n].copy()
        pos = np.ctypeslib.as_array(C.cast(op, C.POINTER(C.c_uint32)), shape=(max(n, 1),))[:n].copy()
        lib().np2o_free(ob)
        lib().np2o_free(op)
        return bases, pos

    def trace(self, pass_idx, name):
        d, n = C.c_void_p(), C.c_uint64()
        rc = lib().np2o_trace_get(self._h, pass_idx, name.encode(), C.byref(d), C.byref(n))
        if rc != 0:
            return None
        dt = np.dtype(TRACE_DTYPES[name])
        if n.value == 0:
            return np.zeros(0, dtype=dt)
        buf = (C.c_uint8 * n.value).from_address(d.value)
        return np.frombuffer(bytes(buf), dtype=dt)

    def stats(self):
        out = np.zeros(5, dtype=np.uint64)
        lib().np2o_last_stats(self._h, out.ctypes.data)
        return dict(zip(["kmer_probes", "n_regions", "n_candidates", "n_nodes", "n_invalid"], out.tolist()))

    def score_strings(self, yak_idx, strings, min_kmer_count=5):
        off = np.zeros(len(strings) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(s) for s in strings])
        blob = np.frombuffer(b"".join(strings) + b"\0", dtype=np.uint8)
        out = np.zeros(len(strings), dtype=np.uint16)
        rc = lib().np2o_score_strings(self._h, yak_idx, blob.ctypes.data, off.ctypes.data, len(strings), min_kmer_count, out.ctypes.data)
        assert rc == 0
        return out

    def lookup_hashes(self, yak_idx, hashes, min_kmer_count=5):
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        out = np.zeros(h.shape[0], dtype=np.uint16)
        rc = lib().np2o_lookup_hashes(self._h, yak_idx, h.ctypes.data, h.shape[0], min_kmer_count, out.ctypes.data)
        assert rc == 0
        return out


def swiss_order(script):
    """Iteration order of the oracle's hashbrown emulation after a script [(op, key)]: 0 insert, 1 remove, 2 entry."""
    ops = np.array([o for o, _ in script], dtype=np.uint32)
    keys = np.array([k for _, k in script], dtype=np.uint32)
    out = np.zeros(max(1, len(script)), dtype=np.uint32)
    n = C.c_uint32()
    lib().np2o_swiss_order(ops.ctypes.data, keys.ctypes.data, len(script), out.ctypes.data, C.byref(n))
    return out[: n.value].tolist()


def yak_hash64(kmer, k):
    return int(lib().np2o_yak_hash64(int(kmer), int(k)))


def phase_communities(edges, ref=None):
    """edges: list of (a, b, w) applied in order with insert_data; ref: dict id->w or None."""
    ea = np.array([e[0] for e in edges], dtype=np.uint32)
    eb = np.array([e[1] for e in edges], dtype=np.uint32)
    ew = np.array([e[2] for e in edges], dtype=np.float32)
    ri = np.array(list(ref.keys()) if ref else [], dtype=np.uint32)
    rw = np.array(list(ref.values()) if ref else [], dtype=np.float32)
    out = np.zeros(max(1, len(edges) * 2 + 8), dtype=np.uint32)
    n = C.c_uint64()
    rc = lib().np2o_phase_communities(ea.ctypes.data, eb.ctypes.data, ew.ctypes.data, len(edges), ri.ctypes.data,
                                      rw.ctypes.data, len(ri), 1 if ref is not None else 0, out.ctypes.data, C.byref(n))
    if rc != 0:
        raise RefPanic("phase_communities")
    return out[: n.value].tolist()


class np2o_front_opts_t(C.Structure):
    _fields_ = [("min_read_len", C.c_uint32), ("min_map_len", C.c_uint32), ("min_map_fra", C.c_float),
                ("min_map_qual", C.c_int16), ("max_clip_len", C.c_uint32), ("use_supplementary", C.c_uint8),
                ("use_secondary", C.c_uint8)]


def front_end(ref, recs, cigar, ascii_seq, ascii_off, fopts):
    """Oracle of the read admission + columnarisation (oracle/np2_oracle_front.cpp) -> Pileup.

    recs: BAMREC_DTYPE array (seq_off is replaced by ascii_off); ascii_seq: SEQ as rust-htslib decodes it."""
    from nextpolish2_amd._types import READ_DTYPE, Pileup
    L_ = lib()
    L_.np2o_front_end.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_char_p,
                                  C.POINTER(np2o_front_opts_t), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L_.np2o_front_free.argtypes = [C.c_void_p]
    ref = bytes(ref)
    r2 = np.array(recs, copy=True)
    r2["seq_off"] = np.asarray(ascii_off, dtype=np.uint64)
    cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
    o = np2o_front_opts_t(fopts.min_read_len, fopts.min_map_len, fopts.min_map_fra, fopts.min_map_qual,
                          fopts.max_clip_len, 1 if fopts.use_supplementary else 0, 1 if fopts.use_secondary else 0)
    pr, pn, nr, nb = C.c_void_p(), C.c_void_p(), C.c_uint32(), C.c_uint64()
    rc = L_.np2o_front_end(ref, len(ref), r2.ctypes.data, r2.shape[0], cigar.ctypes.data, bytes(ascii_seq) + b"\0",
                           C.byref(o), C.byref(pr), C.byref(nr), C.byref(pn), C.byref(nb))
    if rc == -5:
        raise RefPanic("front end")
    if rc != 0:
        raise RuntimeError(f"oracle front end rc={rc}")
    reads = np.frombuffer(C.string_at(pr.value, nr.value * 24), dtype=READ_DTYPE).copy()
    nib = np.frombuffer(C.string_at(pn.value, nb.value), dtype=np.uint8).copy()
    L_.np2o_front_free(pr)
    L_.np2o_front_free(pn)
    return Pileup(np.frombuffer(ref, dtype=np.uint8), reads, nib)


_RC = {"A": "T", "a": "T", "T": "A", "t": "A", "G": "C", "g": "C", "C": "G", "c": "G"}


def reverse_complement(seq):
    """reverse_complement_seq_u8 (secondary.rs:66-80): A<->T, C<->G (either case -> upper), anything else unchanged."""
    return "".join(_RC.get(ch, ch) for ch in reversed(seq))


def secondary_seqs(all_records):
    """retrieve_secondary_seq_from_bam (secondary.rs:8-148) over every record of the BAM (dicts with name, flag, seq):
    {read name: SEQ of its primary record in read orientation} for the reads that have a secondary record."""
    ids = {r["name"] for r in all_records if r["flag"] & 0x100}
    out = {}
    for r in all_records:
        if r["name"] in ids and not (r["flag"] & 0x900):
            assert r["name"] not in out, "reference would panic: two primary records for one read name"
            out[r["name"]] = reverse_complement(r["seq"]) if r["flag"] & 0x10 else r["seq"]
    return out


def with_secondary_seq(records, sec):
    """Records of one contig with the SEQ of every secondary record replaced as main.rs:1775-1784 does before
    fill_with_cigar: the primary's SEQ, reverse-complemented if the secondary record is on the reverse strand.
    A read without a primary record gets an empty SEQ (the reference would panic if the record is admitted)."""
    out = []
    for r in records:
        if r["flag"] & 0x100:
            q = sec.get(r["name"], "")
            r = dict(r, seq=reverse_complement(q) if r["flag"] & 0x10 else q)
        out.append(r)
    return out

