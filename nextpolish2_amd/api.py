"""ctypes binding of libnp2_hip.so — the C-ABI of include/np2.h.

There is no CPU fallback: importing is fine without a GPU (symbols can be inspected), but
creating a Polisher requires the in-tree HIP library and a visible MI355X device.
"""
import ctypes as C
import os
import weakref

import numpy as np

from ._types import (Opts, Pileup, np2_opts_t, np2_read_t, np2_shard_piece_t, np2_shard_plan_t, np2_vote_t, np2_yak_t,
                     yaks_array)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NP2_LIB_PATH") or os.path.join(_HERE, "libnp2_hip.so")  # override: A/B builds
_LIB = None

# every symbol include/np2.h declares
ABI_SYMBOLS = [
    "np2_ctx_create", "np2_ctx_destroy", "np2_last_error", "np2_ctx_stream", "np2_contig_upload",
    "np2_contig_free", "np2_polish_resident", "np2_polish_contig", "np2_free", "np2_score_strings",
    "np2_lookup_hashes", "np2_ctx_set_trace", "np2_ctx_set_timing", "np2_trace_get", "np2_last_timings", "np2_last_span", "np2_last_result_device", "np2_result_fetch_begin", "np2_result_fetch_end", "np2_phase_vote",
    "np2_ctx_create_shared", "np2_batch_create", "np2_batch_destroy", "np2_batch_slots", "np2_batch_slot_ctx", "np2_batch_set_sink",
    "np2_batch_last_error", "np2_batch_polish", "np2_batch_flush_log", "np2_shard_plan", "np2_shard_upload",
    "np2_shard_begin", "np2_shard_passes_left", "np2_shard_vote", "np2_vote_decide", "np2_shard_apply", "np2_shard_final",
    "np2_shard_final_device", "np2_shard_fetch", "np2_alloc_pinned", "np2_trim_device_cache",
    "np2_shard_end", "np2_swiss_order", "np2_batch_set_timing", "np2_batch_set_priority", "np2_batch_last_diff_ms", "np2_batch_stats", "np2_batch_last_call_ms",
]

# include/np2_io.h (input side; bound by nextpolish2_amd.io)
IO_ABI_SYMBOLS = [
    "np2_fasta_open", "np2_fasta_next", "np2_fasta_close", "np2_yak_load", "np2_yak_free", "np2_bam_open", "np2_bam_close",
    "np2_bam_n_refs", "np2_bam_ref_name", "np2_io_last_error", "np2_contig_from_records", "np2_contig_from_bam",
    "np2_contig_export", "np2_shard_bam_begin", "np2_shard_bam_finish", "np2_shard_bam_abort", "np2_ctx_create_from_files",
    "np2_bgzf_inflate_device",
]

ERRORS = {-1: "NP2_E_ARG", -2: "NP2_E_DEVICE", -3: "NP2_E_NOMEM", -4: "NP2_E_UNSUPPORTED", -5: "NP2_E_REFPANIC"}


class Np2Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


_LIB_LOCK = __import__("threading").Lock()


def lib():
    """Load libnp2_hip.so; raises loudly if the HIP extension has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LIB_LOCK:
        return _lib_locked()


def _lib_locked():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with nextpolish2_amd/csrc/build.sh "
                "(the np2 hot path has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, u16 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint16
        L.np2_ctx_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(np2_yak_t), C.c_int]
        L.np2_ctx_destroy.argtypes = [vp]
        L.np2_last_error.restype = C.c_char_p
        L.np2_last_error.argtypes = [vp]
        L.np2_ctx_stream.restype = vp
        L.np2_ctx_stream.argtypes = [vp]
        L.np2_contig_upload.argtypes = [vp, vp, u32, vp, u32, vp, u64, C.POINTER(vp)]
        L.np2_contig_free.argtypes = [vp, vp]
        L.np2_polish_resident.argtypes = [vp, vp, C.POINTER(np2_opts_t), C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
        L.np2_polish_contig.argtypes = [vp, vp, u32, vp, u32, vp, u64, C.POINTER(np2_opts_t), C.POINTER(vp),
                                        C.POINTER(vp), C.POINTER(u64)]
        L.np2_free.argtypes = [vp]
        L.np2_score_strings.argtypes = [vp, C.c_int, vp, vp, u64, u16, vp]
        L.np2_lookup_hashes.argtypes = [vp, C.c_int, vp, u64, u16, vp]
        L.np2_ctx_set_trace.argtypes = [vp, C.c_int]
        L.np2_ctx_set_timing.argtypes = [vp, C.c_int]
        L.np2_ctx_set_timing.restype = None
        L.np2_trace_get.argtypes = [vp, C.c_int, C.c_char_p, C.POINTER(vp), C.POINTER(u64)]
        L.np2_last_timings.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int)]
        L.np2_last_span.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
        L.np2_last_result_device.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.np2_result_fetch_begin.argtypes = [vp]
        L.np2_result_fetch_end.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.np2_phase_vote.argtypes = [vp, u32, vp, vp, vp, u64, vp, vp, u32, C.c_int, vp, C.POINTER(u32)]
        L.np2_ctx_create_shared.argtypes = [C.POINTER(vp), vp]
        L.np2_batch_create.argtypes = [C.POINTER(vp), vp, C.c_int]
        L.np2_batch_destroy.argtypes = [vp]
        L.np2_batch_destroy.restype = None
        L.np2_batch_slots.argtypes = [vp]
        L.np2_batch_set_sink.argtypes = [vp, C.c_int, vp, C.c_uint64]
        L.np2_batch_slot_ctx.argtypes = [vp, C.c_int]
        L.np2_batch_slot_ctx.restype = vp
        L.np2_batch_last_error.argtypes = [vp]
        L.np2_batch_last_error.restype = C.c_char_p
        L.np2_batch_polish.argtypes = [vp, vp, C.c_int, C.POINTER(np2_opts_t), vp, vp, vp, vp, vp]
        L.np2_batch_set_timing.argtypes = [vp, C.c_int]
        L.np2_batch_set_timing.restype = None
        L.np2_batch_set_priority.argtypes = [vp, C.c_int]
        L.np2_batch_set_priority.restype = C.c_int
        L.np2_batch_last_diff_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.np2_batch_last_call_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.np2_batch_flush_log.argtypes = [vp, C.POINTER(vp)]
        L.np2_shard_plan.argtypes = [vp, u32, u32, u32, u32, C.POINTER(np2_shard_plan_t)]
        L.np2_shard_upload.argtypes = [vp, vp, u32, vp, u32, vp, u64, C.POINTER(np2_shard_plan_t), C.POINTER(vp)]
        L.np2_shard_begin.argtypes = [vp, vp, C.POINTER(np2_shard_plan_t), C.POINTER(np2_opts_t), u32, C.POINTER(vp)]
        L.np2_shard_passes_left.argtypes = [vp]
        L.np2_shard_vote.argtypes = [vp, C.POINTER(np2_vote_t)]
        L.np2_vote_decide.argtypes = [C.POINTER(np2_vote_t), C.c_int, u32, C.POINTER(np2_opts_t), vp, C.POINTER(u32)]
        L.np2_shard_apply.argtypes = [vp, vp, u32]
        L.np2_shard_final.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
        L.np2_shard_final_device.argtypes = [vp, C.POINTER(np2_shard_piece_t)]
        L.np2_shard_fetch.argtypes = [vp, vp, vp]
        L.np2_alloc_pinned.argtypes = [u64]
        L.np2_alloc_pinned.restype = vp
        L.np2_trim_device_cache.argtypes = []
        L.np2_trim_device_cache.restype = None
        L.np2_shard_end.argtypes = [vp]
        L.np2_shard_end.restype = None
        L.np2_swiss_order.argtypes = [vp, vp, u32, vp, C.POINTER(u32)]
        L.np2_batch_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
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


class _Raw:
    """`n` elements at a raw address, for np.asarray (array-interface protocol: 3.5 us where np.ctypeslib.as_array on a cast
    pointer takes 33 — seventeen results per step of a batched assembly, on threads that share the interpreter lock)"""
    __slots__ = ("__array_interface__",)

    def __init__(self, addr, n, typestr):
        self.__array_interface__ = {"shape": (n,), "typestr": typestr, "data": (addr, False), "version": 3}


_TYPESTR = {C.c_uint8: "|u1", C.c_uint32: "<u4", C.c_uint64: "<u8", C.c_uint16: "<u2", C.c_int32: "<i4"}


def _owned(ptr, n, ctype):
    """Zero-copy numpy view of a callee-allocated result buffer; np2_free runs when the array is collected."""
    base = np.asarray(_Raw(ptr.value, max(n, 1), _TYPESTR[ctype]))
    weakref.finalize(base, lib().np2_free, C.c_void_p(ptr.value))
    return base[:n]


class ResidentContig:
    """A contig's packed pileup resident in HBM (np2_contig_t)."""

    def __init__(self, polisher, handle, pileup):
        self._p = polisher
        self._h = handle
        self.L = pileup.L
        self.n_reads = pileup.n_reads
        self.n_columns = pileup.n_columns()
        self.name = pileup.name

    def free(self):
        if self._h:
            lib().np2_contig_free(self._p._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Polisher:
    """One np2 context == one HIP device + HBM-resident yak tables.

    Mirrors the reference's per-worker state (one Opt clone with its KmerInfo per worker thread,
    src/main.rs:1724): contexts are independent, calls on one context are serialized."""

    def __init__(self, yaks, device=0):
        self._yaks = list(yaks)
        arr = yaks_array(self._yaks)
        h = C.c_void_p()
        rc = lib().np2_ctx_create(C.byref(h), device, arr, len(self._yaks))
        if rc != 0:
            raise Np2Error(rc, "np2_ctx_create failed (see stderr)")
        self._h = h
        self.device = device

    def clone(self):
        """np2_ctx_create_shared: a further context on the same device over the SAME HBM k-mer tables."""
        p = Polisher.__new__(Polisher)
        p._yaks = self._yaks
        p._parent = self
        p.device = self.device
        h = C.c_void_p()
        rc = lib().np2_ctx_create_shared(C.byref(h), self._h)
        if rc != 0:
            raise Np2Error(rc, "np2_ctx_create_shared failed (see stderr)")
        p._h = h
        return p

    def close(self):
        if getattr(self, "_h", None):
            lib().np2_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise Np2Error(rc, lib().np2_last_error(self._h).decode())

    def set_trace(self, on=True):
        lib().np2_ctx_set_trace(self._h, 1 if on else 0)

    def set_timing(self, on=True):
        """Arm every per-stage HIP-event timer (default: only the dense pass is timed)."""
        lib().np2_ctx_set_timing(self._h, 1 if on else 0)

    def upload(self, pileup: Pileup) -> ResidentContig:
        h = C.c_void_p()
        self._check(lib().np2_contig_upload(self._h, pileup.ref.ctypes.data, pileup.L, pileup.reads.ctypes.data,
                                            pileup.n_reads, pileup.nibbles.ctypes.data, pileup.nibbles.shape[0],
                                            C.byref(h)))
        return ResidentContig(self, h, pileup)

    def polish_resident(self, contig: ResidentContig, opts: Opts = None, want_pos=True, defer_output=False):
        """np2_polish_resident.  want_pos=False returns (bases, (first_pos, last_pos)) — all a FASTA record needs.
        defer_output=True leaves the sequence on the device and returns (None, span): fetch it with fetch_begin() /
        fetch_end() so that its host copy overlaps the next contig."""
        o = (opts or Opts()).c()
        ob, op, on = C.c_void_p(), C.c_void_p(), C.c_uint64()
        if defer_output:
            self._check(lib().np2_polish_resident(self._h, contig._h, C.byref(o), None, None, C.byref(on)))
            return None, self.last_span()
        self._check(lib().np2_polish_resident(self._h, contig._h, C.byref(o), C.byref(ob),
                                              C.byref(op) if want_pos else None, C.byref(on)))
        n = on.value
        if want_pos:
            return _owned(ob, n, C.c_uint8), _owned(op, n, C.c_uint32)
        return _owned(ob, n, C.c_uint8), self.last_span()

    def last_span(self):
        a, b = C.c_uint32(), C.c_uint32()
        self._check(lib().np2_last_span(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def fetch_begin(self):
        """np2_result_fetch_begin: start the host copy of the last (deferred) result on the output stream."""
        self._check(lib().np2_result_fetch_begin(self._h))

    def fetch_end(self):
        """np2_result_fetch_end: wait for the copy; zero-copy view of the context's pinned buffer (valid until the
        second-next fetch_begin)."""
        p, n = C.c_void_p(), C.c_uint64()
        self._check(lib().np2_result_fetch_end(self._h, C.byref(p), C.byref(n)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n.value, 1),))[: n.value]

    def last_result_device(self):
        """(device address, length) of the last polished sequence in HBM; valid until the next call on this context."""
        p, n = C.c_void_p(), C.c_uint64()
        self._check(lib().np2_last_result_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def polish(self, pileup: Pileup, opts: Opts = None):
        c = self.upload(pileup)
        try:
            return self.polish_resident(c, opts)
        finally:
            c.free()

    def trace(self, pass_idx, name):
        d, n = C.c_void_p(), C.c_uint64()
        if lib().np2_trace_get(self._h, pass_idx, name.encode(), C.byref(d), C.byref(n)) != 0:
            return None
        dt = np.dtype(TRACE_DTYPES[name])
        if n.value == 0:
            return np.zeros(0, dtype=dt)
        buf = (C.c_uint8 * n.value).from_address(d.value)
        return np.frombuffer(bytes(buf), dtype=dt)

    def timings(self):
        names, ms, n = C.c_void_p(), C.c_void_p(), C.c_int()
        lib().np2_last_timings(self._h, C.byref(names), C.byref(ms), C.byref(n))
        out = {}
        if n.value:
            vals = np.ctypeslib.as_array(C.cast(ms, C.POINTER(C.c_float)), shape=(n.value,))
            addr = names.value
            for i in range(n.value):  # NUL-separated names: walk them one C string at a time
                nm = C.string_at(addr)
                addr += len(nm) + 1
                out[nm.decode()] = float(vals[i])
        return out

    def score_strings(self, yak_idx, strings, min_kmer_count=5):
        off = np.zeros(len(strings) + 1, dtype=np.uint64)
        if strings:
            off[1:] = np.cumsum([len(s) for s in strings])
        blob = np.frombuffer(b"".join(strings) + b"\0", dtype=np.uint8)
        out = np.zeros(len(strings), dtype=np.uint16)
        self._check(lib().np2_score_strings(self._h, yak_idx, blob.ctypes.data, off.ctypes.data, len(strings),
                                            min_kmer_count, out.ctypes.data))
        return out

    def lookup_hashes(self, yak_idx, hashes, min_kmer_count=5):
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        out = np.zeros(h.shape[0], dtype=np.uint16)
        self._check(lib().np2_lookup_hashes(self._h, yak_idx, h.ctypes.data, h.shape[0], min_kmer_count,
                                            out.ctypes.data))
        return out


class BatchPolisher:
    """np2_batch_t: up to `n_slots` contigs polished at once with one launch per pipeline step for all of them.

    Mirrors the reference's pool of worker threads (src/main.rs:1717-1843); results per contig are exactly those of
    Polisher.polish_resident."""

    def __init__(self, polisher: Polisher, n_slots):
        self._pol = polisher  # keeps the parent context (and its yak tables) alive
        h = C.c_void_p()
        rc = lib().np2_batch_create(C.byref(h), polisher._h, n_slots)
        if rc != 0:
            raise Np2Error(rc, lib().np2_last_error(polisher._h).decode())
        self._h = h
        self.n_slots = n_slots

    def close(self):
        if getattr(self, "_h", None):
            lib().np2_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_priority(self, high):
        """np2_batch_set_priority: stream priority of this batch (alternate it over the batches of one device)."""
        rc = lib().np2_batch_set_priority(self._h, 1 if high else 0)
        if rc != 0:
            raise Np2Error(rc, "np2_batch_set_priority failed")

    def set_timing(self, on=True):
        lib().np2_batch_set_timing(self._h, 1 if on else 0)

    def last_diff_ms(self):
        ms, n = C.c_float(), C.c_int()
        lib().np2_batch_last_diff_ms(self._h, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def last_call_ms_inside(self):
        """(wall ms of the last np2_batch_polish measured inside the call, ms of it after its last flush)"""
        a, t = C.c_double(), C.c_double()
        lib().np2_batch_last_call_ms(self._h, C.byref(a), C.byref(t))
        return a.value, t.value

    def flush_log(self):
        """[(host ms, issue ms, device-wait ms)] per flush of the last polish call."""
        p = C.c_void_p()
        n = lib().np2_batch_flush_log(self._h, C.byref(p))
        if not n:
            return []
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(3 * n,)).copy()
        return [tuple(round(float(x), 3) for x in a[3 * i:3 * i + 3]) for i in range(n)]

    def flush_total_ms(self):
        """host + issue + wait over the flushes of the last polish call (one number: cheap enough for a timed loop)"""
        p = C.c_void_p()
        n = lib().np2_batch_flush_log(self._h, C.byref(p))
        return float(sum((C.c_double * (3 * n)).from_address(p.value))) if n else 0.0

    def stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().np2_batch_stats(self._h, C.byref(a), C.byref(b), C.byref(c))
        return {"launches": a.value, "commands": b.value, "flushes": c.value}

    def polish(self, contigs, opts: Opts = None, want_pos=False, keep_on_device=False):
        """[(bases, pos-or-span)] per contig, like Polisher.polish_resident(want_pos=...).  keep_on_device=True returns
        [(None, span)]: fetch the sequences of the last wave from the slot contexts (slot_fetch_begin / _end)."""
        n = len(contigs)
        o = (opts or Opts()).c()
        hs = (C.c_void_p * n)(*[c._h for c in contigs])
        key = (n, want_pos, keep_on_device)
        if getattr(self, "_arrs_key", None) != key:  # (the argument arrays of the last call shape are kept)
            self._arrs = ((C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_uint64 * n)(), (C.c_int * n)(), (C.c_uint32 * (2 * n))())
            self._arrs_key = key
        ob, op, on, rcs, span = self._arrs
        import time as _t
        _t0 = _t.perf_counter()
        rc = lib().np2_batch_polish(self._h, hs, n, C.byref(o), None if keep_on_device else ob,
                                    op if (want_pos and not keep_on_device) else None, on, span, rcs)
        self.last_call_ms = (_t.perf_counter() - _t0) * 1e3  # the C call alone (bench.py reports it next to the step time)
        if rc != 0:
            raise Np2Error(rc, lib().np2_batch_last_error(self._h).decode())
        out = []
        for i in range(n):
            if keep_on_device:
                out.append((None, (span[2 * i], span[2 * i + 1])))
                continue
            b = _owned(C.c_void_p(ob[i]), on[i], C.c_uint8)
            if want_pos:
                out.append((b, _owned(C.c_void_p(op[i]), on[i], C.c_uint32)))
            else:
                out.append((b, (span[2 * i], span[2 * i + 1])))
        return out

    def set_sink(self, slot, device_ptr, cap):
        """np2_batch_set_sink: the polished bases of the contig on `slot` are also copied to device address `device_ptr`."""
        rc = lib().np2_batch_set_sink(self._h, slot, C.c_void_p(device_ptr) if device_ptr else None, int(cap))
        if rc != 0:
            raise Np2Error(rc, "np2_batch_set_sink")

    def slot_fetch_begin(self, slot):
        rc = lib().np2_result_fetch_begin(lib().np2_batch_slot_ctx(self._h, slot))
        if rc != 0:
            raise Np2Error(rc, "np2_result_fetch_begin")

    def slot_fetch_end(self, slot):
        p, n = C.c_void_p(), C.c_uint64()
        rc = lib().np2_result_fetch_end(lib().np2_batch_slot_ctx(self._h, slot), C.byref(p), C.byref(n))
        if rc != 0:
            raise Np2Error(rc, "np2_result_fetch_end")
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n.value, 1),))[: n.value]


# ---- shards of one contig (include/np2.h np2_shard_*) -------------------------------------------------------------------
def shard_plan(pileup: Pileup, n_shards, halo=65536):
    """np2_shard_plan: cut a contig into n_shards reference intervals -> [np2_shard_plan_t]."""
    plans = (np2_shard_plan_t * n_shards)()
    rc = lib().np2_shard_plan(pileup.reads.ctypes.data, pileup.n_reads, pileup.L, n_shards, halo, plans)
    if rc != 0:
        raise Np2Error(rc, "np2_shard_plan: contig too short for that many shards?")
    return [plans[i] for i in range(n_shards)]


class Vote:
    """Host copy of one shard's np2_vote_t (numpy arrays), serialisable for the exchange between ranks."""

    FIELDS = (("pair_key", np.uint64), ("pair_cnt", np.uint32), ("read_id", np.uint32), ("first_pos", np.uint32),
              ("ref_w", np.int32), ("flags", np.uint8))

    def __init__(self, **arrays):
        for name, dt in self.FIELDS:
            setattr(self, name, np.ascontiguousarray(arrays.get(name, np.zeros(0, dtype=dt)), dtype=dt))

    @classmethod
    def from_c(cls, v: np2_vote_t, copy=True):
        """copy=False: views of the run's arrays (borrowed until the run's next call: a chromosome's pair list is hundreds
        of MB — packed or decided right away, it need not be copied first)."""
        def arr(ptr, n, dt):
            if not n:
                return np.zeros(0, dtype=dt)
            a = np.frombuffer((C.c_uint8 * (n * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt)
            return a.copy() if copy else a
        return cls(pair_key=arr(v.pair_key, v.n_pairs, np.uint64), pair_cnt=arr(v.pair_cnt, v.n_pairs, np.uint32),
                   read_id=arr(v.read_id, v.n_reads, np.uint32), first_pos=arr(v.first_pos, v.n_reads, np.uint32),
                   ref_w=arr(v.ref_w, v.n_reads, np.int32), flags=arr(v.flags, v.n_reads, np.uint8))

    def c(self):
        return np2_vote_t(len(self.pair_key), self.pair_key.ctypes.data, self.pair_cnt.ctypes.data, len(self.read_id),
                          self.read_id.ctypes.data, self.first_pos.ctypes.data, self.ref_w.ctypes.data, self.flags.ctypes.data)

    def pack(self):
        """One uint8 array: [n_pairs, n_reads] + the fields, each padded to 8 bytes (the exchange between ranks)."""
        parts = [np.array([len(self.pair_key), len(self.read_id)], dtype=np.uint64).view(np.uint8)]
        for name, _ in self.FIELDS:
            a = getattr(self, name).view(np.uint8)
            parts.append(a)
            if a.shape[0] & 7:
                parts.append(np.zeros(8 - (a.shape[0] & 7), dtype=np.uint8))
        return np.concatenate(parts)

    @classmethod
    def unpack(cls, raw):
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        n_pairs, n_reads = (int(x) for x in raw[:16].view(np.uint64))
        off, out = 16, {}
        for name, dt in cls.FIELDS:
            n = n_pairs if name.startswith("pair") else n_reads
            nb = n * np.dtype(dt).itemsize
            out[name] = raw[off:off + nb].view(dt)
            off += (nb + 7) & ~7
        return cls(**out)

    def to_bytes(self):
        """Unpadded byte-string form (the one the recorded shard fixtures under tests/golden hold)."""
        hdr = np.array([len(self.pair_key), len(self.read_id)], dtype=np.uint64).tobytes()
        return hdr + b"".join(getattr(self, n).tobytes() for n, _ in self.FIELDS)

    @classmethod
    def from_bytes(cls, raw):
        n_pairs, n_reads = (int(x) for x in np.frombuffer(raw[:16], dtype=np.uint64))
        off, out = 16, {}
        for name, dt in cls.FIELDS:
            n = n_pairs if name.startswith("pair") else n_reads
            nb = n * np.dtype(dt).itemsize
            out[name] = np.frombuffer(raw[off:off + nb], dtype=dt)
            off += nb
        return cls(**out)


def vote_decide(votes, n_reads_total, opts: Opts = None):
    """np2_vote_decide: merge the shards' votes of one phasing pass, Louvain on the merged read graph -> sorted losers
    (contig-wide read ids).  Host only, deterministic: every rank can run it on the same gathered votes."""
    o = (opts or Opts()).c()
    arr = (np2_vote_t * len(votes))(*[v.c() for v in votes])
    out = np.zeros(max(1, n_reads_total), dtype=np.uint32)
    n = C.c_uint32()
    rc = lib().np2_vote_decide(arr, len(votes), n_reads_total, C.byref(o), out.ctypes.data, C.byref(n))
    if rc != 0:
        raise Np2Error(rc, "np2_vote_decide")
    return out[: n.value].copy()


class ShardRun:
    """One shard of a contig on one context: upload (from a host pileup, or an already resident shard built by
    io.shard_from_bam), then vote() / apply() per phasing pass, final()."""

    def __init__(self, polisher: Polisher, pileup, plan: np2_shard_plan_t, opts: Opts = None, verify=1024, resident=None,
                 own_contig=True):
        self._pol = polisher
        self.plan = plan
        self.verify = verify
        self._own = own_contig
        self._c = C.c_void_p()
        if resident is not None:
            self._c = resident  # np2_contig_t* of the shard (ownership taken unless own_contig=False: see upload_shard)
        else:
            polisher._check(lib().np2_shard_upload(polisher._h, pileup.ref.ctypes.data, pileup.L, pileup.reads.ctypes.data,
                                                   pileup.n_reads, pileup.nibbles.ctypes.data, pileup.nibbles.shape[0],
                                                   C.byref(plan), C.byref(self._c)))
        self._o = (opts or Opts()).c()
        self._r = C.c_void_p()
        rc = lib().np2_shard_begin(polisher._h, self._c, C.byref(plan), C.byref(self._o), verify, C.byref(self._r))
        if rc != 0:
            self.close()
            polisher._check(rc)

    def passes_left(self):
        return lib().np2_shard_passes_left(self._r)

    def vote(self, copy=True) -> Vote:
        v = np2_vote_t()
        self._pol._check(lib().np2_shard_vote(self._r, C.byref(v)))
        return Vote.from_c(v, copy)

    def vote_view(self) -> Vote:
        """vote() without the copy: valid until this run's next call (the shard protocol packs or decides it at once)."""
        return self.vote(copy=False)

    def apply(self, losers):
        a = np.ascontiguousarray(losers, dtype=np.uint32)
        self._pol._check(lib().np2_shard_apply(self._r, a.ctypes.data, len(a)))

    def final(self):
        ob, op, on = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._pol._check(lib().np2_shard_final(self._r, C.byref(ob), C.byref(op), C.byref(on)))
        return _owned(ob, on.value, C.c_uint8), _owned(op, on.value, C.c_uint32)

    def final_device(self):
        """The final pass with the polished sub-contig left on the device -> ShardPiece (owned slice: device address +
        length; verification strips on the host)."""
        pc = np2_shard_piece_t()
        self._pol._check(lib().np2_shard_final_device(self._r, C.byref(pc)))
        return ShardPiece(pc)

    def fetch(self, dst_bases, dst_pos=None):
        """Copy the owned slice of the last final_device() into host arrays (slices of a pinned_array for long ones)."""
        self._pol._check(lib().np2_shard_fetch(self._r, dst_bases.ctypes.data, dst_pos.ctypes.data if dst_pos is not None else None))

    def close(self):
        if getattr(self, "_r", None):
            lib().np2_shard_end(self._r)
            self._r = None
        if getattr(self, "_c", None):
            if self._own:
                lib().np2_contig_free(self._pol._h, self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def upload_shard(polisher: Polisher, pileup: Pileup, plan: np2_shard_plan_t):
    """np2_shard_upload on its own: the resident shard (np2_contig_t*) of `plan`, to be polished any number of times by
    ShardRun(polisher, None, plan, ..., resident=h, own_contig=False); release it with free_shard."""
    h = C.c_void_p()
    polisher._check(lib().np2_shard_upload(polisher._h, pileup.ref.ctypes.data, pileup.L, pileup.reads.ctypes.data,
                                           pileup.n_reads, pileup.nibbles.ctypes.data, pileup.nibbles.shape[0],
                                           C.byref(plan), C.byref(h)))
    return h


def free_shard(polisher: Polisher, h):
    lib().np2_contig_free(polisher._h, h)


class ShardPiece:
    """Host view of np2_shard_piece_t: the owned slice stays on the device (dev_bases / dev_pos, own_len), the two strips
    around the cuts are numpy arrays (contig coordinates) freed with the object."""

    def __init__(self, pc: np2_shard_piece_t):
        self.own_len = int(pc.own_len)
        self.dev_bases, self.dev_pos = pc.dev_bases, pc.dev_pos
        self.first_pos, self.last_pos = int(pc.first_pos), int(pc.last_pos)

        def take(ptr, n, ct, dt):
            if not n or not ptr:
                return np.zeros(0, dtype=dt)
            return _owned(C.c_void_p(ptr), n, ct)
        self.lo_bases, self.lo_pos = take(pc.lo_bases, pc.lo_len, C.c_uint8, np.uint8), take(pc.lo_pos, pc.lo_len, C.c_uint32, np.uint32)
        self.hi_bases, self.hi_pos = take(pc.hi_bases, pc.hi_len, C.c_uint8, np.uint8), take(pc.hi_pos, pc.hi_len, C.c_uint32, np.uint32)

    def strips(self):
        """(lo_bases, lo_pos, hi_bases, hi_pos) as one uint8 array (exchange between ranks) — see unpack_strips."""
        hdr = np.array([self.own_len, self.first_pos, self.last_pos, len(self.lo_bases), len(self.hi_bases)], dtype=np.uint64)
        return np.concatenate([hdr.view(np.uint8), self.lo_pos.view(np.uint8), self.hi_pos.view(np.uint8), self.lo_bases, self.hi_bases])

    @staticmethod
    def unpack_strips(raw):
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        own_len, first, last, nl, nh = (int(x) for x in raw[:40].view(np.uint64))
        o = 40
        lo_pos = raw[o:o + 4 * nl].view(np.uint32); o += 4 * nl
        hi_pos = raw[o:o + 4 * nh].view(np.uint32); o += 4 * nh
        lo_b = raw[o:o + nl]; o += nl
        hi_b = raw[o:o + nh]
        return {"own_len": own_len, "first_pos": first, "last_pos": last, "lo_bases": lo_b, "lo_pos": lo_pos,
                "hi_bases": hi_b, "hi_pos": hi_pos}


def pinned_array(n, dtype=np.uint8):
    """A page-locked host array from the library's result pool (np2_alloc_pinned): large device-to-host copies into
    pageable memory stall on lazy pinning."""
    dt = np.dtype(dtype)
    p = lib().np2_alloc_pinned(max(1, int(n)) * dt.itemsize)
    if not p:  # (no device in this process, e.g. the CPU-only replay tests: plain memory does the same job, slower)
        return np.empty(int(n), dtype=dt)
    ct = {1: C.c_uint8, 4: C.c_uint32, 8: C.c_uint64}[dt.itemsize]
    return _owned(C.c_void_p(p), int(n), ct).view(dt)


def fasta_record(name, bases, pos):
    """display_consensusbase_vec (src/main.rs:607-645): '>{name} start:{first} end:{last}\\n{seq}\\n'."""
    return b">%s start:%d end:%d\n%s\n" % (name.encode() if isinstance(name, str) else name, int(pos[0]), int(pos[-1]),
                                           bytes(bases))


def swiss_order(script):
    """np2_swiss_order: iteration order of the product's SwissTable order model after [(op, key)] (0 insert, 1 remove,
    2 entry)."""
    ops = np.array([o for o, _ in script], dtype=np.uint32)
    keys = np.array([k for _, k in script], dtype=np.uint32)
    out = np.zeros(max(1, len(script)), dtype=np.uint32)
    n = C.c_uint32()
    if lib().np2_swiss_order(ops.ctypes.data, keys.ctypes.data, len(script), out.ctypes.data, C.byref(n)) != 0:
        raise Np2Error(-1, "np2_swiss_order")
    return out[: n.value].tolist()


def phase_vote(keys, pairs, ref=None):
    """Host-only phasing vote of the product library (np2_phase_vote): keys in creation order, pairs [(a, b, w)],
    ref = {read: weight} or None -> sorted losing reads."""
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    pa = np.array([p[0] for p in pairs], dtype=np.uint32)
    pb = np.array([p[1] for p in pairs], dtype=np.uint32)
    pw = np.array([p[2] for p in pairs], dtype=np.float32)
    ri = np.array(list(ref.keys()) if ref else [], dtype=np.uint32)
    rw = np.array(list(ref.values()) if ref else [], dtype=np.float32)
    out = np.zeros(max(1, len(keys)), dtype=np.uint32)
    n = C.c_uint32()
    rc = lib().np2_phase_vote(keys.ctypes.data, len(keys), pa.ctypes.data, pb.ctypes.data, pw.ctypes.data, len(pairs),
                              ri.ctypes.data, rw.ctypes.data, len(ri), 1 if ref is not None else 0, out.ctypes.data,
                              C.byref(n))
    if rc != 0:
        raise Np2Error(rc, "np2_phase_vote")
    return out[: n.value].tolist()
