"""ctypes binding of include/np2_io.h: FASTA[.gz] / yak / indexed BAM readers and the GPU columnariser."""
import ctypes as C

import numpy as np

from ._types import READ_DTYPE, Pileup, Yak, np2_read_t, np2_yak_t
from .api import Np2Error, ResidentContig, lib

BAMREC_DTYPE = np.dtype([("pos", "<i4"), ("flag", "<u2"), ("mapq", "u1"), ("pad", "u1"), ("n_cigar", "<u4"),
                         ("pad2", "<u4"), ("cigar_off", "<u8"), ("l_seq", "<u4"), ("pad3", "<u4"), ("seq_off", "<u8")])
assert BAMREC_DTYPE.itemsize == 40

IO_SYMBOLS = ["np2_fasta_open", "np2_fasta_next", "np2_fasta_close", "np2_yak_load", "np2_yak_free", "np2_bam_open",
              "np2_bam_close", "np2_bam_n_refs", "np2_bam_ref_name", "np2_io_last_error", "np2_contig_from_records",
              "np2_contig_from_bam", "np2_contig_export", "np2_ctx_create_from_files", "np2_bgzf_inflate_device"]


class np2_front_opts_t(C.Structure):
    _fields_ = [("min_read_len", C.c_uint32), ("min_map_len", C.c_uint32), ("min_map_fra", C.c_float),
                ("min_map_qual", C.c_int16), ("max_clip_len", C.c_uint32), ("use_supplementary", C.c_uint8),
                ("use_secondary", C.c_uint8)]


class FrontOpts:
    """Read-admission options with the reference defaults (src/utils/option.rs:267-292; -a 500.5 splits into
    min_map_len = 500 and min_map_fra = 0.5, option.rs:232,258-259)."""

    def __init__(self, min_read_len=1000, min_map_len=500, min_map_fra=0.5, min_map_qual=1, max_clip_len=100,
                 use_supplementary=False, use_secondary=False):
        self.min_read_len, self.min_map_len, self.min_map_fra = min_read_len, min_map_len, min_map_fra
        self.min_map_qual, self.max_clip_len = min_map_qual, max_clip_len
        self.use_supplementary, self.use_secondary = use_supplementary, use_secondary

    def c(self):
        return np2_front_opts_t(self.min_read_len, self.min_map_len, self.min_map_fra, self.min_map_qual,
                                self.max_clip_len, 1 if self.use_supplementary else 0, 1 if self.use_secondary else 0)


_BOUND = False


_BIND_LOCK = __import__("threading").Lock()


def _bind():
    """The library with the argument types of include/np2_io.h declared.  Under a lock: the command line's stages call in
    from several threads at start-up, and a function called while another thread is still declaring its argtypes gets
    its pointers truncated to C ints."""
    global _BOUND
    L = lib()
    if _BOUND:
        return L
    with _BIND_LOCK:
        if _BOUND:
            return L
        _bind_locked(L)
        _BOUND = True
    return L


def _bind_locked(L):
    if True:
        vp = C.c_void_p
        L.np2_fasta_open.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.np2_fasta_next.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.np2_fasta_close.argtypes = [vp]
        L.np2_yak_load.argtypes = [C.c_char_p, C.POINTER(np2_yak_t)]
        L.np2_yak_free.argtypes = [C.POINTER(np2_yak_t)]
        L.np2_ctx_create_from_files.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_char_p), C.c_int]
        L.np2_bam_open.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.np2_bam_close.argtypes = [vp]
        L.np2_bam_n_refs.argtypes = [vp]
        L.np2_bam_ref_name.restype = C.c_char_p
        L.np2_bam_ref_name.argtypes = [vp, C.c_int, C.POINTER(C.c_uint32)]
        L.np2_io_last_error.restype = C.c_char_p
        L.np2_contig_from_records.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, vp, vp, C.POINTER(np2_front_opts_t),
                                              C.POINTER(vp)]
        L.np2_contig_from_bam.argtypes = [vp, vp, C.c_char_p, vp, C.c_uint32, C.POINTER(np2_front_opts_t), C.POINTER(vp)]
        L.np2_contig_export.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.np2_bgzf_inflate_device.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
        _bind_shard(L)


def _bind_shard(L):
    from ._types import np2_shard_plan_t
    vp = C.c_void_p
    L.np2_shard_bam_begin.argtypes = [vp, vp, C.c_char_p, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.np2_shard_bam_finish.argtypes = [vp, vp, C.c_uint64, C.POINTER(np2_shard_plan_t), C.POINTER(vp), C.POINTER(C.c_uint32)]
    L.np2_shard_bam_abort.argtypes = [vp]
    L.np2_shard_bam_abort.restype = None


def _io_check(rc):
    if rc != 0:
        raise Np2Error(rc, _bind().np2_io_last_error().decode())


def read_fasta(path):
    """Yield (name, sequence bytes) like kseq (src/main.rs:1705-1714): name = header up to the first whitespace."""
    L = _bind()
    h = C.c_void_p()
    _io_check(L.np2_fasta_open(path.encode(), C.byref(h)))
    try:
        name, seq, n = C.c_char_p(), C.c_void_p(), C.c_uint64()
        while True:
            rc = L.np2_fasta_next(h, C.byref(name), C.byref(seq), C.byref(n))
            if rc == 0:
                break
            if rc < 0:
                _io_check(rc)
            yield name.value.decode(), C.string_at(seq.value, n.value) if n.value else b""
    finally:
        L.np2_fasta_close(h)


def load_yak(path):
    """yak v2 dump -> Yak (src/utils/kmer.rs:72-170)."""
    import weakref
    L = _bind()
    y = np2_yak_t()
    _io_check(L.np2_yak_load(path.encode(), C.byref(y)))
    nb = (1 << y.pre) + 1
    off = np.ctypeslib.as_array(y.bucket_off, shape=(nb,)).copy()
    # the words are used where the loader put them (no second copy of a dump of tens of GB): np2_yak_free runs when the
    # array is collected
    base = np.ctypeslib.as_array(y.words, shape=(max(int(y.n_words), 1),))
    yk = Yak(y.k, base[: int(y.n_words)], off, pre=y.pre)
    # (tied to the Yak object, not to `base`: a view's .base is the memory's owner, not the intermediate array, so `base`
    # itself may be collected while its views live on)
    weakref.finalize(yk, L.np2_yak_free, y)
    return yk


def check_yak_header(path):
    """The cheap part of a dump's validation (kmer.rs:73-90: magic, counter bits; what this implementation supports:
    k < 32, pre == 10) without reading its words: the command line runs it on every dump BEFORE it opens its output, like
    the reference, which loads its yak files before the first contig.  Returns k; raises ValueError with the message."""
    import struct
    with open(path, "rb") as f:
        hd = f.read(16)
    if len(hd) != 16 or hd[:4] != b"YAK\x02":
        raise ValueError("The input binary k-mer dump file is incompatible.")
    k, pre, cbits = struct.unpack("<III", hd[4:])
    if cbits != 10:
        raise ValueError("different YAK_COUNTER_BITS")
    if k >= 32 or k < 2 or pre != 10:
        raise ValueError(f"{path}: k = {k}, prefix bits = {pre}: only k < 32 with the default 10 prefix bits is supported")
    return k


def polisher_from_yak_files(paths, device=0):
    """np2_ctx_create_from_files: a Polisher whose HBM k-mer tables are built from the dumps as they are read (no host
    copy of the words; tables ordered by k, option.rs:238)."""
    from .api import Polisher
    L = _bind()
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    h = C.c_void_p()
    _io_check(L.np2_ctx_create_from_files(C.byref(h), device, arr, len(paths)))
    p = Polisher.__new__(Polisher)
    p._yaks = []
    p._h = h
    p.device = device
    return p


def write_yak(path, yak):
    """Write a Yak as a yak v2 dump ("YAK\\2", k, pre, counter_bits = 10, then per bucket: u32, u32 n, n x u64)."""
    import struct
    with open(path, "wb") as f:
        f.write(b"YAK\x02" + struct.pack("<III", yak.k, yak.pre, 10))
        for b in range(1 << yak.pre):
            s, e = int(yak.bucket_off[b]), int(yak.bucket_off[b + 1])
            f.write(struct.pack("<II", 0, e - s))
            f.write(yak.words[s:e].tobytes())


class Bam:
    def __init__(self, path):
        L = _bind()
        self._h = C.c_void_p()
        _io_check(L.np2_bam_open(path.encode(), C.byref(self._h)))

    def refs(self):
        L = _bind()
        out = []
        for i in range(L.np2_bam_n_refs(self._h)):
            n = C.c_uint32()
            out.append((L.np2_bam_ref_name(self._h, i, C.byref(n)).decode(), n.value))
        return out

    def close(self):
        if self._h:
            _bind().np2_bam_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _resident(pol, h, name, L_, n_reads=0, n_cols=0):
    rc = ResidentContig.__new__(ResidentContig)
    rc._p, rc._h, rc.L, rc.n_reads, rc.n_columns, rc.name = pol, h, L_, n_reads, n_cols, name
    return rc


def contig_from_bam(pol, bam, name, ref, opts=None):
    """BAM records of contig `name` -> packed pileup resident in HBM (GPU columnariser)."""
    L = _bind()
    ref = np.frombuffer(ref, dtype=np.uint8) if isinstance(ref, (bytes, bytearray)) else np.ascontiguousarray(ref, dtype=np.uint8)
    o = (opts or FrontOpts()).c()
    h = C.c_void_p()
    pol._check(L.np2_contig_from_bam(pol._h, bam._h, name.encode(), ref.ctypes.data, ref.shape[0], C.byref(o), C.byref(h)))
    return _resident(pol, h, name, int(ref.shape[0]))


def contig_from_records(pol, ref, recs, cigar, seq4, opts=None, name="ctg"):
    L = _bind()
    ref = np.frombuffer(ref, dtype=np.uint8) if isinstance(ref, (bytes, bytearray)) else np.ascontiguousarray(ref, dtype=np.uint8)
    recs = np.ascontiguousarray(recs, dtype=BAMREC_DTYPE)
    cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
    seq4 = np.ascontiguousarray(seq4, dtype=np.uint8)
    o = (opts or FrontOpts()).c()
    h = C.c_void_p()
    pol._check(L.np2_contig_from_records(pol._h, ref.ctypes.data, ref.shape[0], recs.ctypes.data, recs.shape[0],
                                         cigar.ctypes.data, seq4.ctypes.data, C.byref(o), C.byref(h)))
    return _resident(pol, h, name, int(ref.shape[0]))


def shard_cuts(L_, n_shards):
    """The owned intervals np2_shard_plan gives a contig of L_ positions: cuts at L * k / n rounded down to 1024."""
    cuts = [0] + [((L_ * k) // n_shards) & ~1023 for k in range(1, n_shards)] + [L_]
    return list(zip(cuts[:-1], cuts[1:]))


class ShardFromBam:
    """One reference interval of a contig read straight from the BAM (np2_shard_bam_*): begin() fetches, admits and
    columnarises the records overlapping the interval +- halo and returns the file offsets this rank contributes to
    the numbering exchange; finish(all_offsets) returns (np2_contig_t* of the shard, its plan, the contig's read count)."""

    def __init__(self, pol, bam, name, ref, own_lo, own_hi, halo=65536, opts=None):
        from ._types import np2_shard_plan_t
        L = _bind()
        self._pol = pol
        ref = np.frombuffer(ref, dtype=np.uint8) if isinstance(ref, (bytes, bytearray)) else np.ascontiguousarray(ref, dtype=np.uint8)
        self._ref = ref
        o = (opts or FrontOpts()).c()
        self._io = C.c_void_p()
        pv, n = C.c_void_p(), C.c_uint64()
        pol._check(L.np2_shard_bam_begin(pol._h, bam._h, name.encode(), ref.ctypes.data, ref.shape[0], own_lo, own_hi, halo,
                                         C.byref(o), C.byref(self._io), C.byref(pv), C.byref(n)))
        self.own_offsets = (np.frombuffer((C.c_uint8 * (8 * n.value)).from_address(pv.value), dtype=np.uint64).copy()
                            if n.value else np.zeros(0, dtype=np.uint64))
        self._plan_t = np2_shard_plan_t

    def finish(self, all_offsets):
        L = _bind()
        a = np.ascontiguousarray(all_offsets, dtype=np.uint64)
        plan = self._plan_t()
        h = C.c_void_p()
        n_total = C.c_uint32()
        io, self._io = self._io, None
        self._pol._check(L.np2_shard_bam_finish(io, a.ctypes.data, len(a), C.byref(plan), C.byref(h), C.byref(n_total)))
        return h, plan, n_total.value

    def abort(self):
        """Give the half-built shard up (another rank's shard failed: the contig is polished unsharded)."""
        if getattr(self, "_io", None):
            _bind().np2_shard_bam_abort(self._io)
            self._io = None

    def __del__(self):
        try:
            self.abort()
        except Exception:
            pass


def export_contig(pol, contig, ref):
    """Copy a resident packed pileup back to the host as a Pileup (parity tests)."""
    L = _bind()
    pr, pn, nr, nb = C.c_void_p(), C.c_void_p(), C.c_uint32(), C.c_uint64()
    pol._check(L.np2_contig_export(pol._h, contig._h, C.byref(pr), C.byref(nr), C.byref(pn), C.byref(nb)))
    reads = np.frombuffer(C.string_at(pr.value, nr.value * C.sizeof(np2_read_t)), dtype=READ_DTYPE).copy()
    nib = np.frombuffer(C.string_at(pn.value, nb.value), dtype=np.uint8).copy()
    L.np2_free(pr)
    L.np2_free(pn)
    return Pileup(ref, reads, nib)


def bgzf_inflate_device(pol, data):
    """np2_bgzf_inflate_device: a run of whole BGZF blocks (bytes / uint8 array) inflated on the polisher's device by the
    kernel np2_contig_from_bam uses -> (inflated bytes as a uint8 array, kernel milliseconds)."""
    L = _bind()
    src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else data, dtype=np.uint8)
    # ISIZE of every block bounds the output: 64 KiB a block, a block is at least 28 bytes
    cap = max(1, (len(src) // 28 + 1)) * 65536
    cap = min(cap, max(1 << 20, len(src) * 1100))  # (deflate expands at most ~1032 x)
    out = np.empty(cap, dtype=np.uint8)
    n, ms = C.c_uint64(), C.c_float()
    rc = L.np2_bgzf_inflate_device(pol._h, src.ctypes.data, len(src), out.ctypes.data, cap, C.byref(n), C.byref(ms))
    if rc != 0:
        raise Np2Error(rc, "np2_bgzf_inflate_device: " + (L.np2_io_last_error() or b"").decode())
    return out[: n.value].copy(), float(ms.value)
