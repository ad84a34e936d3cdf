"""ctypes mirrors of the structs in include/np2.h (the C-ABI boundary of the hot path)."""
import ctypes as C

import numpy as np


class np2_read_t(C.Structure):
    # include/np2.h: np2_read_t  == reference AlignSeq (src/main.rs:272-276)
    _fields_ = [
        ("aln_t_s", C.c_uint32),
        ("aln_t_e", C.c_uint32),
        ("nib_off", C.c_uint64),
        ("n_cols", C.c_uint32),
        ("flags", C.c_uint32),
    ]


READ_DTYPE = np.dtype(
    [("aln_t_s", "<u4"), ("aln_t_e", "<u4"), ("nib_off", "<u8"), ("n_cols", "<u4"), ("flags", "<u4")]
)
assert READ_DTYPE.itemsize == C.sizeof(np2_read_t) == 24


class np2_yak_t(C.Structure):
    # include/np2.h: np2_yak_t == yak v2 dump words (src/utils/kmer.rs:72-170)
    _fields_ = [
        ("k", C.c_uint32),
        ("pre", C.c_uint32),
        ("n_words", C.c_uint64),
        ("words", C.POINTER(C.c_uint64)),
        ("bucket_off", C.POINTER(C.c_uint64)),
    ]


class np2_opts_t(C.Structure):
    # include/np2.h: np2_opts_t == the Option fields used on the path (src/utils/option.rs:267-292)
    _fields_ = [
        ("min_kmer_count", C.c_uint16),
        ("max_indel_len", C.c_int32),
        ("iter_count", C.c_uint32),
        ("model_ref", C.c_uint8),
        ("use_all_reads", C.c_uint8),
    ]


class np2_shard_plan_t(C.Structure):
    # include/np2.h: one reference interval of a contig cut over several GPUs
    _fields_ = [("own_lo", C.c_uint32), ("own_hi", C.c_uint32), ("sub_lo", C.c_uint32), ("sub_hi", C.c_uint32),
                ("zone_lo", C.c_uint32), ("zone_hi", C.c_uint32), ("read_lo", C.c_uint32), ("read_hi", C.c_uint32)]


class np2_vote_t(C.Structure):
    # include/np2.h: what one shard contributes to a phasing pass
    _fields_ = [("n_pairs", C.c_uint64), ("pair_key", C.c_void_p), ("pair_cnt", C.c_void_p), ("n_reads", C.c_uint32),
                ("read_id", C.c_void_p), ("first_pos", C.c_void_p), ("ref_w", C.c_void_p), ("flags", C.c_void_p)]


class np2_shard_piece_t(C.Structure):
    # include/np2.h: the device-resident result of one shard's final pass (owned slice + the two verification strips)
    _fields_ = [("own_len", C.c_uint64), ("dev_bases", C.c_void_p), ("dev_pos", C.c_void_p), ("first_pos", C.c_uint32),
                ("last_pos", C.c_uint32), ("lo_len", C.c_uint32), ("hi_len", C.c_uint32), ("lo_bases", C.c_void_p),
                ("hi_bases", C.c_void_p), ("lo_pos", C.c_void_p), ("hi_pos", C.c_void_p)]


class Opts:
    """Defaults of the reference CLI (src/utils/option.rs:267-292)."""

    def __init__(self, min_kmer_count=5, max_indel_len=20, iter_count=2, model="ref", use_all_reads=False):
        self.min_kmer_count = min_kmer_count
        self.max_indel_len = max_indel_len
        self.iter_count = iter_count
        self.model = model
        self.use_all_reads = use_all_reads

    def c(self):
        # main.rs:1547 compares opt.model == "ref" case-sensitively
        return np2_opts_t(
            self.min_kmer_count, self.max_indel_len, self.iter_count, 1 if self.model == "ref" else 0,
            1 if self.use_all_reads else 0,
        )


class Yak:
    """One yak table in boundary form: file words grouped by bucket (pre = 10)."""

    def __init__(self, k, words, bucket_off, pre=10):
        self.k = int(k)
        self.pre = int(pre)
        self.words = np.ascontiguousarray(words, dtype=np.uint64)
        self.bucket_off = np.ascontiguousarray(bucket_off, dtype=np.uint64)
        assert self.bucket_off.shape[0] == (1 << self.pre) + 1

    def c(self):
        return np2_yak_t(
            self.k,
            self.pre,
            self.words.shape[0],
            self.words.ctypes.data_as(C.POINTER(C.c_uint64)),
            self.bucket_off.ctypes.data_as(C.POINTER(C.c_uint64)),
        )


class Pileup:
    """One contig's packed pileup: ref bytes + np2_read_t[] + nibble buffer (reads[0] = the contig)."""

    def __init__(self, ref, reads, nibbles, name="ctg"):
        self.ref = np.frombuffer(ref, dtype=np.uint8) if isinstance(ref, (bytes, bytearray)) else np.ascontiguousarray(ref, dtype=np.uint8)
        self.reads = np.ascontiguousarray(reads, dtype=READ_DTYPE)
        self.nibbles = np.ascontiguousarray(nibbles, dtype=np.uint8)
        self.name = name

    @property
    def L(self):
        return int(self.ref.shape[0])

    @property
    def n_reads(self):
        return int(self.reads.shape[0])

    def n_columns(self):
        return int(self.reads["n_cols"].astype(np.int64).sum())


def yaks_array(yaks):
    arr = (np2_yak_t * len(yaks))(*[y.c() for y in yaks])
    return arr
