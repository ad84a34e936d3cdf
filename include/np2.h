/*
 * np2.h — C ABI of the MI355X-native NextPolish2 consensus hot path.
 *
 * The reference (Nextomics/NextPolish2 v0.2.2) has no FFI: the hot path is the block
 * src/main.rs:1819-1836 (graph build -> best-path DP -> LQ candidates -> yak k-mer
 * scoring -> phasing vote -> splice), run per contig inside the worker closure
 * src/main.rs:1726-1837.  This header is the boundary a maintainer would bind from
 * Rust (`extern "C"`): the cut is *after* AlignSeq construction
 * (src/main.rs:1805-1817) and *before* output (src/main.rs:1837).
 *
 * All entry points return 0 on success and a negative NP2_E_* code otherwise; nothing
 * panics or aborts across the ABI (the reference aborts: Cargo.toml:26-27 panic='abort').
 * np2_last_error() returns a human-readable message for the last failure on a context.
 */
#ifndef NP2_H
#define NP2_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NP2_OK 0
#define NP2_E_ARG -1      /* bad argument / inconsistent packed read */
#define NP2_E_DEVICE -2   /* HIP runtime failure */
#define NP2_E_NOMEM -3
#define NP2_E_UNSUPPORTED -4 /* e.g. k >= 32, pre != 10 */
#define NP2_E_REFPANIC -5 /* input on which the reference itself would panic */

/* One packed alignment == the reference's AlignSeq (src/main.rs:272-338).
 * Nibble stream: 1 nibble per alignment column, high nibble first; low 3 bits = base
 * code (A0 C1 G2 T3 '-'4 N5 M6, src/utils/kmer.rs:11-22), bit 3 = insertion column
 * (target '-'); terminator nibble 0xF (src/main.rs:306-310).
 * reads[0] must be the contig aligned to itself ("order 0", src/main.rs:1732-1739). */
typedef struct np2_read {
    uint32_t aln_t_s;  /* first reference position (src/main.rs:273) */
    uint32_t aln_t_e;  /* INCLUSIVE last reference position (src/main.rs:274,295-297) */
    uint64_t nib_off;  /* byte offset of this read's nibble stream; must be a multiple of 16 */
    uint32_t n_cols;   /* number of alignment columns before the 0xF terminator */
    uint32_t flags;    /* bit0: dropped (align_bases == [], src/main.rs:571): index retained */
} np2_read_t;
#define NP2_READ_DROPPED 1u

/* One yak v2 table, file words verbatim (src/utils/kmer.rs:72-170):
 * bucket b in [0, 1<<pre) holds words[bucket_off[b] .. bucket_off[b+1]);
 * word = (hash >> 10) << 10 | count, hash & ((1<<pre)-1) == b.  Only pre == 10 is
 * meaningful to the reference's lookup (kmer.rs:52-54,123-125). */
typedef struct np2_yak {
    uint32_t k;
    uint32_t pre;
    uint64_t n_words;
    const uint64_t *words;
    const uint64_t *bucket_off; /* (1<<pre)+1 entries */
} np2_yak_t;

/* Options that reach the hot path (src/utils/option.rs:267-292). */
typedef struct np2_opts {
    uint16_t min_kmer_count; /* -k, default 5; words with count < this are ignored (kmer.rs:160) */
    int32_t max_indel_len;   /* -n, default 20 (main.rs:903) */
    uint32_t iter_count;     /* -i, default 2 (main.rs:1819-1836); must be >= 1 */
    uint8_t model_ref;       /* -m ref (1, default) | len (0) (main.rs:1547) */
    uint8_t use_all_reads;   /* -r (main.rs:948-1010) */
} np2_opts_t;

typedef struct np2_ctx np2_ctx_t;
typedef struct np2_contig np2_contig_t;

/* Create a context on HIP device `device`; builds the HBM-resident open-addressed k-mer
 * tables from `yaks` (ascending k, option.rs:238).  Replaces KmerInfo::new +
 * retrieve_kmers file re-streaming (kmer.rs:72-170). */
int np2_ctx_create(np2_ctx_t **out, int device, const np2_yak_t *yaks, int n_yak);
/* A further context on the device of `parent` that shares its HBM k-mer tables (reference-counted: the tables are freed
 * with the last context that uses them).  Mirrors the reference's worker threads, which clone the option set but read
 * the same yak files (main.rs:1724). */
int np2_ctx_create_shared(np2_ctx_t **out, np2_ctx_t *parent);
void np2_ctx_destroy(np2_ctx_t *ctx);
const char *np2_last_error(np2_ctx_t *ctx);
/* The HIP stream every kernel of this context is launched on (hipStream_t as void*). */
void *np2_ctx_stream(np2_ctx_t *ctx);

/* Upload one contig's packed pileup into HBM (borrowed host buffers, copied).
 * Replaces the in-memory Vec<AlignSeq> + tseq (main.rs:1732-1817). */
int np2_contig_upload(np2_ctx_t *ctx, const uint8_t *ref, uint32_t L, const np2_read_t *reads,
                      uint32_t n_reads, const uint8_t *nibbles, uint64_t nib_bytes,
                      np2_contig_t **out);
void np2_contig_free(np2_ctx_t *ctx, np2_contig_t *c);

/* The hot path on an HBM-resident contig: the loop main.rs:1819-1836.
 * Outputs (callee-allocated, release with np2_free): consensus bases (ASCII) and their
 * reference positions (ConsensusBase, main.rs:591-596).
 * Errors: results never depend on it, but WHICH error is reported can: without -r a pass starts on the reads the vote kernel
 * flagged while the previous vote is still being decided, so an error of that later pass (NP2_E_NOMEM, a reference panic in
 * its consensus) may be reported where the reference's sequential loop would have stopped at the vote's own panic
 * ("weight of two conflicting community").  NP2_NO_SPECULATE=1 restores the reference's order. */
int np2_polish_resident(np2_ctx_t *ctx, np2_contig_t *c, const np2_opts_t *opts,
                        uint8_t **out_bases, uint32_t **out_pos, uint64_t *out_len);

/* FASTA header span of the last successful polish on this context: pos of the first / last consensus
 * base (display_consensusbase_vec, main.rs:627-632).  `out_pos` of np2_polish_resident may be NULL when
 * only the sequence and this span are needed (FASTA output), which skips a 4 B/bp device-to-host copy. */
int np2_last_span(np2_ctx_t *ctx, uint32_t *first_pos, uint32_t *last_pos);

/* Device-resident copy of the last polished sequence (`len` ASCII bases in HBM on the context's device), valid until
 * the next call on this context.  Multi-GPU drivers hand it to RCCL directly (all-gather of the per-contig polished
 * sequences over xGMI) instead of sending the host copy back up. */
int np2_last_result_device(np2_ctx_t *ctx, const uint8_t **dev_bases, uint64_t *len);

/* Deferred output for back-to-back contigs: call np2_polish_resident with out_bases == NULL (the sequence then stays
 * on the device), then np2_result_fetch_begin: it snapshots the sequence and starts its device-to-host copy on a
 * stream of its own, so the copy overlaps the next contig's kernels instead of ending this one's.
 * np2_result_fetch_end waits for the copy and returns the host bytes: pinned memory owned by the context, valid
 * until the second-next np2_result_fetch_begin.  One fetch may be in flight per context. */
int np2_result_fetch_begin(np2_ctx_t *ctx);
int np2_result_fetch_end(np2_ctx_t *ctx, const uint8_t **bases, uint64_t *len);

/* Convenience: upload + polish + free (PCIe-inclusive). */
int np2_polish_contig(np2_ctx_t *ctx, const uint8_t *ref, uint32_t L, const np2_read_t *reads,
                      uint32_t n_reads, const uint8_t *nibbles, uint64_t nib_bytes,
                      const np2_opts_t *opts, uint8_t **out_bases, uint32_t **out_pos,
                      uint64_t *out_len);
void np2_free(void *p);

/* Batched min-count k-mer scoring of byte strings against yak table `yak_idx`:
 * scores[i] = min over k-mers of strs[off[i]..off[i+1]) of count (0 if none / absent /
 * below min_kmer_count).  Mirrors the scoring closure main.rs:761-769 / 1300-1315
 * (iter2kmer kmer.rs:255-287 + to_hash kmer.rs:102-110 + get kmer.rs:123-125). */
int np2_score_strings(np2_ctx_t *ctx, int yak_idx, const uint8_t *strs, const uint64_t *off,
                      uint64_t n, uint16_t min_kmer_count, uint16_t *scores);
/* Point lookups of already-hashed k-mers (kmer.rs:123-125). */
int np2_lookup_hashes(np2_ctx_t *ctx, int yak_idx, const uint64_t *hashes, uint64_t n,
                      uint16_t min_kmer_count, uint16_t *counts);

/* Stage-level exports for kernel parity tests and profiling.
 * After np2_polish_resident with tracing enabled, np2_trace_get returns a pointer to a
 * host copy of intermediate `name` of pass `pass` (valid until the next polish call). */
void np2_ctx_set_trace(np2_ctx_t *ctx, int enable);
int np2_trace_get(np2_ctx_t *ctx, int pass, const char *name, const void **data,
                  uint64_t *nbytes);

/* Host-side phasing vote (louvain.rs:290-356) on an explicit signed read graph — no device needed.
 * keys[n_keys]: graph nodes in the creation order of the reference's outer HashMap keys; pairs (pa[i], pb[i], pw[i])
 * are undirected weights; ref_ids/ref_w: the reference haplotype's row (ref_data[0]) or n_ref = 0 with has_ref = 0.
 * Writes the losing reads (sorted) to out_ids (capacity n_keys) and their number to n_out. */
int np2_phase_vote(const uint32_t *keys, uint32_t n_keys, const uint32_t *pa, const uint32_t *pb, const float *pw,
                   uint64_t n_pairs, const uint32_t *ref_ids, const float *ref_w, uint32_t n_ref, int has_ref,
                   uint32_t *out_ids, uint32_t *n_out);

/* ---- batch of contigs (the reference's N worker threads, main.rs:1717-1843, as ONE launch stream) -------------------
 * np2_batch_polish runs np2_polish_resident for up to `n_slots` resident contigs at a time: every contig keeps its own
 * pipeline, scratch context and host thread (so the host side of the phasing vote runs concurrently), but their
 * kernels are issued as one grid per pipeline step for the whole batch.  Results per contig are exactly those of
 * np2_polish_resident: out_bases[i] / out_pos[i] (either array may be NULL; release entries with np2_free),
 * out_len[i], out_span[2i], out_span[2i+1] (np2_last_span), rcs[i].  With out_bases == NULL the sequences stay on the device: slot context i % n_slots
 * (np2_batch_slot_ctx) then serves np2_last_span / np2_last_result_device / np2_result_fetch_begin for contig i of
 * the last wave.  Returns 0 or the code of the last failing contig (np2_batch_last_error). */
typedef struct np2_batch np2_batch_t;
int np2_batch_create(np2_batch_t **out, np2_ctx_t *parent, int n_slots);
void np2_batch_destroy(np2_batch_t *b);
/* Stream priority of this batch (high != 0: the device's greatest).  Several batches driven by one host thread each
 * overlap their host phases (the phasing vote) with each other's kernels; alternate their priorities so that they do
 * not advance in lockstep.  Call while no np2_batch_polish is in flight. */
int np2_batch_set_priority(np2_batch_t *b, int high);
int np2_batch_slots(np2_batch_t *b);
np2_ctx_t *np2_batch_slot_ctx(np2_batch_t *b, int slot);
const char *np2_batch_last_error(np2_batch_t *b);
/* A copy of the polished bases of whatever contig runs on `slot` (contig i of a wave: slot i % n_slots) goes, device to
 * device and at its polished length (at most `cap` bytes: compare out_len), to `device_ptr` — a slot of the buffer a rank's
 * all-gather sends (what replaces the reference's channel to the writer thread, main.rs:1838-1853, between GPUs) —, complete
 * when np2_batch_polish returns.  device_ptr == NULL: no copy.  Call while no np2_batch_polish is in flight. */
int np2_batch_set_sink(np2_batch_t *b, int slot, void *device_ptr, uint64_t cap);
int np2_batch_polish(np2_batch_t *b, np2_contig_t *const *contigs, int n, const np2_opts_t *opts, uint8_t **out_bases,
                     uint32_t **out_pos, uint64_t *out_len, uint32_t *out_span /* first / last position per contig, or NULL */,
                     int *rcs);
/* HIP-event timing of the batched dense kernel (k_diff_reads) on the batch's stream: total ms and launches of the last
 * np2_batch_polish (roofline bookkeeping of bench.py). */
void np2_batch_set_timing(np2_batch_t *b, int enable);
int np2_batch_last_diff_ms(np2_batch_t *b, float *ms, int *launches);
/* cumulative counters: kernel launches issued, commands recorded by the pipelines, device flushes */
/* per flush of the last np2_batch_polish: (host phase before it, command issue, device wait) in ms; returns the count */
int np2_batch_flush_log(np2_batch_t *b, const double **log);
/* wall time of the last np2_batch_polish measured inside the call (ms), and of it the part after the last flush
 * (results handed over, workers parked): what a caller's own clock adds on top is its language runtime's */
int np2_batch_last_call_ms(np2_batch_t *b, double *total_ms, double *tail_ms);
int np2_batch_stats(np2_batch_t *b, uint64_t *launches, uint64_t *commands, uint64_t *flushes);

/* ---- shards of one contig: reference-interval sharding of a long contig over several GPUs ------------------------------
 * The reference polishes a contig on one thread (main.rs:1726-1837).  Here a contig can be cut into reference intervals,
 * one per GPU: a shard holds every read overlapping its interval widened by `halo` (whole reads, global read numbering
 * kept), polishes that sub-contig like a contig of its own, votes only over the HETE regions it owns and emits only the
 * consensus bases of its interval.  Per phasing pass the shards' votes are merged and decided once per contig
 * (np2_vote_decide: the Louvain of louvain.rs:290-356 on the merged read graph), and the reads it removes are applied to
 * every shard.  Inside [own_lo - halo, own_hi + halo) a shard sees exactly the reads of the whole contig, so the stitched
 * result is the unsharded one; `verify` extra positions on either side are emitted too so that the stitcher can check
 * that neighbouring shards agree there (nextpolish2_amd.dist.stitch_shards). */
typedef struct np2_shard_plan {
    uint32_t own_lo, own_hi;   /* contig positions [own_lo, own_hi) this shard emits and votes over */
    uint32_t sub_lo, sub_hi;   /* sub-contig [sub_lo, sub_hi) it polishes (holds its reads entirely) */
    uint32_t zone_lo, zone_hi; /* [own_lo - halo, own_hi + halo) clipped to the contig: reads overlapping it are held */
    uint32_t read_lo, read_hi; /* reads [read_lo, read_hi) of the contig (read 0, the contig itself, is replaced by the
                                * sub-contig); local read i >= 1 is contig read read_lo + i - 1 */
} np2_shard_plan_t;
/* host only: cut a contig (read descriptors in alignment-start order, as the BAM gives them) into n_shards intervals */
int np2_shard_plan(const np2_read_t *reads, uint32_t n_reads, uint32_t L, uint32_t n_shards, uint32_t halo,
                   np2_shard_plan_t *out /* [n_shards] */);
/* upload the shard's part of a host pileup (same arguments as np2_contig_upload + the plan entry) */
int np2_shard_upload(np2_ctx_t *ctx, const uint8_t *ref, uint32_t L, const np2_read_t *reads, uint32_t n_reads,
                     const uint8_t *nibbles, uint64_t nib_bytes, const np2_shard_plan_t *plan, np2_contig_t **out);
/* what one shard contributes to a phasing pass (borrowed from the run until its next call); read ids are contig-wide */
typedef struct np2_vote {
    uint64_t n_pairs;
    const uint64_t *pair_key;  /* a << 32 | b, a < b */
    const uint32_t *pair_cnt;  /* owned HETE regions in which the pair agrees | disagrees << 16 (main.rs:982-1002) */
    uint32_t n_reads;
    const uint32_t *read_id;
    const uint32_t *first_pos; /* start of the rightmost owned HETE region the read votes in (0xFFFFFFFF: none) */
    const int32_t *ref_w;      /* summed weight against the contig's own candidate (ref_data[0], main.rs:972-976) */
    const uint8_t *flags;      /* 1: votes, 2: has a ref_data entry, 4: invalid (disagrees with the contig, main.rs:977) */
} np2_vote_t;
typedef struct np2_shard_run np2_shard_run_t;
int np2_shard_begin(np2_ctx_t *ctx, np2_contig_t *shard, const np2_shard_plan_t *plan, const np2_opts_t *opts,
                    uint32_t verify, np2_shard_run_t **out);
int np2_shard_passes_left(np2_shard_run_t *run);          /* iter_count - passes done; 1 = only the final pass is left */
/* a phasing pass up to its votes.  Without -r the shard starts its NEXT pass at once on the reads its vote kernel flagged
 * (kernels only: `out` stays valid) and np2_shard_apply settles it; between the two calls np2_shard_passes_left already counts
 * that pass as begun: fix the number of phasing passes before the loop (iter_count - 1 on every rank), as dist.py does */
int np2_shard_vote(np2_shard_run_t *run, np2_vote_t *out);
/* host only: merge the shards' votes of one pass and decide (reads of the whole contig: n_reads_total); losers: capacity
 * n_reads_total */
int np2_vote_decide(const np2_vote_t *votes, int n_votes, uint32_t n_reads_total, const np2_opts_t *opts, uint32_t *losers,
                    uint32_t *n_losers);
/* the decision's removed reads (contig-wide ids; any list: np2_vote_decide's, a replayed or a custom one).  If the pass
 * np2_shard_vote started early went by other reads — the list removes more than the vote kernel flagged, or KEEPS a flagged
 * read (np2_vote_decide never does: main.rs:977) — the difference is applied (reads removed / brought back) and the pass is
 * started again; the result is the one of the list either way */
int np2_shard_apply(np2_shard_run_t *run, const uint32_t *losers, uint32_t n_losers);
/* the final pass; bases / positions (contig coordinates) of [own_lo - verify, own_hi + verify); release with np2_free */
int np2_shard_final(np2_shard_run_t *run, uint8_t **out_bases, uint32_t **out_pos, uint64_t *out_len);
void np2_shard_end(np2_shard_run_t *run);

/* The same final pass with the polished sub-contig left ON THE DEVICE.  The consensus is ordered by position, so what a
 * shard owns — the bases emitted by positions [own_lo, own_hi) — is one contiguous slice of it: nothing is filtered, a
 * few searches find the slice, and the stitched contig is the shards' slices end to end.  Neighbours are checked against
 * each other on two short strips only (positions within `verify` of a cut, bases + contig positions, host memory from
 * the pinned pool: np2_free): the high strip of shard k must equal the low strip of shard k + 1 base for base.
 * np2_shard_fetch copies the owned slice to host memory the caller provides — for multi-megabase slices memory from
 * np2_alloc_pinned (any offset inside such a block: the shards of one process fetch straight into their places of ONE
 * buffer) — or the slice is gathered from dev_bases by a collective (dist.py: RCCL over xGMI).  dev_bases / dev_pos stay
 * valid until the next call on the run's context; dev_pos holds SUB-CONTIG positions (add plan.sub_lo). */
typedef struct np2_shard_piece {
    uint64_t own_len;            /* bases of the owned interval */
    const uint8_t *dev_bases;    /* device address of the first of them */
    const uint32_t *dev_pos;     /* ... and of its position (sub-contig coordinates) */
    uint32_t first_pos, last_pos; /* contig positions of the first / last owned base (own_len > 0) */
    uint32_t lo_len, hi_len;     /* strips: bases at positions [own_lo - verify, own_lo + verify) / [own_hi - verify, own_hi + verify) */
    uint8_t *lo_bases, *hi_bases;
    uint32_t *lo_pos, *hi_pos;   /* contig coordinates */
} np2_shard_piece_t;
int np2_shard_final_device(np2_shard_run_t *run, np2_shard_piece_t *out);
int np2_shard_fetch(np2_shard_run_t *run, uint8_t *dst_bases, uint32_t *dst_pos /* NULL: bases only; contig coordinates */);
void *np2_alloc_pinned(uint64_t bytes); /* page-locked host memory from the result pool; np2_free releases it */
/* The library keeps released device blocks (a twelfth of the device's memory at most, NP2_DEV_CACHE_GB) for its next
 * allocations instead of returning them to the driver; a caller about to allocate device memory by other means (torch,
 * RCCL buffers) hands the idle ones back with this.  Blocks in use are not touched. */
void np2_trim_device_cache(void);

/* Host-only test hook: key iteration order of the SwissTable order model behind np2_phase_vote after a script of
 * operations (0 insert, 1 remove, 2 entry().or_insert) — pinned by hand-traced vectors in tests/test_swiss_vectors.py. */
int np2_swiss_order(const uint32_t *ops, const uint32_t *keys, uint32_t n, uint32_t *out, uint32_t *n_out);

/* Per-stage device timings of the last np2_polish_resident (HIP events on the ctx stream).
 * names: NUL-separated list terminated by an empty string; ms[i] matches names[i].
 * By default only the dense pass ("diff_reads") is timed; np2_ctx_set_timing(ctx, 1) arms every stage timer
 * (each one costs two event packets on the stream). */
void np2_ctx_set_timing(np2_ctx_t *ctx, int enable);
int np2_last_timings(np2_ctx_t *ctx, const char **names, const float **ms, int *n);

#ifdef __cplusplus
}
#endif
#endif
