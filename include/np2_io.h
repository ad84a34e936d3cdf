/*
 * np2_io.h — input side of the NextPolish2 hot path.
 *
 * Replaces the reference's rust-htslib / kseq / KmerInfo::new call sites:
 *   FASTA[.gz] records            src/main.rs:1705-1714   (kseq: name = header up to first whitespace)
 *   indexed BAM fetch per contig  src/main.rs:1745-1758   (IndexedReader::from_path + fetch(tid, 0, len))
 *   record admission + packing    src/main.rs:1758-1817   (filters, fill_with_cigar, is_clip, trim(8),
 *                                                          AlignSeq::new, filter_alignseqs_by_clip)
 *   yak v2 dump header + buckets  src/utils/kmer.rs:72-170
 * BGZF inflate and BAM parsing run on the host pool (libdeflate / zlib) or — NP2_INFLATE=gpu, and by default when this
 * rank has fewer than twelve host CPUs to itself, or for a reference with 128 MB of BAM and more — ON THE DEVICE: the contig's BGZF blocks are uploaded as they lie in the
 * file, inflated one wavefront per block (csrc/np2_inflate.hip), the records found by walking the inflated stream along
 * the .bai linear index, and the SEQ bytes read by the columnariser where the inflater left them.  The CIGAR walk /
 * trim(8) / nibble packing is a HIP kernel that writes the packed pileup straight into HBM (np2_contig_t) either way.
 */
#ifndef NP2_IO_H
#define NP2_IO_H
#include "np2.h"
#ifdef __cplusplus
extern "C" {
#endif

/* read-admission options (src/utils/option.rs:267-292) */
typedef struct np2_front_opts {
    uint32_t min_read_len;     /* -l 1000 */
    uint32_t min_map_len;      /* -a integer part, 500 */
    float min_map_fra;         /* -a fractional part, 0.5 */
    int16_t min_map_qual;      /* -q 1 */
    uint32_t max_clip_len;     /* -c 100 */
    uint8_t use_supplementary; /* -s */
    uint8_t use_secondary;     /* -S: np2_contig_from_bam recovers the SEQ of secondary records from the primary record
                                * of the same read (secondary.rs; two passes over the BAM on first use);
                                * np2_contig_from_records expects the caller to pass the recovered SEQ */
} np2_front_opts_t;

/* one alignment record, fields as in the BAM record (what rust-htslib's Record exposes, main.rs:1751-1797) */
typedef struct np2_bamrec {
    int32_t pos;        /* 0-based leftmost position */
    uint16_t flag;
    uint8_t mapq;
    uint8_t pad;
    uint32_t n_cigar;
    uint64_t cigar_off; /* index into cigar[] (u32 each: len << 4 | op, BAM encoding) */
    uint32_t l_seq;
    uint64_t seq_off;   /* byte offset into seq4[] (4-bit packed, BAM encoding, high nibble first) */
} np2_bamrec_t;

/* ---- FASTA[.gz] ---- */
typedef struct np2_fasta np2_fasta_t;
int np2_fasta_open(const char *path, np2_fasta_t **out);
/* returns 1 and borrows name/seq until the next call, 0 at end of file, <0 on error */
int np2_fasta_next(np2_fasta_t *f, const char **name, const uint8_t **seq, uint64_t *len);
void np2_fasta_close(np2_fasta_t *f);

/* ---- yak v2 dump ---- */
/* fills *out with malloc'ed arrays (release with np2_yak_free) */
int np2_yak_load(const char *path, np2_yak_t *out);
void np2_yak_free(np2_yak_t *y);
/* np2_ctx_create for dumps on disk: the files go to the device in pieces as they are read and the HBM tables are built
 * from the file images, without the host arrays of np2_yak_load (KmerInfo::new + the per-phase file reads of
 * kmer.rs:72-170 collapsed into one load).  Tables are ordered by k (option.rs:238).  Errors: np2_io_last_error(). */
int np2_ctx_create_from_files(np2_ctx_t **out, int device, const char *const *paths, int n_paths);

/* ---- indexed BAM ---- */
typedef struct np2_bam np2_bam_t;
/* needs <path>.bai (or <stem>.bai).  A handle keeps device-side staging of the FIRST context it is used with (stream, device):
 * use one handle with contexts of one device only, one thread at a time (nextpolish2_amd.cli: a handle per front-end thread) */
int np2_bam_open(const char *path, np2_bam_t **out);
void np2_bam_close(np2_bam_t *b);
int np2_bam_n_refs(np2_bam_t *b);
const char *np2_bam_ref_name(np2_bam_t *b, int tid, uint32_t *len);
const char *np2_io_last_error(void);

/* Packed pileup of one contig, resident in HBM, built from its alignment records.
 * ref = contig bytes exactly as in the FASTA (case matters for trim, main.rs:447-513). */
int np2_contig_from_records(np2_ctx_t *ctx, const uint8_t *ref, uint32_t L, const np2_bamrec_t *recs,
                            uint32_t n_recs, const uint32_t *cigar, const uint8_t *seq4,
                            const np2_front_opts_t *opts, np2_contig_t **out);
/* same, reading the records of contig `name` from an indexed BAM (must be coordinate sorted) */
int np2_contig_from_bam(np2_ctx_t *ctx, np2_bam_t *bam, const char *name, const uint8_t *ref, uint32_t L,
                        const np2_front_opts_t *opts, np2_contig_t **out);
/* ---- one reference interval of a contig straight from the BAM (multi-GPU: every rank parses only its part) -----------
 * begin: fetch the records overlapping [own_lo - halo, own_hi + halo) through the .bai linear index, admit + columnarise
 *        them; returns the BGZF virtual offsets of the pushed records that START in [own_lo, own_hi) (file order).
 *        The ranks' own intervals must tile the contig (np2_shard_plan's cuts: L * k / n rounded down to 1024).
 * exchange (caller): all-gather those lists in rank order -> the contig's pushed records in file order.
 * finish: numbers the shard's reads contig-wide (1 + place in that list; the reference numbers alignseqs in file order,
 *        main.rs:1813), applies the clip filter in contig coordinates, returns the shard's resident pileup, its plan for
 *        np2_shard_begin and the contig's read count for np2_vote_decide.  Consumes `io` (also on error). */
typedef struct np2_shard_io np2_shard_io_t;
int np2_shard_bam_begin(np2_ctx_t *ctx, np2_bam_t *bam, const char *name, const uint8_t *ref, uint32_t L, uint32_t own_lo,
                        uint32_t own_hi, uint32_t halo, const np2_front_opts_t *opts, np2_shard_io_t **io,
                        const uint64_t **own_voffsets, uint64_t *n_own);
int np2_shard_bam_finish(np2_shard_io_t *io, const uint64_t *all_voffsets, uint64_t n_all, np2_shard_plan_t *plan,
                         np2_contig_t **contig, uint32_t *n_reads_total);
void np2_shard_bam_abort(np2_shard_io_t *io);
/* A run of whole BGZF blocks (`bgzf`, `n` bytes: a BAM file or a piece of one that starts at a block) inflated on ctx's
 * device by the kernel np2_contig_from_bam uses, the bytes copied back to `out` (capacity out_cap; *out_len = what the
 * blocks hold).  kernel_ms (optional): the inflate kernel alone (HIP events).  For parity tests against zlib and for
 * measuring the kernel; the reference's counterpart is rust-htslib's bgzf reader (main.rs:1745-1757). */
int np2_bgzf_inflate_device(np2_ctx_t *ctx, const uint8_t *bgzf, uint64_t n, uint8_t *out, uint64_t out_cap, uint64_t *out_len,
                            float *kernel_ms);
/* copy a resident packed pileup back to the host (parity tests / debugging); free both with np2_free */
int np2_contig_export(np2_ctx_t *ctx, np2_contig_t *c, np2_read_t **reads, uint32_t *n_reads,
                      uint8_t **nibbles, uint64_t *nib_bytes);

#ifdef __cplusplus
}
#endif
#endif
