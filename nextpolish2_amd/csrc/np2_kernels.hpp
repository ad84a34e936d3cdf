// Launcher declarations for np2_kernels.hip / np2_prims.hip (host-callable, all asynchronous on `s`).
#pragma once
#include <cstddef>
#include <cstdint>
#include <hip/hip_runtime.h>
#include "../../include/np2.h"
#include "np2_lookback.hpp"
#include "np2_launch.hpp"

namespace np2 {

static constexpr uint32_t TILE_SHIFT = 10;  // contig tile = 1024 positions: unit of the bucketed exception sort
static constexpr uint32_t TILE = 1u << TILE_SHIFT;
static constexpr uint32_t TILE_CAP = 4096;  // records per tile bucket = what a tile can sort inside LDS (larger
                                            // tiles spill to the overflow area and take the device-wide sort)

static constexpr uint32_t DENSE_COLS = 4096; // columns of a chunk of the dense pass: 64 per lane of a wavefront
struct ChunkDesc { // one DENSE_COLS-column chunk of a streamed read (built on the host at upload)
    uint64_t nib_off;     // byte offset of the READ's nibble stream
    uint64_t ckbase;      // first checkpoint slot of the read
    uint32_t read, ts;    // read index, aln_t_s
    uint32_t c0, ncols;   // first column of the chunk, columns of the read
    uint32_t first_chunk; // index of the read's first chunk
    uint32_t aln_t_e, nck;
    uint32_t pad0;
    uint32_t pad[4];
};

struct NodeArrays { // exception nodes, grouped by position, ordered like Msa::sort (main.rs:227-229)
    uint32_t *pos;
    uint16_t *bases;
    uint16_t *delta;
    uint32_t *count;
    uint32_t *minr;
};
struct GraphPtrs {
    const uint8_t *refnib;
    const uint32_t *node_off;
    NodeArrays nd;
    const int32_t *cov;
    uint32_t L;
    const uint2 *nrec; // packed node records {bases | delta << 16, count}
    const uint32_t *deep; // per-pass flag: a position covered 65536x or more (set by the tile builder)
    const uint8_t *pflag; // per position: bit 0 = has exception nodes, bit 1 = coverage below 2
};
// what candidate extraction needs to know about a read, in one 32-byte line (built per pass by k_pair_count: a
// (region, read) pair then costs one memory transaction for the read instead of six scattered ones)
struct __attribute__((aligned(16))) ReadInfo {
    // first half: what decoding a candidate needs
    uint32_t aln_t_s, n_cols;
    uint32_t nib16;        // nib_off / 16 (nibble streams are 16-byte aligned; a contig's pileup stays below 64 GiB)
    uint32_t ck_off;       // first checkpoint of the read (a contig has fewer than 2^32 of them: 137 G pileup columns)
    // second half: what pairing a read with a region needs — ONE 16-byte request per (read, region) pair
    uint32_t aln_t_e;      // inclusive last position
    uint32_t pj, pcount;   // the read's region interval [pj, pj + pcount); 0 regions for a dropped read
    uint32_t pad;
};
struct CandPtrs {
    const np2_read_t *reads;
    const uint8_t *nib;
    const uint64_t *ck_off;
    const uint32_t *ckpt;
    const uint32_t *lq_start;
    const uint32_t *lq_end;
    const uint32_t *pj;
    const uint32_t *pcount;
    const uint8_t *alive;
    const ReadInfo *rinfo;       // per read, packed (k_pair_count)
    const uint32_t *tile_rd_off; // per contig tile: reads overlapping it, ascending read index (built at upload)
    const uint32_t *tile_rd;
    uint32_t n_tiles;
    uint32_t ksize;
    // the sorted exception records of the dense pass, by contig tile (bucketed layout), and the index k_tile_sort leaves:
    // rec_pidx[tile * 64 + j] = first record of the tile at or beyond its position 16 j (nullptr: not available — the
    // device-wide sort took over —, every candidate is then decoded from its read)
    const uint64_t *rec_key;
    const uint32_t *rec_read;
    const uint32_t *tile_n;
    const uint16_t *rec_pidx;
    uint32_t bucket_cap;
    const uint32_t *refnib;
    uint32_t L;
};
struct YakDev {
    const uint64_t *table; // 1024 sub-tables of (1 << cap_log2) slots
    uint32_t cap_log2;
    uint32_t k;
    const uint32_t *ord; // nullptr unless the dump repeated a key: a slot's index in its bucket's file order (k_yak_insert_dup)
};

// refnib: [0, stride) the contig's codes, position p in nibble p & 1 of byte p >> 1; [stride, 2 stride) and [2 stride,
// 3 stride) the same codes in the packed streams' own order (even column in the high nibble) for windows that start at an
// even / odd position (k_diff_reads)
void launch_encode_ref(hipStream_t s, const uint8_t *read0, uint32_t L, uint8_t *refnib, uint32_t nbytes, uint32_t stride, uint32_t *err);
void launch_chunk_counts(hipStream_t s, const ChunkDesc *descs, uint32_t n_chunks, const uint8_t *nib, uint64_t *chunk_st, uint32_t epoch);
void launch_diff_reads(hipStream_t s, const ChunkDesc *descs, uint32_t n_chunks, const uint8_t *nib, const uint64_t *refw,
                       const uint8_t *refnib, const uint8_t *refeo, uint32_t eo_stride, uint32_t L, uint64_t *keys, uint32_t *vals, uint32_t *tile_cur,
                       uint32_t n_tiles, uint32_t bucket_cap, uint64_t ovf_base, uint32_t ovf_cap, uint32_t *ovf_cnt,
                       uint32_t *ckpt, uint64_t *chunk_st, uint32_t epoch, uint32_t *err, uint32_t probe = 0);
void launch_post(hipStream_t s, uint32_t *scal, uint32_t n_scal, uint32_t *mbox, uint32_t seq, uint32_t *d0 = nullptr,
                 const uint32_t *s0 = nullptr, uint32_t *d1 = nullptr, const uint32_t *s1 = nullptr, uint32_t *d2 = nullptr,
                 const uint32_t *s2 = nullptr, uint32_t *d3 = nullptr, const uint32_t *s3 = nullptr,
                 const uint32_t *ends_of = nullptr, uint32_t *ends_dst = nullptr);
void launch_fill(hipStream_t s, uint8_t *p, uint64_t bytes, uint8_t byte);
void launch_copy(hipStream_t s, uint8_t *dst, const uint8_t *src, uint64_t bytes);
void launch_copy_len(hipStream_t s, uint8_t *dst, const uint8_t *src, const uint32_t *n_dev, uint32_t elem, uint64_t cap_bytes);
// min(*n_dev, cap) 32-bit words, the count read on the device
void launch_copy_counted(hipStream_t s, uint32_t *dst, const uint32_t *src, const uint32_t *n_dev, uint32_t cap);
void launch_init_alive(hipStream_t s, const np2_read_t *reads, uint32_t R, uint8_t *alive);
void launch_kill_reads(hipStream_t s, const uint32_t *ids, uint32_t n, uint8_t *alive);
void launch_revive_reads(hipStream_t s, const uint32_t *ids, uint32_t n, uint8_t *alive); // alive[ids[i]] = 1 (np2_shard_apply)
void launch_kill_flagged(hipStream_t s, const uint8_t *flag, uint32_t n, uint8_t *alive); // alive[i] = 0 where flag[i]
// DP + backtrack of the dirty runs.  Short runs: one fused on-chip kernel; long runs and the run reaching the contig end:
// the generic kernel (independent of the first: the two may run on different streams); finish: score total, best end
// node, backtrack of the contig-end run, emission fix-up left of the path start.
void launch_dp_short(hipStream_t s, const GraphPtrs &gp, const void *refw, const uint32_t *run_start,
                     const uint32_t *n_runs, uint32_t max_runs, uint32_t *run_end, int64_t *run_gain, uint32_t *emit,
                     uint32_t *path_begin, uint64_t *path, uint32_t *dp_list, uint32_t *n_dp_list);
void launch_dp_long(hipStream_t s, const GraphPtrs &gp, const uint32_t *run_start, const uint32_t *n_runs,
                    uint32_t max_runs, const uint2 *nrec, int64_t *nscore, uint32_t *nbesti, uint32_t *n0_besti,
                    uint32_t *run_end, int64_t *last_n0_score, int64_t *run_gain, uint32_t *emit, uint32_t *path_begin,
                    uint64_t *path, uint8_t *run_flag, const uint32_t *dp_list, const uint32_t *n_dp_list);
void launch_dp_finish(hipStream_t s, const GraphPtrs &gp, const uint32_t *run_start, const uint32_t *n_runs,
                      const int64_t *nscore, const uint32_t *nbesti, const uint32_t *n0_besti, const int64_t *last_n0_score,
                      unsigned long long *total_gain, uint32_t *blocks_done, uint32_t *best_idx, const int64_t *run_gain,
                      const long long *tile_gain, uint32_t n_tiles, uint32_t *emit, uint32_t *path_begin, uint64_t *path);
// consensus write-out: clean positions + the recorded run paths, one thread per contig position
// consensus write-out: clean positions by position, dirty runs by run (with the number of low-quality bases each
// wrote: lqc[0 .. run_bound], cleared past the device-side run count); the scanned counts place the consensus indices of
// the low-quality bases in lq_list (count on the device, bounded by lq_cap) and the LQ kernels run over that list.
// Region heads are marked in a bitmap over the emission indices (zeroed by the caller), counted per word, scanned, and
// written out in bit order = the reference's order.
void launch_bt_write(hipStream_t s, const GraphPtrs &gp, const uint32_t *run_start, const uint32_t *n_runs,
                     uint32_t run_bound, const uint32_t *emit, const uint32_t *eoff, const uint64_t *path,
                     uint32_t *cns_pos, uint8_t *cns_base, uint8_t *cns_cls, uint8_t *lq_nothead, uint32_t *lqc);
void launch_lq_list(hipStream_t s, const GraphPtrs &gp, const uint32_t *run_start, const uint32_t *n_runs,
                    uint32_t run_bound, const uint32_t *emit, const uint32_t *eoff, const uint64_t *path,
                    const uint32_t *lqoff, uint32_t cap, uint32_t *lq_list, uint32_t *err);
// main.rs:1651,1680 with no end node at score >= 0: the default node's 'A' at a clean last position (see k_dp_finish)
void launch_default_tail(hipStream_t s, const GraphPtrs &gp, const uint32_t *best_idx, const uint32_t *M_p, uint8_t *cns_base,
                         uint8_t *cns_cls, uint32_t *lq_list, uint32_t *n_lq, uint32_t cap, uint32_t *err);
void launch_lq_scan(hipStream_t s, const uint32_t *cns_pos, const uint8_t *cns_base, const uint8_t *cns_cls,
                    const uint32_t *M_p, const uint32_t *lq_list, const uint32_t *n_lq, uint32_t lq_cap, uint8_t *lq_kind,
                    uint32_t *lq_next, uint8_t *lq_nothead, uint32_t *hbits, uint32_t n_hwords, uint32_t *rstart, uint32_t *rend);
void launch_lq_merge_scan_lb(hipStream_t s, const Lookback &lb, const uint32_t *raw_start, const uint32_t *raw_end, const uint32_t *n_raw,
                             uint32_t n_host, uint32_t *headflag, uint32_t *hidx, uint32_t *err);
uint32_t lq_merge_lb_blocks();
void launch_lq_merge_scan(hipStream_t s, const uint32_t *raw_start, const uint32_t *raw_end, const uint32_t *n_raw, uint32_t n_host,
                          uint32_t *headflag, uint32_t *hidx);
// exclusive sums of popcount(bits[w]), w < n_words, into out[0 .. n_words] (out[n_words] = the total)
void launch_scan_lb_popc(hipStream_t s, const Lookback &lb, const uint32_t *bits, uint32_t *out, uint32_t n_words, uint32_t *err);
void launch_lq_bits_count(hipStream_t s, const uint32_t *hbits, uint32_t n_words, uint32_t *wcnt);
void launch_scatter_regions(hipStream_t s, const uint32_t *hbits, uint32_t n_words, const uint32_t *woff,
                            const uint32_t *rstart, const uint32_t *rend, uint32_t *raw_start, uint32_t *raw_end,
                            uint32_t *n_raw);
void launch_lq_merge_flag(hipStream_t s, const uint32_t *raw_start, const uint32_t *raw_end, const uint32_t *n_raw,
                          uint32_t *headflag);
void launch_lq_merge_write(hipStream_t s, const uint32_t *raw_start, const uint32_t *raw_end, const uint32_t *n_raw,
                           const uint32_t *headflag, const uint32_t *hidx, uint32_t *lq_start, uint32_t *lq_end,
                           uint32_t *n_reg);
void launch_read_m(hipStream_t s, const np2_read_t *reads, uint32_t R, const uint8_t *alive, const uint32_t *lq_start,
                   uint32_t n_reg, int32_t *mval);
void launch_yak_insert(hipStream_t s, const uint64_t *words, const uint64_t *bucket_off, uint32_t n_buckets,
                       uint64_t max_bucket, uint64_t *table, uint32_t cap_log2, uint32_t *dup_flag, uint32_t gap = 0);
// a dump that repeats a key: the table refilled (the caller has reset it to EMPTY) with a slot per WORD and `ord[slot]` =
// the word's index in its bucket (kmer.rs:148-167: the last word that passes min_count wins, decided at lookup)
void launch_yak_insert_dup(hipStream_t s, const uint64_t *words, const uint64_t *bucket_off, uint32_t n_buckets,
                           uint64_t max_bucket, uint64_t *table, uint32_t cap_log2, uint32_t *ord, uint32_t gap = 0);
void launch_lookup(hipStream_t s, const YakDev &y, const uint64_t *hashes, uint64_t n, uint16_t min_count, uint16_t *out);
void launch_score_strings(hipStream_t s, const YakDev &y, const uint8_t *strs, const uint64_t *off, uint64_t n,
                          uint16_t min_count, uint16_t *out, bool own_strings);
void launch_cand_score(hipStream_t s, const YakDev &y, const uint32_t *cand_seq_off, const uint8_t *cand_seq,
                       const uint64_t *cand_kmer, const uint32_t *n_cand_p, uint32_t cand_cap, uint16_t min_count,
                       uint16_t *kscore, uint32_t *long_list, uint32_t *n_long);


// ---- np2_cand.hip: region-major candidate extraction, single-block scans -----------------------------
void launch_pair_count(hipStream_t s, const np2_read_t *reads, uint32_t R, const uint8_t *alive,
                       const uint32_t *lq_start, const uint32_t *lq_end, uint32_t n_reg, const int32_t *smin,
                       uint32_t *pj, uint32_t *pcount, const uint64_t *ck_off, ReadInfo *rinfo);
void launch_scan_lb_excl(hipStream_t s, const Lookback &lb, const uint32_t *in, uint32_t *out, uint32_t n, bool write_end,
                          uint32_t *err);
// exclusive sums of any length (reduce-then-scan over 4096-element tiles); part / part_off: scan3_tiles(n) + 1 words each
uint32_t scan3_tiles(uint32_t n);
uint32_t scan_lb_blocks(uint64_t n); // blocks of launch_scan_lb_excl over n elements (its look-back descriptor is sized by this)
void launch_scan3_excl(hipStream_t s, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *part, uint32_t *part_off,
                       bool write_end);
void launch_scan_small_excl(hipStream_t s, const uint32_t *in, uint32_t *out, uint32_t n, const uint32_t *n_dev,
                            uint32_t *total_out, bool write_end); // write_end: also out[n] = total
void launch_scan_small_incl(hipStream_t s, const int32_t *in, int32_t *out, uint32_t n, const uint32_t *n_dev);
void launch_scan_small_min(hipStream_t s, const int32_t *in, int32_t *out, uint32_t n, const uint32_t *n_dev);
void launch_region_measure(hipStream_t s, const CandPtrs &c, uint32_t n_reg, uint32_t *kept_read, uint32_t *kept_len,
                           uint32_t *kept_col, uint32_t *reg_ncand, uint32_t *reg_bytes, uint32_t *reg_maxlen,
                           uint32_t *blk_sum);
// blk_sum: 3 x ceil(n_reg / 4) sums per block of 4 regions (candidates, bytes, longest kept string)
uint32_t cand_offsets_blocks(uint32_t n_reg); // blocks of the look-back variant
void launch_cand_offsets(hipStream_t s, const uint32_t *blk_sum, uint32_t n_reg, uint32_t *blk_coff, uint32_t *blk_soff,
                         uint32_t *cand_off, uint32_t *reg_soff, uint32_t *n_cand, uint32_t *n_bytes, uint32_t *grow,
                         const Lookback *lb = nullptr, uint32_t *err = nullptr);
// also fills cand_off[g] / reg_soff[g] (block prefix + the regions before g inside its block)
void launch_region_write(hipStream_t s, const CandPtrs &c, uint32_t n_reg, const uint32_t *kept_read,
                         const uint32_t *kept_len, const uint32_t *kept_col, const uint32_t *reg_ncand,
                         const uint32_t *reg_bytes, const uint32_t *blk_coff, const uint32_t *blk_soff, uint32_t *cand_off,
                         uint32_t *reg_soff, uint32_t cand_cap, uint32_t seq_cap, uint32_t *cand_order, uint64_t *cand_kmer,
                         uint32_t *cand_seq_off, uint8_t *cand_seq);

// ---- np2_graph.hip: tile-bucketed exception sort and per-pass graph construction ---------------------
uint32_t tile_scan_blocks(uint32_t n_tiles); // blocks of the look-back variants of the two per-tile scans below
void launch_tile_layout(hipStream_t s, uint32_t *tile_cur, uint32_t n_tiles, uint32_t bucket_cap, uint32_t *tile_n,
                        uint32_t *tile_scan, uint32_t *tile_scanb, const uint32_t *ovf_cnt, uint32_t *out,
                        const Lookback *lb = nullptr, uint32_t *err = nullptr);
// (pidx: 64 entries per tile, see CandPtrs::rec_pidx)
void launch_tile_sort(hipStream_t s, uint16_t *pidx, const np2_read_t *reads, const uint8_t *nib, const uint32_t *tile_n,
                      uint32_t n_tiles, uint32_t bucket_cap, uint32_t max_tile, uint64_t *keys, uint32_t *vals,
                      uint32_t *err);
void launch_gather_buckets(hipStream_t s, const np2_read_t *reads, const uint8_t *nib, const uint32_t *tile_n,
                           const uint32_t *tile_scanb, uint32_t n_tiles, uint32_t bucket_cap, const uint64_t *bkeys,
                           const uint32_t *bvals, uint64_t *keys, uint32_t *vals);
void launch_gather_spill(hipStream_t s, const np2_read_t *reads, const uint8_t *nib, const uint64_t *okeys,
                         const uint32_t *ovals, uint32_t n, uint64_t *keys, uint32_t *vals);
// tile t's sorted records: [a, a + tile_n[t]) with a = t * bucket_cap, or tile_scan[t] when bucket_cap == 0
void launch_tile_count(hipStream_t s, const uint64_t *keys, const uint32_t *vals, const uint32_t *tile_n,
                       const uint32_t *tile_scan, uint32_t bucket_cap, uint32_t n_tiles, const uint8_t *alive,
                       uint32_t *tile_nn, uint32_t *tile_nr);
// (tile_gain != nullptr: the per-tile shares of the path score are added up into *gain_total as well — the fused pass front)
void launch_tile_offsets(hipStream_t s, const uint32_t *tile_nn, const uint32_t *tile_nr, uint32_t n_tiles,
                         uint32_t *tile_noff, uint32_t *tile_roff, uint32_t *n_nodes, uint32_t *n_runs, uint32_t *reset,
                         uint32_t n_reset, const Lookback *lb = nullptr, uint32_t *err = nullptr,
                         const long long *tile_gain = nullptr, unsigned long long *gain_total = nullptr);
void launch_tile_write(hipStream_t s, const uint64_t *keys, const uint32_t *vals, const uint32_t *tile_n,
                       const uint32_t *tile_scan, uint32_t bucket_cap, const uint32_t *tile_noff, const uint32_t *tile_roff,
                       uint32_t n_tiles, const uint8_t *alive, uint32_t L, NodeArrays nd, uint2 *nrec, uint32_t *node_off,
                       uint32_t *run_start, const np2_read_t *reads, const uint32_t *tile_rd_off, const uint32_t *tile_rd,
                       int32_t *cov, const uint8_t *refnib, uint32_t *emit, long long *tile_gain, uint32_t *deep_flag, uint32_t deep_min,
                       uint8_t *pflag);

// ---- np2_passfront.hip: the front of a pass on chip (records of a tile -> the tile's piece of the consensus) -----------
static constexpr uint32_t PF_CAP = 960;      // records (tile + halo) the ordinary variant stages in LDS
static constexpr uint32_t PF_HALO = 64;      // positions of the next tile it sees
static constexpr uint32_t PF_CAP_MID = 2048; // the middle variant (tiles the ordinary one listed): records; the same halo
static constexpr uint32_t PF_CAP_BIG = 3584; // the big variant (tiles the middle one listed): records, and a whole tile of halo
static constexpr uint32_t PF_COV_MAX = 8192; // coverage from which a pass goes through the unfused kernels (32-bit scores, 14-bit counts)
static constexpr uint32_t PF_REDO = 1u;      // flag word: this pass has to be redone by the unfused kernels
static constexpr uint32_t PF_NEED_BIG = 2u;  // ... because a tile needed the big variant, which was not launched (it is from then on)
struct PfTile {
    const uint64_t *keys; // sorted records, bucketed layout: tile t at [t * bucket_cap, + tile_n[t])
    const uint32_t *vals;
    const uint32_t *tile_n, *tile_scan;
    const uint16_t *pidx;  // per tile: first record at or beyond every 16th position (k_tile_sort)
    const uint8_t *alive;
    const np2_read_t *reads;
    const uint32_t *tile_rd_off, *tile_rd;
    const uint8_t *refnib;
    uint16_t *slots;       // per-tile consensus entries: position in the tile | base code << 11 | class << 14
    uint32_t *tile_cnt, *tile_lq; // entries / low-quality entries per tile
    long long *tile_gain;  // the tile's share of the path score
    uint32_t *flags, *n_bad, *bad_list, *n_bad2, *bad_list2;
    long long *end_rel;    // score of the best end node relative to the total of the gains
    unsigned long long *gain_total; // (zeroed here for k_tile_offsets)
    unsigned long long *prof;       // nullptr, or 8 clock stamps per tile (NP2_PF_PROF)
    uint32_t L, n_tiles, bucket_cap;
    uint32_t cap_lim, cap_lim_big, halo_lim, cov_max; // PF_CAP, PF_CAP_BIG, PF_HALO, PF_COV_MAX unless a test lowers them
    uint32_t big_enabled;  // the big variant is part of the launch sequence (once a contig of this context has needed it)
};
uint64_t pf_slot_entries(uint32_t n_tiles, uint64_t T); // 16-bit entries the slot array needs
void launch_pf_tile(hipStream_t s, const PfTile &a);
void launch_pf_compact(hipStream_t s, uint32_t n_tiles, const uint16_t *slots, const uint32_t *tile_scan, const uint32_t *tile_cnt,
                       const uint32_t *tile_coff, const uint32_t *tile_lqoff, uint32_t *cns_pos, uint8_t *cns_base,
                       uint8_t *cns_cls, uint8_t *lq_nothead, uint32_t *lq_list, uint32_t lq_cap, uint32_t *err, uint32_t *flags,
                       uint32_t *flags_out, uint32_t *n_bad, uint32_t *n_bad2);

// ---- np2_regions.hip: region-logic kernels --------------------------------------------------------
struct RegionTables { // GPU-resident candidate tables of one pass (LqSeqs / LqSeq, main.rs:647-667)
    const uint32_t *cand_off; // [n_reg + 1]
    const uint32_t *order;    // read index of each candidate
    uint16_t *kscore;
    const uint32_t *seq_off;  // [n_cand + 1]
    const uint8_t *seq;
    uint32_t n_reg;
};
struct RechPtrs {
    const void *groups;
    const uint32_t *job_off;
    const uint32_t *rech;
    const uint32_t *cand_off;
    const uint32_t *keep_list;
    const uint32_t *seq_off;
    const uint8_t *seq;
    const uint8_t *cns_base;
    uint32_t n_groups;
};
// votes are collected over the regions whose start lies in [own_lo, own_hi) (everything for a whole contig)
void launch_vote_phase(hipStream_t s, const RegionTables &rt, bool asref, bool use_all, const uint32_t *lq_start,
                       uint32_t own_lo, uint32_t own_hi, uint8_t *reg_lable, uint8_t *grp, uint32_t *ecount, int32_t *ref_w,
                       uint8_t *ref_seen, uint8_t *bad, uint32_t *first_reg, uint32_t *err);
void launch_vote_counts(hipStream_t s, const uint32_t *first_reg, const uint8_t *bad, uint32_t R, uint32_t *out);
void launch_edges_write(hipStream_t s, const RegionTables &rt, const uint8_t *reg_lable, const uint8_t *grp,
                        const uint32_t *ecount, const uint32_t *eoff, uint64_t *ekey, uint32_t *eval);
// banded pair accumulator (np2_regions.hip): EDGE_BAND partners per read, 256 = one uint4 per lane of a wavefront
static constexpr uint32_t EDGE_BAND = 256;
// pj / pcount: the region interval every read spans (candidate extraction); rows of reads without partners stay unwritten
void launch_edges_row(hipStream_t s, const RegionTables &rt, const uint8_t *grp, const uint32_t *ecount, const uint32_t *pj,
                      const uint32_t *pcount, const uint8_t *alive, uint32_t R, uint32_t *band, uint32_t *row_n, uint32_t *ovf);
void launch_band_emit(hipStream_t s, const uint32_t *band, uint32_t R, const uint32_t *row_off, uint64_t *ukey, uint32_t *uw,
                      uint32_t *n_out, uint64_t key_add = 0);
// the same pairs in 4 bytes each, row by row: word = (b - a - 1) | agreeing regions << 8 | disagreeing regions << 20
// (12 bits each; a larger count bumps *ovf and the host takes the sort path), row a = [row_off[a], row_off[a + 1])
static constexpr uint32_t VOTE_CNT_MAX = 0xFFFu;
void launch_band_emit_compact(hipStream_t s, const uint32_t *band, uint32_t R, const uint32_t *row_off, uint32_t *pairs,
                              uint32_t *n_out, uint32_t *ovf);
void launch_edge_reduce(hipStream_t s, const uint64_t *ekey, const uint32_t *eval, uint32_t n, uint32_t *flag, uint32_t *wout);
void launch_edge_compact(hipStream_t s, const uint64_t *ekey, const uint32_t *flag, const uint32_t *idx, const uint32_t *wout,
                         uint32_t n, uint64_t *ukey, uint32_t *uw, uint32_t *n_out);
void launch_seed(hipStream_t s, const RegionTables &rt, int32_t max_indel_len, uint8_t *reg_lable, uint32_t *seed_cand,
                 uint32_t *keep_n, uint32_t *keep_list, uint16_t *keep_ks, uint32_t *err);
// blocks of the look-back compaction kernels over regions (k_splice_plan, k_rech_list)
uint32_t region_lb_blocks(uint32_t n_reg);
void launch_splice_find(hipStream_t s, const uint32_t *cns_pos, const uint32_t *M_p, const uint32_t *lq_start,
                        const uint32_t *lq_end, const uint8_t *reg_lable, uint8_t lable, uint32_t n_reg, uint32_t *idx_s,
                        uint32_t *idx_e, uint32_t *stuck);
void launch_splice_plan(hipStream_t s, const Lookback &lb, const uint8_t *reg_lable, uint8_t lable, uint32_t n_reg,
                        const uint32_t *stuck, const uint32_t *idx_s, const uint32_t *idx_e, const uint32_t *seed_cand,
                        const uint32_t *seq_off, uint32_t *ap_g, uint32_t *ap_s, uint32_t *ap_e, int32_t *ap_delta,
                        int32_t *ap_shift_incl, uint32_t *n_ap, const uint32_t *M_in, uint32_t *M_out, uint32_t out_cap /* elements the
                        output buffers hold: a round that would outgrow them splices nothing and raises GROW_ERR */, uint32_t *err);
// M_p: consensus length on the device; M_cap: host-side upper bound used for the launch
void launch_splice_write(hipStream_t s, const uint32_t *in_pos, const uint8_t *in_base, const uint32_t *M_p, uint32_t M_cap,
                         const uint32_t *ap_g, const uint32_t *ap_s, const uint32_t *ap_e, const int32_t *ap_delta,
                         const int32_t *ap_shift_incl, const uint32_t *n_ap, uint32_t max_ap, const uint32_t *lq_start,
                         const uint32_t *seed_cand, const uint32_t *seq_off, const uint8_t *seq, uint32_t *out_pos,
                         uint8_t *out_base);
// out[0..6): index of the first consensus base at a position >= t[k]; out[6], out[7]: positions of the first / last base of
// [out[1], out[4])
void launch_shard_bounds(hipStream_t s, const uint32_t *cns_pos, const uint32_t *M_p, const uint32_t *t, uint32_t *out);
void launch_rech_list(hipStream_t s, const Lookback &lb, const uint8_t *reg_lable, uint32_t n_reg, uint32_t *rech,
                      uint32_t *n_rech, unsigned long long *blob_bound, uint32_t *err);
// groups of chained RECH regions + per-group job offsets (job_off[n_groups] = *n_jobs); max_rech = launch bound
void launch_rech_groups(hipStream_t s, const Lookback &lb, const uint32_t *rech, const uint32_t *n_rech_p, uint32_t max_rech,
                        const uint32_t *cns_pos, const uint32_t *M_p, const uint32_t *lq_start, const uint32_t *lq_end,
                        const uint32_t *keep_n, uint32_t ksize, const uint32_t *reg_maxlen, void *tmp_groups,
                        uint32_t *head_jobs, void *groups, uint32_t *job_off, uint32_t *n_groups, uint32_t *n_jobs,
                        unsigned long long *blob_bound, uint32_t *err);
size_t rech_group_bytes();
void launch_rech_job_len(hipStream_t s, const RechPtrs &p, uint32_t n_jobs, uint32_t *len);
void launch_rech_job_build(hipStream_t s, const RechPtrs &p, uint32_t n_jobs, const uint32_t *soff32, uint64_t *soff64,
                           uint8_t *blob);
void launch_rech_apply(hipStream_t s, const RechPtrs &p, const uint16_t *score, uint16_t *keep_ks);
void launch_rech_select(hipStream_t s, const uint32_t *rech, const uint32_t *n_rech_p, uint32_t max_rech,
                        const uint32_t *cand_off, const uint32_t *keep_n, const uint32_t *keep_list, const uint16_t *keep_ks,
                        const uint32_t *order, bool first_yak, uint8_t *reg_lable, uint32_t *seed_cand);
void launch_rech_relabel(hipStream_t s, uint8_t *reg_lable, uint32_t n_reg);

// ---- np2_front.hip: BAM record -> packed pileup columnariser ---------------------------------------
struct FrontOp { // one column-producing CIGAR op (M = X I D), prefix sums precomputed by the host
    uint32_t col0, q0, t0; // first alignment column / query index / target offset (relative to pos)
    uint32_t len_type;     // len << 4 | BAM op code
};
struct FrontRec {
    uint32_t pos, n_ops;
    uint64_t op_off;  // index into ops[]
    uint64_t seq_off; // byte offset into the 4-bit SEQ buffer
    uint64_t out_off; // byte offset of the output nibble slot (16-B aligned, pre-zeroed)
    uint32_t n_cols;  // untrimmed alignment columns
    uint32_t pad;
};
struct FrontOut {
    uint32_t aln_t_s, aln_t_e, n_cols, pad;
};
void launch_columnarise(hipStream_t s, const FrontRec *recs, uint32_t n_recs, const FrontOp *ops, const uint8_t *ref,
                        const uint8_t *seq4, uint8_t *nib, FrontOut *out);
void launch_pack_ref(hipStream_t s, const uint8_t *ref, uint32_t L, uint8_t *dst);

// ---- np2_prims.hip: device-wide sort / scan plumbing (rocPRIM) -------------------------------
// All take a caller-provided temp buffer; *_temp_bytes report the requirement for n elements.
size_t prim_temp_bytes(size_t n);
int prim_sort_pairs_u64_u32(hipStream_t s, void *tmp, size_t tmp_bytes, const uint64_t *kin, uint64_t *kout,
                            const uint32_t *vin, uint32_t *vout, size_t n, unsigned end_bit);
int prim_exclusive_sum_u32(hipStream_t s, void *tmp, size_t tmp_bytes, const uint32_t *in, uint32_t *out, size_t n);
int prim_inclusive_sum_i32(hipStream_t s, void *tmp, size_t tmp_bytes, const int32_t *in, int32_t *out, size_t n);
int prim_inclusive_min_i32(hipStream_t s, void *tmp, size_t tmp_bytes, const int32_t *in, int32_t *out, size_t n);

} // namespace np2
