// Dense pass of the MI355X NextPolish2 hot path (gfx950, wave64): every packed read column is compared with the contig
// once (update_msas / Kmer::new / AlignSeq decode of the reference: src/main.rs:576-589, 84-102, 314-338), and only the
// *exception* columns — those whose 3-column-mer differs from the one the contig itself contributes — leave the kernel,
// as raw records (t_pos << 32 | column, read) filed under their contig tile, next to the per-read checkpoints.
//
// k_diff_reads is the kernel the benchmark's roofline is quoted on: 0.5 B per column + 0.5 B per contig base of
// algorithmic HBM traffic.  The work is split by what a 32-column piece of a read holds:
//   * phase 1, every lane, branch-free, two pieces per lane (64 columns, two 16-byte loads), in the packed stream's own
//     nibble order: count insertion columns, one wave scan -> t_pos of each piece's first column, 16 bytes of the contig
//     copy of the matching parity at that position, four XORs.  A piece without insertion columns whose codes all match and
//     with nothing bad in the two columns before it (90 % of the pieces of a haploid pileup, ~80 % of a diploid one) is
//     *done* after writing its checkpoint;
//   * every other piece ("dirty") is queued in LDS and handled in phase 2 by ONE THREAD per piece, the block's threads
//     taking the queue in order, in the 128-bit nibble domain of np2_nib128.hpp: insertion runs shift the contig window,
//     exact bad-column mask, exception mask, records (staged in LDS by contig tile and written out as contiguous pieces
//     while the bucket reservation is in flight), the checkpoint of an insertion-bearing or partial piece.  The divergent
//     code runs in densely packed wavefronts instead of in all of them.
// Round 5 measured what bounds it: not instruction issue (367 -> 206 VALU per 2048 columns changed
// nothing), not the stream (the first part runs at 5.5 TB/s) but its phases in sequence at eight resident blocks per CU.
#include "np2_common.hpp"
#include "np2_kernels.hpp"
#include "np2_blockscan.hpp"
#include "np2_nib128.hpp"
#include "../../include/np2.h"
#include <cstdlib>

namespace np2 {

__device__ __forceinline__ uint8_t dense_ref_code(const uint8_t *__restrict__ refnib, uint32_t p) {
    return (refnib[p >> 1] >> (4 * (p & 1))) & 7;
}

struct DenseChunk { // what phase 2 needs to know about a chunk of the block (LDS): the first 48 bytes of its ChunkDesc,
    // stored as loaded (three 16-byte pieces from the lane that fetched them) instead of field by field from scalars
    uint64_t nib_off, ckbase;
    uint32_t read, ts, c0, ncols, first_chunk, aln_t_e, nck, pad;
};
static_assert(sizeof(DenseChunk) == 48, "DenseChunk mirrors the head of ChunkDesc");
#ifndef NP2_DENSE_CPW
#define NP2_DENSE_CPW 2
#endif
static constexpr uint32_t DENSE_CPW = NP2_DENSE_CPW;     // chunks per wavefront: their loads are all in flight together
static constexpr uint32_t DENSE_CHUNKS = 4 * DENSE_CPW;  // chunks per 256-thread block
static constexpr uint32_t DENSE_HALVES = DENSE_COLS / 32; // 32-column pieces of a chunk: phase 2's unit (two per lane)
static constexpr uint32_t DENSE_TSLOTS = DENSE_CPW > 2 ? 128 : 64; // tile table of a block (a chunk's columns lie in <= 5 tiles)
static constexpr uint32_t DENSE_STAGE = 1024;            // records of a round of phase 2 staged in LDS
static_assert(DENSE_COLS == 4096, "a lane of phase 1 holds 64 columns");
static_assert(DENSE_CHUNKS * DENSE_HALVES <= 1024 && DENSE_TSLOTS <= 128, "stage entry: piece in 10 bits, column in 5, tile slot above");
static_assert(DENSE_CHUNKS * (DENSE_COLS / TILE + 1) <= DENSE_TSLOTS, "tile table too small");

typedef uint32_t dense_u32x4 __attribute__((ext_vector_type(4)));
// 16 bytes from any byte address (gfx950 in the HSA's unaligned access mode: one global_load_dwordx4)
__device__ __forceinline__ dense_u32x4 dense_load16u(const uint8_t *p) {
    dense_u32x4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
// the packed stream's own nibble order (even column in the high nibble): bits of dword k (columns 8 k .. 8 k + 7) that
// belong to the first nv columns of a 32-column piece
__device__ __forceinline__ uint32_t dense_native_below(uint32_t nv, uint32_t k) {
    const uint32_t m = nv > 8 * k ? min(8u, nv - 8 * k) : 0u;
    const uint32_t by = m >> 1;
    uint32_t r = by >= 4 ? 0xFFFFFFFFu : ((1u << (8 * by)) - 1u);
    if (m & 1) r |= 0xF0u << (8 * by);
    return r;
}
static constexpr uint32_t NF3W = 0x88888888u;

// non-insertion columns of one chunk, by one wavefront (uniform result)
__device__ __forceinline__ uint32_t dense_chunk_total(const ChunkDesc *__restrict__ dp, const uint8_t *__restrict__ nib, uint32_t lane) {
    const uint32_t c0 = dp->c0, ncols = dp->ncols;
    const uint32_t lc0 = c0 + lane * 64;
    const uint32_t nv = lc0 < ncols ? min(64u, ncols - lc0) : 0u;
    const uint32_t nvA = min(nv, 32u), nvB = nv - nvA;
    dense_u32x4 a{0, 0, 0, 0}, b{0, 0, 0, 0};
    const uint8_t *p = nib + dp->nib_off + (lc0 >> 1);
    if (nvA) a = *reinterpret_cast<const dense_u32x4 *>(p);
    if (nvB) b = *reinterpret_cast<const dense_u32x4 *>(p + 16);
    if (c0 == 0 && lane == 0) a.x &= ~0x80u; // column 0 is never an insertion column (main.rs:325,332-335)
    const uint32_t ins = (uint32_t)__builtin_popcount(a.x & NF3W & dense_native_below(nvA, 0)) + (uint32_t)__builtin_popcount(a.y & NF3W & dense_native_below(nvA, 1)) +
                         (uint32_t)__builtin_popcount(a.z & NF3W & dense_native_below(nvA, 2)) + (uint32_t)__builtin_popcount(a.w & NF3W & dense_native_below(nvA, 3)) +
                         (uint32_t)__builtin_popcount(b.x & NF3W & dense_native_below(nvB, 0)) + (uint32_t)__builtin_popcount(b.y & NF3W & dense_native_below(nvB, 1)) +
                         (uint32_t)__builtin_popcount(b.z & NF3W & dense_native_below(nvB, 2)) + (uint32_t)__builtin_popcount(b.w & NF3W & dense_native_below(nvB, 3));
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan<OpAdd>(nv - ins), 63);
}

// Phase 1 works on the stream as it lies in memory: 64 columns = two 16-byte pieces per lane (pieces l and 64 + l of the chunk), compared with a copy of the
// contig in the SAME nibble order (k_encode_ref writes two: codes 2k | 2k+1 per byte for an even first position, 2k+1 |
// 2k+2 for an odd one), so that "this piece agrees with the contig" is four XORs and two ORs per piece, with no nibble
// swap, no alignment shifts and no column masks; one wave scan serves both pieces of a lane.
__device__ __forceinline__ void k_diff_reads(const uint32_t np2_bid, const uint32_t np2_nb, const ChunkDesc *__restrict__ descs, uint32_t n_chunks, const uint8_t *__restrict__ nib,
    const uint32_t *__restrict__ refw32, const uint8_t *__restrict__ refnib, const uint8_t *__restrict__ refeo, uint32_t eo_stride, uint32_t L,
    uint64_t *__restrict__ out_keys, uint32_t *__restrict__ out_vals, uint32_t *__restrict__ tile_cur, uint32_t n_tiles,
    uint32_t bucket_cap, uint64_t ovf_base, uint32_t ovf_cap, uint32_t *__restrict__ ovf_cnt,
    uint32_t *__restrict__ ckpt, uint64_t *__restrict__ chunk_st, uint32_t epoch, uint32_t *__restrict__ err, uint32_t probe) {
    // (probe != 0: a timing experiment — tools/dense_probe.sh — that stops after a part of the kernel; launched after the
    // real pass, it rewrites what that one wrote and nothing else)
    const uint32_t lane = threadIdx.x & 63;
    // Which chunks a block takes: workgroups go round the 8 XCDs in turn, each with an L2 of its own, and the reads lie
    // in contig order — taken in launch order every XCD would pull the whole contig (three copies of it) through its L2 and
    // every tile's bucket would be written through all eight.  Block b takes the (b / 8)-th block of chunks of the
    // (b % 8)-th eighth of the list instead: an XCD works on one stretch of the contig.  Inside an eighth the order is the
    // launch order, so a chunk still only waits for blocks dispatched before its own (np2_lookback.hpp); the counts of a
    // read's chunks that lie in the eighth BEFORE — a handful of chunks per launch — are not waited for but counted again.
    uint32_t lb = np2_bid, x_first = 0; // this block's place in the list, the first block of its eighth
    if (np2_nb >= 64) {
        const uint32_t xq = np2_nb >> 3, xr = np2_nb & 7u, xcd = np2_bid & 7u;
        x_first = xcd * xq + min(xcd, xr);
        lb = x_first + (np2_bid >> 3);
    }
    const uint32_t pw = __builtin_amdgcn_readfirstlane(lb * (blockDim.x >> 6) + (threadIdx.x >> 6));
    __shared__ uint32_t s_total[DENSE_CHUNKS];        // non-insertion columns of the block's chunks
    __shared__ uint2 s_q[DENSE_CHUNKS * DENSE_HALVES]; // dirty pieces: {t0, chunk in block << 7 | piece}
    __shared__ __attribute__((aligned(16))) DenseChunk s_desc[DENSE_CHUNKS];
    __shared__ uint32_t s_nq;
    __shared__ uint32_t s_tile[DENSE_TSLOTS], s_cnt[DENSE_TSLOTS], s_base[DENSE_TSLOTS], s_off[DENSE_TSLOTS + 1]; // phase 2: the tiles the block adds records to
    __shared__ uint2 s_stage[DENSE_STAGE];            // phase 2: records of a round, grouped by tile: {t_pos, piece | column << 10 | tile slot << 15}
    const uint32_t blk_first = lb * DENSE_CHUNKS;
    if (threadIdx.x == 0) s_nq = 0;
#ifdef NP2_DENSE_PAD_LDS // (occupancy experiment: fewer resident blocks per CU)
    __shared__ uint32_t s_pad[NP2_DENSE_PAD_LDS / 4];
    if (probe == 77) s_pad[threadIdx.x] = epoch, atomicOr(err, s_pad[(threadIdx.x * 7) & 255]);
#endif
    // A wave's time is a chain of memory round trips on top of its instructions: everything that can be requested together
    // is.  Round trip 1: the descriptors of the wave's chunks (scalar loads); 2: the chunks' 32 bytes per lane; 3 (after
    // the block-wide exchange of the chunk totals): the status words of chunks in earlier blocks, where a read started
    // there; 4: the contig windows of all pieces.
    struct Desc {
        uint64_t nib_off, ckbase;
        uint32_t read, ts, c0, ncols, first_chunk;
        bool live;
    } dd[DENSE_CPW];
    {
        // (the wave's chunk indices are uniform: the descriptors arrive as scalar loads, all requested together, and cost
        // no VALU issue slot; the copy phase 2 reads goes to LDS from lanes 0..CPW-1 as three 16-byte pieces each)
        const uint32_t ch0 = DENSE_CPW * pw;
        if (lane < DENSE_CPW && ch0 + lane < n_chunks) {
            const uint4 *p = reinterpret_cast<const uint4 *>(descs + ch0 + lane);
            const uint4 q0 = p[0], q1 = p[1], q2 = p[2];
            uint4 *d = reinterpret_cast<uint4 *>(&s_desc[ch0 + lane - blk_first]); // (read after the block's barriers)
            d[0] = q0, d[1] = q1, d[2] = q2;
        }
#pragma unroll
        for (uint32_t it = 0; it < DENSE_CPW; ++it) {
            const uint32_t ci = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(ch0 + it, n_chunks - 1));
            const ChunkDesc *dp = descs + ci;
            dd[it].nib_off = dp->nib_off, dd[it].ckbase = dp->ckbase;
            dd[it].read = dp->read, dd[it].ts = dp->ts, dd[it].c0 = dp->c0, dd[it].ncols = dp->ncols;
            dd[it].first_chunk = dp->first_chunk; // (nck, aln_t_e: read from the LDS copy where they are used)
            dd[it].live = ch0 + it < n_chunks;
        }
    }
    // ---- phase A: load the chunks, count their non-insertion columns and publish the counts at once.  A chunk needs
    //      the counts of the read's earlier chunks (status word: launch epoch | count); they belong to lower-numbered,
    //      already running waves, which publish within a microsecond of starting ---------------------------------------
    // A lane holds pieces `lane` and 64 + `lane` of its chunk (each of the two loads of a wavefront is 1 KiB of consecutive
    // bytes); positions run through pieces 0 .. 127, so the lane's two counts are scanned side by side in one word.
    dense_u32x4 va_[DENSE_CPW], vb_[DENSE_CPW];
    uint32_t nvA_[DENSE_CPW], nvB_[DENSE_CPW], nA_[DENSE_CPW], nB_[DENSE_CPW], exA_[DENSE_CPW], exB_[DENSE_CPW], total_[DENSE_CPW];
    bool full_[DENSE_CPW];
#pragma unroll
    for (uint32_t it = 0; it < DENSE_CPW; ++it) { // (all loads first)
        const uint32_t lcA = dd[it].c0 + lane * 32, lcB = lcA + DENSE_COLS / 2;
        const bool full = dd[it].live && dd[it].ncols - dd[it].c0 >= DENSE_COLS; // every piece of the chunk holds 32 columns
        full_[it] = full;
        nvA_[it] = !dd[it].live ? 0u : (full ? 32u : (lcA < dd[it].ncols ? min(32u, dd[it].ncols - lcA) : 0u));
        nvB_[it] = !dd[it].live ? 0u : (full ? 32u : (lcB < dd[it].ncols ? min(32u, dd[it].ncols - lcB) : 0u));
        // (unconditional loads, all in flight together: a piece past the read's end reads the stream's last bytes — 16 bytes
        // of padding follow every stream — and what it gets is masked by its column count)
        const uint32_t last = (max(dd[it].ncols, 1u) - 1u) >> 1;
        const uint8_t *p = nib + dd[it].nib_off;
        va_[it] = dense_load16u(p + min(lcA >> 1, last));
        vb_[it] = dense_load16u(p + min(lcB >> 1, last));
    }
#pragma unroll
    for (uint32_t it = 0; it < DENSE_CPW; ++it) {
        const uint32_t ch = DENSE_CPW * pw + it;
        if (dd[it].c0 == 0 && lane == 0) va_[it].x &= ~0x80u; // column 0 is never an insertion column (main.rs:325,332-335)
        const dense_u32x4 a = va_[it], b = vb_[it];
        uint32_t iA = (uint32_t)__builtin_popcount(a.x & NF3W) + (uint32_t)__builtin_popcount(a.y & NF3W) +
                      (uint32_t)__builtin_popcount(a.z & NF3W) + (uint32_t)__builtin_popcount(a.w & NF3W);
        uint32_t iB = (uint32_t)__builtin_popcount(b.x & NF3W) + (uint32_t)__builtin_popcount(b.y & NF3W) +
                      (uint32_t)__builtin_popcount(b.z & NF3W) + (uint32_t)__builtin_popcount(b.w & NF3W);
        uint32_t nA = 32u - iA, nB = 32u - iB;
        if (!full_[it]) { // (uniform) the read ends in this chunk: one piece is partial, the pieces after it hold nothing
            const uint32_t nvA = nvA_[it], nvB = nvB_[it];
            if (nvA != 32u) {
                iA = nvA ? (uint32_t)__builtin_popcount(a.x & NF3W & dense_native_below(nvA, 0)) + (uint32_t)__builtin_popcount(a.y & NF3W & dense_native_below(nvA, 1)) +
                               (uint32_t)__builtin_popcount(a.z & NF3W & dense_native_below(nvA, 2)) + (uint32_t)__builtin_popcount(a.w & NF3W & dense_native_below(nvA, 3))
                         : 0u;
                nA = nvA - iA;
            }
            if (nvB != 32u) {
                iB = nvB ? (uint32_t)__builtin_popcount(b.x & NF3W & dense_native_below(nvB, 0)) + (uint32_t)__builtin_popcount(b.y & NF3W & dense_native_below(nvB, 1)) +
                               (uint32_t)__builtin_popcount(b.z & NF3W & dense_native_below(nvB, 2)) + (uint32_t)__builtin_popcount(b.w & NF3W & dense_native_below(nvB, 3))
                         : 0u;
                nB = nvB - iB;
            }
        }
        const uint32_t incl = wave_incl_scan<OpAdd>(nA | (nB << 16)); // (64 x 32 < 2^16: the halves do not meet)
        const uint32_t tot2 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t totA = tot2 & 0xFFFFu, total = totA + (tot2 >> 16);
        if (lane == 0 && dd[it].live) {
            __hip_atomic_store(&chunk_st[ch], ((uint64_t)epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_total[ch - blk_first] = total;
        }
        nA_[it] = nA, nB_[it] = nB, total_[it] = total;
        exA_[it] = (incl & 0xFFFFu) - nA, exB_[it] = totA + (incl >> 16) - nB; // non-insertion columns of the chunk before the piece
    }
    if (probe == 1) return;
    __syncthreads();
    // ---- phase B: t_pos, contig codes, compare; clean pieces finish here, dirty ones are queued ---------------------------
    uint32_t carry_[DENSE_CPW];
    {
        bool timeout = false;
#pragma unroll
        for (uint32_t it = 0; it < DENSE_CPW; ++it) {
            const uint32_t ch = DENSE_CPW * pw + it;
            carry_[it] = 0;
            if (!dd[it].live) continue;
            const bool cont = it != 0 && dd[it].read == dd[it ? it - 1 : 0].read && dd[it].c0 != 0;
            if (cont) {
                carry_[it] = carry_[it ? it - 1 : 0] + total_[it ? it - 1 : 0];
                continue;
            }
            uint32_t carryN = 0; // non-insertion columns of the read before this chunk
            // earlier chunks of the read inside this block: from LDS; the ones in earlier blocks: from their status words
            {
                const uint32_t lo = max(dd[it].first_chunk, blk_first) - blk_first, hi = ch - blk_first; // (hi <= DENSE_CHUNKS <= 16)
                uint32_t v = lane >= lo && lane < hi ? s_total[lane] : 0u;
                if (lo < hi) { // a row's worth of lanes: four DPP steps, lane 15 holds the sum
                    static_assert(DENSE_CHUNKS <= 16, "the block's chunk totals are summed inside one DPP row");
                    v += dpp_get<0x111, 0xF>(0u, v);
                    v += dpp_get<0x112, 0xF>(0u, v);
                    v += dpp_get<0x114, 0xF>(0u, v);
                    v += dpp_get<0x118, 0xF>(0u, v);
                    carryN += (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
                }
            }
            const uint32_t jend = min(ch, blk_first);
            const uint32_t jown = min(jend, max(dd[it].first_chunk, x_first * DENSE_CHUNKS)); // chunks before it: another XCD's
            for (uint32_t j = dd[it].first_chunk; j < jown; ++j) carryN += dense_chunk_total(descs + j, nib, lane);
            for (uint32_t j0 = jown; j0 < jend; j0 += 64) {
                const uint32_t j = j0 + lane;
                uint32_t v = 0;
                if (j < jend) {
                    uint32_t spins = 0;
                    uint64_t t_wait0 = 0;
                    for (;;) {
                        const uint64_t sw = __hip_atomic_load(&chunk_st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((uint32_t)(sw >> 32) == epoch) {
                            v = (uint32_t)sw;
                            break;
                        }
                        if (lb_gave_up(spins, t_wait0)) { // (4 s of wall clock: np2_lookback.hpp)
                            timeout = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                carryN += (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan<OpAdd>(v), 63);
            }
            carry_[it] = carryN;
        }
        if (__ballot(timeout) && lane == 0) atomicOr(err, LB_ERR);
    }
    // the 32 contig codes of every piece, starting at the t_pos of its first non-insertion column: 16 bytes of the copy
    // whose nibble parity matches, at whatever byte they start; all of them requested before the first is used.
    // (a stream that disagrees with its descriptor could push t0 past the contig: stay inside the padded buffer; such a
    // read is reported by the descriptor check at the end of its last chunk)
    uint32_t tA_[DENSE_CPW], tB_[DENSE_CPW];
    dense_u32x4 ra_[DENSE_CPW], rb_[DENSE_CPW];
    const uint32_t win_lim = (L >> 1) + 32;
#pragma unroll
    for (uint32_t it = 0; it < DENSE_CPW; ++it) {
        const uint32_t tA = dd[it].ts + carry_[it] + exA_[it], tB = dd[it].ts + carry_[it] + exB_[it];
        tA_[it] = tA, tB_[it] = tB;
        ra_[it] = dense_load16u(refeo + ((tA & 1) ? eo_stride : 0u) + min(tA >> 1, win_lim));
        rb_[it] = dense_load16u(refeo + ((tB & 1) ? eo_stride : 0u) + min(tB >> 1, win_lim));
    }
    uint32_t prev_top = 0; // "one of the last two columns of the previous chunk is bad" (when this one continues it)
#pragma unroll
    for (uint32_t it = 0; it < DENSE_CPW; ++it) {
        const uint32_t ch = DENSE_CPW * pw + it;
        if (!dd[it].live) break;
        const uint32_t ncols = dd[it].ncols, ts = dd[it].ts, c0 = dd[it].c0;
        const uint32_t lcA = c0 + lane * 32, lcB = lcA + DENSE_COLS / 2;
        const uint32_t nA = nA_[it], nB = nB_[it], total = total_[it], carryN = carry_[it];
        const uint32_t tA = tA_[it], tB = tB_[it];
        const bool cont = it != 0 && dd[it].read == dd[it ? it - 1 : 0].read && c0 != 0;
        // columns that differ from the contig, insertion columns among them (the contig's copies carry no flag bit)
        const dense_u32x4 xa = va_[it] ^ ra_[it], xb = vb_[it] ^ rb_[it];
        uint32_t oA = xa.x | xa.y | xa.z | xa.w, oB = xb.x | xb.y | xb.z | xb.w;
        // A bad column marks itself and the two columns after it (3-column-mers): the last two columns of a piece reach
        // into the next one.  After an insertion column a piece's window is no longer the contig's: assume the worst.
        uint32_t topA = nA != 32u ? 1u : xa.w >> 24, topB = nB != 32u ? 1u : xb.w >> 24;
        bool okA = true, okB = true; // the piece holds 32 columns
        if (!full_[it]) { // (uniform) a partial piece goes to phase 2, which masks; an empty one nowhere
            okA = nvA_[it] == 32u, okB = nvB_[it] == 32u;
            if (!okA) oA = nvA_[it] ? 1u : 0u, topA = 0;
            if (!okB) oB = nvB_[it] ? 1u : 0u, topB = 0;
        }
        // checkpoints: column of the reference column at the next multiple of CKPT (a whole piece without insertion columns
        // holds exactly one, and column index and position advance together; phase 2 writes the others'); consecutive lanes
        // write consecutive slots
        {
            const uint64_t ck_first = (ts + CKPT - 1) >> CKPT_SHIFT;
            uint32_t *const ckb = ckpt + (dd[it].ckbase - ck_first);
            const uint32_t nck = s_desc[ch - blk_first].nck;
            const uint32_t tsA = (tA + CKPT - 1) & ~(CKPT - 1), tsB = (tB + CKPT - 1) & ~(CKPT - 1);
            if (probe != 2) {
                if (okA && nA == 32u && (tsA >> CKPT_SHIFT) - (uint32_t)ck_first < nck) ckb[tsA >> CKPT_SHIFT] = lcA + (tsA - tA);
                if (okB && nB == 32u && (tsB >> CKPT_SHIFT) - (uint32_t)ck_first < nck) ckb[tsB >> CKPT_SHIFT] = lcB + (tsB - tB);
            }
        }
        // the piece before piece A of lane l is A of lane l - 1 (lane 0: the previous chunk's last piece), before B of lane l
        // B of lane l - 1 (lane 0: A of lane 63)
        const uint32_t tops = (topA ? 1u : 0u) | (topB ? 2u : 0u);
        uint32_t pb = wave_prev_lane(0u, tops);
        const uint32_t t63 = (uint32_t)__builtin_amdgcn_readlane((int)tops, 63);
        if (lane == 0) pb = (cont ? prev_top : (c0 > 0 ? 1u : 0u)) | ((t63 & 1u) << 1); // (chunk start inside a read: phase 2 looks)
        prev_top = t63 >> 1;
        const bool dirtyA = oA != 0 || ((pb & 1u) != 0 && nvA_[it] != 0) || (lcA == 0 && ts != 0);
        const bool dirtyB = oB != 0 || ((pb & 2u) != 0 && nvB_[it] != 0);
        const uint64_t dmA = __ballot(dirtyA), dmB = __ballot(dirtyB);
        if (probe == 2 || probe == 3) { // (keep the comparison alive)
            if ((dmA ^ dmB) == 0x123456789ABCDEFull && lane == 0) atomicOr(err, 0x80000000u);
            continue;
        }
        if (dmA | dmB) {
            const uint32_t cA = (uint32_t)__builtin_popcountll(dmA);
            uint32_t qb = 0;
            if (lane == 0) qb = atomicAdd(&s_nq, cA + (uint32_t)__builtin_popcountll(dmB));
            qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
            const uint32_t tag = ((ch - blk_first) << 7) | lane; // chunk in block | piece
            if (dirtyA) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dmA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dmA, 0u));
                s_q[qb + rank] = make_uint2(tA, tag);
            }
            if (dirtyB) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dmB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dmB, 0u));
                s_q[qb + cA + rank] = make_uint2(tB, tag | 64u);
            }
        }
        if (lane == 0) {
            if (c0 + DENSE_COLS >= ncols) {
                // last chunk: the packed stream must agree with its descriptor (AlignSeq::new, main.rs:279-312)
                const uint8_t *base = nib + dd[it].nib_off; // start of the READ's stream
                const uint32_t te = s_desc[ch - blk_first].aln_t_e;
                if (ncols == 0 || ts + carryN + total - 1 != te || te >= L) atomicOr(err, 2u);
                if ((nib_at(base, ncols) & 15) != 15) atomicOr(err, 2u);
            }
        }
    }
    if (probe == 2 || probe == 3 || probe == 4) return;
    __syncthreads();
    // ---- phase 2: one thread per dirty piece --------------------------------------------------------------------------------
    // Records go into the bucket of their contig tile; the block reserves its place in a bucket ONCE per tile (a tile
    // table in LDS: a chunk's columns lie in at most 5 tiles), so that what a block adds to a tile is one contiguous piece
    // written through one L2.  (Reserving per lane left every bucket line shared by fragments
    // of a dozen blocks on different XCDs: the scattered partial-line stores cost 5x the rest of the kernel.)
    const uint32_t nq = s_nq;
    for (uint32_t q0 = 0; q0 < nq; q0 += 256) { // (uniform; one round unless more than half of the lanes are dirty)
        if (threadIdx.x < DENSE_TSLOTS) {
            s_tile[threadIdx.x] = 0xFFFFFFFFu;
            s_cnt[threadIdx.x] = 0;
        }
        __syncthreads();
        const uint32_t qi = q0 + threadIdx.x;
        N128 E{0, 0}, NI{0, 0};
        uint32_t t0 = 0, lc0 = 0, read = 0, cnt = 0, n_lo = 0, tA = 0, sl_lo = 0, sl_hi = 0, loc_lo = 0, loc_hi = 0, q_tag = 0;
        if (qi < nq) {
            const uint2 qe = s_q[qi];
            q_tag = qe.y;
            const uint32_t ln = qe.y & (DENSE_HALVES - 1);
            t0 = qe.x;
            const DenseChunk dc = s_desc[qe.y >> 7];
            const uint32_t ts = dc.ts;
            lc0 = dc.c0 + ln * 32;
            read = dc.read;
            const uint32_t nv = min(32u, dc.ncols - lc0);
            const uint8_t *base = nib + dc.nib_off;
            // everything this lane reads, requested up front: its 16 bytes, the contig window, the byte with the two
            // columns before it and the contig codes those two can sit at
            const uint4 v = *reinterpret_cast<const uint4 *>(base + (lc0 >> 1));
            const uint32_t q = min(t0 >> 3, (L >> 3) + 8), sh = (t0 & 7) * 4;
            const uint32_t r0 = refw32[q], r1 = refw32[q + 1], r2 = refw32[q + 2], r3 = refw32[q + 3], r4 = refw32[q + 4];
            uint8_t pbyte = 0, rc1 = 8, rc2 = 8;
            const uint32_t t1 = t0 - 1; // t_pos of the column before the lane, insertion column or not
            if (lc0 > 0) {
                pbyte = base[(lc0 - 2) >> 1];
                if (t1 < L) rc1 = dense_ref_code(refnib, t1);
                if (t1 >= 1 && t1 - 1 < L) rc2 = dense_ref_code(refnib, t1 - 1);
            }
            N128 w;
            w.lo = (uint64_t)swap_nib(v.x) | ((uint64_t)swap_nib(v.y) << 32);
            w.hi = (uint64_t)swap_nib(v.z) | ((uint64_t)swap_nib(v.w) << 32);
            const N128 vm = n_below(nv);
            const N128 V{NF3 & vm.lo, NF3 & vm.hi};
            N128 I{w.lo & V.lo, w.hi & V.hi};
            if (lc0 == 0) I.lo &= ~8ULL; // column 0 is never an insertion column (main.rs:325,332-335)
            NI = N128{~I.lo & V.lo, ~I.hi & V.hi};
            const uint32_t nonins = n_popc(NI);
            const N128 codes{w.lo & ~NF3, w.hi & ~NF3};
            N128 R;
            R.lo = (uint64_t)__builtin_amdgcn_alignbit(r1, r0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(r2, r1, sh) << 32);
            R.hi = (uint64_t)__builtin_amdgcn_alignbit(r3, r2, sh) | ((uint64_t)__builtin_amdgcn_alignbit(r4, r3, sh) << 32);
            // insertion runs push the rest of the contig window up by their length (one iteration per run)
            {
                N128 J = I;
                while (J.lo | J.hi) {
                    const N128 bp = n_mask_before_first(J);                    // columns before the run
                    const N128 K{~J.lo & NF3 & ~bp.lo, ~J.hi & NF3 & ~bp.hi}; // non-insertion flags at / above it
                    if (K.lo | K.hi) {
                        const N128 up = n_shl(N128{R.lo & ~bp.lo, R.hi & ~bp.hi}, n_ctz(K) - n_ctz(J)); // 4 * run length
                        R.lo = (R.lo & bp.lo) | up.lo;
                        R.hi = (R.hi & bp.hi) | up.hi;
                        const N128 bq = n_mask_before_first(K); // columns before the first one past the run
                        J.lo &= ~bq.lo;
                        J.hi &= ~bq.hi;
                    } else { // the run reaches the lane's last column
                        R.lo &= bp.lo;
                        R.hi &= bp.hi;
                        J.lo = J.hi = 0;
                    }
                }
            }
            // bad columns: insertion, or code differs from the contig
            N128 B;
            B.lo = ((((codes.lo ^ R.lo) & ~NF3) + ~NF3) | I.lo) & V.lo;
            B.hi = ((((codes.hi ^ R.hi) & ~NF3) + ~NF3) | I.hi) & V.hi;
            // the two columns before the lane (t_pos of the nearer one is t0 - 1 whether it is an insertion column or not)
            uint32_t pb = 0;
            if (lc0 > 0) {
                const uint8_t n2 = pbyte >> 4, n1 = pbyte & 15;
                const bool b1 = (n1 & 8) || (n1 & 7) != rc1;                    // (rc = 8: position outside the contig)
                const bool b2 = (n2 & 8) || (n2 & 7) != ((n1 & 8) ? rc1 : rc2); // an insertion column shares its t_pos
                pb = (b1 ? 0x80000000u : 0u) | (b2 ? 0x08000000u : 0u);
            }
            // exception columns: x | x << 1 column | x << 2 columns, carrying across dwords (alignbit(hi, lo, s) = {hi, lo} >> s)
            {
                const uint32_t b0 = (uint32_t)B.lo, b1 = (uint32_t)(B.lo >> 32), b2 = (uint32_t)B.hi, b3 = (uint32_t)(B.hi >> 32);
                const uint32_t e0 = b0 | __builtin_amdgcn_alignbit(b0, pb, 28) | __builtin_amdgcn_alignbit(b0, pb, 24);
                const uint32_t e1 = b1 | __builtin_amdgcn_alignbit(b1, b0, 28) | __builtin_amdgcn_alignbit(b1, b0, 24);
                const uint32_t e2 = b2 | __builtin_amdgcn_alignbit(b2, b1, 28) | __builtin_amdgcn_alignbit(b2, b1, 24);
                const uint32_t e3 = b3 | __builtin_amdgcn_alignbit(b3, b2, 28) | __builtin_amdgcn_alignbit(b3, b2, 24);
                E.lo = (uint64_t)e0 | ((uint64_t)e1 << 32);
                E.hi = (uint64_t)e2 | ((uint64_t)e3 << 32);
                if (lc0 == 0 && ts != 0) E.lo |= 0x88ULL; // head sentinels differ from the contig's own (main.rs:579-580)
                E.lo &= V.lo;
                E.hi &= V.hi;
            }
            // checkpoint of a piece with insertion columns, or of a partial one: the (nth + 1)-th non-insertion column
            if (nonins != nv || nv != 32u) {
                const uint32_t tstar = (t0 + CKPT - 1) & ~(CKPT - 1);
                const uint32_t nth = tstar - t0;
                if (nth < nonins) {
                    const uint32_t ck_first = (ts + CKPT - 1) >> CKPT_SHIFT;
                    const uint32_t idx = (tstar >> CKPT_SHIFT) - ck_first;
                    if (idx < dc.nck) ckpt[dc.ckbase + idx] = lc0 + n_kth_flag(NI, nth + 1);
                }
            }
            cnt = n_popc(E);
        }
        // t_pos of column c = t0 - 1 + the non-insertion columns up to and including c (a leading insertion column belongs
        // to t0 - 1); the lane's columns span at most 33 positions, i.e. at most two tiles
        auto t_of = [&](uint32_t col) -> uint32_t {
            const N128 thr = n_below(col + 1);
            return t0 + n_popc(N128{NI.lo & thr.lo, NI.hi & thr.hi}) - 1;
        };
        auto tile_slot = [&](uint32_t tile) -> uint32_t { // the tile's entry in the block's table (insert if new)
            uint32_t h = tile & (DENSE_TSLOTS - 1);
            for (;;) {
                const uint32_t old = atomicCAS(&s_tile[h], 0xFFFFFFFFu, tile);
                if (old == 0xFFFFFFFFu || old == tile) return h;
                h = (h + 1) & (DENSE_TSLOTS - 1);
            }
        };
        if (cnt) {
            const uint32_t col_f = n_ctz(E) >> 2;
            const uint32_t col_l = E.hi ? 16u + ((63u - (uint32_t)__builtin_clzll(E.hi)) >> 2) : (63u - (uint32_t)__builtin_clzll(E.lo)) >> 2;
            tA = t_of(col_f) >> TILE_SHIFT;
            const uint32_t P = (tA + 1) << TILE_SHIFT;
            n_lo = cnt;
            if (t_of(col_l) >= P) { // the (P - t0 + 1)-th non-insertion column is the first one at P
                const N128 lowm = n_below(n_kth_flag(NI, P - t0 + 1));
                n_lo = n_popc(N128{E.lo & lowm.lo, E.hi & lowm.hi});
            }
            // (a position >= L has no tile: the descriptor check reports the read)
            if (n_lo && tA < n_tiles) {
                sl_lo = tile_slot(tA);
                loc_lo = atomicAdd(&s_cnt[sl_lo], n_lo);
            }
            if (cnt - n_lo && tA + 1 < n_tiles) {
                sl_hi = tile_slot(tA + 1);
                loc_hi = atomicAdd(&s_cnt[sl_hi], cnt - n_lo);
            }
        }
        if (probe == 5) { // (everything up to the reservation; keep the counts alive)
            if (cnt == 0x7FFFFFFFu) atomicOr(err, 0x80000000u);
            return;
        }
        __syncthreads();
        // The block's place in every bucket, ONE atomic per tile — whose round trip to memory (agent scope: past the L2s)
        // nobody waits for: the records are formed meanwhile, into an LDS stage grouped by tile, and leave it as whole
        // contiguous pieces once the places are known.
        uint32_t gbase = 0;
        if (threadIdx.x < 64) { // (wave 0: DENSE_TSLOTS is one or two slots per lane)
            uint32_t c[DENSE_TSLOTS / 64], sum = 0;
#pragma unroll
            for (uint32_t k = 0; k < DENSE_TSLOTS / 64; ++k) {
                c[k] = s_cnt[threadIdx.x * (DENSE_TSLOTS / 64) + k];
                sum += c[k];
            }
            const uint32_t incl = wave_incl_scan<OpAdd>(sum);
            uint32_t o = incl - sum;
#pragma unroll
            for (uint32_t k = 0; k < DENSE_TSLOTS / 64; ++k) {
                const uint32_t sl = threadIdx.x * (DENSE_TSLOTS / 64) + k;
                s_off[sl] = o;
                o += c[k];
                if (c[k]) { // (k = 0 only keeps its result in a register; the rare second slot of a lane waits)
                    const uint32_t b = atomicAdd(&tile_cur[s_tile[sl]], c[k]);
                    if (k == 0) gbase = b; else s_base[sl] = b;
                }
            }
            if (threadIdx.x == 63) s_off[DENSE_TSLOTS] = incl;
        }
        __syncthreads();
        const uint32_t n_rec = s_off[DENSE_TSLOTS];
        const bool staged = n_rec <= DENSE_STAGE; // (uniform)
        if (!staged) { // more records than the stage holds (pileups far from HiFi statistics): straight to memory
            if (threadIdx.x < 64 && s_cnt[threadIdx.x * (DENSE_TSLOTS / 64)]) s_base[threadIdx.x * (DENSE_TSLOTS / 64)] = gbase;
            __syncthreads();
        }
        auto put = [&](uint32_t tile, uint32_t slot, uint64_t key, uint32_t val) {
            uint64_t dst;
            bool ok = true;
            if (slot < bucket_cap) {
                dst = (uint64_t)tile * bucket_cap + slot;
            } else { // the tile's bucket is full: spill (rare; the host then takes the device-wide sort)
                const uint32_t x = atomicAdd(ovf_cnt, 1u);
                dst = ovf_base + x;
                ok = x < ovf_cap;
            }
            if (ok) {
                out_keys[dst] = key;
                out_vals[dst] = val;
            }
        };
        if (cnt) {
            // (a position >= L has no tile: the descriptor check reports the read)
            const bool ok_lo = n_lo && tA < n_tiles, ok_hi = cnt - n_lo && tA + 1 < n_tiles;
            const uint32_t b_lo = (staged ? s_off[sl_lo] : s_base[sl_lo]) + loc_lo, b_hi = (staged ? s_off[sl_hi] : s_base[sl_hi]) + loc_hi;
            N128 e = E;
            for (uint32_t i = 0; i < cnt; ++i) {
                // t_pos of column c = t0 - 1 + the non-insertion columns up to and including c (a leading insertion column
                // belongs to t0 - 1)
                const N128 m = n_mask_through_first(e);
                const uint32_t col = (n_popc(N128{m.lo & NF3, m.hi & NF3})) - 1u;
                const uint32_t t = t0 + n_popc(N128{NI.lo & m.lo, NI.hi & m.hi}) - 1u;
                e.lo &= ~m.lo, e.hi &= ~m.hi;
                const bool hi = i >= n_lo;
                if (!(hi ? ok_hi : ok_lo)) continue;
                const uint32_t slot = hi ? b_hi + (i - n_lo) : b_lo + i;
                if (staged)
                    s_stage[slot] = make_uint2(t, q_tag | (col << 10) | ((hi ? sl_hi : sl_lo) << 15));
                else
                    put(hi ? tA + 1 : tA, slot, ((uint64_t)t << 32) | (lc0 + col), read);
            }
        }
        if (staged) {
            if (threadIdx.x < 64 && s_cnt[threadIdx.x * (DENSE_TSLOTS / 64)]) s_base[threadIdx.x * (DENSE_TSLOTS / 64)] = gbase;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n_rec; i += 256) {
                const uint2 r = s_stage[i];
                const uint32_t sl = r.y >> 15, pc = r.y & 1023u, col = (r.y >> 10) & 31u;
                const DenseChunk &dc = s_desc[pc >> 7];
                put(s_tile[sl], s_base[sl] + (i - s_off[sl]), ((uint64_t)r.x << 32) | (dc.c0 + (pc & (DENSE_HALVES - 1)) * 32u + col), dc.read);
            }
        }
        if (q0 + 256 < nq) __syncthreads(); // (the tables and the stage are reused by the next round)
    }
}

// The chunks' non-insertion column counts ahead of the dense pass (one wavefront per chunk, one pass over the nibbles):
// with every status word already carrying this launch's epoch no chunk of k_diff_reads ever waits for another.  The
// dense pass publishes the counts itself and normally needs no help — its waits are for lower-numbered, already
// running blocks —, except when ANOTHER PROCESS shares the device: a queue preempted by draining stops dispatching blocks
// that resident ones wait for (np2_lookback.hpp).  Used then (NP2_DENSE_PRECOUNT, or after a wait that gave up).
__device__ __forceinline__ void k_chunk_counts(const uint32_t np2_bid, const uint32_t np2_nb, const ChunkDesc *__restrict__ descs, uint32_t n_chunks,
                                               const uint8_t *__restrict__ nib, uint64_t *__restrict__ chunk_st, uint32_t epoch) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t ch = (uint32_t)__builtin_amdgcn_readfirstlane((int)(np2_bid * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (ch >= n_chunks) return;
    const uint32_t total = dense_chunk_total(descs + ch, nib, lane);
    if (lane == 0) __hip_atomic_store(&chunk_st[ch], ((uint64_t)epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
void launch_chunk_counts(hipStream_t s, const ChunkDesc *descs, uint32_t n_chunks, const uint8_t *nib, uint64_t *chunk_st, uint32_t epoch) {
    if (n_chunks) NP2_LAUNCH(k_chunk_counts, dim3((n_chunks + 3) / 4), 256, s, descs, n_chunks, nib, chunk_st, epoch);
}

void launch_diff_reads(hipStream_t s, const ChunkDesc *descs, uint32_t n_chunks, const uint8_t *nib,
                       const uint64_t *refw, const uint8_t *refnib, const uint8_t *refeo, uint32_t eo_stride, uint32_t L, uint64_t *keys, uint32_t *vals,
                       uint32_t *tile_cur, uint32_t n_tiles, uint32_t bucket_cap, uint64_t ovf_base, uint32_t ovf_cap,
                       uint32_t *ovf_cnt, uint32_t *ckpt, uint64_t *chunk_st, uint32_t epoch, uint32_t *err, uint32_t probe) {
    if (n_chunks)
        NP2_LAUNCH(k_diff_reads, dim3((n_chunks + DENSE_CHUNKS - 1) / DENSE_CHUNKS), 256, s, descs, n_chunks, nib, (const uint32_t *)refw, refnib, refeo, eo_stride, L, keys, vals, tile_cur, n_tiles, bucket_cap, ovf_base, ovf_cap, ovf_cnt, ckpt, chunk_st, epoch, err, probe);
}

} // namespace np2
