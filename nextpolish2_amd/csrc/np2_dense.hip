// Dense pass of the MI355X NextPolish2 hot path (gfx950, wave64): every packed read column is compared with the contig
// once (update_msas / Kmer::new / AlignSeq decode of the reference: src/main.rs:576-589, 84-102, 314-338), and only the
// *exception* columns — those whose 3-column-mer differs from the one the contig itself contributes — leave the kernel,
// as raw records (t_pos << 32 | column, read) filed under their contig tile, next to the per-read checkpoints.
//
// k_diff_reads is the kernel the benchmark's roofline is quoted on: 0.5 B per column + 0.5 B per contig base of
// algorithmic HBM traffic.  Round 2's version did everything wave-wide and was VALU-issue bound at ~500 VALU
// instructions per 2048-column chunk (12 % of the HBM peak), most of them on paths a wave enters because ONE of its 64
// lanes needs them (an insertion run, an exception to emit).  This version splits the work by what a lane holds:
//   * phase 1, every lane, branch-free: load 16 B, count non-insertion columns, wave scan -> t_pos, fetch the 32 contig
//     codes at t_pos, one nibble-domain compare.  A lane without insertion columns whose codes all match (94 % of
//     the lanes of a haploid pileup, ~80 % of a diploid one) is *done* after writing its checkpoint;
//   * every other lane ("dirty": a mismatch, an insertion, a bad column in the two columns before it, a read start) is
//     queued in LDS and handled in phase 2 by ONE THREAD per dirty lane, the block's threads taking the queue in
//     order: insertion runs shift the contig window, exact bad-column mask, exception mask, records, the checkpoint
//     of an insertion-bearing lane.  The divergent code runs in one or two densely packed wavefronts per block instead
//     of in all of them.
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
#define NP2_DENSE_CPW 4
#endif
static constexpr uint32_t DENSE_CPW = NP2_DENSE_CPW;                 // chunks per wavefront: their loads are all in flight together
static constexpr uint32_t DENSE_CHUNKS = 4 * DENSE_CPW;  // chunks per 256-thread block
static constexpr uint32_t DENSE_TSLOTS = DENSE_CPW > 4 ? 128 : 64;             // tile table of a block (>= 3 tiles per chunk)

__device__ __forceinline__ void k_diff_reads(const uint32_t np2_bid, const uint32_t np2_nb, const ChunkDesc *__restrict__ descs, uint32_t n_chunks, const uint8_t *__restrict__ nib,
    const uint32_t *__restrict__ refw32, const uint8_t *__restrict__ refnib, uint32_t L,
    uint64_t *__restrict__ out_keys, uint32_t *__restrict__ out_vals, uint32_t *__restrict__ tile_cur, uint32_t n_tiles,
    uint32_t bucket_cap, uint64_t ovf_base, uint32_t ovf_cap, uint32_t *__restrict__ ovf_cnt,
    uint32_t *__restrict__ ckpt, uint64_t *__restrict__ chunk_st, uint32_t epoch, uint32_t *__restrict__ err) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t pw = __builtin_amdgcn_readfirstlane(np2_bid * (blockDim.x >> 6) + (threadIdx.x >> 6));
    __shared__ uint32_t s_total[DENSE_CHUNKS];        // non-insertion columns of the block's chunks
    __shared__ uint2 s_q[DENSE_CHUNKS * 64];          // dirty lanes: {t0, chunk in block << 6 | lane}
    __shared__ __attribute__((aligned(16))) DenseChunk s_desc[DENSE_CHUNKS];
    __shared__ uint32_t s_nq;
    __shared__ uint32_t s_tile[DENSE_TSLOTS], s_cnt[DENSE_TSLOTS], s_base[DENSE_TSLOTS]; // phase 2: the tiles the block adds records to
    const uint32_t blk_first = np2_bid * DENSE_CHUNKS;
    if (threadIdx.x == 0) s_nq = 0;
    // A wave's time is a chain of memory round trips, not instructions (phase 1 issues ~160 VALU per chunk): everything
    // that can be requested together is.  Round trip 1: the descriptors of the wave's chunks, one per lane, handed
    // round by readlane; 2: the chunks' 16 bytes per lane; 3 (after the block-wide exchange of the chunk totals): the
    // status words of chunks in earlier blocks, where a read started there; 4: the contig windows of all chunks.
    struct Desc {
        uint64_t nib_off;
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
            dd[it].nib_off = dp->nib_off;
            dd[it].read = dp->read, dd[it].ts = dp->ts, dd[it].c0 = dp->c0, dd[it].ncols = dp->ncols;
            dd[it].first_chunk = dp->first_chunk; // (ckbase, nck, aln_t_e: read from the LDS copy where they are used —
            // held in scalar registers across both phases they were a good part of ~100 spill moves per wave)
            dd[it].live = ch0 + it < n_chunks;
        }
    }
    // ---- phase A: load the chunks, count their non-insertion columns and publish the counts at once.  A chunk needs
    //      the counts of the read's earlier chunks (status word: launch epoch | count); they belong to lower-numbered,
    //      already running waves, which publish within a microsecond of starting ---------------------------------------
    N128 w_[DENSE_CPW];
    uint32_t nv_[DENSE_CPW], nonins_[DENSE_CPW], incl_[DENSE_CPW], total_[DENSE_CPW];
    uint4 v_[DENSE_CPW];
    bool full_[DENSE_CPW];
#pragma unroll
    for (uint32_t it = 0; it < DENSE_CPW; ++it) { // (all loads first)
        const uint32_t lc0 = dd[it].c0 + lane * 32;
        const bool full = dd[it].live && dd[it].ncols - dd[it].c0 >= 2048; // every lane of the wave holds 32 columns
        full_[it] = full;
        nv_[it] = !dd[it].live ? 0u : (full ? 32u : (lc0 < dd[it].ncols ? min(32u, dd[it].ncols - lc0) : 0u));
        v_[it] = make_uint4(0, 0, 0, 0);
        if (nv_[it]) v_[it] = *reinterpret_cast<const uint4 *>(nib + dd[it].nib_off + (lc0 >> 1));
    }
#pragma unroll
    for (uint32_t it = 0; it < DENSE_CPW; ++it) {
        const uint32_t ch = DENSE_CPW * pw + it;
        const uint32_t nv = nv_[it];
        N128 w;
        w.lo = (uint64_t)swap_nib(v_[it].x) | ((uint64_t)swap_nib(v_[it].y) << 32);
        w.hi = (uint64_t)swap_nib(v_[it].z) | ((uint64_t)swap_nib(v_[it].w) << 32);
        if (dd[it].c0 == 0 && lane == 0) w.lo &= ~8ULL; // column 0 is never an insertion column (main.rs:325,332-335)
        N128 I{w.lo & NF3, w.hi & NF3}; // insertion columns
        if (!full_[it]) { // (uniform: 97 % of the chunks are whole, their lanes need no column masks)
            const N128 m = n_below(nv);
            I.lo &= m.lo, I.hi &= m.hi;
        }
        const uint32_t nonins = nv - n_popc(I);
        const uint32_t incl = wave_incl_scan<OpAdd>(nonins);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (lane == 0 && dd[it].live) {
            __hip_atomic_store(&chunk_st[ch], ((uint64_t)epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_total[ch - blk_first] = total;
        }
        w_[it] = w;
        nonins_[it] = nonins, incl_[it] = incl, total_[it] = total;
    }
    __syncthreads();
    // ---- phase B: t_pos, contig codes, compare; clean lanes finish here, dirty ones are queued ----------------------------
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
                const uint32_t lo = max(dd[it].first_chunk, blk_first) - blk_first, hi = ch - blk_first; // (hi <= DENSE_CHUNKS <= 64)
                uint32_t v = lane >= lo && lane < hi ? s_total[lane] : 0u;
                if (lo < hi) {
                    if (DENSE_CHUNKS <= 16) { // a row's worth of lanes: four DPP steps, lane 15 holds the sum
                        v += dpp_get<0x111, 0xF>(0u, v);
                        v += dpp_get<0x112, 0xF>(0u, v);
                        v += dpp_get<0x114, 0xF>(0u, v);
                        v += dpp_get<0x118, 0xF>(0u, v);
                        carryN += (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
                    } else {
                        carryN += (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan<OpAdd>(v), 63);
                    }
                }
            }
            const uint32_t jend = min(ch, blk_first);
            for (uint32_t j0 = dd[it].first_chunk; j0 < jend; j0 += 64) {
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
    // the 32 contig codes starting at t0 (t_pos of the lane's first non-insertion column): one unaligned 20-byte window
    // of the nibble-packed contig per chunk, all of them requested before the first is used
    uint32_t t0_[DENSE_CPW], r_[DENSE_CPW][5];
#pragma unroll
    for (uint32_t it = 0; it < DENSE_CPW; ++it) {
        t0_[it] = dd[it].ts + carry_[it] + (incl_[it] - nonins_[it]);
        // (a stream that disagrees with its descriptor could push t0 past the contig: stay inside the padded buffer;
        // such a read is reported by the descriptor check at the end of its last chunk)
        const uint32_t q = min(t0_[it] >> 3, (L >> 3) + 8);
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) r_[it][k] = refw32[q + k];
    }
    uint32_t prev_top = 0; // flags of the last two columns of the previous chunk (when this one continues it)
#pragma unroll
    for (uint32_t it = 0; it < DENSE_CPW; ++it) {
        const uint32_t ch = DENSE_CPW * pw + it;
        if (!dd[it].live) break;
        const uint8_t *base = nib + dd[it].nib_off; // start of the READ's stream
        const uint32_t ncols = dd[it].ncols, ts = dd[it].ts, c0 = dd[it].c0;
        const uint32_t lc0 = c0 + lane * 32;
        const N128 w = w_[it];
        const uint32_t nv = nv_[it], nonins = nonins_[it], total = total_[it], carryN = carry_[it], t0 = t0_[it];
        const bool cont = it != 0 && dd[it].read == dd[it ? it - 1 : 0].read && c0 != 0;
        N128 R;
        {
            const uint32_t sh = (t0 & 7) * 4;
            const uint32_t a0 = __builtin_amdgcn_alignbit(r_[it][1], r_[it][0], sh), a1 = __builtin_amdgcn_alignbit(r_[it][2], r_[it][1], sh);
            const uint32_t a2 = __builtin_amdgcn_alignbit(r_[it][3], r_[it][2], sh), a3 = __builtin_amdgcn_alignbit(r_[it][4], r_[it][3], sh);
            R.lo = (uint64_t)a0 | ((uint64_t)a1 << 32);
            R.hi = (uint64_t)a2 | ((uint64_t)a3 << 32);
        }
        // bad columns as far as a lane WITHOUT insertion columns is concerned (exact for those lanes; a lane with an
        // insertion column is dirty whatever the rest says): code differs from the contig (nibble != 0 -> + 7 carries
        // into bit 3) or the insertion flag itself; columns past the read's end are cleared
        N128 B0;
        B0.lo = ((((w.lo ^ R.lo) & ~NF3) + ~NF3) | w.lo) & NF3;
        B0.hi = ((((w.hi ^ R.hi) & ~NF3) + ~NF3) | w.hi) & NF3;
        if (!full_[it]) {
            const N128 m = n_below(nv);
            B0.lo &= m.lo, B0.hi &= m.hi;
        }
        const uint32_t n_ins = nv - nonins;
        // checkpoint: column of the reference column at the next multiple of CKPT (lanes without insertion columns:
        // column index and position advance together)
        {
            const uint32_t tstar = (t0 + CKPT - 1) & ~(CKPT - 1);
            const uint32_t nth = tstar - t0; // 0-based index among the lane's non-insertion columns
            if (n_ins == 0 && nth < nonins) {
                const uint32_t ck_first = (ts + CKPT - 1) >> CKPT_SHIFT;
                const uint32_t idx = (tstar >> CKPT_SHIFT) - ck_first;
                const DenseChunk &dl = s_desc[ch - blk_first];
                if (idx < dl.nck) ckpt[dl.ckbase + idx] = lc0 + nth;
            }
        }
        // A bad column marks itself and the two columns after it (3-column-mers): the flags of the lane's last two
        // columns reach into the next lane.  An insertion-bearing lane's flags are not exact here: assume the worst.
        const uint32_t top = n_ins ? 0x88000000u : (uint32_t)(B0.hi >> 32);
        uint32_t pb = wave_prev_lane(0u, top);
        if (lane == 0) pb = cont ? prev_top : (c0 > 0 ? 0x88000000u : 0u); // (chunk start inside a read: phase 2 looks)
        const bool dirty = nv != 0 && ((B0.lo | B0.hi) != 0 || (pb & 0x88000000u) != 0 || (lc0 == 0 && ts != 0));
        prev_top = (uint32_t)__builtin_amdgcn_readlane((int)top, 63);
        const uint64_t dm = __ballot(dirty);
        if (dm) {
            uint32_t qb = 0;
            if (lane == 0) qb = atomicAdd(&s_nq, (uint32_t)__builtin_popcountll(dm));
            qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
            if (dirty) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));
                s_q[qb + rank] = make_uint2(t0, ((ch - blk_first) << 6) | lane);
            }
        }
        if (lane == 0) {
            if (c0 + 2048 >= ncols) {
                // last chunk: the packed stream must agree with its descriptor (AlignSeq::new, main.rs:279-312)
                const uint32_t te = s_desc[ch - blk_first].aln_t_e;
                if (ncols == 0 || ts + carryN + total - 1 != te || te >= L) atomicOr(err, 2u);
                if ((nib_at(base, ncols) & 15) != 15) atomicOr(err, 2u);
            }
        }
    }
    __syncthreads();
    // ---- phase 2: one thread per dirty lane ---------------------------------------------------------------------------------
    // Records go straight into the bucket of their contig tile; the block reserves its place in a bucket ONCE per tile
    // (a 32-entry tile table in LDS: the block's 8 chunks touch at most 24 tiles), so that what a block adds to a tile
    // is one contiguous piece written through one L2.  (Reserving per lane left every bucket line shared by fragments
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
        uint32_t t0 = 0, lc0 = 0, read = 0, cnt = 0, n_lo = 0, tA = 0, sl_lo = 0, sl_hi = 0, loc_lo = 0, loc_hi = 0;
        if (qi < nq) {
            const uint2 qe = s_q[qi];
            const uint32_t ln = qe.y & 63;
            t0 = qe.x;
            const DenseChunk dc = s_desc[qe.y >> 6];
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
            // checkpoint of a lane with insertion columns: the (nth + 1)-th non-insertion column
            if (nonins != nv) {
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
        __syncthreads();
        if (threadIdx.x < DENSE_TSLOTS && s_tile[threadIdx.x] != 0xFFFFFFFFu)
            s_base[threadIdx.x] = atomicAdd(&tile_cur[s_tile[threadIdx.x]], s_cnt[threadIdx.x]);
        __syncthreads();
        if (cnt) {
            const uint32_t b_lo = s_base[sl_lo] + loc_lo, b_hi = s_base[sl_hi] + loc_hi;
            N128 e = E;
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint32_t col = n_ctz(e) >> 2;
                if (e.lo) e.lo &= e.lo - 1ULL; else e.hi &= e.hi - 1ULL;
                const uint32_t t = t_of(col);
                const bool hi = i >= n_lo;
                const uint32_t tile = hi ? tA + 1 : tA;
                const uint32_t slot = hi ? b_hi + (i - n_lo) : b_lo + i;
                bool ok = tile < n_tiles;
                uint64_t dst = 0;
                if (ok) {
                    if (slot < bucket_cap) {
                        dst = (uint64_t)tile * bucket_cap + slot;
                    } else { // the tile's bucket is full: spill (rare; the host then takes the device-wide sort)
                        const uint32_t x = atomicAdd(ovf_cnt, 1u);
                        dst = ovf_base + x;
                        ok = x < ovf_cap;
                    }
                }
                if (ok) {
                    out_keys[dst] = ((uint64_t)t << 32) | (lc0 + col);
                    out_vals[dst] = read;
                }
            }
        }
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
    const ChunkDesc *dp = descs + ch;
    const uint32_t c0 = dp->c0, ncols = dp->ncols;
    const uint32_t lc0 = c0 + lane * 32;
    const uint32_t nv = lc0 < ncols ? min(32u, ncols - lc0) : 0u;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (nv) v = *reinterpret_cast<const uint4 *>(nib + dp->nib_off + (lc0 >> 1));
    N128 w;
    w.lo = (uint64_t)swap_nib(v.x) | ((uint64_t)swap_nib(v.y) << 32);
    w.hi = (uint64_t)swap_nib(v.z) | ((uint64_t)swap_nib(v.w) << 32);
    if (c0 == 0 && lane == 0) w.lo &= ~8ULL; // column 0 is never an insertion column (main.rs:325,332-335)
    const N128 m = n_below(nv);
    const N128 I{w.lo & NF3 & m.lo, w.hi & NF3 & m.hi};
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan<OpAdd>(nv - n_popc(I)), 63);
    if (lane == 0) __hip_atomic_store(&chunk_st[ch], ((uint64_t)epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
void launch_chunk_counts(hipStream_t s, const ChunkDesc *descs, uint32_t n_chunks, const uint8_t *nib, uint64_t *chunk_st, uint32_t epoch) {
    if (n_chunks) NP2_LAUNCH(k_chunk_counts, dim3((n_chunks + 3) / 4), 256, s, descs, n_chunks, nib, chunk_st, epoch);
}

void launch_diff_reads(hipStream_t s, const ChunkDesc *descs, uint32_t n_chunks, const uint8_t *nib,
                       const uint64_t *refw, const uint8_t *refnib, uint32_t L, uint64_t *keys, uint32_t *vals,
                       uint32_t *tile_cur, uint32_t n_tiles, uint32_t bucket_cap, uint64_t ovf_base, uint32_t ovf_cap,
                       uint32_t *ovf_cnt, uint32_t *ckpt, uint64_t *chunk_st, uint32_t epoch, uint32_t *err) {
    if (n_chunks)
        NP2_LAUNCH(k_diff_reads, dim3((n_chunks + DENSE_CHUNKS - 1) / DENSE_CHUNKS), 256, s, descs, n_chunks, nib, (const uint32_t *)refw, refnib, L, keys, vals, tile_cur, n_tiles, bucket_cap, ovf_base, ovf_cap, ovf_cnt, ckpt, chunk_st, epoch, err);
}

} // namespace np2
