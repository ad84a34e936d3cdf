// Candidate extraction (get_lqseqs_from_align_tags, src/main.rs:1433-1530), region-major.
//
// Every live read covers a contiguous interval [pj, pj + pcount) of the (descending) LQ region list.  One wavefront
// per region finds its reads through the per-tile read lists built at upload, measures their candidate strings
// 8 columns at a time, keeps the first 60 non-empty ones (main.rs:1474,1509) in per-region slots, and a second pass
// writes strings and first k-mers once the offsets are known.  No (read, region) pair list is ever materialised.
#include "np2_common.hpp"
#include "np2_kernels.hpp"
#include "np2_blockscan.hpp"
#include "np2_lookback.hpp"

namespace np2 {

// ------------------------------------------------------------------------------------------------------
// single-block scans for short arrays (region / read / pair counts): no temp storage, no init launch, the
// element count may live on the device
// ------------------------------------------------------------------------------------------------------
template <int MODE> // 0: exclusive sum (optionally out[n] = total), 1: inclusive sum, 2: inclusive min (signed)
__device__ __forceinline__ void k_scan_small(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                     uint32_t n_host, const uint32_t *__restrict__ n_dev,
                                                     uint32_t *__restrict__ total_out, bool write_end) {
    __shared__ uint32_t sh[16];
    const uint32_t n = n_dev ? min(*n_dev, n_host) : n_host;
    uint32_t total;
    if (MODE == 2) {
        total = block_scan_array<OpMinI32>(
            n, sh, [&](uint32_t i) { return in[i]; },
            [&](uint32_t i, uint32_t pre, uint32_t v) { out[i] = OpMinI32::apply(pre, v); });
    } else {
        total = block_scan_array<OpAdd>(
            n, sh, [&](uint32_t i) { return in[i]; },
            [&](uint32_t i, uint32_t pre, uint32_t v) { out[i] = MODE == 0 ? pre : pre + v; });
    }
    if (threadIdx.x == 0) {
        if (MODE == 0 && write_end) out[n] = total;
        if (total_out) *total_out = total;
    }
}

// Exclusive sums whose length is known on the host, any length: blocks of 1024 threads x 8 elements chained by the
// decoupled look-back (relaxed status words, np2_lookback.hpp).  One kernel for every length on purpose: the contigs of a
// batch must not pick different kernels for the same step (their queues fall out of step and every later stage goes
// out in more, smaller launches), and a 1.5 M-element scan is 183 blocks = three look-back hops.
static constexpr uint32_t SCAN_LB_ITEMS = 8;   // elements per thread, blocked: thread t owns [8t, 8t + 8) of its block's 8192
static constexpr uint32_t SCAN_LB_THREADS = 1024;
static constexpr uint32_t SCAN_LB_BLOCK = SCAN_LB_ITEMS * SCAN_LB_THREADS;
// (POPC: the elements are the population counts of the words of `in` — the LQ head bitmap —, element n - 1 is a closing 0)
template <bool POPC>
__device__ __forceinline__ void scan_lb_excl_body(const uint32_t np2_bid, const uint32_t np2_nb, Lookback lb, uint32_t n_blocks, const uint32_t *__restrict__ in,
                                                      uint32_t *__restrict__ out, uint32_t n, bool write_end,
                                                      uint32_t *__restrict__ err) {
    __shared__ uint32_t sh[16];
    const uint32_t bid = lb_block_id(lb, sh);
    const uint32_t i0 = bid * SCAN_LB_BLOCK + threadIdx.x * SCAN_LB_ITEMS;
    uint32_t v[SCAN_LB_ITEMS], sum = 0;
    const bool wide = (((uintptr_t)in | (uintptr_t)out) & 15) == 0; // (uniform) 16-byte accesses for whole octets
    if (wide && i0 + SCAN_LB_ITEMS <= (POPC ? n - 1 : n)) {
        const uint4 a = *reinterpret_cast<const uint4 *>(in + i0), b = *reinterpret_cast<const uint4 *>(in + i0 + 4);
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < SCAN_LB_ITEMS; ++k) v[k] = i0 + k < (POPC ? n - 1 : n) ? in[i0 + k] : 0u;
    }
    if (POPC) {
#pragma unroll
        for (uint32_t k = 0; k < SCAN_LB_ITEMS; ++k) v[k] = (uint32_t)__builtin_popcount(v[k]);
    }
#pragma unroll
    for (uint32_t k = 0; k < SCAN_LB_ITEMS; ++k) sum += v[k];
    uint32_t total, pre, dummy;
    uint32_t run = block_excl_scan<OpAdd, 16>(sum, sh, total);
    lb_exclusive2(lb, bid, total, 0u, sh, err, pre, dummy);
    run += pre;
    if (wide && i0 + SCAN_LB_ITEMS <= n) {
        uint4 a, b;
        a.x = run, a.y = a.x + v[0], a.z = a.y + v[1], a.w = a.z + v[2];
        b.x = a.w + v[3], b.y = b.x + v[4], b.z = b.y + v[5], b.w = b.z + v[6];
        *reinterpret_cast<uint4 *>(out + i0) = a;
        *reinterpret_cast<uint4 *>(out + i0 + 4) = b;
        run = b.w + v[7];
    } else {
#pragma unroll
        for (uint32_t k = 0; k < SCAN_LB_ITEMS; ++k) {
            if (i0 + k < n) out[i0 + k] = run;
            run += v[k];
        }
    }
    // (the thread holding element n - 1 ends with the total)
    if (write_end && n && i0 <= n - 1 && n - 1 < i0 + SCAN_LB_ITEMS) out[n] = run;
    if (write_end && n == 0 && bid == 0 && threadIdx.x == 0) out[0] = 0;
}

__device__ __forceinline__ void k_scan_lb_excl(const uint32_t np2_bid, const uint32_t np2_nb, Lookback lb, uint32_t n_blocks, const uint32_t *__restrict__ in,
                                               uint32_t *__restrict__ out, uint32_t n, bool write_end, uint32_t *__restrict__ err) {
    scan_lb_excl_body<false>(np2_bid, np2_nb, lb, n_blocks, in, out, n, write_end, err);
}
__device__ __forceinline__ void k_scan_lb_popc(const uint32_t np2_bid, const uint32_t np2_nb, Lookback lb, uint32_t n_blocks, const uint32_t *__restrict__ in,
                                               uint32_t *__restrict__ out, uint32_t n, bool write_end, uint32_t *__restrict__ err) {
    scan_lb_excl_body<true>(np2_bid, np2_nb, lb, n_blocks, in, out, n, write_end, err);
}

// Long exclusive sums (one element per contig position / consensus base / band slot): reduce-then-scan.  Tiles of 4096
// elements: (1) tile sums, (2) a single-block scan of the tile sums (k_scan_small), (3) tile-local scans seeded with
// the tile offsets.  12 B of traffic per element, three launches whatever the number of contigs in the batch, no
// inter-block waiting (a chained look-back over thousands of tiles measured 100x slower here).
static constexpr uint32_t SCAN3_ITEMS = 16;
static constexpr uint32_t SCAN3_TILE = 256 * SCAN3_ITEMS;
__device__ __forceinline__ void scan3_load(const uint32_t *__restrict__ in, uint32_t i0, uint32_t n, uint32_t (&v)[SCAN3_ITEMS]) {
    if (i0 + SCAN3_ITEMS <= n) {
#pragma unroll
        for (uint32_t q = 0; q < SCAN3_ITEMS / 4; ++q) {
            const uint4 w = *reinterpret_cast<const uint4 *>(in + i0 + 4 * q);
            v[4 * q] = w.x, v[4 * q + 1] = w.y, v[4 * q + 2] = w.z, v[4 * q + 3] = w.w;
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < SCAN3_ITEMS; ++k) v[k] = i0 + k < n ? in[i0 + k] : 0u;
    }
}
__device__ __forceinline__ void k_scan3_part(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ in, uint32_t n,
                                             uint32_t *__restrict__ part) {
    __shared__ uint32_t sh[4];
    uint32_t v[SCAN3_ITEMS], sum = 0;
    scan3_load(in, np2_bid * SCAN3_TILE + threadIdx.x * SCAN3_ITEMS, n, v);
#pragma unroll
    for (uint32_t k = 0; k < SCAN3_ITEMS; ++k) sum += v[k];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) part[np2_bid] = sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ void k_scan3_apply(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                              uint32_t n, const uint32_t *__restrict__ part_off, bool write_end) {
    __shared__ uint32_t sh[4];
    const uint32_t i0 = np2_bid * SCAN3_TILE + threadIdx.x * SCAN3_ITEMS;
    uint32_t v[SCAN3_ITEMS], sum = 0;
    scan3_load(in, i0, n, v);
#pragma unroll
    for (uint32_t k = 0; k < SCAN3_ITEMS; ++k) sum += v[k];
    uint32_t total;
    uint32_t run = part_off[np2_bid] + block_excl_scan<OpAdd, 4>(sum, sh, total);
    if (i0 + SCAN3_ITEMS <= n) {
#pragma unroll
        for (uint32_t q = 0; q < SCAN3_ITEMS / 4; ++q) {
            uint4 w;
            w.x = run, run += v[4 * q];
            w.y = run, run += v[4 * q + 1];
            w.z = run, run += v[4 * q + 2];
            w.w = run, run += v[4 * q + 3];
            *reinterpret_cast<uint4 *>(out + i0 + 4 * q) = w;
        }
        if (write_end && i0 + SCAN3_ITEMS == n) out[n] = run;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < SCAN3_ITEMS; ++k) {
            if (i0 + k < n) out[i0 + k] = run;
            run += v[k];
            if (write_end && i0 + k + 1 == n) out[n] = run;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// candidate decode
// ------------------------------------------------------------------------------------------------------
struct CandCtx {
    const np2_read_t *reads;
    const uint8_t *nib;
    const uint64_t *ck_off;
    const uint32_t *ckpt;
    const uint32_t *lq_start;
    const uint32_t *lq_end;
    const uint32_t *pj;
    const uint32_t *pcount;
    const uint8_t *alive;
    const ReadInfo *rinfo;
    const uint32_t *tile_rd_off; // reads overlapping each contig tile, ascending read index (built at upload)
    const uint32_t *tile_rd;
    uint32_t n_tiles;
    uint32_t ksize;
    const uint64_t *rec_key;  // sorted exception records by contig tile (bucketed layout) ...
    const uint32_t *rec_read;
    const uint32_t *tile_n;
    const uint16_t *rec_pidx; // ... and where the records of every 16th position of a tile begin (nullptr: not available)
    uint32_t bucket_cap;
    const uint32_t *refnib;
    uint32_t L;
};

// checkpoint lookup: a non-insertion column at or before the first column of t_pos == start, and its t_pos
__device__ __forceinline__ void cand_anchor(const CandCtx &cx, uint32_t r, const np2_read_t &rd, uint64_t ck_off, uint32_t start,
                                            uint32_t &col, uint32_t &t) {
    col = 0;
    t = rd.aln_t_s;
    const uint32_t ck_first = (rd.aln_t_s + CKPT - 1) >> CKPT_SHIFT;
    const uint32_t cki = start >> CKPT_SHIFT;
    if (r == 0) { // the contig aligned to itself: column index == position (the dense pass skips read 0)
        col = start;
        t = start;
    } else if (cki >= ck_first) {
        col = cx.ckpt[ck_off + (cki - ck_first)];
        t = cki << CKPT_SHIFT;
    }
}

// position (0..15) of the n-th (1-based) set bit of a nibble-spaced 64-bit mask (bits 0, 4, 8, ...): binary descent
__device__ __forceinline__ uint32_t nth_col16(uint64_t m64, uint32_t n) {
    uint32_t col = 0;
    uint32_t m = (uint32_t)m64;
    uint32_t c = (uint32_t)__builtin_popcount(m);
    if (n > c) n -= c, col = 8, m = (uint32_t)(m64 >> 32);
    c = (uint32_t)__builtin_popcount(m & 0xFFFFu);
    if (n > c) n -= c, col += 4, m >>= 16;
    c = (uint32_t)__builtin_popcount(m & 0xFFu);
    if (n > c) n -= c, col += 2, m >>= 8;
    c = (uint32_t)__builtin_popcount(m & 0xFu);
    if (n > c) col += 1;
    return col;
}

// Length of the candidate string of (read, region) and the column it starts at, 16 columns per step
// (the emission rule of main.rs:1478-1521: every non-gap column whose t_pos lies in [start, end], from the reference
// column of `start` on; the early exits of that loop all sit behind t_pos > end).
__device__ uint32_t cand_measure(const CandCtx &cx, uint32_t r, const np2_read_t &rd, uint64_t ck_off, uint32_t start,
                                 uint32_t end, uint32_t &col_start) {
    uint32_t col_ck, t_ck;
    cand_anchor(cx, r, rd, ck_off, start, col_ck, t_ck);
    col_start = col_ck;
    if (end < t_ck) return 0;                                   // the read begins behind the region
    const uint32_t want_s = (start > t_ck ? start - t_ck : 0u) + 1; // col_start = want_s-th non-insertion column from col_ck
    const uint32_t want_e = end + 1 - t_ck + 1;                 // first column past the region, counted the same way
    const uint8_t *base = cx.nib + rd.nib_off;
    const uint32_t n_cols = rd.n_cols;
    uint32_t seen = 0, len = 0;
    bool started = false;
    uint4 win = make_uint4(0, 0, 0, 0);
    uint32_t win_blk = 0xFFFFFFFFu;
    const uint64_t N1 = 0x1111111111111111ULL;
    for (uint32_t wi = col_ck >> 4;; ++wi) { // 16 columns = 8 bytes of the read's stream per step
        const uint32_t c0 = wi << 4;
        if (c0 >= n_cols) break;
        if ((wi >> 1) != win_blk) {
            win_blk = wi >> 1;
            win = *reinterpret_cast<const uint4 *>(base + ((size_t)win_blk << 4));
        }
        const uint32_t r0 = (wi & 1) ? win.z : win.x, r1 = (wi & 1) ? win.w : win.y;
        const uint64_t w = (uint64_t)(((r0 & 0x0F0F0F0Fu) << 4) | ((r0 >> 4) & 0x0F0F0F0Fu)) |
                           ((uint64_t)(((r1 & 0x0F0F0F0Fu) << 4) | ((r1 >> 4) & 0x0F0F0F0Fu)) << 32); // column j at bits 4j
        uint64_t valid = N1;
        if (c0 < col_ck) valid &= ~((1ULL << (4 * (col_ck - c0))) - 1ULL);          // columns before the anchor
        if (n_cols - c0 < 16) valid &= (1ULL << (4 * (n_cols - c0))) - 1ULL;        // columns past the read
        uint64_t nonins = (~(w >> 3)) & valid;
        if (c0 == 0) nonins |= valid & 1ULL; // column 0 is never an insertion column (main.rs:325,332-335)
        const uint64_t x = (w & 0x7777777777777777ULL) ^ 0x4444444444444444ULL;
        const uint64_t nongap = (x | (x >> 1) | (x >> 2)) & valid;
        const uint32_t cn = (uint32_t)__builtin_popcountll(nonins);
        uint32_t lo = 0, hi = 16;
        bool done = false;
        if (!started && seen + cn >= want_s) {
            lo = nth_col16(nonins, want_s - seen);
            col_start = c0 + lo;
            started = true;
        }
        if (seen + cn >= want_e) {
            hi = nth_col16(nonins, want_e - seen);
            done = true;
        }
        if (started) {
            uint64_t m = nongap;
            if (lo) m &= ~((1ULL << (4 * lo)) - 1ULL);
            if (hi < 16) m &= (1ULL << (4 * hi)) - 1ULL;
            len += (uint32_t)__builtin_popcountll(m);
        }
        seen += cn;
        if (done) break;
    }
    return started ? len : 0u;
}

// sequential nibble reader with a 16-column (8-byte) register window; `base` is 16-byte aligned
struct NibReader {
    const uint8_t *base;
    uint64_t w;     // current window, nibble of column c at bits 4*(c & 15)
    uint32_t wbase; // first column of the window (multiple of 16), 0xFFFFFFFF = empty
    __device__ __forceinline__ uint8_t get(uint32_t c) {
        const uint32_t b = c & ~15u;
        if (b != wbase) {
            const uint2 v = *reinterpret_cast<const uint2 *>(base + (b >> 1));
            const uint32_t x = ((v.x & 0x0F0F0F0Fu) << 4) | ((v.x >> 4) & 0x0F0F0F0Fu);
            const uint32_t y = ((v.y & 0x0F0F0F0Fu) << 4) | ((v.y >> 4) & 0x0F0F0F0Fu);
            w = (uint64_t)x | ((uint64_t)y << 32);
            wbase = b;
        }
        return (uint8_t)((w >> (4 * (c & 15))) & 15);
    }
};

// Write the candidate string and the hashed first k-mer (main.rs:1478-1521), starting at the candidate's first
// column `col` whose t_pos is `t`.  The string is the first `len` non-gap columns from `col` (cand_measure counted
// exactly those), so emission needs no t_pos tracking and goes 8 columns per step: nibbles -> byte selectors ->
// ASCII through v_perm_b32 with the 8-entry code table in two registers, dword stores.  Only words holding a gap code
// or the string's end take the per-column path.  The k-mer needs the first k non-gap codes and the decode limit.
__device__ void cand_write(const CandCtx &cx, uint32_t r, const np2_read_t &rd, uint32_t pj, uint32_t g, uint32_t col, uint32_t t,
                           uint32_t len, uint8_t *__restrict__ seq_out, uint64_t *kmer_out) {
    const uint8_t *base = cx.nib + rd.nib_off;
    // ---- first k-mer -------------------------------------------------------------------------------------------------
    const uint32_t limit = cx.lq_end[pj] + cx.ksize; // decode stops after t_pos > end[j] + k (main.rs:1467)
    bool have_kmer = false;
    if (col + cx.ksize <= rd.n_cols) {
        // Common cases without the per-column loop: the first k non-gap codes sit inside the 32 columns from `col` and
        // none of the columns up to the k-th one holds N / M (codes 5 / 6, whose third bit would smear into the
        // neighbour, main.rs:1488-1492).  48 columns as three 8-byte words -> the 32 columns from `col`; gap columns
        // (code 4: the read deletes a contig base) are taken out one at a time — a candidate has none or a few —, then
        // the 2-bit codes are squeezed together; the forward k-mer is their pair-wise bit reversal, the reverse
        // complement their complement in place (kmer.rs:255-287).
        const uint32_t k = cx.ksize, W0 = col >> 4, sh = (col & 15) * 4;
        auto ld = [&](uint32_t W) -> uint64_t {
            const uint2 v = *reinterpret_cast<const uint2 *>(base + ((size_t)W << 3));
            const uint32_t x = ((v.x & 0x0F0F0F0Fu) << 4) | ((v.x >> 4) & 0x0F0F0F0Fu);
            const uint32_t y = ((v.y & 0x0F0F0F0Fu) << 4) | ((v.y >> 4) & 0x0F0F0F0Fu);
            return (uint64_t)x | ((uint64_t)y << 32);
        };
        const uint64_t x0 = ld(W0), x1 = ld(W0 + 1), x2 = ld(W0 + 2);
        uint64_t lo = sh ? (x0 >> sh) | (x1 << (64 - sh)) : x0, hi = sh ? (x1 >> sh) | (x2 << (64 - sh)) : x1;
        auto nmask = [](uint32_t n) -> uint64_t { return n >= 16 ? ~0ULL : ((1ULL << (4 * n)) - 1ULL); }; // nibbles [0, n)
        const uint64_t N1 = 0x1111111111111111ULL;
        // the columns the read still has (at most 32 of them here)
        const uint32_t avail = min(32u, rd.n_cols - col);
        const uint64_t a_lo = nmask(avail), a_hi = avail > 16 ? nmask(avail - 16) : 0ULL;
        const uint64_t ng_lo = ~(lo >> 2) & N1 & a_lo, ng_hi = ~(hi >> 2) & N1 & a_hi; // non-gap-ish columns (code < 4)
        const uint32_t c_lo = (uint32_t)__builtin_popcountll(ng_lo);
        if (c_lo + (uint32_t)__builtin_popcountll(ng_hi) >= k) {
            // C: column (0 .. 31) of the k-th code below 4
            uint32_t kk = k, C = 0;
            uint64_t m = ng_lo;
            if (kk > c_lo) kk -= c_lo, C = 16, m = ng_hi;
            uint32_t c = (uint32_t)__builtin_popcount((uint32_t)m);
            if (kk > c) kk -= c, C += 8, m >>= 32;
            uint32_t m32 = (uint32_t)m;
            c = (uint32_t)__builtin_popcount(m32 & 0xFFFFu);
            if (kk > c) kk -= c, C += 4, m32 >>= 16;
            c = (uint32_t)__builtin_popcount(m32 & 0xFFu);
            if (kk > c) kk -= c, C += 2, m32 >>= 8;
            c = (uint32_t)__builtin_popcount(m32 & 0xFu);
            if (kk > c) C += 1;
            const uint64_t u_lo = nmask(C + 1), u_hi = C + 1 > 16 ? nmask(C + 1 - 16) : 0ULL; // columns 0 .. C
            // codes >= 4 among them must all be plain gaps (4): bit 2 set, bits 0 and 1 clear
            const uint64_t hi4_lo = (lo >> 2) & N1 & u_lo, hi4_hi = (hi >> 2) & N1 & u_hi;
            const bool plain = (((lo | (lo >> 1)) & hi4_lo) | ((hi | (hi >> 1)) & hi4_hi)) == 0;
            // the loop also ends once t_pos runs past `limit`: harmless iff that has not happened by column C - 1
            // (columns 1 .. C - 1 advance t_pos unless they are insertion columns)
            const uint64_t r_lo = nmask(C) & ~0xFULL, r_hi = C > 16 ? nmask(C - 16) : 0ULL;
            const uint32_t ins = (uint32_t)__builtin_popcountll(lo & r_lo & 0x8888888888888888ULL) +
                                 (uint32_t)__builtin_popcountll(hi & r_hi & 0x8888888888888888ULL);
            const uint32_t steps = C ? C - 1 : 0u;
            if (plain && t + (steps - ins) <= limit) {
                lo &= u_lo, hi &= u_hi;
                uint64_t g_lo = hi4_lo, g_hi = hi4_hi; // gap columns still in the window (flags at bit 0)
                while (g_lo | g_hi) { // take the lowest gap column out: everything above it moves down one column
                    if (g_lo) {
                        const uint64_t below = (g_lo & (0 - g_lo)) - 1; // bits under the gap's nibble
                        lo = (lo & below) | (((lo >> 4) | (hi << 60)) & ~below);
                        g_lo = ((g_lo >> 4) | (g_hi << 60)) & ~below; // (no flag below; the removed gap's own falls under `below`)
                        hi >>= 4;
                        g_hi >>= 4;
                    } else {
                        const uint64_t below = (g_hi & (0 - g_hi)) - 1;
                        hi = (hi & below) | ((hi >> 4) & ~below);
                        g_hi = (g_hi >> 4) & ~below;
                    }
                }
                auto squeeze = [](uint64_t x) -> uint64_t { // 16 nibbles -> 16 two-bit codes in the low 32 bits
                    uint64_t y = x & 0x3333333333333333ULL;
                    y = (y | (y >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
                    y = (y | (y >> 4)) & 0x00FF00FF00FF00FFULL;
                    y = (y | (y >> 8)) & 0x0000FFFF0000FFFFULL;
                    return (y | (y >> 16)) & 0xFFFFFFFFULL;
                };
                const uint64_t mask = (1ULL << (2 * (uint64_t)k)) - 1;
                const uint64_t codes = (squeeze(lo) | (squeeze(hi) << 32)) & mask; // code of the j-th kept column at bits 2j
                uint64_t rb = __builtin_bitreverse64(codes);
                rb = ((rb >> 1) & 0x5555555555555555ULL) | ((rb & 0x5555555555555555ULL) << 1); // pairs back in bit order
                const uint64_t fw = rb >> (64 - 2 * k), rv = ~codes & mask;
                *kmer_out = yak_hash64(fw < rv ? fw : rv, mask);
                have_kmer = true;
            }
        }
    }
    if (!have_kmer) {
        NibReader nr{base, 0, 0xFFFFFFFFu};
        const uint64_t ksize = cx.ksize, shift = 2 * (ksize - 1), mask = (1ULL << (2 * ksize)) - 1;
        uint64_t fw = 0, rv = 0, l = 0;
        for (uint32_t c = col; c < rd.n_cols && l < ksize; ++c) {
            const uint8_t nb = nr.get(c);
            if (c != col && !(nb & 8)) ++t;
            const uint8_t q = nb & 7;
            if (q != 4) { // N/M codes are not filtered here (main.rs:1488-1492)
                fw = ((fw << 2) | (uint64_t)q) & mask;
                rv = (rv >> 2) | ((3ULL ^ (uint64_t)q) << shift);
                ++l;
            }
            if (t > limit) break; // this column was the last one decoded
        }
        uint64_t km = INVALID_KMER;
        if (l >= ksize) km = yak_hash64(fw < rv ? fw : rv, mask);
        *kmer_out = km;
    }
    // ---- the string ----------------------------------------------------------------------------------------------------
    const uint32_t LUT_LO = 0x54474341u, LUT_HI = 0x004D4E2Du; // code_to_ascii: A C G T | - N M
    uint32_t o = 0;
    for (uint32_t wi = col >> 3; o < len; ++wi) {
        const uint32_t raw = *reinterpret_cast<const uint32_t *>(base + ((size_t)wi << 2));
        uint32_t w = (((raw & 0x0F0F0F0Fu) << 4) | ((raw >> 4) & 0x0F0F0F0Fu)) & 0x77777777u; // column j at bits 4j
        uint32_t nc = 8;
        if (wi == (col >> 3)) { // drop the columns before the start
            const uint32_t skip = col & 7;
            w >>= 4 * skip;
            nc = 8 - skip;
        }
        // gap columns are taken out of the word (usually none), the rest becomes ASCII eight at a time; a full word goes
        // out as two dword stores, a partial one (first / last word of the string, words that held gaps) byte by byte.
        // (A per-column fallback loop here ran on nearly every iteration: some lane of the wave is always at its tail.)
        const uint32_t vm = nc == 8 ? 0x11111111u : ((1u << (4 * nc)) - 1u) & 0x11111111u;
        const uint32_t x = w ^ 0x44444444u;
        uint32_t gap = ~(x | (x >> 1) | (x >> 2)) & vm;
        const uint32_t cnt = min(nc - (uint32_t)__builtin_popcount(gap), len - o);
        while (gap) {
            const uint32_t below = (gap & (0u - gap)) - 1u;
            w = (w & below) | ((w >> 4) & ~below);
            gap = (gap >> 4) & ~below;
        }
        const uint32_t lo4 = w & 0xFFFFu, hi4 = w >> 16;
        uint32_t s0 = (lo4 | (lo4 << 8)) & 0x00FF00FFu, s1 = (hi4 | (hi4 << 8)) & 0x00FF00FFu;
        s0 = (s0 | (s0 << 4)) & 0x0F0F0F0Fu;
        s1 = (s1 | (s1 << 4)) & 0x0F0F0F0Fu;
        const uint32_t a0 = __builtin_amdgcn_perm(LUT_HI, LUT_LO, s0), a1 = __builtin_amdgcn_perm(LUT_HI, LUT_LO, s1);
        if (cnt == 8) {
            __builtin_memcpy(seq_out + o, &a0, 4);
            __builtin_memcpy(seq_out + o + 4, &a1, 4);
        } else {
            const uint64_t a = (uint64_t)a0 | ((uint64_t)a1 << 32);
#pragma unroll
            for (uint32_t k = 0; k < 7; ++k)
                if (k < cnt) seq_out[o + k] = (uint8_t)(a >> (8 * k));
        }
        o += cnt;
    }
}

// ------------------------------------------------------------------------------------------------------
// one wavefront per region
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_excl(uint32_t v) { // exclusive prefix sum across the wave
    const uint32_t lane = threadIdx.x & 63;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(x, o);
        if (lane >= (uint32_t)o) x += t;
    }
    return x - v;
}

// ------------------------------------------------------------------------------------------------------
// The contig's own candidate for most reads.  A read without an exception record at the positions [start, end + k + 2]
// carries, column for column, the contig's bases there (a mismatch, a deleted base or an insertion column makes the
// dense pass file records at its position and the two after it), so its candidate string is the contig's substring
// and its first k-mer the contig's: nothing of the read has to be decoded.  That holds for 2/3 of the (read, region)
// pairs of a diploid phasing pass and 9/10 of the pairs of a haploid or final pass.  A block takes 16 regions, four
// per wavefront: (A) every region looks up the records around it (k_tile_sort's index of every 16th position) and marks
// the reads that have one — a binary search of the record's read in the region's read list, in LDS —; (B) the pairs
// that do need their read ("dirty": a record in the window, a read ending inside it) go through one queue for the
// whole block and are decoded by consecutive threads: full wavefronts of the expensive path instead of one or two
// such lanes holding up every wavefront; (C) the regions' waves rank their candidates and write the kept slots.
// A region the shortcut does not cover (its window leaves its tile, the contig has a non-ACGT letter there, more than
// 64 reads over the tile, no record index) decodes every pair the old way.
// ------------------------------------------------------------------------------------------------------
#ifndef NP2_RM_WAVES
#define NP2_RM_WAVES 8 // floor on k_region_measure's resident waves per SIMD
#endif
#ifndef NP2_RM_RPW
#define NP2_RM_RPW 4
#endif
static constexpr uint32_t RM_RPW = NP2_RM_RPW;       // regions per wavefront (a multiple of 4: the offsets go by groups of 4 regions)
static_assert(RM_RPW % 4 == 0 && RM_RPW <= 16, "regions per wavefront");
static constexpr uint32_t RM_REG = 4 * RM_RPW;       // regions per 256-thread block
static constexpr uint32_t CAND_CLEAN = 0xFFFFFFFFu;  // kept_col of a candidate that is the contig's own string
static constexpr uint32_t CLEAN_MAX_LEN = 64;

__device__ __forceinline__ void wave_lds_sync() { // LDS written by this wavefront is read by this wavefront
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) { // set bits of m at lanes below this one
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ np2_read_t rinfo_read(const ReadInfo &ri) {
    return np2_read_t{ri.aln_t_s, ri.aln_t_e, (uint64_t)ri.nib16 << 4, ri.n_cols, 0u};
}

// every pair of a region decoded by the region's wavefront, 64 reads per round (the path regions outside the shortcut take)
__device__ __forceinline__ void region_measure_inline(const CandCtx &cx, uint32_t g, uint32_t st, uint32_t en, uint32_t la, uint32_t lb,
                                                      uint32_t *__restrict__ kept_read, uint32_t *__restrict__ kept_len,
                                                      uint32_t *__restrict__ kept_col, uint32_t &kept, uint32_t &bytes, uint32_t &mx) {
    const uint32_t lane = threadIdx.x & 63;
    kept = 0, bytes = 0, mx = 0;
    for (uint32_t c0 = la; c0 < lb && kept < LQSEQ_MAX_CAN_COUNT; c0 += 64) {
        const uint32_t i = c0 + lane;
        uint32_t r = 0, len = 0, col = 0;
        if (i < lb) {
            r = cx.tile_rd[i];
            const ReadInfo ri = cx.rinfo[r]; // (pcount is 0 for a dropped read)
            if (ri.pcount != 0 && ri.pj <= g && g - ri.pj < ri.pcount) len = cand_measure(cx, r, rinfo_read(ri), ri.ck_off, st, en, col);
        }
        const uint64_t ne = __ballot(len > 0);
        const uint32_t before = kept + lanes_below(ne);
        const bool keep = len > 0 && before < LQSEQ_MAX_CAN_COUNT;
        if (keep) {
            const size_t slot = (size_t)g * LQSEQ_MAX_CAN_COUNT + before;
            kept_read[slot] = r;
            kept_len[slot] = len;
            kept_col[slot] = col;
        }
        kept = min(kept + (uint32_t)__builtin_popcountll(ne), (uint32_t)LQSEQ_MAX_CAN_COUNT);
        bytes += wave_sum(keep ? len : 0u);
        mx = max(mx, keep ? len : 0u);
    }
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, o));
}

// Reads paired with region g (main.rs:1445-1476): those whose region interval [pj, pj + pcount) holds g.  A pair that
// yields a non-empty string has lq_end[g] inside the read, so the reads overlapping that position's contig tile are
// the only ones to test; the list is ascending in read index = the order the reference visits alignseqs in.
// Keeps the first 60 non-empty candidates (main.rs:1474,1509): per region slots kept_read / kept_len / kept_col
// (kept_col == CAND_CLEAN: the contig's own string, nothing to decode).
__device__ __forceinline__ void k_region_measure(const uint32_t np2_bid0, const uint32_t np2_nb, CandCtx cx, uint32_t n_reg, uint32_t *__restrict__ kept_read,
                                                        uint32_t *__restrict__ kept_len, uint32_t *__restrict__ kept_col,
                                                        uint32_t *__restrict__ reg_ncand, uint32_t *__restrict__ reg_bytes,
                                                        uint32_t *__restrict__ reg_maxlen, uint32_t *__restrict__ blk_sum) {
    const uint32_t np2_bid = xcd_order(np2_bid0, np2_nb); // (neighbouring items on one XCD: np2_common.hpp)
    // blk_sum: per group of 4 regions (= one wavefront's), three arrays of n_mb entries: candidates, bytes, longest kept strings
    __shared__ uint32_t s_reads[RM_REG][64];
    __shared__ uint32_t s_dm[RM_REG][2];
    __shared__ uint32_t s_len[RM_REG][64];
    __shared__ uint32_t s_col[RM_REG][64];
    __shared__ uint32_t s_q[4][RM_RPW * 64]; // one queue per wavefront: the wavefronts of a block never wait for each other
    const uint32_t lane = threadIdx.x & 63, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (uniform: the region-level loads below are scalar)
    const uint32_t n_mb = (n_reg + 3) / 4, g0 = (np2_bid * 4 + wv) * RM_RPW, mb0 = g0 / 4;
    uint32_t nq = 0; // (uniform)
    // (A) The four regions of a wavefront go through every step TOGETHER: each step is a chain link of dependent loads
    // (region -> tile -> read list -> read info / records), and a wavefront that walked the chain region by region spent
    // its time waiting for one load at a time (measured: slower than decoding every pair).
    uint32_t st[RM_RPW], en[RM_RPW], tile[RM_RPW], la[RM_RPW], lb[RM_RPW], tn[RM_RPW];
    bool live[RM_RPW];
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) {
        const uint32_t g = g0 + h;
        live[h] = g < n_reg;
        const uint32_t gg = live[h] ? g : n_reg - 1;
        st[h] = cx.lq_start[gg], en[h] = cx.lq_end[gg];
    }
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) {
        tile[h] = min(en[h] >> TILE_SHIFT, cx.n_tiles - 1);
        la[h] = cx.tile_rd_off[tile[h]], lb[h] = cx.tile_rd_off[tile[h] + 1];
        tn[h] = cx.rec_pidx ? cx.tile_n[tile[h]] : 0u;
    }
    const ReadInfo r0 = cx.rinfo[0]; // the contig itself: paired with every region it is to lend its string to
    uint32_t rr[RM_RPW], lo[RM_RPW], hi[RM_RPW], refw[RM_RPW];
    bool wok[RM_RPW];
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) {
        const uint32_t g = g0 + h, nlist = lb[h] - la[h], we = en[h] + cx.ksize + 2;
        wok[h] = live[h] && cx.rec_pidx != nullptr && nlist != 0 && nlist <= 64 && en[h] - st[h] + 1 <= CLEAN_MAX_LEN && we < cx.L &&
                 (st[h] >> TILE_SHIFT) == tile[h] && (we >> TILE_SHIFT) == tile[h] && tn[h] <= cx.bucket_cap &&
                 r0.pcount != 0 && r0.pj <= g && g - r0.pj < r0.pcount;
        rr[h] = (wok[h] && lane < nlist) ? cx.tile_rd[la[h] + lane] : 0xFFFFFFFFu;
        // the contig's letters at [start, end + k]: A/C/G/T only (nibble p at bits 4 (p & 7) of word p >> 3; <= 13 words)
        const uint32_t w0 = st[h] >> 3, w1 = (en[h] + cx.ksize) >> 3;
        refw[h] = 0;
        if (wok[h] && w0 + lane <= w1) {
            uint32_t m = 0x44444444u;
            if (lane == 0) m &= 0xFFFFFFFFu << (4 * (st[h] & 7));
            if (w0 + lane == w1) m &= 0xFFFFFFFFu >> (4 * (7 - ((en[h] + cx.ksize) & 7)));
            refw[h] = cx.refnib[w0 + lane] & m;
        }
        // the tile's records around [start, end + k + 2]: from the index of every 16th position
        lo[h] = 0, hi[h] = 0;
        if (wok[h] && tn[h]) {
            const uint32_t tstart = tile[h] << TILE_SHIFT;
            const uint32_t j0 = (st[h] - tstart) >> 4, j1 = ((we - tstart) >> 4) + 1;
            lo[h] = cx.rec_pidx[(size_t)tile[h] * (TILE / 16) + j0];
            hi[h] = j1 < TILE / 16 ? (uint32_t)cx.rec_pidx[(size_t)tile[h] * (TILE / 16) + j1] : tn[h];
        }
    }
    uint32_t pjv[RM_RPW], pcv[RM_RPW], tev[RM_RPW], rpos[RM_RPW], rq[RM_RPW];
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) {
        pjv[h] = 0, pcv[h] = 0, tev[h] = 0, rpos[h] = 0xFFFFFFFFu, rq[h] = 0;
        if (rr[h] != 0xFFFFFFFFu) {
            const uint4 q = *reinterpret_cast<const uint4 *>(&cx.rinfo[rr[h]].aln_t_e); // (the pairing half: one request)
            tev[h] = q.x, pjv[h] = q.y, pcv[h] = q.z;
        }
        const uint32_t i = lo[h] + lane; // first 64 records of the window
        if (i < hi[h]) {
            const uint64_t a = (uint64_t)tile[h] * cx.bucket_cap + i;
            rpos[h] = (uint32_t)(cx.rec_key[a] >> 32), rq[h] = cx.rec_read[a];
        }
    }
    uint32_t fastmask = 0;
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) {
        const uint32_t slot = wv * RM_RPW + h;
        const bool fast = wok[h] && (uint32_t)__builtin_amdgcn_readfirstlane((int)rr[h]) == 0 && __ballot(refw[h] != 0) == 0;
        if (fast) fastmask |= 1u << h;
        s_reads[slot][lane] = rr[h];
        if (lane < 2) s_dm[slot][lane] = 0;
    }
    wave_lds_sync();
    auto mark = [&](uint32_t slot, uint32_t nlist, uint32_t pos, uint32_t q, uint32_t ws, uint32_t we) {
        if (pos >= ws && pos <= we) { // the record's read in the region's (ascending) read list
            uint32_t x = 0, y = nlist;
            while (x < y) {
                const uint32_t m = (x + y) >> 1;
                if (s_reads[slot][m] < q) x = m + 1; else y = m;
            }
            if (x < nlist && s_reads[slot][x] == q) atomicOr(&s_dm[slot][x >> 5], 1u << (x & 31));
        }
    };
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) {
        if (!((fastmask >> h) & 1u)) continue;
        const uint32_t slot = wv * RM_RPW + h, nlist = lb[h] - la[h], ws = st[h], we = en[h] + cx.ksize + 2;
        mark(slot, nlist, rpos[h], rq[h], ws, we);
        for (uint32_t c0 = lo[h] + 64; c0 < hi[h]; c0 += 64) { // (a window of more than 64 records)
            const uint32_t i = c0 + lane;
            if (i < hi[h]) {
                const uint64_t a = (uint64_t)tile[h] * cx.bucket_cap + i;
                mark(slot, nlist, (uint32_t)(cx.rec_key[a] >> 32), cx.rec_read[a], ws, we);
            }
        }
    }
    wave_lds_sync();
    uint32_t state[RM_RPW]; // 0 no pair, 1 the contig's string, 2 queued
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) {
        state[h] = 0;
        if (!((fastmask >> h) & 1u)) continue;
        const uint32_t g = g0 + h, slot = wv * RM_RPW + h;
        const bool paired = rr[h] != 0xFFFFFFFFu && pcv[h] != 0 && pjv[h] <= g && g - pjv[h] < pcv[h];
        const bool marked = (s_dm[slot][lane >> 5] >> (lane & 31)) & 1u;
        const bool ends_inside = tev[h] < en[h] + cx.ksize; // (the first k-mer needs the columns up to start + k - 1)
        const uint32_t stt = !paired ? 0u : ((marked || ends_inside) && rr[h] != 0) ? 2u : 1u; // (the contig has no records)
        state[h] = stt;
        const uint64_t dq = __ballot(stt == 2u);
        if (stt == 2u) s_q[wv][nq + lanes_below(dq)] = (slot << 6) | lane;
        nq += (uint32_t)__builtin_popcountll(dq);
    }
    uint32_t sum_k[RM_RPW / 4] = {}, sum_b[RM_RPW / 4] = {}, sum_m[RM_RPW / 4] = {}; // totals of the groups of 4 regions (uniform)
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) { // regions outside the shortcut: every pair decoded here
        if (!live[h] || ((fastmask >> h) & 1u)) continue;
        const uint32_t g = g0 + h;
        uint32_t kept, bytes, mx;
        region_measure_inline(cx, g, st[h], en[h], la[h], lb[h], kept_read, kept_len, kept_col, kept, bytes, mx);
        if (lane == 0) {
            reg_ncand[g] = kept;
            reg_bytes[g] = bytes;
            reg_maxlen[g] = mx;
        }
        sum_k[h / 4] += kept, sum_b[h / 4] += bytes, sum_m[h / 4] += mx;
    }
    wave_lds_sync();
    // (B) the pairs that need their read, one per lane
    for (uint32_t e = lane; e < nq; e += 64) {
        const uint32_t w = s_q[wv][e], slot = w >> 6, ln = w & 63;
        const uint32_t g = (np2_bid * 4 + slot / RM_RPW) * RM_RPW + slot % RM_RPW;
        const uint32_t r = s_reads[slot][ln];
        const ReadInfo ri = cx.rinfo[r];
        uint32_t col = 0;
        const uint32_t len = cand_measure(cx, r, rinfo_read(ri), ri.ck_off, cx.lq_start[g], cx.lq_end[g], col);
        s_len[slot][ln] = len;
        s_col[slot][ln] = col;
    }
    wave_lds_sync();
    // (C) rank and keep
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) {
        if (!((fastmask >> h) & 1u)) continue;
        const uint32_t g = g0 + h, slot = wv * RM_RPW + h;
        uint32_t len = 0, col = CAND_CLEAN;
        if (state[h] == 1u) len = en[h] - st[h] + 1;
        else if (state[h] == 2u) len = s_len[slot][lane], col = s_col[slot][lane];
        const uint64_t ne = __ballot(len > 0);
        const uint32_t before = lanes_below(ne);
        const bool keep = len > 0 && before < LQSEQ_MAX_CAN_COUNT;
        if (keep) {
            const size_t ks = (size_t)g * LQSEQ_MAX_CAN_COUNT + before;
            kept_read[ks] = rr[h];
            kept_len[ks] = len;
            kept_col[ks] = col;
        }
        const uint32_t kept = min((uint32_t)__builtin_popcountll(ne), (uint32_t)LQSEQ_MAX_CAN_COUNT);
        const uint32_t bytes = wave_sum(keep ? len : 0u);
        uint32_t mx = keep ? len : 0u;
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, o));
        if (lane == 0) {
            reg_ncand[g] = kept;
            reg_bytes[g] = bytes;
            reg_maxlen[g] = mx; // the longest string a splice can put in place of this region
        }
        sum_k[h / 4] += kept, sum_b[h / 4] += bytes, sum_m[h / 4] += mx;
    }
#pragma unroll
    for (uint32_t q = 0; q < RM_RPW / 4; ++q)
        if (lane == 0 && mb0 + q < n_mb) {
            blk_sum[mb0 + q] = sum_k[q];
            blk_sum[n_mb + mb0 + q] = sum_b[q];
            blk_sum[2 * n_mb + mb0 + q] = sum_m[q];
        }
}

// candidate / sequence offsets of every region; totals -> *n_cand, *n_bytes and the closing cand_seq_off entry
__device__ __forceinline__ void k_cand_offsets(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ blk_sum, uint32_t n_blk,
                                                       uint32_t n_reg, uint32_t *__restrict__ blk_coff,
                                                       uint32_t *__restrict__ blk_soff, uint32_t *__restrict__ cand_off,
                                                       uint32_t *__restrict__ reg_soff, uint32_t *__restrict__ n_cand,
                                                       uint32_t *__restrict__ n_bytes, uint32_t *__restrict__ grow) {
    __shared__ uint32_t sh[16];
    const uint32_t ta = block_scan_array<OpAdd>(
        n_blk, sh, [&](uint32_t i) { return blk_sum[i]; }, [&](uint32_t i, uint32_t pre, uint32_t) { blk_coff[i] = pre; });
    const uint32_t tb = block_scan_array<OpAdd>(
        n_blk, sh, [&](uint32_t i) { return blk_sum[n_blk + i]; },
        [&](uint32_t i, uint32_t pre, uint32_t) { blk_soff[i] = pre; });
    const uint32_t tc = block_scan_array<OpAdd>(
        n_blk, sh, [&](uint32_t i) { return blk_sum[2 * n_blk + i]; }, [&](uint32_t, uint32_t, uint32_t) {});
    if (threadIdx.x == 0) {
        cand_off[n_reg] = ta;
        reg_soff[n_reg] = tb;
        *n_cand = ta;
        *n_bytes = tb;
        *grow = tc; // upper bound of the consensus growth of one splice round
    }
}

// the same with a few blocks chained by the look-back (1024 region blocks each); block 0 also adds up the growth bound
__device__ __forceinline__ void k_cand_offsets_lb(const uint32_t np2_bid, const uint32_t np2_nb, Lookback lb, uint32_t n_blocks, const uint32_t *__restrict__ blk_sum,
                                                         uint32_t n_blk, uint32_t n_reg, uint32_t *__restrict__ blk_coff,
                                                         uint32_t *__restrict__ blk_soff, uint32_t *__restrict__ cand_off,
                                                         uint32_t *__restrict__ reg_soff, uint32_t *__restrict__ n_cand,
                                                         uint32_t *__restrict__ n_bytes, uint32_t *__restrict__ grow,
                                                         uint32_t *__restrict__ err) {
    __shared__ uint32_t sh[8];
    const uint32_t bid = lb_block_id(lb, sh);
    const uint32_t i0 = (bid * 256 + threadIdx.x) * 4;
    uint32_t a[4], b[4], sa = 0, sb = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        a[k] = i0 + k < n_blk ? blk_sum[i0 + k] : 0u;
        b[k] = i0 + k < n_blk ? blk_sum[n_blk + i0 + k] : 0u;
        sa += a[k];
        sb += b[k];
    }
    uint32_t ta, tb, pa, pb;
    uint32_t ra = block_excl_scan<OpAdd, 4>(sa, sh, ta);
    uint32_t rb = block_excl_scan<OpAdd, 4>(sb, sh, tb);
    lb_exclusive2(lb, bid, ta, tb, sh, err, pa, pb);
    ra += pa, rb += pb;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (i0 + k < n_blk) {
            blk_coff[i0 + k] = ra;
            blk_soff[i0 + k] = rb;
            ra += a[k];
            rb += b[k];
        }
    if (bid == n_blocks - 1 && threadIdx.x == 255) {
        cand_off[n_reg] = ra;
        reg_soff[n_reg] = rb;
        *n_cand = ra;
        *n_bytes = rb;
    }
    if (bid == 0) { // upper bound of the consensus growth of one splice round
        uint32_t c = 0, tc;
        for (uint32_t i = threadIdx.x; i < n_blk; i += 256) c += blk_sum[2 * n_blk + i];
        (void)block_excl_scan<OpAdd, 4>(c, sh, tc);
        if (threadIdx.x == 0) *grow = tc;
    }
}

// Merge rule of the raw LQ regions (main.rs:1613-1615: region j merges into j - 1 iff end_j >= start_{j-1}): head flags and
// their exclusive scan.  The number of raw regions lives on the device, so the grid is a FIXED handful of blocks that
// split whatever it turns out to be into equal stretches: a block counts its stretch's heads, the blocks chain their
// counts (a look-back over at most MERGE_LB_BLOCKS predecessors: one round), then it scans and writes its stretch.
// (One block of 1024 threads walking the whole array — a contig's 10^4 - 10^5 regions, 1024 at a time with a block scan
// each — was 25 - 29 us per pass, 4 % of a bacterial contig's step.)
static constexpr uint32_t MERGE_LB_BLOCKS = 32;
__device__ __forceinline__ void k_lq_merge_scan_lb(const uint32_t np2_bid, const uint32_t np2_nb, Lookback lb, const uint32_t *__restrict__ raw_start,
                                                   const uint32_t *__restrict__ raw_end, const uint32_t *__restrict__ n_raw, uint32_t n_host,
                                                   uint32_t *__restrict__ headflag, uint32_t *__restrict__ hidx, uint32_t *__restrict__ err) {
    __shared__ uint32_t sh[16];
    const uint32_t bid = lb_block_id(lb, sh);
    const uint32_t n = min(*n_raw, n_host);
    const uint32_t seg = ((n + MERGE_LB_BLOCKS - 1) / MERGE_LB_BLOCKS + 1023u) & ~1023u; // a multiple of the block's width
    const uint32_t lo = min(bid * seg, n), hi = min(lo + seg, n);
    auto flag = [&](uint32_t j) { return (j >= 1 && raw_end[j] >= raw_start[j - 1]) ? 0u : 1u; };
    uint32_t cnt = 0;
    for (uint32_t j = lo + threadIdx.x; j < hi; j += 1024) cnt += flag(j);
    uint32_t total, pre, dummy;
    (void)block_excl_scan<OpAdd, 16>(cnt, sh, total);
    lb_exclusive2(lb, bid, total, 0u, sh, err, pre, dummy);
    uint32_t carry = pre;
    for (uint32_t j0 = lo; j0 < hi; j0 += 1024) { // (uniform)
        const uint32_t j = j0 + threadIdx.x;
        const uint32_t v = j < hi ? flag(j) : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<OpAdd, 16>(v, sh, tot);
        if (j < hi) {
            hidx[j] = carry + ex;
            headflag[j] = v;
        }
        carry += tot;
    }
}

// Second pass, once the offsets are known: order, string offset, string and first k-mer of every kept candidate.  One
// wavefront per region (four regions per wavefront, one after the other), one lane per kept candidate (at most 60).  The
// candidates that are the contig's own string (kept_col == CAND_CLEAN) copy it from LDS, where the region's wavefront
// put it straight from the nibble-packed contig, and take the k-mer of the contig's own candidate (read 0, always the
// first of such a region); the others — and read 0 — go through the block's queue and are decoded by consecutive threads.
__device__ __forceinline__ void k_region_write(const uint32_t np2_bid0, const uint32_t np2_nb, CandCtx cx, uint32_t n_reg, const uint32_t *__restrict__ kept_read,
                                                      const uint32_t *__restrict__ kept_len,
                                                      const uint32_t *__restrict__ kept_col,
                                                      const uint32_t *__restrict__ reg_ncand,
                                                      const uint32_t *__restrict__ reg_bytes,
                                                      const uint32_t *__restrict__ blk_coff,
                                                      const uint32_t *__restrict__ blk_soff,
                                                      uint32_t *__restrict__ cand_off, uint32_t *__restrict__ reg_soff,
                                                      uint32_t cand_cap,
                                                      uint32_t seq_cap, uint32_t *__restrict__ cand_order,
                                                      uint64_t *__restrict__ cand_kmer, uint32_t *__restrict__ cand_seq_off,
                                                      uint8_t *__restrict__ cand_seq) {
    const uint32_t np2_bid = xcd_order(np2_bid0, np2_nb); // (neighbouring items on one XCD: np2_common.hpp)
    __shared__ __attribute__((aligned(4))) uint8_t s_str[RM_REG][CLEAN_MAX_LEN];
    __shared__ uint64_t s_km[RM_REG];
    __shared__ uint32_t s_so[RM_REG][64];
    __shared__ uint32_t s_oc[RM_REG];
    __shared__ uint32_t s_q[4][RM_RPW * 64]; // one queue per wavefront
    const uint32_t lane = threadIdx.x & 63, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t g0 = (np2_bid * 4 + wv) * RM_RPW, mb0 = g0 / 4;
    uint32_t nq = 0; // (uniform)
    uint32_t oc = 0, ob = 0; // (running over the wavefront's regions: the groups of 4 follow each other)
    if (g0 < n_reg) oc = blk_coff[mb0], ob = blk_soff[mb0];
    uint32_t clean_h = 0;        // regions of this wavefront with candidates that are the contig's string (uniform)
    uint32_t my_len[RM_RPW], my_so[RM_RPW], my_ci[RM_RPW];
    bool my_clean[RM_RPW];
    // (the four regions' loads level by level, not region after region: see k_region_measure)
    uint32_t n[RM_RPW], rb[RM_RPW], st[RM_RPW], ln0[RM_RPW];
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) {
        const uint32_t g = g0 + h;
        n[h] = 0, rb[h] = 0, st[h] = 0, ln0[h] = 0;
        if (g < n_reg) n[h] = reg_ncand[g], rb[h] = reg_bytes[g], st[h] = cx.lq_start[g], ln0[h] = cx.lq_end[g] - st[h] + 1;
    }
    uint32_t len[RM_RPW], col[RM_RPW], rd_[RM_RPW], refc[RM_RPW];
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) {
        const size_t ks = (size_t)(g0 + h) * LQSEQ_MAX_CAN_COUNT + lane;
        const bool act = lane < n[h];
        len[h] = act ? kept_len[ks] : 0u;
        col[h] = act ? kept_col[ks] : 0u;
        rd_[h] = act ? kept_read[ks] : 0u;
        const uint32_t p = st[h] + lane; // (used only by regions with clean candidates: lane < ln0 <= 64 there)
        refc[h] = (n[h] && lane < ln0[h] && ln0[h] <= CLEAN_MAX_LEN) ? cx.refnib[p >> 3] >> (4 * (p & 7)) : 0u;
    }
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) {
        const uint32_t g = g0 + h, slot = wv * RM_RPW + h;
        my_len[h] = 0, my_so[h] = 0, my_ci[h] = 0, my_clean[h] = false;
        if (g >= n_reg) continue;
        if (g == n_reg - 1 && lane == 0 && cand_off[n_reg] < cand_cap) cand_seq_off[cand_off[n_reg]] = reg_soff[n_reg];
        if (lane == 0) {
            cand_off[g] = oc;
            reg_soff[g] = ob;
            s_oc[slot] = oc;
        }
        const bool act = lane < n[h];
        const uint32_t so = ob + wave_excl(len[h]);
        const uint32_t ci = oc + lane;
        const bool ok = act && ci < cand_cap && (uint64_t)so + len[h] <= seq_cap;
        if (ok) {
            cand_order[ci] = rd_[h];
            cand_seq_off[ci] = so;
        }
        const bool clean = ok && col[h] == CAND_CLEAN;
        const uint64_t cm = __ballot(clean);
        // read 0's kept_col says "clean" too (its string is the contig's by definition): it is decoded for the k-mer
        const bool queued = ok && (!clean || (lane == 0 && rd_[h] == 0));
        my_len[h] = len[h], my_so[h] = so, my_ci[h] = ci, my_clean[h] = clean && !queued;
        s_so[slot][lane] = so;
        if (cm) {
            clean_h |= 1u << h;
            // the contig's bases at [start, end]: one per lane, straight from the nibble-packed contig (all A/C/G/T here)
            if (lane < ln0[h]) s_str[slot][lane] = (uint8_t)((0x54474341u >> (8 * (refc[h] & 3))) & 0xFFu);
        }
        const uint64_t dq = __ballot(queued);
        if (queued) s_q[wv][nq + lanes_below(dq)] = (slot << 6) | lane;
        nq += (uint32_t)__builtin_popcountll(dq);
        oc += n[h];
        ob += rb[h];
    }
    wave_lds_sync();
    for (uint32_t e = lane; e < nq; e += 64) {
        const uint32_t w = s_q[wv][e], slot = w >> 6, li = w & 63;
        const uint32_t g = (np2_bid * 4 + slot / RM_RPW) * RM_RPW + slot % RM_RPW;
        const size_t ks = (size_t)g * LQSEQ_MAX_CAN_COUNT + li;
        const uint32_t r = kept_read[ks], len = kept_len[ks];
        uint32_t col = kept_col[ks];
        const ReadInfo ri = cx.rinfo[r];
        const np2_read_t rd = rinfo_read(ri);
        const uint32_t t0 = max(cx.lq_start[g], rd.aln_t_s);
        if (col == CAND_CLEAN) col = t0; // (read 0: column index == position)
        const uint32_t ci = s_oc[slot] + li;
        uint64_t km;
        cand_write(cx, r, rd, ri.pj, g, col, t0, len, cand_seq + s_so[slot][li], &km);
        cand_kmer[ci] = km;
        if (li == 0 && r == 0) s_km[slot] = km;
    }
    wave_lds_sync();
#pragma unroll
    for (uint32_t h = 0; h < RM_RPW; ++h) {
        if (!((clean_h >> h) & 1u)) continue;
        const uint32_t slot = wv * RM_RPW + h;
        if (my_clean[h]) {
            cand_kmer[my_ci[h]] = s_km[slot];
            // (four bases per store: the string starts at any byte of cand_seq, the hardware takes unaligned dwords)
            typedef uint32_t u32_any __attribute__((aligned(1)));
            uint8_t *dst = cand_seq + my_so[h];
            const uint32_t L4 = my_len[h] & ~3u;
            for (uint32_t j = 0; j < L4; j += 4) *reinterpret_cast<u32_any *>(dst + j) = *reinterpret_cast<const uint32_t *>(&s_str[slot][j]);
            for (uint32_t j = L4; j < my_len[h]; ++j) dst[j] = s_str[slot][j];
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------
void launch_scan_small_excl(hipStream_t s, const uint32_t *in, uint32_t *out, uint32_t n, const uint32_t *n_dev,
                            uint32_t *total_out, bool write_end) {
    NP2_LAUNCH(k_scan_small<0>, dim3(1), 1024, s, in, out, n, n_dev, total_out, write_end);
}
void launch_scan_lb_excl(hipStream_t s, const Lookback &lb, const uint32_t *in, uint32_t *out, uint32_t n, bool write_end,
                          uint32_t *err) {
    const uint32_t nb = scan_lb_blocks(n);
    NP2_LAUNCH(k_scan_lb_excl, dim3(nb), SCAN_LB_THREADS, s, lb, nb, in, out, n, write_end, err);
}
uint32_t scan3_tiles(uint32_t n) { return (n + SCAN3_TILE - 1) / SCAN3_TILE; }
void launch_scan_lb_popc(hipStream_t s, const Lookback &lb, const uint32_t *bits, uint32_t *out, uint32_t n_words, uint32_t *err) {
    const uint32_t nb = scan_lb_blocks((uint64_t)n_words + 1);
    NP2_LAUNCH(k_scan_lb_popc, dim3(nb), SCAN_LB_THREADS, s, lb, nb, bits, out, n_words + 1, false, err);
}
uint32_t scan_lb_blocks(uint64_t n) { return (uint32_t)std::max<uint64_t>(1, (n + SCAN_LB_BLOCK - 1) / SCAN_LB_BLOCK); }
void launch_scan3_excl(hipStream_t s, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *part, uint32_t *part_off,
                       bool write_end) {
    if (!n) return;
    const uint32_t nt = scan3_tiles(n);
    NP2_LAUNCH(k_scan3_part, nt, 256, s, in, n, part);
    NP2_LAUNCH(k_scan_small<0>, 1, 1024, s, (const uint32_t *)part, part_off, nt, (const uint32_t *)nullptr, (uint32_t *)nullptr, false);
    NP2_LAUNCH(k_scan3_apply, nt, 256, s, in, out, n, (const uint32_t *)part_off, write_end);
}
void launch_scan_small_incl(hipStream_t s, const int32_t *in, int32_t *out, uint32_t n, const uint32_t *n_dev) {
    NP2_LAUNCH(k_scan_small<1>, dim3(1), 1024, s, (const uint32_t *)in, (uint32_t *)out, n, n_dev, (uint32_t *)nullptr, false);
}
void launch_scan_small_min(hipStream_t s, const int32_t *in, int32_t *out, uint32_t n, const uint32_t *n_dev) {
    NP2_LAUNCH(k_scan_small<2>, dim3(1), 1024, s, (const uint32_t *)in, (uint32_t *)out, n, n_dev, (uint32_t *)nullptr, false);
}
static CandCtx mk_cand(const CandPtrs &c) {
    return CandCtx{c.reads, c.nib,    c.ck_off, c.ckpt,        c.lq_start, c.lq_end, c.pj,
                   c.pcount, c.alive, c.rinfo, c.tile_rd_off, c.tile_rd, c.n_tiles,  c.ksize,
                   c.rec_key, c.rec_read, c.tile_n, c.rec_pidx, c.bucket_cap, c.refnib, c.L};
}
void launch_region_measure(hipStream_t s, const CandPtrs &c, uint32_t n_reg, uint32_t *kept_read, uint32_t *kept_len,
                           uint32_t *kept_col, uint32_t *reg_ncand, uint32_t *reg_bytes, uint32_t *reg_maxlen,
                           uint32_t *blk_sum) {
    if (n_reg)
        NP2_LAUNCH_WAVES(k_region_measure, NP2_RM_WAVES, dim3((n_reg + RM_REG - 1) / RM_REG), 256, s, // (65 registers without the floor: 7 waves)
                         mk_cand(c), n_reg, kept_read, kept_len, kept_col, reg_ncand, reg_bytes, reg_maxlen, blk_sum);
}
uint32_t cand_offsets_blocks(uint32_t n_reg) { return ((n_reg + 3) / 4 + 1023) / 1024; }
void launch_cand_offsets(hipStream_t s, const uint32_t *blk_sum, uint32_t n_reg, uint32_t *blk_coff, uint32_t *blk_soff,
                         uint32_t *cand_off, uint32_t *reg_soff, uint32_t *n_cand, uint32_t *n_bytes, uint32_t *grow,
                         const Lookback *lb, uint32_t *err) {
    if (lb)
        NP2_LAUNCH(k_cand_offsets_lb, dim3(cand_offsets_blocks(n_reg)), 256, s, *lb, cand_offsets_blocks(n_reg), blk_sum, (n_reg + 3) / 4, n_reg, blk_coff, blk_soff, cand_off, reg_soff, n_cand, n_bytes, grow, err);
    else
        NP2_LAUNCH(k_cand_offsets, dim3(1), 1024, s, blk_sum, (n_reg + 3) / 4, n_reg, blk_coff, blk_soff, cand_off, reg_soff, n_cand, n_bytes, grow);
}
void launch_lq_merge_scan_lb(hipStream_t s, const Lookback &lb, const uint32_t *raw_start, const uint32_t *raw_end, const uint32_t *n_raw,
                             uint32_t n_host, uint32_t *headflag, uint32_t *hidx, uint32_t *err) {
    NP2_LAUNCH(k_lq_merge_scan_lb, dim3(MERGE_LB_BLOCKS), 1024, s, lb, raw_start, raw_end, n_raw, n_host, headflag, hidx, err);
}
uint32_t lq_merge_lb_blocks() { return MERGE_LB_BLOCKS; }
void launch_region_write(hipStream_t s, const CandPtrs &c, uint32_t n_reg, const uint32_t *kept_read,
                         const uint32_t *kept_len, const uint32_t *kept_col, const uint32_t *reg_ncand,
                         const uint32_t *reg_bytes, const uint32_t *blk_coff, const uint32_t *blk_soff, uint32_t *cand_off,
                         uint32_t *reg_soff, uint32_t cand_cap, uint32_t seq_cap, uint32_t *cand_order, uint64_t *cand_kmer,
                         uint32_t *cand_seq_off, uint8_t *cand_seq) {
    if (n_reg)
        NP2_LAUNCH(k_region_write, dim3((n_reg + RM_REG - 1) / RM_REG), 256, s, mk_cand(c), n_reg, kept_read, kept_len, kept_col, reg_ncand, reg_bytes, blk_coff, blk_soff, cand_off, reg_soff, cand_cap, seq_cap, cand_order, cand_kmer, cand_seq_off, cand_seq);
}

} // namespace np2
