// Sparse-graph construction by contig tiles (gfx950, wave64).
//
// The exception records of the dense pass are bucketed by contig tile (TILE positions), sorted inside LDS
// (one workgroup per tile) and, per pass, turned into the exception nodes of the 3-column-mer graph
// (Msa::push / Msa::sort, main.rs:193-229), the per-position node offsets and the list of dirty runs by
// one workgroup per tile.  This replaces a device-wide radix sort, three device-wide scans over the
// contig and half a dozen L-sized passes by three short launches per pass.
#include "np2_common.hpp"
#include "np2_kernels.hpp"
#include "np2_blockscan.hpp"
#include "np2_lookback.hpp"

namespace np2 {

// ------------------------------------------------------------------------------------------------------
// small block-level helpers (256-thread blocks = 4 wavefronts)
// ------------------------------------------------------------------------------------------------------
// exclusive scan of one value per thread over a 256-thread block; `sh` = 8 words of LDS scratch
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *sh, uint32_t &total) {
    return block_excl_scan<OpAdd, 4>(v, sh, total);
}

// ------------------------------------------------------------------------------------------------------
// raw exception record (read, column, t_pos) -> node key pos << 32 | bases << 16 | delta1
// (Kmer::new over the columns c-2, c-1, c; head sentinels before a read's first column, main.rs:579-585)
// ------------------------------------------------------------------------------------------------------
__device__ uint64_t make_node_key(const np2_read_t *__restrict__ reads, const uint8_t *__restrict__ nib, uint64_t rec,
                                  uint32_t r) {
    const uint32_t col = (uint32_t)rec, t3 = (uint32_t)(rec >> 32);
    const np2_read_t rd = reads[r];
    const uint8_t *base = nib + rd.nib_off;
    const uint32_t ts = rd.aln_t_s;
    // AlignBase of column c whose t_pos is known; delta = insertion run length ending at c
    auto mk = [&](int64_t c, uint32_t t) -> AlignBase {
        if (c == -2) return ab_head(ts - 1, 0);
        if (c == -1) return ab_head(ts - 1, 1);
        const uint8_t nb = nib_at(base, (uint32_t)c);
        AlignBase a;
        a.q = nb & 7;
        a.t_pos = t;
        a.delta = 0;
        if (c > 0 && (nb & 8)) {
            uint16_t dl = 1;
            int64_t x = c - 1;
            while (x > 0 && (nib_at(base, (uint32_t)x) & 8)) {
                dl = (uint16_t)(dl + 1);
                --x;
            }
            a.delta = dl;
        }
        return a;
    };
    auto is_ins = [&](int64_t c) -> bool { return c > 0 && (nib_at(base, (uint32_t)c) & 8); };
    const int64_t c3 = col;
    const AlignBase b3 = mk(c3, t3);
    const uint32_t t2 = is_ins(c3) ? t3 : t3 - 1;
    const AlignBase b2 = mk(c3 - 1, t2);
    const uint32_t t1 = is_ins(c3 - 1) ? t2 : t2 - 1;
    const AlignBase b1 = mk(c3 - 2, t1);
    return ((uint64_t)b3.t_pos << 32) | ((uint64_t)node_bases(b1, b2, b3) << 16) | b1.delta;
}

// ------------------------------------------------------------------------------------------------------
// once per contig: bucket layout, in-LDS tile sort (or the device-wide sort for oversized tiles)
// ------------------------------------------------------------------------------------------------------
// single block: per-tile record counts -> tile_n (the cursors are reset for the next contig), exclusive scans of
// the true counts (tile_scan: compact layout) and of the bucket-resident counts (tile_scanb), and the mailbox:
// out[0] = T, out[1] = largest tile, out[2] = records spilled to the overflow area
__device__ __forceinline__ void k_tile_layout(const uint32_t np2_bid, const uint32_t np2_nb, uint32_t *__restrict__ tile_cur, uint32_t n_tiles,
                                                      uint32_t bucket_cap, uint32_t *__restrict__ tile_n,
                                                      uint32_t *__restrict__ tile_scan, uint32_t *__restrict__ tile_scanb,
                                                      const uint32_t *__restrict__ ovf_cnt, uint32_t *__restrict__ out) {
    __shared__ uint32_t sh[16];
    const uint32_t ta = block_scan_array<OpAdd>(
        n_tiles, sh, [&](uint32_t i) { return tile_cur[i]; },
        [&](uint32_t i, uint32_t pre, uint32_t v) {
            tile_scan[i] = pre;
            tile_n[i] = v;
        });
    const uint32_t tb = block_scan_array<OpAdd>(
        n_tiles, sh, [&](uint32_t i) { return min(tile_n[i], bucket_cap); },
        [&](uint32_t i, uint32_t pre, uint32_t) { tile_scanb[i] = pre; });
    const uint32_t mx = block_scan_array<OpMaxU32>(
        n_tiles, sh, [&](uint32_t i) { return tile_n[i]; }, [&](uint32_t i, uint32_t, uint32_t) { tile_cur[i] = 0; });
    if (threadIdx.x == 0) {
        tile_scan[n_tiles] = ta;
        tile_scanb[n_tiles] = tb;
        out[0] = ta;
        out[1] = mx;
        out[2] = *ovf_cnt;
    }
}

// the same for contigs of more than a couple of thousand tiles: 1024 tiles per block, the two running sums chained across
// the blocks by the look-back (a handful of light, uniform blocks), the maximum by one atomic per wave (out[1] is zero
// before: the scalar block is cleared ahead of the dense pass)
static constexpr uint32_t TLB_ITEMS = 4;
__device__ __forceinline__ void k_tile_layout_lb(const uint32_t np2_bid, const uint32_t np2_nb, Lookback lb, uint32_t n_blocks, uint32_t *__restrict__ tile_cur,
                                                        uint32_t n_tiles, uint32_t bucket_cap, uint32_t *__restrict__ tile_n,
                                                        uint32_t *__restrict__ tile_scan, uint32_t *__restrict__ tile_scanb,
                                                        const uint32_t *__restrict__ ovf_cnt, uint32_t *__restrict__ out,
                                                        uint32_t *__restrict__ err) {
    __shared__ uint32_t sh[8];
    const uint32_t bid = lb_block_id(lb, sh);
    const uint32_t i0 = (bid * 256 + threadIdx.x) * TLB_ITEMS;
    uint32_t v[TLB_ITEMS], sa = 0, sb = 0, mx = 0;
#pragma unroll
    for (uint32_t k = 0; k < TLB_ITEMS; ++k) {
        v[k] = i0 + k < n_tiles ? tile_cur[i0 + k] : 0u;
        sa += v[k];
        sb += min(v[k], bucket_cap);
        mx = max(mx, v[k]);
    }
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&out[1], mx);
    uint32_t ta, tb, pa, pb;
    uint32_t ra = block_excl_scan<OpAdd, 4>(sa, sh, ta);
    uint32_t rb = block_excl_scan<OpAdd, 4>(sb, sh, tb);
    lb_exclusive2(lb, bid, ta, tb, sh, err, pa, pb);
    ra += pa, rb += pb;
#pragma unroll
    for (uint32_t k = 0; k < TLB_ITEMS; ++k)
        if (i0 + k < n_tiles) {
            tile_scan[i0 + k] = ra;
            tile_scanb[i0 + k] = rb;
            tile_n[i0 + k] = v[k];
            tile_cur[i0 + k] = 0;
            ra += v[k];
            rb += min(v[k], bucket_cap);
        }
    if (bid == n_blocks - 1 && threadIdx.x == 255) { // (the last thread's running sums are the totals)
        tile_scan[n_tiles] = ra;
        tile_scanb[n_tiles] = rb;
        out[0] = ra;
        out[2] = *ovf_cnt;
    }
}

// one block per tile: raw records of the tile's bucket -> node keys, sorted by (key, read) and written back in place.
// A node key leads with its contig position and a tile has only TILE of them, so the sort is a counting sort by
// position (LDS histogram, block scan, scatter of record indices) followed by a rank inside each position's group
// (one thread per record counts the smaller members of its group: a position holds a handful of records, a
// heterozygous site a few dozen).  ~5x fewer LDS operations than the bitonic network this replaces, whose 66 stages
// at 2048 records made this the longest kernel of the diploid workload.  Pairs are unique, so the result does not
// depend on the order the bucket was filled in.
template <uint32_t CAP>
__device__ __forceinline__ void k_tile_sort(const uint32_t np2_bid, const uint32_t np2_nb, uint16_t *__restrict__ pidx, const np2_read_t *__restrict__ reads,
                                                   const uint8_t *__restrict__ nib, const uint32_t *__restrict__ tile_n,
                                                   uint32_t bucket_cap, uint64_t *__restrict__ keys,
                                                   uint32_t *__restrict__ vals, uint32_t *__restrict__ err,
                                                   uint32_t n_above, uint32_t n_upto) {
    // (tiles with n_above < records <= n_upto are this launch's: a contig whose fullest tile needs the big variant
    // still sorts its ordinary tiles with the small one, at more blocks per CU)
    __shared__ uint32_t s_klo[CAP];  // bases << 16 | delta1 of the node key (its position is the group)
    __shared__ uint32_t s_v[CAP];    // read
    __shared__ uint16_t s_q[CAP];    // position inside the tile
    __shared__ uint16_t s_idx[CAP];  // record indices grouped by position
    __shared__ uint32_t s_cnt[TILE]; // records per position (also the scatter cursor)
    __shared__ uint16_t s_off[TILE]; // (16 bits are enough below 64 k records and make the 1024 variant 18 KB: 8 blocks per CU, not 7;
                                     // the 2048 variant 30 KB: 5, not 4)
    __shared__ uint32_t sh[8];
    const uint32_t n = tile_n[np2_bid], tid = threadIdx.x;
    const uint64_t a = (uint64_t)np2_bid * bucket_cap;
    if (n <= n_above || n > n_upto) return;
    if (n > CAP) {
        if (tid == 0) atomicOr(err, 8u);
        return;
    }
    const uint32_t start = np2_bid << TILE_SHIFT;
    for (uint32_t i = tid; i < TILE; i += 256) s_cnt[i] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 256) {
        const uint32_t r = vals[a + i];
        const uint64_t k = make_node_key(reads, nib, keys[a + i], r);
        uint32_t q = (uint32_t)(k >> 32) - start;
        if (q >= TILE) { // (the dense pass files a record under the tile of its position)
            atomicOr(err, 8u);
            q = TILE - 1;
        }
        s_klo[i] = (uint32_t)k;
        s_v[i] = r;
        s_q[i] = (uint16_t)q;
        atomicAdd(&s_cnt[q], 1u);
    }
    __syncthreads();
    {
        const uint32_t q0 = tid * 4;
        const uint32_t c0 = s_cnt[q0], c1 = s_cnt[q0 + 1], c2 = s_cnt[q0 + 2], c3 = s_cnt[q0 + 3];
        uint32_t tot;
        const uint32_t l0 = block_excl_scan_256(c0 + c1 + c2 + c3, sh, tot);
        s_off[q0] = (uint16_t)l0, s_off[q0 + 1] = (uint16_t)(l0 + c0), s_off[q0 + 2] = (uint16_t)(l0 + c0 + c1), s_off[q0 + 3] = (uint16_t)(l0 + c0 + c1 + c2);
        s_cnt[q0] = 0, s_cnt[q0 + 1] = 0, s_cnt[q0 + 2] = 0, s_cnt[q0 + 3] = 0;
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 256) {
        const uint32_t q = s_q[i];
        s_idx[s_off[q] + atomicAdd(&s_cnt[q], 1u)] = (uint16_t)i;
    }
    // where the records of every 16th position begin: candidate extraction looks up the records around an LQ region
    if (tid < TILE / 16) pidx[(size_t)np2_bid * (TILE / 16) + tid] = s_off[tid * 16];
    __syncthreads();
    for (uint32_t d = tid; d < n; d += 256) {
        const uint32_t i = s_idx[d], q = s_q[i], b = s_off[q], m = s_cnt[q];
        const uint32_t klo = s_klo[i], v = s_v[i];
        uint32_t rank = 0;
        for (uint32_t e = b; e < b + m; ++e) {
            const uint32_t j = s_idx[e];
            const uint32_t kj = s_klo[j], vj = s_v[j];
            rank += (kj < klo || (kj == klo && vj < v)) ? 1u : 0u;
        }
        keys[a + b + rank] = ((uint64_t)(start + q) << 32) | klo;
        vals[a + b + rank] = v;
    }
}

// oversized tiles: gather bucket-resident and spilled raw records into one compact array of node keys
// (sorted device-wide afterwards)
__device__ __forceinline__ void k_gather_buckets(const uint32_t np2_bid, const uint32_t np2_nb, const np2_read_t *__restrict__ reads,
                                                        const uint8_t *__restrict__ nib,
                                                        const uint32_t *__restrict__ tile_n,
                                                        const uint32_t *__restrict__ tile_scanb, uint32_t bucket_cap,
                                                        const uint64_t *__restrict__ bkeys,
                                                        const uint32_t *__restrict__ bvals, uint64_t *__restrict__ keys,
                                                        uint32_t *__restrict__ vals) {
    const uint32_t n = min(tile_n[np2_bid], bucket_cap);
    const uint64_t a = (uint64_t)np2_bid * bucket_cap;
    const uint32_t dst = tile_scanb[np2_bid];
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint32_t r = bvals[a + i];
        keys[dst + i] = make_node_key(reads, nib, bkeys[a + i], r);
        vals[dst + i] = r;
    }
}
__device__ __forceinline__ void k_gather_spill(const uint32_t np2_bid, const uint32_t np2_nb, const np2_read_t *__restrict__ reads, const uint8_t *__restrict__ nib,
                               const uint64_t *__restrict__ okeys, const uint32_t *__restrict__ ovals, uint32_t n,
                               uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t i = np2_bid * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = ovals[i];
    keys[i] = make_node_key(reads, nib, okeys[i], r);
    vals[i] = r;
}

// ------------------------------------------------------------------------------------------------------
// per pass: nodes of the live reads, node offsets, dirty runs
// ------------------------------------------------------------------------------------------------------
// is record i the head of a key group with at least one live read?  (count / first read of the group)
__device__ __forceinline__ bool group_head(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                           const uint8_t *__restrict__ alive, uint64_t i, uint64_t a, uint64_t b,
                                           uint32_t &cnt, uint32_t &mn) {
    const uint64_t k = keys[i];
    if (i > a && keys[i - 1] == k) return false;
    cnt = 0;
    mn = 0xFFFFFFFFu;
    for (uint64_t j = i; j < b && keys[j] == k; ++j) {
        const uint32_t r = vals[j];
        if (alive[r]) {
            ++cnt;
            mn = min(mn, r);
        }
    }
    return cnt != 0;
}
// does position p (the last position of the previous tile, whose records are [a, b)) hold a live exception node?
__device__ __forceinline__ bool prev_pos_dirty(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                               const uint8_t *__restrict__ alive, uint64_t a, uint64_t b, uint32_t p) {
    for (uint64_t i = b; i-- > a;) {
        if ((uint32_t)(keys[i] >> 32) != p) return false;
        if (alive[vals[i]]) return true;
    }
    return false;
}

// records of tile t: [a, a + tile_n[t]) with a = t * bucket_cap (bucketed layout) or tile_scan[t] (compact layout)
struct TileLayout {
    const uint32_t *tile_n;
    const uint32_t *tile_scan;
    uint32_t bucket_cap; // 0 = compact layout
    __device__ __forceinline__ uint64_t begin(uint32_t t) const {
        return bucket_cap ? (uint64_t)t * bucket_cap : (uint64_t)tile_scan[t];
    }
};

static constexpr uint32_t TC_CAP = 2048; // records of a tile staged in LDS by k_tile_count

__device__ __forceinline__ void k_tile_count(const uint32_t np2_bid, const uint32_t np2_nb, const uint64_t *__restrict__ keys,
                                                    const uint32_t *__restrict__ vals, TileLayout tl,
                                                    const uint8_t *__restrict__ alive,
                                                    uint32_t *__restrict__ tile_nn, uint32_t *__restrict__ tile_nr) {
    __shared__ uint32_t dirty[TILE / 32];
    __shared__ uint32_t acc[2];
    __shared__ uint32_t prevd;
    __shared__ uint64_t s_k[TC_CAP];
    __shared__ uint8_t s_live[TC_CAP];
    const uint32_t tid = threadIdx.x;
    const uint64_t a = tl.begin(np2_bid);
    const uint32_t n = tl.tile_n[np2_bid];
    const uint64_t b = a + n;
    const uint32_t start = np2_bid << TILE_SHIFT;
    const bool fast = n <= TC_CAP;
    if (tid < TILE / 32) dirty[tid] = 0;
    if (tid < 2) acc[tid] = 0;
    if (tid == 0) {
        prevd = 0;
        if (start && n) {
            const uint64_t pa = tl.begin(np2_bid - 1);
            prevd = prev_pos_dirty(keys, vals, alive, pa, pa + tl.tile_n[np2_bid - 1], start - 1) ? 1u : 0u;
        }
    }
    if (fast) { // (all of a thread's records requested before the first is used: keys and reads, then the reads' liveness)
        uint64_t rk[TC_CAP / 256];
        uint32_t rv[TC_CAP / 256];
#pragma unroll
        for (uint32_t j = 0; j < TC_CAP / 256; ++j) {
            const uint32_t i = tid + 256 * j;
            rk[j] = 0, rv[j] = 0;
            if (i < n) rk[j] = keys[a + i], rv[j] = vals[a + i];
        }
        uint8_t rl[TC_CAP / 256];
#pragma unroll
        for (uint32_t j = 0; j < TC_CAP / 256; ++j) {
            const uint32_t i = tid + 256 * j;
            rl[j] = i < n ? alive[rv[j]] : 0;
        }
#pragma unroll
        for (uint32_t j = 0; j < TC_CAP / 256; ++j) {
            const uint32_t i = tid + 256 * j;
            if (i < n) s_k[i] = rk[j], s_live[i] = rl[j];
        }
    }
    __syncthreads();
    uint32_t nn = 0;
    for (uint32_t i = tid; i < n; i += 256) {
        bool isnode = false;
        uint64_t k;
        if (fast) {
            k = s_k[i];
            if (i == 0 || s_k[i - 1] != k)
                for (uint32_t j = i; j < n && s_k[j] == k && !isnode; ++j) isnode = s_live[j] != 0;
        } else {
            uint32_t cnt, mn;
            k = keys[a + i];
            isnode = group_head(keys, vals, alive, a + i, a, b, cnt, mn);
        }
        if (isnode) {
            ++nn;
            const uint32_t q = (uint32_t)(k >> 32) - start;
            atomicOr(&dirty[q >> 5], 1u << (q & 31));
        }
    }
    if (nn) atomicAdd(&acc[0], nn);
    __syncthreads();
    if (tid < TILE / 32) {
        const uint32_t d = dirty[tid];
        const uint32_t carry = tid ? dirty[tid - 1] >> 31 : prevd;
        const uint32_t starts = d & ~((d << 1) | carry);
        if (starts) atomicAdd(&acc[1], (uint32_t)__builtin_popcount(starts));
    }
    __syncthreads();
    if (tid == 0) {
        tile_nn[np2_bid] = acc[0];
        tile_nr[np2_bid] = acc[1];
    }
}

// single block: exclusive scans of the per-tile node and run counts; totals -> n_nodes / n_runs
__device__ __forceinline__ void k_tile_offsets(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ tile_nn,
                                                       const uint32_t *__restrict__ tile_nr, uint32_t n_tiles,
                                                       uint32_t *__restrict__ tile_noff, uint32_t *__restrict__ tile_roff,
                                                       uint32_t *__restrict__ n_nodes, uint32_t *__restrict__ n_runs,
                                                       uint32_t *__restrict__ reset, uint32_t n_reset,
                                                       const long long *__restrict__ tile_gain,
                                                       unsigned long long *__restrict__ gain_total) {
    __shared__ uint32_t sh[16];
    if (threadIdx.x < n_reset) reset[threadIdx.x] = 0; // per-pass device scalars (best, path begin, gains, ...)
    if (tile_gain) { // the fused pass front: the tiles' shares of the path score, summed (one block: no atomics)
        __shared__ long long sg[16];
        long long v = 0;
        for (uint32_t i = threadIdx.x; i < n_tiles; i += blockDim.x) v += tile_gain[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0) sg[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long tsum = 0;
            for (uint32_t w = 0; w < blockDim.x / 64; ++w) tsum += sg[w];
            *gain_total = (unsigned long long)tsum;
        }
    }
    const uint32_t ta = block_scan_array<OpAdd>(
        n_tiles, sh, [&](uint32_t i) { return tile_nn[i]; }, [&](uint32_t i, uint32_t pre, uint32_t) { tile_noff[i] = pre; });
    const uint32_t tb = block_scan_array<OpAdd>(
        n_tiles, sh, [&](uint32_t i) { return tile_nr[i]; }, [&](uint32_t i, uint32_t pre, uint32_t) { tile_roff[i] = pre; });
    if (threadIdx.x == 0) {
        *n_nodes = ta;
        *n_runs = tb;
    }
}

__device__ __forceinline__ void k_tile_offsets_lb(const uint32_t np2_bid, const uint32_t np2_nb, Lookback lb, uint32_t n_blocks, const uint32_t *__restrict__ tile_nn,
                                                         const uint32_t *__restrict__ tile_nr, uint32_t n_tiles,
                                                         uint32_t *__restrict__ tile_noff, uint32_t *__restrict__ tile_roff,
                                                         uint32_t *__restrict__ n_nodes, uint32_t *__restrict__ n_runs,
                                                         uint32_t *__restrict__ reset, uint32_t n_reset,
                                                         uint32_t *__restrict__ err, const long long *__restrict__ tile_gain,
                                                         unsigned long long *__restrict__ gain_total) {
    __shared__ uint32_t sh[8];
    if (np2_bid == 0 && threadIdx.x < n_reset) reset[threadIdx.x] = 0; // per-pass device scalars
    const uint32_t bid = lb_block_id(lb, sh);
    const uint32_t i0 = (bid * 256 + threadIdx.x) * TLB_ITEMS;
    if (tile_gain) { // (*gain_total was zeroed by k_pf_tile; one atomic per wavefront of a few dozen blocks)
        long long v = 0;
#pragma unroll
        for (uint32_t k = 0; k < TLB_ITEMS; ++k)
            if (i0 + k < n_tiles) v += tile_gain[i0 + k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(gain_total, (unsigned long long)v);
    }
    uint32_t a[TLB_ITEMS], b[TLB_ITEMS], sa = 0, sb = 0;
#pragma unroll
    for (uint32_t k = 0; k < TLB_ITEMS; ++k) {
        a[k] = i0 + k < n_tiles ? tile_nn[i0 + k] : 0u;
        b[k] = i0 + k < n_tiles ? tile_nr[i0 + k] : 0u;
        sa += a[k];
        sb += b[k];
    }
    uint32_t ta, tb, pa, pb;
    uint32_t ra = block_excl_scan<OpAdd, 4>(sa, sh, ta);
    uint32_t rb = block_excl_scan<OpAdd, 4>(sb, sh, tb);
    lb_exclusive2(lb, bid, ta, tb, sh, err, pa, pb);
    ra += pa, rb += pb;
#pragma unroll
    for (uint32_t k = 0; k < TLB_ITEMS; ++k)
        if (i0 + k < n_tiles) {
            tile_noff[i0 + k] = ra;
            tile_roff[i0 + k] = rb;
            ra += a[k];
            rb += b[k];
        }
    if (bid == n_blocks - 1 && threadIdx.x == 255) {
        *n_nodes = ra;
        *n_runs = rb;
    }
}

// one block per tile: write the tile's nodes (ordered like Msa::sort over first-seen order: delta3, first read),
// the packed DP records, node_off for every position of the tile and the tile's dirty-run starts.
// Fast path (tile has at most TW_CAP records): records and nodes are staged in LDS, grouping and per-position
// ordering never touch global memory.  Larger tiles take the same steps on the global arrays.
// LDS per block decides how many tiles a CU works on at once, and a tile's time is a chain of a dozen barriers and three
// levels of dependent loads: the node arrays (key, count, first read) live in the memory of the staged records, which are
// dead by the time the nodes are written (every thread keeps its own four record keys in registers across the barrier
// in between) — 19.9 KB per block = 8 blocks per CU (with separate arrays: 32 KB = 5; round 4)
static constexpr uint32_t TW_CAP = 960;

__device__ __forceinline__ void k_tile_write(const uint32_t np2_bid, const uint32_t np2_nb, const uint64_t *__restrict__ keys,
                                                    const uint32_t *__restrict__ vals, TileLayout tl,
                                                    const uint32_t *__restrict__ tile_noff,
                                                    const uint32_t *__restrict__ tile_roff,
                                                    const uint8_t *__restrict__ alive, uint32_t L,
                                                    uint32_t n_tiles, NodeArrays nd, uint2 *__restrict__ nrec,
                                                    uint32_t *__restrict__ node_off, uint32_t *__restrict__ run_start,
                                                    const np2_read_t *__restrict__ reads,
                                                    const uint32_t *__restrict__ tile_rd_off,
                                                    const uint32_t *__restrict__ tile_rd, int32_t *__restrict__ cov,
                                                    const uint8_t *__restrict__ refnib, uint32_t *__restrict__ emit,
                                                    long long *__restrict__ tile_gain, uint32_t *__restrict__ deep_flag,
                                                    int32_t deep_min, uint8_t *__restrict__ pflag) {
    __shared__ uint32_t cnt[TILE];
    __shared__ int32_t dcov[TILE + 1];
    __shared__ uint32_t sh[8];
    __shared__ uint32_t prevd;
    __shared__ __attribute__((aligned(8))) uint32_t s_raw[3 * TW_CAP];
    uint64_t *const s_k = reinterpret_cast<uint64_t *>(s_raw); // record keys          } until the barrier that follows the
    uint32_t *const s_v = s_raw + 2 * TW_CAP;                  // record read | live << 31 } grouping of the records
    uint32_t *const s_ncnt = s_raw;                            // node: live members      } from then on
    uint32_t *const s_nmin = s_raw + TW_CAP;                   // node: first live read   }
    uint32_t *const s_nkey = s_raw + 2 * TW_CAP;               // node: bases | delta << 16 }
    __shared__ long long s_gain[4];
    __shared__ uint32_t s_wt[16];
    const uint32_t tid = threadIdx.x;
    const uint64_t a = tl.begin(np2_bid);
    const uint32_t n = tl.tile_n[np2_bid];
    const uint64_t b = a + n;
    const uint32_t start = np2_bid << TILE_SHIFT;
    const uint32_t npos = min((uint32_t)TILE, L - start);
    const uint32_t nbase = tile_noff[np2_bid];
    const bool fast = n <= TW_CAP;
    for (uint32_t i = tid; i < TILE; i += 256) {
        cnt[i] = 0;
        dcov[i] = 0;
    }
    if (tid == 0) dcov[TILE] = 0;
    __syncthreads(); // (LDS only: nothing in flight yet)
    // Two chains of dependent loads — the tile's records (key, read -> alive) and its read list (read -> alive, span: the
    // coverage, Msa::coverage, main.rs:232-241, as a difference array) — and a third on thread 0 (is the position left
    // of the tile dirty?): requested level by level for all of them, not one chain after the other (by the phase
    // timers the read list alone, behind a barrier of its own, was 18 % of a tile's time).
    const uint32_t ro0 = tile_rd_off[np2_bid], ro1 = tile_rd_off[np2_bid + 1];
    uint64_t rk[4];
    uint32_t rv[4];
    const uint32_t ci = ro0 + tid;
    const bool hasc = ci < ro1;
    uint32_t cr = 0;
    if (hasc) cr = tile_rd[ci];
    if (fast) {
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t i = tid + 256 * j;
            rk[j] = 0, rv[j] = 0;
            if (i < n) rk[j] = keys[a + i], rv[j] = vals[a + i];
        }
    }
    uint8_t cal = 0;
    uint32_t cts = 0, cte = 0;
    if (hasc) cal = alive[cr], cts = reads[cr].aln_t_s, cte = reads[cr].aln_t_e;
    if (fast) {
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t i = tid + 256 * j;
            if (i < n) {
                s_k[i] = rk[j];
                s_v[i] = rv[j] | (alive[rv[j]] ? 0x80000000u : 0u);
            }
        }
    }
    if (tid == 0) {
        prevd = 0;
        if (start) {
            const uint64_t pa = tl.begin(np2_bid - 1);
            prevd = prev_pos_dirty(keys, vals, alive, pa, pa + tl.tile_n[np2_bid - 1], start - 1) ? 1u : 0u;
        }
    }
    if (hasc && cal) {
        atomicAdd(&dcov[max(cts, start) - start], 1);
        atomicAdd(&dcov[min(cte, start + TILE - 1) - start + 1], -1);
    }
    for (uint32_t i = ci + 256; i < ro1; i += 256) { // (a tile under more than 256 reads)
        const uint32_t r = tile_rd[i];
        if (!alive[r]) continue;
        const uint32_t ts = reads[r].aln_t_s, te = reads[r].aln_t_e;
        atomicAdd(&dcov[max(ts, start) - start], 1);
        atomicAdd(&dcov[min(te, start + TILE - 1) - start + 1], -1);
    }
    __syncthreads();
    int32_t cv[4]; // coverage of this thread's four positions
    {
        const uint32_t q = tid * 4;
        const int32_t d0 = dcov[q], d1 = dcov[q + 1], d2 = dcov[q + 2], d3 = dcov[q + 3];
        uint32_t tot;
        const int32_t pre = (int32_t)block_excl_scan_256((uint32_t)(d0 + d1 + d2 + d3), sh, tot);
        cv[0] = pre + d0, cv[1] = cv[0] + d1, cv[2] = cv[1] + d2, cv[3] = cv[2] + d3;
        // (stored at the end of the kernel with everything else that goes to memory: a barrier waits for the stores
        // issued before it, and there are a dozen barriers to come — by the phase timers a third of a tile's time was
        // spent at barriers waiting for the coverage / node_off / emission stores of the phase before)
    }
    // ---- nodes in key order ---------------------------------------------------------------------------
    uint32_t carry = 0;
    if (fast) {
        // all (up to four) records of a thread at once: group heads count their live members, then ONE exchange of the
        // per-wave node counts of the four 256-record slices (ballots) places every node — instead of a block scan
        // (two barriers) per slice
        bool isn[4];
        uint32_t gcv[4], gmv[4], lr[4];
        uint64_t kown[4]; // (the thread's own record keys: the node arrays overwrite the staged records below)
        const uint32_t lane = tid & 63, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t i = tid + 256 * j;
            isn[j] = false, gcv[j] = 0, gmv[j] = 0xFFFFFFFFu, kown[j] = 0;
            if (i < n) {
                const uint64_t k = s_k[i];
                kown[j] = k;
                if (i == 0 || s_k[i - 1] != k) {
                    uint32_t gc = 0, gm = 0xFFFFFFFFu;
                    for (uint32_t jj = i; jj < n && s_k[jj] == k; ++jj) {
                        const uint32_t v = s_v[jj];
                        if (v >> 31) {
                            ++gc;
                            gm = min(gm, v & 0x7FFFFFFFu);
                        }
                    }
                    isn[j] = gc != 0, gcv[j] = gc, gmv[j] = gm;
                }
            }
            const uint64_t bal = __ballot(isn[j]);
            lr[j] = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (lane == 0) s_wt[wv * 4 + j] = (uint32_t)__builtin_popcountll(bal);
        }
        __syncthreads();
        uint32_t tot_j[4], before_w[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            tot_j[j] = 0, before_w[j] = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) {
                const uint32_t t = s_wt[w * 4 + j];
                tot_j[j] += t;
                if (w < wv) before_w[j] += t;
            }
        }
        uint32_t base = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            if (isn[j]) {
                const uint64_t k = kown[j];
                const uint32_t li = base + before_w[j] + lr[j]; // node index inside the tile
                s_nkey[li] = (uint32_t)k;                       // bases << 16 | delta1 in the key's low word
                s_ncnt[li] = gcv[j];
                s_nmin[li] = gmv[j];
                atomicAdd(&cnt[(uint32_t)(k >> 32) - start], 1u);
            }
            base += tot_j[j];
        }
        carry = base;
    }
    for (uint32_t c0 = 0; !fast && c0 < n; c0 += 256) { // uniform trip count: the block scans below need every thread
        const uint32_t i = c0 + tid;
        uint32_t gc = 0, gm = 0xFFFFFFFFu;
        bool isnode = false;
        uint64_t k = 0;
        if (i < n) {
            k = keys[a + i];
            isnode = group_head(keys, vals, alive, a + i, a, b, gc, gm);
        }
        uint32_t tot;
        const uint32_t rank = block_excl_scan_256(isnode ? 1u : 0u, sh, tot);
        if (isnode) {
            const uint32_t li = carry + rank; // node index inside the tile
            const uint32_t q = (uint32_t)(k >> 32) - start;
            const uint32_t o = nbase + li;
            nd.bases[o] = (uint16_t)(k >> 16);
            nd.delta[o] = (uint16_t)k;
            nd.count[o] = gc;
            nd.minr[o] = gm;
            atomicAdd(&cnt[q], 1u);
        }
        carry += tot;
    }
    __syncthreads();
    const uint32_t nn = carry; // nodes of the tile
    // ---- node_off of the tile's positions (4 consecutive positions per thread) --------------------------
    const uint32_t q0 = tid * 4;
    const uint32_t c0 = cnt[q0], c1 = cnt[q0 + 1], c2 = cnt[q0 + 2], c3 = cnt[q0 + 3];
    uint32_t tot;
    const uint32_t l0 = block_excl_scan_256(c0 + c1 + c2 + c3, sh, tot); // tile-local node index of position q0
    const uint32_t off[5] = {l0, l0 + c0, l0 + c0 + c1, l0 + c0 + c1 + c2, l0 + c0 + c1 + c2 + c3};
    // ---- order the nodes of each position, emit the packed records ---------------------------------------
    const uint32_t cj[4] = {c0, c1, c2, c3};
    if (fast) {
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t o0 = off[j], o1 = off[j + 1];
            for (uint32_t i = o0 + 1; i < o1; ++i) {
                const uint32_t kk = s_nkey[i], c = s_ncnt[i], m = s_nmin[i];
                const uint32_t kd = node_delta3((uint16_t)(kk >> 16), (uint16_t)kk);
                uint32_t x = i;
                while (x > o0) {
                    const uint32_t pk = s_nkey[x - 1];
                    const uint32_t pd = node_delta3((uint16_t)(pk >> 16), (uint16_t)pk);
                    if (pd < kd || (pd == kd && s_nmin[x - 1] < m)) break;
                    s_nkey[x] = pk;
                    s_ncnt[x] = s_ncnt[x - 1];
                    s_nmin[x] = s_nmin[x - 1];
                    --x;
                }
                s_nkey[x] = kk;
                s_ncnt[x] = c;
                s_nmin[x] = m;
            }
        }
        // (the sorted nodes are written out at the end: they stay in LDS untouched until then)
    } else {
        for (uint32_t j = 0; j < 4; ++j) {
            if (cj[j] == 0) continue;
            const uint32_t o0 = nbase + off[j], o1 = nbase + off[j + 1];
            for (uint32_t i = o0 + 1; i < o1; ++i) {
                const uint16_t bb = nd.bases[i], d = nd.delta[i];
                const uint32_t c = nd.count[i], m = nd.minr[i];
                const uint32_t kd = node_delta3(bb, d);
                uint32_t x = i;
                while (x > o0) {
                    const uint32_t pd = node_delta3(nd.bases[x - 1], nd.delta[x - 1]);
                    if (pd < kd || (pd == kd && nd.minr[x - 1] < m)) break;
                    nd.bases[x] = nd.bases[x - 1];
                    nd.delta[x] = nd.delta[x - 1];
                    nd.count[x] = nd.count[x - 1];
                    nd.minr[x] = nd.minr[x - 1];
                    --x;
                }
                nd.bases[x] = bb;
                nd.delta[x] = d;
                nd.count[x] = c;
                nd.minr[x] = m;
            }
            for (uint32_t i = o0; i < o1; ++i)
                nrec[i] = make_uint2((uint32_t)nd.bases[i] | ((uint32_t)nd.delta[i] << 16), nd.count[i]);
        }
    }
    // ---- clean positions: consensus emission flag (the contig base, unless it is a gap code) and their share of the
    //      best-path score: a clean position after a clean one adds 10 * c0 - 4 * cov = 6 * cov -------------------
    // per position: bit 0 = has exception nodes, bit 1 = coverage below 2 (what the consensus write-out needs to know
    // about a position, in one byte instead of two node offsets and the coverage); four positions per store
    uint32_t pf = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) pf |= ((cj[j] ? 1u : 0u) | (cv[j] < 2 ? 2u : 0u)) << (8 * j);
    const bool pd0 = q0 ? cnt[q0 - 1] != 0 : prevd != 0;
    // the contig codes of the thread's four positions: two bytes of the nibble-packed contig (start + q0 is a multiple of 4)
    const uint32_t ref4 = q0 < npos ? *reinterpret_cast<const uint16_t *>(refnib + ((start + q0) >> 1)) : 0u;
    uint32_t em[4];
    {
        long long gain = 0;
        bool pdirty = pd0;
        for (uint32_t j = 0; j < 4; ++j) {
            const bool d = cj[j] != 0;
            em[j] = d ? 0u : (((ref4 >> (4 * j)) & 7) != 4 ? 1u : 0u);
            if (q0 + j < npos) {
                if (!d && !pdirty) gain += 6LL * cv[j];
            }
            pdirty = d;
        }
        for (int o = 32; o > 0; o >>= 1) gain += __shfl_down(gain, o);
        if ((tid & 63) == 0) s_gain[tid >> 6] = gain;
        __syncthreads();
    }
    // ---- dirty-run starts ----------------------------------------------------------------------------------
    const bool s0 = c0 && !pd0, s1 = c1 && !c0, s2 = c2 && !c1, s3 = c3 && !c2;
    uint32_t r = tile_roff[np2_bid] +
                 block_excl_scan_256((uint32_t)s0 + (uint32_t)s1 + (uint32_t)s2 + (uint32_t)s3, sh, tot);
    // ---- everything that goes to memory, after the last barrier ------------------------------------------------
    if (s0) run_start[r++] = start + q0;
    if (s1) run_start[r++] = start + q0 + 1;
    if (s2) run_start[r++] = start + q0 + 2;
    if (s3) run_start[r++] = start + q0 + 3;
    if (q0 < npos) {
        *reinterpret_cast<uint32_t *>(pflag + start + q0) = pf; // (padded past L)
        if (q0 + 4 <= npos) { // whole quads (start + q0 is a multiple of 4: 16-byte stores)
            *reinterpret_cast<int4 *>(cov + start + q0) = make_int4(cv[0], cv[1], cv[2], cv[3]);
            *reinterpret_cast<uint4 *>(node_off + start + q0) = make_uint4(nbase + off[0], nbase + off[1], nbase + off[2], nbase + off[3]);
            *reinterpret_cast<uint4 *>(emit + start + q0) = make_uint4(em[0], em[1], em[2], em[3]);
        } else {
            for (uint32_t j = 0; j < 4; ++j)
                if (q0 + j < npos) {
                    cov[start + q0 + j] = cv[j];
                    node_off[start + q0 + j] = nbase + off[j];
                    emit[start + q0 + j] = em[j];
                }
        }
    }
    if (np2_bid == n_tiles - 1 && tid == 255) node_off[L] = nbase + off[4];
    // the on-chip DP of short runs keeps coverages and counts in 16 bits: tell it when that does not hold
    if (max(max(cv[0], cv[1]), max(cv[2], cv[3])) >= deep_min) atomicOr(deep_flag, 1u); // (65536; lower in a test)
    // one value per tile, summed later (same-address atomics from every tile would serialise at L2)
    if (tid == 0) tile_gain[np2_bid] = s_gain[0] + s_gain[1] + s_gain[2] + s_gain[3];
    if (fast) {
        for (uint32_t i = tid; i < nn; i += 256) { // coalesced write-out of the tile's nodes
            const uint32_t kk = s_nkey[i], c = s_ncnt[i];
            const uint32_t o = nbase + i;
            nd.bases[o] = (uint16_t)(kk >> 16);
            nd.delta[o] = (uint16_t)kk;
            nd.count[o] = c;
            nrec[o] = make_uint2((kk >> 16) | (kk << 16), c);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------
uint32_t tile_scan_blocks(uint32_t n_tiles) { return (n_tiles + 256 * TLB_ITEMS - 1) / (256 * TLB_ITEMS); }
void launch_tile_layout(hipStream_t s, uint32_t *tile_cur, uint32_t n_tiles, uint32_t bucket_cap, uint32_t *tile_n,
                        uint32_t *tile_scan, uint32_t *tile_scanb, const uint32_t *ovf_cnt, uint32_t *out,
                        const Lookback *lb, uint32_t *err) {
    if (lb)
        NP2_LAUNCH(k_tile_layout_lb, dim3(tile_scan_blocks(n_tiles)), 256, s, *lb, tile_scan_blocks(n_tiles), tile_cur, n_tiles, bucket_cap, tile_n, tile_scan, tile_scanb, ovf_cnt, out, err);
    else
        NP2_LAUNCH(k_tile_layout, dim3(1), 1024, s, tile_cur, n_tiles, bucket_cap, tile_n, tile_scan, tile_scanb, ovf_cnt, out);
}
void launch_tile_sort(hipStream_t s, uint16_t *pidx, const np2_read_t *reads, const uint8_t *nib, const uint32_t *tile_n,
                      uint32_t n_tiles, uint32_t bucket_cap, uint32_t max_tile, uint64_t *keys, uint32_t *vals,
                      uint32_t *err) {
    const uint32_t ALL = 0xFFFFFFFFu;
    if (max_tile <= 1024) {
        NP2_LAUNCH(k_tile_sort<1024>, dim3(n_tiles), 256, s, pidx, reads, nib, tile_n, bucket_cap, keys, vals, err, 0u, ALL);
    } else {
        NP2_LAUNCH(k_tile_sort<1024>, dim3(n_tiles), 256, s, pidx, reads, nib, tile_n, bucket_cap, keys, vals, err, 0u, 1024u);
        if (max_tile <= 2048 && TILE_CAP > 2048)
            NP2_LAUNCH(k_tile_sort<2048>, dim3(n_tiles), 256, s, pidx, reads, nib, tile_n, bucket_cap, keys, vals, err, 1024u, ALL);
        else
            NP2_LAUNCH(k_tile_sort<TILE_CAP>, dim3(n_tiles), 256, s, pidx, reads, nib, tile_n, bucket_cap, keys, vals, err, 1024u, ALL);
    }
}
void launch_gather_buckets(hipStream_t s, const np2_read_t *reads, const uint8_t *nib, const uint32_t *tile_n,
                           const uint32_t *tile_scanb, uint32_t n_tiles, uint32_t bucket_cap, const uint64_t *bkeys,
                           const uint32_t *bvals, uint64_t *keys, uint32_t *vals) {
    NP2_LAUNCH(k_gather_buckets, dim3(n_tiles), 256, s, reads, nib, tile_n, tile_scanb, bucket_cap, bkeys, bvals, keys, vals);
}
void launch_gather_spill(hipStream_t s, const np2_read_t *reads, const uint8_t *nib, const uint64_t *okeys,
                         const uint32_t *ovals, uint32_t n, uint64_t *keys, uint32_t *vals) {
    if (n) NP2_LAUNCH(k_gather_spill, dim3((n + 255) / 256), 256, s, reads, nib, okeys, ovals, n, keys, vals);
}
void launch_tile_count(hipStream_t s, const uint64_t *keys, const uint32_t *vals, const uint32_t *tile_n,
                       const uint32_t *tile_scan, uint32_t bucket_cap, uint32_t n_tiles, const uint8_t *alive,
                       uint32_t *tile_nn, uint32_t *tile_nr) {
    NP2_LAUNCH(k_tile_count, dim3(n_tiles), 256, s, keys, vals, TileLayout{tile_n, tile_scan, bucket_cap}, alive, tile_nn, tile_nr);
}
void launch_tile_offsets(hipStream_t s, const uint32_t *tile_nn, const uint32_t *tile_nr, uint32_t n_tiles,
                         uint32_t *tile_noff, uint32_t *tile_roff, uint32_t *n_nodes, uint32_t *n_runs, uint32_t *reset,
                         uint32_t n_reset, const Lookback *lb, uint32_t *err, const long long *tile_gain,
                         unsigned long long *gain_total) {
    if (lb)
        NP2_LAUNCH(k_tile_offsets_lb, dim3(tile_scan_blocks(n_tiles)), 256, s, *lb, tile_scan_blocks(n_tiles), tile_nn, tile_nr, n_tiles, tile_noff, tile_roff, n_nodes, n_runs, reset, n_reset, err, tile_gain, gain_total);
    else
        NP2_LAUNCH(k_tile_offsets, dim3(1), 1024, s, tile_nn, tile_nr, n_tiles, tile_noff, tile_roff, n_nodes, n_runs, reset, n_reset, tile_gain, gain_total);
}
void launch_tile_write(hipStream_t s, const uint64_t *keys, const uint32_t *vals, const uint32_t *tile_n,
                       const uint32_t *tile_scan, uint32_t bucket_cap, const uint32_t *tile_noff, const uint32_t *tile_roff,
                       uint32_t n_tiles, const uint8_t *alive, uint32_t L, NodeArrays nd, uint2 *nrec, uint32_t *node_off,
                       uint32_t *run_start, const np2_read_t *reads, const uint32_t *tile_rd_off, const uint32_t *tile_rd,
                       int32_t *cov, const uint8_t *refnib, uint32_t *emit, long long *tile_gain, uint32_t *deep_flag, uint32_t deep_min,
                       uint8_t *pflag) {
    NP2_LAUNCH(k_tile_write, dim3(n_tiles), 256, s, keys, vals, TileLayout{tile_n, tile_scan, bucket_cap}, tile_noff, tile_roff, alive, L, n_tiles, nd, nrec, node_off, run_start, reads, tile_rd_off, tile_rd, cov, refnib, emit, tile_gain, deep_flag, (int32_t)deep_min, pflag);
}

} // namespace np2
