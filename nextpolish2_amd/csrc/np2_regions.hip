// Region-logic kernels (gfx950, wave64): one wavefront per LQ region, one lane per candidate
// (<= 60 candidates per region, main.rs:30).  Reproduces, on the GPU-resident candidate tables:
//   fill_order_stat (main.rs:813-849), mark_hete_lqseqs (916-946), the +-1 read-pair edges of
//   phase_reads_by_lqseqs (948-1002), fill_seed_lqseqs + retain_sort_seqs (862-914, 714-726),
//   update_consensus_with_lqseqs (1027-1058) and reupdate_consensus_with_lqseqs (1060-1420).
#include "np2_common.hpp"
#include "np2_kernels.hpp"
#include "np2_blockscan.hpp"
#include "np2_lookback.hpp"

namespace np2 {

static constexpr uint8_t LB_TEMP = 0x01, LB_SUCC = 0x80, LB_HETE = 0x40, LB_RECH = 0x20; // main.rs:655-658

__device__ __forceinline__ uint32_t min_support(uint32_t n) { return n >= 9 ? 3u : (n >= 6 ? 2u : 1u); }

// ---- per-region wave statistics ------------------------------------------------------------
// A region's candidates occupy either a whole wavefront (base 0, width 64) or one 32-lane half of it (base 0 / 32):
// ballots come back relative to the region's first lane, shuffles take region-relative source lanes.
struct SubWave {
    uint32_t base;  // first lane of the region's lanes
    bool half;      // 32 lanes instead of 64
    __device__ __forceinline__ uint64_t ballot(bool p) const {
        const uint64_t b = __ballot(p);
        return half ? (b >> base) & 0xFFFFFFFFull : b;
    }
    template <class T> __device__ __forceinline__ T shfl(T v, uint32_t j) const { return __shfl(v, (int)(base + j)); }
};

struct WaveStats {
    uint64_t eqmask;  // per lane: candidates with an identical sequence (bit j)
    uint32_t stat;    // per lane: stats[p] of fill_order_stat (0 = not grouped)
    uint32_t c;       // per lane: group size if this lane is a group head
    bool head;        // per lane: first kscore>0 member of its class
    uint32_t max1_c, max1_p, max2_c, max2_p; // wave-uniform
};

// exact byte-wise class computation (fallback; also the definition the fast path must reproduce)
__device__ __forceinline__ uint64_t eqmask_exact(const SubWave &sw, uint32_t lane, uint32_t n, uint32_t so, uint32_t len,
                                                  const uint8_t *__restrict__ seq) {
    uint64_t m = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const uint32_t sj = sw.shfl(so, j), lj = sw.shfl(len, j);
        bool eq = lane < n && lj == len;
        if (eq)
            for (uint32_t t = 0; t < len; ++t)
                if (seq[so + t] != seq[sj + t]) {
                    eq = false;
                    break;
                }
        if (eq) m |= 1ull << j;
    }
    return m;
}

__device__ __forceinline__ WaveStats wave_group_stats(const SubWave &sw, uint32_t lane, uint32_t n, uint32_t so, uint32_t len,
                                                      const uint8_t *__restrict__ seq, bool kpos, uint32_t order) {
    WaveStats w;
    // classes by (length, 64-bit hash); every lane then verifies byte-wise against its class head.  A hash
    // collision (never observed) falls back to the exact O(n^2) comparison, so the result is always exact.
    // (8 bytes per step through unaligned 64-bit loads; the candidate pool is padded, the tail is masked)
    uint64_t h = 0x9E3779B97F4A7C15ull ^ len;
    if (lane < n)
        for (uint32_t t = 0; t < len; t += 8) {
            uint64_t x;
            __builtin_memcpy(&x, seq + so + t, 8);
            if (len - t < 8) x &= (1ull << (8 * (len - t))) - 1ull;
            h ^= x;
            h *= 0x100000001B3ull;
            h ^= h >> 29;
        }
    const uint32_t hlo = (uint32_t)h, hhi = (uint32_t)(h >> 32);
    w.eqmask = 0;
    { // one ballot per distinct hash (a region rarely holds more than a few different strings), not one per candidate
        uint64_t todo = sw.ballot(lane < n);
        while (todo) {
            const uint32_t hd = (uint32_t)__builtin_ctzll(todo);
            const uint32_t jl = sw.shfl(hlo, hd), jh = sw.shfl(hhi, hd);
            const uint64_t m = sw.ballot(lane < n && jl == hlo && jh == hhi);
            if ((m >> lane) & 1ull) w.eqmask = m;
            todo &= ~m;
        }
    }
    bool ok = true;
    if (lane < n) {
        const uint32_t head = __builtin_ctzll(w.eqmask);
        const uint32_t sh = sw.shfl(so, head), lh = sw.shfl(len, head);
        ok = lh == len;
        if (ok && head != lane)
            for (uint32_t t = 0; t < len; t += 8) {
                uint64_t x, y;
                __builtin_memcpy(&x, seq + so + t, 8);
                __builtin_memcpy(&y, seq + sh + t, 8);
                x ^= y;
                if (len - t < 8) x &= (1ull << (8 * (len - t))) - 1ull;
                if (x) {
                    ok = false;
                    break;
                }
            }
    } else {
        (void)sw.shfl(so, 0);
        (void)sw.shfl(len, 0);
    }
    if (sw.ballot(!ok)) w.eqmask = eqmask_exact(sw, lane, n, so, len, seq);
    const uint64_t kmask = sw.ballot(lane < n && kpos);
    const uint64_t valid = w.eqmask & kmask;
    uint32_t p1 = 64;
    w.c = 0;
    if (lane < n && valid) {
        p1 = __builtin_ctzll(valid);
        w.c = __builtin_popcountll(w.eqmask >> p1);
    }
    w.stat = (p1 < 64 && lane >= p1) ? w.c : 0;
    w.head = lane < n && p1 == lane;
    w.max1_c = w.max1_p = w.max2_c = w.max2_p = 0;
    uint64_t hm = sw.ballot(w.head);
    while (hm) {
        const uint32_t hd = __builtin_ctzll(hm);
        hm &= hm - 1;
        const uint32_t ch = sw.shfl(w.c, hd), oh = sw.shfl(order, hd);
        if (ch > w.max1_c || (ch == w.max1_c && oh == 0)) {
            w.max2_c = w.max1_c, w.max2_p = w.max1_p;
            w.max1_c = ch, w.max1_p = hd;
        } else if (w.max1_p == w.max2_p || ch > w.max2_c) {
            w.max2_c = ch, w.max2_p = hd;
        }
    }
    return w;
}

// is_valid_snp (main.rs:780-801): differ after homopolymer compression
__device__ bool hp_differs(const uint8_t *a, uint32_t na, const uint8_t *b, uint32_t nb) {
    uint32_t i = 0, j = 0;
    while (i < na && j < nb) {
        if (a[i] != b[j]) return true;
        while (i + 1 < na && a[i] == a[i + 1]) ++i;
        while (j + 1 < nb && b[j] == b[j + 1]) ++j;
        ++i, ++j;
    }
    return false;
}

// the same across a wavefront (every lane calls it with the same arguments): lane i holds byte i of both strings, the
// run starts are a ballot, lane t fetches the t-th run's base of either string by shuffle.  (One lane walking the two
// strings byte by byte is a chain of ~60 dependent loads while 63 lanes wait: it was half of k_vote_phase's time.)
__device__ __forceinline__ uint32_t kth_set_bit(uint64_t m, uint32_t k) { // position of the k-th (0-based) set bit
    uint32_t pos = 0;
    uint32_t c = (uint32_t)__builtin_popcount((uint32_t)m);
    if (k >= c) k -= c, pos = 32, m >>= 32;
    uint32_t m32 = (uint32_t)m;
    c = (uint32_t)__builtin_popcount(m32 & 0xFFFFu);
    if (k >= c) k -= c, pos += 16, m32 >>= 16;
    c = (uint32_t)__builtin_popcount(m32 & 0xFFu);
    if (k >= c) k -= c, pos += 8, m32 >>= 8;
    c = (uint32_t)__builtin_popcount(m32 & 0xFu);
    if (k >= c) k -= c, pos += 4, m32 >>= 4;
    c = (uint32_t)__builtin_popcount(m32 & 0x3u);
    if (k >= c) k -= c, pos += 2, m32 >>= 2;
    if (k >= (m32 & 1u)) pos += 1;
    return pos;
}
__device__ __forceinline__ bool hp_differs_wave(const SubWave &sw, uint32_t lane, const uint8_t *a, uint32_t na, const uint8_t *b,
                                                uint32_t nb) {
    const uint32_t W = sw.half ? 32u : 64u;
    if (na > W || nb > W) { // (uniform over the region's lanes) longer than they are many: the serial walk
        uint32_t d = 0;
        if (lane == 0) d = hp_differs(a, na, b, nb) ? 1u : 0u;
        return sw.shfl(d, 0) != 0;
    }
    const uint32_t ca = lane < na ? a[lane] : 0u, cb = lane < nb ? b[lane] : 0u;
    const uint32_t pa = sw.shfl(ca, lane ? lane - 1 : 0u), pb = sw.shfl(cb, lane ? lane - 1 : 0u);
    const uint64_t ma = sw.ballot(lane < na && (lane == 0 || ca != pa)), mb = sw.ballot(lane < nb && (lane == 0 || cb != pb));
    const uint32_t runs = min((uint32_t)__builtin_popcountll(ma), (uint32_t)__builtin_popcountll(mb));
    const bool act = lane < runs;
    const uint32_t xa = sw.shfl(ca, act ? kth_set_bit(ma, lane) : 0u), xb = sw.shfl(cb, act ? kth_set_bit(mb, lane) : 0u);
    return sw.ballot(act && xa != xb) != 0;
}

// ---- phasing pass -----------------------------------------------------------------------------
__device__ __forceinline__ void k_vote_phase(const uint32_t np2_bid0, const uint32_t np2_nb, RegionTables rt, uint32_t asref, uint32_t use_all,
                                                    const uint32_t *__restrict__ lq_start, uint32_t own_lo, uint32_t own_hi,
                                                    uint8_t *__restrict__ reg_lable, uint8_t *__restrict__ grp,
                                                    uint32_t *__restrict__ ecount, int32_t *__restrict__ ref_w,
                                                    uint8_t *__restrict__ ref_seen, uint8_t *__restrict__ bad,
                                                    uint32_t *__restrict__ first_reg, uint32_t *__restrict__ err) {
    const uint32_t np2_bid = xcd_order(np2_bid0, np2_nb); // (neighbouring items on one XCD: np2_common.hpp)
    // a wavefront owns two consecutive regions: side by side in its two halves when both have at most 32 candidates
    // (the usual case at 30x), otherwise one after the other over all 64 lanes
    const uint32_t wl = threadIdx.x & 63;
    const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((np2_bid * blockDim.x + threadIdx.x) >> 6)) * 2; // (uniform)
    if (g0 >= rt.n_reg) return;
    const bool has1 = g0 + 1 < rt.n_reg;
    const bool packed = has1 && rt.cand_off[g0 + 1] - rt.cand_off[g0] <= 32 && rt.cand_off[g0 + 2] - rt.cand_off[g0 + 1] <= 32;
    const uint32_t rounds = (has1 && !packed) ? 2u : 1u;
    for (uint32_t round = 0; round < rounds; ++round) {
        const SubWave sw{packed ? (wl & 32u) : 0u, packed};
        const uint32_t lane = packed ? (wl & 31u) : wl;              // lane inside the region
        const uint32_t g = packed ? g0 + (wl >> 5) : g0 + round;
        const uint32_t c0 = rt.cand_off[g], n = rt.cand_off[g + 1] - c0;
        uint32_t so = 0, len = 0, order = 0xFFFFFFFFu;
        uint16_t ks = 0;
        if (lane < n) {
            so = rt.seq_off[c0 + lane];
            len = rt.seq_off[c0 + lane + 1] - so;
            order = rt.order[c0 + lane];
            ks = rt.kscore[c0 + lane];
        }
        WaveStats w = wave_group_stats(sw, lane, n, so, len, rt.seq, ks > 0, order);
        const uint32_t min_c = min_support(n);
        uint8_t lable = 0;
        uint32_t ne = 0;
        if (lane < n) grp[c0 + lane] = (uint8_t)__builtin_ctzll(w.eqmask);
        if (w.max2_c >= min_c && n > 0) {
            const uint32_t l1 = sw.shfl(len, w.max1_p), l2 = sw.shfl(len, w.max2_p);
            const uint32_t s1 = sw.shfl(so, w.max1_p), s2 = sw.shfl(so, w.max2_p);
            bool het = (l1 == l2) || (n >= 6 && w.max2_c >= w.max1_c / 2);
            if (het) het = hp_differs_wave(sw, lane, rt.seq + s1, l1, rt.seq + s2, l2); // (uniform over the region's lanes)
            if (het) {
                lable = LB_HETE;
                if (lane < n && ks > 0 && w.stat < min_c) { // main.rs:934-943
                    ks = 0;
                    rt.kscore[c0 + lane] = 0;
                }
                const uint64_t valid = sw.ballot(lane < n && ks > 0);
                const bool ref_valid = (valid & 1ull) && sw.shfl(order, 0) == 0;
                if (lane < n && ks > 0 && order == 0 && lane != 0) atomicOr(err, 4u); // seq2 order == 0 assertion
                // a shard of a contig (np2_shard_*) votes only over the regions it owns; its neighbours see the same region
                // in their halo and stay silent there, so every HETE region of the contig is counted exactly once
                const uint32_t gs = lq_start[g];
                const bool owned = gs >= own_lo && gs < own_hi;
                if (owned && ref_valid && lane >= 1 && lane < n && ks > 0) { // pairs (ref, j): main.rs:972-980
                    const int wgt = ((w.eqmask & 1ull) != 0) ? 1 : -1;
                    if (asref) {
                        atomicAdd(&ref_w[order], wgt);
                        ref_seen[order] = 1;
                    }
                    if (wgt < 0 && !use_all) bad[order] = 1;
                }
                const uint64_t V = ref_valid ? (valid & ~1ull) : valid;
                const uint32_t m = __builtin_popcountll(V);
                if (m >= 2 && owned) {
                    ne = m * (m - 1) / 2;
                    if ((V >> lane) & 1ull) atomicMin(&first_reg[order], g);
                }
            }
        }
        if (lane == 0) {
            reg_lable[g] = lable;
            ecount[g] = ne;
        }
    }
}

// edges among the valid non-ref candidates of HETE regions: key = order_i << 32 | order_j (i < j), val = +1 / -1
__device__ __forceinline__ void k_edges_write(const uint32_t np2_bid, const uint32_t np2_nb, RegionTables rt, const uint8_t *__restrict__ reg_lable,
                                                     const uint8_t *__restrict__ grp,
                                                     const uint32_t *__restrict__ ecount,
                                                     const uint32_t *__restrict__ eoff, uint64_t *__restrict__ ekey,
                                                     uint32_t *__restrict__ eval) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)((np2_bid * blockDim.x + threadIdx.x) >> 6)); // (uniform)
    if (g >= rt.n_reg || ecount[g] == 0) return;
    const uint32_t c0 = rt.cand_off[g], n = rt.cand_off[g + 1] - c0;
    uint32_t order = 0xFFFFFFFFu, gi = 0;
    bool v = false;
    if (lane < n) {
        order = rt.order[c0 + lane];
        gi = grp[c0 + lane];
        v = rt.kscore[c0 + lane] > 0;
    }
    uint64_t V = __ballot(v);
    if ((V & 1ull) && __shfl(order, 0) == 0) V &= ~1ull;
    const bool mine = (V >> lane) & 1ull;
    const uint32_t cnt = mine ? __builtin_popcountll(lane == 63 ? 0ull : (V >> (lane + 1))) : 0u;
    uint32_t inc = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= (uint32_t)o) inc += t;
    }
    uint32_t o = eoff[g] + inc - cnt;
    // shuffles must be wave-uniform: iterate j uniformly
    for (uint32_t j = 1; j < n; ++j) {
        const uint32_t oj = __shfl(order, j), gj = __shfl(gi, j);
        if (mine && j > lane && ((V >> j) & 1ull)) {
            ekey[o] = ((uint64_t)order << 32) | oj;
            eval[o] = (gj == gi) ? 1u : 0xFFFFFFFFu;
            ++o;
        }
    }
}

// ---- banded pair accumulator --------------------------------------------------------------------------------
// Reads are numbered in alignment-start order, so the partners b > a of read a (reads sharing a HETE region with it)
// lie within a short index distance.  band[a * EDGE_BAND + (b - a - 1)] counts the regions in which the pair agrees
// (low half-word) and disagrees (high half-word): a diploid yeast contig produces ~3 M raw pair votes per Mb but only
// ~0.1 M distinct pairs, so accumulating in place replaces a 64-bit radix sort of the raw votes; the rows, read in
// order, ARE the sorted unique pair list.  Pairs further apart than the band bump *ovf: the host then takes the
// sort-based path below for the whole contig (deep pileups).
// One wavefront per read a: walk the HETE regions the read spans ([pj, pj + pcount) of the region list, from candidate
// extraction), and for every region in which a is a valid candidate add one vote per valid partner b > a to an LDS row
// (256 partners = 1 KiB; a region's partners are distinct, so a wave-wide ds_add never collides).  No global atomics:
// the ~3 M raw pair votes per Mb of a diploid contig stay on chip, the finished row is written once, together with
// its number of distinct partners.  Pairs further apart than the band bump *ovf (the host then sorts raw votes).
__device__ __forceinline__ void k_edges_row(const uint32_t np2_bid0, const uint32_t np2_nb, RegionTables rt, const uint8_t *__restrict__ grp,
                                            const uint32_t *__restrict__ ecount, const uint32_t *__restrict__ pj,
                                            const uint32_t *__restrict__ pcount, const uint8_t *__restrict__ alive, uint32_t R,
                                            uint32_t *__restrict__ band, uint32_t *__restrict__ row_n, uint32_t *__restrict__ ovf) {
    const uint32_t np2_bid = xcd_order(np2_bid0, np2_nb); // (neighbouring items on one XCD: np2_common.hpp)
    __shared__ uint32_t s_row[4][EDGE_BAND];
    const uint32_t lane = threadIdx.x & 63, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t a = np2_bid * 4 + wv;
    if (a >= R) return;
    uint32_t *row = s_row[wv];
    *reinterpret_cast<uint4 *>(row + lane * 4) = make_uint4(0, 0, 0, 0);
    uint32_t far = 0;
    const uint32_t n_span = (a != 0 && alive[a]) ? pcount[a] : 0u; // (the contig itself never enters a pair: main.rs:972-980)
    const uint32_t j0 = n_span ? pj[a] : 0u;
    // The walk over the read's HETE regions is a chain of dependent loads per region (candidate range -> read ids, groups,
    // k-scores): the candidate ranges of 64 regions are fetched at once (one coalesced load per lane), and two regions
    // of at most 32 candidates each (the usual case at 30x) share an iteration, one per half of the wave.
    const uint32_t hl = lane & 31, hsel = lane >> 5;
    for (uint32_t base = 0; base < n_span; base += 64) {
        const uint32_t gl = j0 + base + lane;
        const bool in = base + lane < n_span && gl < rt.n_reg;
        const uint32_t c0l = in ? rt.cand_off[gl] : 0u, c1l = in ? rt.cand_off[gl + 1] : 0u;
        uint64_t het = __ballot(in && ecount[gl] != 0);
        while (het) {
            const uint32_t i0 = (uint32_t)__builtin_ctzll(het); // (uniform)
            het &= het - 1;
            const uint32_t c0a = (uint32_t)__builtin_amdgcn_readlane((int)c0l, (int)i0);
            const uint32_t na = (uint32_t)__builtin_amdgcn_readlane((int)c1l, (int)i0) - c0a;
            uint32_t c0b = 0, nb = 0;
            bool two = false;
            if (het && na <= 32) {
                const uint32_t i1 = (uint32_t)__builtin_ctzll(het);
                c0b = (uint32_t)__builtin_amdgcn_readlane((int)c0l, (int)i1);
                nb = (uint32_t)__builtin_amdgcn_readlane((int)c1l, (int)i1) - c0b;
                two = nb <= 32;
                if (two) het &= het - 1;
            }
            // this lane's candidate: region a over the whole wave, or region a / b in the lower / upper half
            const uint32_t li = two ? hl : lane;
            const uint32_t c0 = (two && hsel) ? c0b : c0a, n = (two && hsel) ? nb : na;
            uint32_t order = 0xFFFFFFFFu, gi = 0;
            bool v = false;
            if (li < n) {
                order = rt.order[c0 + li];
                gi = grp[c0 + li];
                v = rt.kscore[c0 + li] > 0;
            }
            uint64_t V = __ballot(v);
            // the contig's own candidate (read 0, always first) takes no part in the pairs (main.rs:972-980)
            if ((V & 1ull) && __shfl(order, 0) == 0) V &= ~1ull;
            if (two && ((V >> 32) & 1ull) && __shfl(order, 32) == 0) V &= ~(1ull << 32);
            const uint64_t me = __ballot(order == a) & V;
            const uint64_t half_mask = two ? (hsel ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull) : ~0ull;
            const uint64_t mine = me & half_mask;
            if (!me) continue; // a is not a (valid) candidate of these regions
            const uint32_t ga = __shfl(gi, mine ? (int)__builtin_ctzll(mine) : 0);
            if (mine && ((V >> lane) & 1ull) && order > a) {
                const uint32_t d = order - a - 1;
                if (d < EDGE_BAND)
                    atomicAdd(&row[d], gi == ga ? 1u : 0x10000u);
                else
                    ++far;
            }
        }
    }
    const uint4 w = *reinterpret_cast<const uint4 *>(row + lane * 4); // (same lane wrote / the wave's ds_adds are done)
    uint32_t c = (w.x != 0) + (w.y != 0) + (w.z != 0) + (w.w != 0);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (c) *reinterpret_cast<uint4 *>(band + (uint64_t)a * EDGE_BAND + lane * 4) = w; // (rows without partners are never read)
    if (lane == 0) row_n[a] = c;
    for (int o = 32; o > 0; o >>= 1) far += __shfl_xor(far, o);
    if (far && lane == 0) atomicAdd(ovf, far);
}
// ... and their emission in (a, b) order: ukey = a << 32 | b, ucnt = agreeing regions | disagreeing regions << 16 (the
// host turns the counts into the weight sum(w), or -(#-1) if #(-1) >= 3, main.rs:996-1002 — after merging shards)
__device__ __forceinline__ void k_band_emit(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ band, uint32_t R,
                                            const uint32_t *__restrict__ row_off, uint64_t *__restrict__ ukey,
                                            uint32_t *__restrict__ ucnt, uint32_t *__restrict__ n_out, uint64_t key_add) {
    // (key_add: a shard's local read numbers become the contig's on the way out — s << 32 | s for a shift of s)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)((np2_bid * blockDim.x + threadIdx.x) >> 6)); // (uniform)
    if (a >= R) return;
    if (a == R - 1 && lane == 0) *n_out = row_off[R];
    if (row_off[a + 1] == row_off[a]) return;
    const uint4 w = *reinterpret_cast<const uint4 *>(band + (uint64_t)a * EDGE_BAND + lane * 4);
    const uint32_t v[4] = {w.x, w.y, w.z, w.w};
    const uint32_t c = (w.x != 0) + (w.y != 0) + (w.z != 0) + (w.w != 0);
    uint32_t inc = c;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= (uint32_t)o) inc += t;
    }
    uint32_t o = row_off[a] + inc - c;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (v[k]) {
            ukey[o] = (((uint64_t)a << 32) | (a + 1 + lane * 4 + k)) + key_add;
            ucnt[o] = v[k];
            ++o;
        }
}

// the compact form the plain pipeline reads back (4 bytes per pair instead of 12: the pair list is the largest thing a
// phasing pass sends over the bus): the rows keep their order, a row's read is implied by row_off
__device__ __forceinline__ void k_band_emit_compact(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ band, uint32_t R,
                                                    const uint32_t *__restrict__ row_off, uint32_t *__restrict__ pairs,
                                                    uint32_t *__restrict__ n_out, uint32_t *__restrict__ ovf) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)((np2_bid * blockDim.x + threadIdx.x) >> 6)); // (uniform)
    if (a >= R) return;
    if (a == R - 1 && lane == 0) *n_out = row_off[R];
    if (row_off[a + 1] == row_off[a]) return;
    const uint4 w = *reinterpret_cast<const uint4 *>(band + (uint64_t)a * EDGE_BAND + lane * 4);
    const uint32_t v[4] = {w.x, w.y, w.z, w.w};
    const uint32_t c = (w.x != 0) + (w.y != 0) + (w.z != 0) + (w.w != 0);
    uint32_t inc = c;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= (uint32_t)o) inc += t;
    }
    uint32_t o = row_off[a] + inc - c;
    bool big = false;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (v[k]) {
            const uint32_t same = v[k] & 0xFFFFu, neg = v[k] >> 16;
            big = big || same > VOTE_CNT_MAX || neg > VOTE_CNT_MAX;
            pairs[o] = (lane * 4 + k) | ((same & VOTE_CNT_MAX) << 8) | ((neg & VOTE_CNT_MAX) << 20);
            ++o;
        }
    if (big) atomicAdd(ovf, 1u);
}

// reduce sorted edges to per-pair counts: agreeing regions | disagreeing regions << 16
__device__ __forceinline__ void k_edge_reduce(const uint32_t np2_bid, const uint32_t np2_nb, const uint64_t *__restrict__ ekey, const uint32_t *__restrict__ eval, uint32_t n,
                              uint32_t *__restrict__ flag, uint32_t *__restrict__ wout) {
    uint32_t i = np2_bid * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = ekey[i];
    if (i > 0 && ekey[i - 1] == k) {
        flag[i] = 0;
        return;
    }
    int32_t sum = 0, neg = 0;
    for (uint32_t j = i; j < n && ekey[j] == k; ++j) {
        const int32_t w = (int32_t)eval[j];
        sum += w;
        neg += w < 0;
    }
    flag[i] = 1;
    wout[i] = (uint32_t)(sum + neg) | ((uint32_t)neg << 16);
}
__device__ __forceinline__ void k_edge_compact(const uint32_t np2_bid, const uint32_t np2_nb, const uint64_t *__restrict__ ekey, const uint32_t *__restrict__ flag,
                               const uint32_t *__restrict__ idx, const uint32_t *__restrict__ wout, uint32_t n,
                               uint64_t *__restrict__ ukey, uint32_t *__restrict__ uw, uint32_t *__restrict__ n_out) {
    uint32_t i = np2_bid * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) {
        ukey[idx[i]] = ekey[i];
        uw[idx[i]] = wout[i];
    }
    if (i == n - 1) *n_out = idx[i] + flag[i];
}

// ---- final pass: seeds -------------------------------------------------------------------------
__device__ __forceinline__ void k_seed(const uint32_t np2_bid0, const uint32_t np2_nb, RegionTables rt, int32_t max_indel_len,
                                              uint8_t *__restrict__ reg_lable, uint32_t *__restrict__ seed_cand,
                                              uint32_t *__restrict__ keep_n, uint32_t *__restrict__ keep_list,
                                              uint16_t *__restrict__ keep_ks, uint32_t *__restrict__ err) {
    const uint32_t np2_bid = xcd_order(np2_bid0, np2_nb); // (neighbouring items on one XCD: np2_common.hpp)
    // a wavefront owns two consecutive regions: side by side in its two halves when both have at most 32 candidates
    // (the usual case at 30x), otherwise one after the other over all 64 lanes
    const uint32_t wl = threadIdx.x & 63;
    const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((np2_bid * blockDim.x + threadIdx.x) >> 6)) * 2; // (uniform)
    if (g0 >= rt.n_reg) return;
    const bool has1 = g0 + 1 < rt.n_reg;
    const bool packed = has1 && rt.cand_off[g0 + 1] - rt.cand_off[g0] <= 32 && rt.cand_off[g0 + 2] - rt.cand_off[g0 + 1] <= 32;
    const uint32_t rounds = (has1 && !packed) ? 2u : 1u;
    for (uint32_t round = 0; round < rounds; ++round) {
        const SubWave sw{packed ? (wl & 32u) : 0u, packed};
        const uint32_t lane = packed ? (wl & 31u) : wl;              // lane inside the region
        const uint32_t g = packed ? g0 + (wl >> 5) : g0 + round;
        const uint32_t c0 = rt.cand_off[g], n = rt.cand_off[g + 1] - c0;
        if (n == 0) { // lqseq.seqs[max1_p] would be out of bounds
            if (lane == 0) {
                atomicOr(err, 8u);
                reg_lable[g] = 0;
                keep_n[g] = 0;
                seed_cand[g] = 0; // in-bounds dummy; the error flag aborts the polish at the next read-back
            }
            continue;
        }
        uint32_t so = 0, len = 0, order = 0xFFFFFFFFu;
        uint16_t ks = 0;
        if (lane < n) {
            so = rt.seq_off[c0 + lane];
            len = rt.seq_off[c0 + lane + 1] - so;
            order = rt.order[c0 + lane];
            ks = rt.kscore[c0 + lane];
        }
        WaveStats w = wave_group_stats(sw, lane, n, so, len, rt.seq, ks > 0, order);
        const uint32_t min_c = min_support(n);
        if (sw.shfl(order, 0) != 0) { // "the first lqseq is not ref."
            if (lane == 0) atomicOr(err, 16u);
        }
        // order_stat as a per-lane key (each candidate has its own read index)
        uint32_t key = w.head ? w.c : 0;
        {
            const bool has0 = sw.shfl((uint32_t)w.head, 0) != 0;
            uint32_t k0 = sw.shfl(key, 0);
            if (has0) {
                if (k0 > 1 && k0 < min_c) k0 = min_c;
            } else {
                const uint32_t cnt0 = __builtin_popcountll(sw.shfl((uint32_t)(w.eqmask & 0xFFFFFFFFu), 0)) +
                                      __builtin_popcountll(sw.shfl((uint32_t)(w.eqmask >> 32), 0));
                if (cnt0 > 1) k0 = min_c;
            }
            // no_dupseq_lqseq (main.rs:851-860): no two equal sequences among candidates 1..
            const bool dup = lane >= 1 && lane < n && lane < 63 && ((w.eqmask >> (lane + 1)) != 0);
            const bool nodup = sw.ballot(dup) == 0;
            if (w.max1_p != 0 && w.max1_c < min_c && (w.max1_c > 1 || nodup)) {
                if (lane == w.max1_p) key = min_c;
                k0 = min_c;
            } else if (w.max1_c < min_c) {
                k0 = min_c;
            }
            if (lane == 0) key = k0;
        }
        // retain_sort_seqs: stable sort by key descending, keep key >= min_c
        // (only kept candidates are ranked, and a candidate below min_c never outranks a kept one: walk the kept ones only)
        const bool kept = lane < n && key >= min_c;
        uint32_t rank = 0;
        for (uint64_t it = sw.ballot(kept); it; it &= it - 1) {
            const uint32_t j = (uint32_t)__builtin_ctzll(it);
            const uint32_t kj = sw.shfl(key, j);
            if (kept && (kj > key || (kj == key && j < lane))) ++rank;
        }
        uint32_t kn = __builtin_popcountll(sw.ballot(kept));
        if (kn == 0) {
            if (lane == 0) atomicOr(err, 32u); // lqseq.seqs[0] out of bounds after retain_sort_seqs
            kn = 0;
        }
        // candidate of rank 0
        const uint64_t r0mask = sw.ballot(kept && rank == 0);
        const uint32_t first = r0mask ? __builtin_ctzll(r0mask) : 0;
        uint32_t seed = w.max1_p;
        const int32_t d = (int32_t)sw.shfl(len, w.max1_p) - (int32_t)sw.shfl(len, first);
        const bool too_long = (d < 0 ? -d : d) > max_indel_len;
        uint8_t lable = LB_SUCC | LB_RECH;
        if (kn <= 1 || too_long) {
            seed = first;
            lable = LB_SUCC;
            kn = 0;
        }
        if (kept && kn) {
            keep_list[c0 + rank] = c0 + lane;
            keep_ks[c0 + rank] = ks;
        }
        if (lane == 0) {
            reg_lable[g] = lable;
            seed_cand[g] = c0 + seed;
            keep_n[g] = kn;
        }
    }
}

// ---- splice (update_consensus_with_lqseqs, main.rs:1027-1058) -----------------------------------------
// Searches in the (multi-megabyte) consensus position array are latency chains: a binary search is ~22 dependent
// loads.  Probe 16 evenly spaced points per round instead (independent loads, one round trip): 6 rounds for 4.6 M.
template <bool UPPER>
__device__ __forceinline__ uint32_t bound16_u32(const uint32_t *__restrict__ a, uint32_t n, uint32_t v, uint32_t lo0 = 0) {
    uint32_t lo = lo0, hi = n; // the answer lies in [lo, hi]
    while (lo < hi) {
        const uint32_t step = (hi - lo + 15) >> 4;
        uint32_t x[16]; // unconditional (clamped) loads: all 16 are in flight together
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) x[k] = a[min(lo + k * step, hi - 1)];
        uint32_t c = 0; // probes whose element lies before the answer (a prefix of the probes: the array is sorted)
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) c += (lo + k * step < hi && (UPPER ? x[k] <= v : x[k] < v)) ? 1u : 0u;
        const uint32_t n_probe = (hi - lo + step - 1) / step; // probes with m < hi
        const uint32_t nlo = c ? lo + (c - 1) * step + 1 : lo;
        const uint32_t nhi = c < n_probe ? lo + c * step : hi;
        lo = nlo, hi = nhi;
    }
    return lo;
}
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint32_t v) {
    return bound16_u32<false>(a, n, v);
}
__device__ __forceinline__ uint32_t upper_bound_u32(const uint32_t *a, uint32_t n, uint32_t v) {
    return bound16_u32<true>(a, n, v);
}

// per labelled region: [idx_s, idx_e) to delete; the cursor gets stuck at the leftmost (highest index)
// labelled region whose start position no longer exists in the consensus
__device__ __forceinline__ void k_splice_find(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ cns_pos, const uint32_t *__restrict__ M_p,
                              const uint32_t *__restrict__ lq_start,
                              const uint32_t *__restrict__ lq_end, const uint8_t *__restrict__ reg_lable,
                              uint8_t lable, uint32_t n_reg, uint32_t *__restrict__ idx_s, uint32_t *__restrict__ idx_e,
                              uint32_t *__restrict__ stuck) {
    uint32_t g = np2_bid * blockDim.x + threadIdx.x;
    if (g >= n_reg || !(reg_lable[g] & lable)) return;
    const uint32_t M = *M_p;
    const uint32_t s = lower_bound_u32(cns_pos, M, lq_start[g]);
    const bool found = s < M && cns_pos[s] == lq_start[g];
    idx_s[g] = s;
    // (the region's end lies a region's length behind its start: one or two rounds over that stretch instead of six over the
    // whole consensus, whenever the element that closes the stretch is already past the end)
    uint32_t e = s;
    if (found) {
        const uint32_t en = lq_end[g];
        const uint32_t near = (uint32_t)min((uint64_t)M, (uint64_t)s + (en - lq_start[g]) + 65u);
        e = (near == M || cns_pos[near - 1] > en) ? bound16_u32<true>(cns_pos, near, en, s) : upper_bound_u32(cns_pos, M, en);
        e = max(s, e);
    }
    idx_e[g] = e;
    if (!found) atomicMax(stuck, g + 1);
}
static constexpr uint32_t LB_REG_ITEMS = 4; // regions per thread of the look-back compaction kernels
// Applied regions in left -> right order (reverse region index): flag each region, compact the applied ones into
// slots with their payload and prefix-sum the length changes -- one pass with a decoupled look-back across blocks.
// The consensus length is chained on the device (*M_out = *M_in + total shift), so the host does not have to read
// anything back between splice rounds.
__device__ __forceinline__ void k_splice_plan(const uint32_t np2_bid, const uint32_t np2_nb, Lookback lb, uint32_t n_blocks, const uint8_t *__restrict__ reg_lable,
                                                     uint8_t lable, uint32_t n_reg, const uint32_t *__restrict__ stuck,
                                                     const uint32_t *__restrict__ idx_s, const uint32_t *__restrict__ idx_e,
                                                     const uint32_t *__restrict__ seed_cand,
                                                     const uint32_t *__restrict__ seq_off, uint32_t *__restrict__ ap_g,
                                                     uint32_t *__restrict__ ap_s, uint32_t *__restrict__ ap_e,
                                                     int32_t *__restrict__ ap_delta, int32_t *__restrict__ ap_shift_incl,
                                                     uint32_t *__restrict__ n_ap, const uint32_t *__restrict__ M_in,
                                                     uint32_t *__restrict__ M_out, uint32_t out_cap, uint32_t *__restrict__ err) {
    __shared__ uint32_t sh[8];
    const uint32_t bid = lb_block_id(lb, sh);
    // four consecutive regions per thread (a quarter of the blocks: the look-back is a chain over the blocks)
    uint32_t flag[LB_REG_ITEMS], g[LB_REG_ITEMS], s[LB_REG_ITEMS], e[LB_REG_ITEMS];
    int32_t delta[LB_REG_ITEMS];
    uint32_t fsum = 0;
    int32_t dsum_t = 0;
#pragma unroll
    for (uint32_t k = 0; k < LB_REG_ITEMS; ++k) {
        const uint32_t rr = (bid * 256 + threadIdx.x) * LB_REG_ITEMS + k;
        flag[k] = 0, g[k] = 0, s[k] = 0, e[k] = 0, delta[k] = 0;
        if (rr < n_reg) {
            g[k] = n_reg - 1 - rr;
            flag[k] = ((reg_lable[g[k]] & lable) && g[k] + 1 > *stuck) ? 1u : 0u;
            if (flag[k]) {
                const uint32_t c = seed_cand[g[k]];
                s[k] = idx_s[g[k]], e[k] = idx_e[g[k]];
                delta[k] = (int32_t)(seq_off[c + 1] - seq_off[c]) - (int32_t)(e[k] - s[k]);
            }
        }
        fsum += flag[k];
        dsum_t += delta[k];
    }
    uint32_t cnt, dsum;
    uint32_t lo = block_excl_scan<OpAdd, 4>(fsum, sh, cnt);
    uint32_t ld = block_excl_scan<OpAdd, 4>((uint32_t)dsum_t, sh, dsum);
    uint32_t pre_c, pre_d;
    lb_exclusive2(lb, bid, cnt, dsum, sh, err, pre_c, pre_d);
#pragma unroll
    for (uint32_t k = 0; k < LB_REG_ITEMS; ++k) {
        if (flag[k]) {
            const uint32_t o = pre_c + lo;
            ap_g[o] = g[k];
            ap_s[o] = s[k];
            ap_e[o] = e[k];
            ap_delta[o] = delta[k];
            ap_shift_incl[o] = (int32_t)(pre_d + ld) + delta[k];
        }
        lo += flag[k];
        ld += (uint32_t)delta[k];
    }
    if (bid == n_blocks - 1 && threadIdx.x == 0) {
        const uint32_t m_out = *M_in + pre_d + dsum;
        if (m_out > out_cap) {
            // the output buffers were sized with a guessed growth allowance and this round would outgrow them: nothing is
            // spliced (the consensus is copied as it is — that fits), the host sees GROW_ERR and repeats the pass with the
            // exact allowance
            atomicOr(err, GROW_ERR);
            *n_ap = 0;
            *M_out = *M_in;
        } else {
            *n_ap = pre_c + cnt;
            *M_out = m_out;
        }
    }
}
// copy the bases outside the applied regions to their shifted places.  A block covers SPLICE_SPAN consecutive consensus
// indices (coalesced, several per thread): the slot search — a latency chain over the applied-region starts — is done once
// per block for its first and last index with 16 probes in flight per round, threads only search the (usually empty)
// slot range in between.
static constexpr uint32_t SPLICE_SPAN = 2048;
static constexpr uint32_t SPLICE_LDS_SLOTS = 256; // slots of a block's span staged in LDS (a span of 2048 bases holds ~20 of them)
__device__ __forceinline__ void k_splice_bases(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ in_pos,
                                                      const uint8_t *__restrict__ in_base,
                                                      const uint32_t *__restrict__ M_p,
                                                      const uint32_t *__restrict__ ap_s, const uint32_t *__restrict__ ap_e,
                                                      const int32_t *__restrict__ ap_shift_incl,
                                                      const uint32_t *__restrict__ n_ap_p, uint32_t *__restrict__ out_pos,
                                                      uint8_t *__restrict__ out_base) {
    __shared__ uint32_t s_lo[2];
    __shared__ uint32_t s_s[SPLICE_LDS_SLOTS], s_e[SPLICE_LDS_SLOTS];
    __shared__ int32_t s_sh[SPLICE_LDS_SLOTS];
    const uint32_t i0 = np2_bid * SPLICE_SPAN;
    const uint32_t n_ap = *n_ap_p, M = *M_p;
    if (i0 >= M) return;
    // (the block's elements first: their loads are in flight while the slot range is found and staged)
    uint32_t pv[SPLICE_SPAN / 256];
    uint8_t bv[SPLICE_SPAN / 256];
#pragma unroll
    for (uint32_t k = 0; k < SPLICE_SPAN / 256; ++k) {
        const uint32_t i = min(i0 + k * 256 + threadIdx.x, M - 1);
        pv[k] = in_pos[i], bv[k] = in_base[i];
    }
    if (threadIdx.x < 2) // number of slots with ap_s <= first / last index of the block
        s_lo[threadIdx.x] = upper_bound_u32(ap_s, n_ap, threadIdx.x == 0 ? i0 : min(M - 1, i0 + SPLICE_SPAN - 1));
    __syncthreads();
    // slots blo - 1 (the one the block's first index may lie in or behind) .. bhi - 1
    const uint32_t blo = s_lo[0], bhi = s_lo[1], b0 = blo ? blo - 1 : 0u;
    const bool staged = bhi - b0 <= SPLICE_LDS_SLOTS; // (uniform)
    if (staged && b0 + threadIdx.x < bhi) {
        s_s[threadIdx.x] = ap_s[b0 + threadIdx.x];
        s_e[threadIdx.x] = ap_e[b0 + threadIdx.x];
        s_sh[threadIdx.x] = ap_shift_incl[b0 + threadIdx.x];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < SPLICE_SPAN / 256; ++k) {
        const uint32_t i = i0 + k * 256 + threadIdx.x;
        if (i >= M) break;
        uint32_t lo = blo, hi = bhi;
        int64_t o = i;
        if (staged) {
            while (lo < hi) { // last slot with ap_s <= i
                const uint32_t mid = (lo + hi) >> 1;
                if (s_s[mid - b0] <= i) lo = mid + 1; else hi = mid;
            }
            if (lo > 0) {
                if (i < s_e[lo - 1 - b0]) continue; // replaced by the region's seed
                o += s_sh[lo - 1 - b0];
            }
        } else {
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ap_s[mid] <= i) lo = mid + 1; else hi = mid;
            }
            if (lo > 0) {
                const uint32_t sl = lo - 1;
                if (i < ap_e[sl]) continue;
                o += ap_shift_incl[sl];
            }
        }
        out_pos[o] = pv[k];
        out_base[o] = bv[k];
    }
}
__device__ __forceinline__ void k_splice_seeds(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ ap_g, const uint32_t *__restrict__ ap_s,
                               const int32_t *__restrict__ ap_delta, const int32_t *__restrict__ ap_shift_incl,
                               const uint32_t *__restrict__ n_ap_p, const uint32_t *__restrict__ lq_start,
                               const uint32_t *__restrict__ seed_cand, const uint32_t *__restrict__ seq_off,
                               const uint8_t *__restrict__ seq, uint32_t *__restrict__ out_pos,
                               uint8_t *__restrict__ out_base) {
    // eight lanes per applied region (a seed string is a few dozen bases: one or two rounds of stores instead of a serial walk)
    const uint32_t sl = (np2_bid * blockDim.x + threadIdx.x) >> 3, q = threadIdx.x & 7u;
    if (sl >= *n_ap_p) return;
    const uint32_t g = ap_g[sl], c = seed_cand[g];
    const uint32_t so = seq_off[c], len = seq_off[c + 1] - so;
    const int64_t o = (int64_t)ap_s[sl] + (ap_shift_incl[sl] - ap_delta[sl]);
    const uint32_t p = lq_start[g];
    for (uint32_t t = q; t < len; t += 8) { // spliced bases all carry pos == start (main.rs:1039-1045)
        out_pos[o + t] = p;
        out_base[o + t] = seq[so + t];
    }
}

// shards of one contig: where the positions t[0..6) begin in the (position-ordered) consensus, and the positions of the
// first / last base of [idx(t[1]), idx(t[4])) — the slice a shard owns
__device__ __forceinline__ void k_shard_bounds(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ cns_pos, const uint32_t *__restrict__ M_p,
                                               uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t t4, uint32_t t5,
                                               uint32_t *__restrict__ out) {
    __shared__ uint32_t s_i[6];
    const uint32_t M = *M_p;
    const uint32_t t[6] = {t0, t1, t2, t3, t4, t5};
    if (threadIdx.x < 6) {
        uint32_t tv = t[0];
#pragma unroll
        for (uint32_t k = 1; k < 6; ++k) tv = threadIdx.x == k ? t[k] : tv;
        const uint32_t i = lower_bound_u32(cns_pos, M, tv);
        out[threadIdx.x] = i;
        s_i[threadIdx.x] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t a = s_i[1], b = s_i[4];
        out[6] = b > a ? cns_pos[a] : 0u;
        out[7] = b > a ? cns_pos[b - 1] : 0u;
    }
}

// ---- recheck (reupdate_consensus_with_lqseqs, main.rs:1060-1420) ------------------------------------
// RECH regions in left -> right order (reverse region index), compacted with a look-back across blocks
__device__ __forceinline__ void k_rech_list(const uint32_t np2_bid, const uint32_t np2_nb, Lookback lb, uint32_t n_blocks, const uint8_t *__restrict__ reg_lable,
                                                   uint32_t n_reg, uint32_t *__restrict__ rech, uint32_t *__restrict__ n_rech,
                                                   unsigned long long *__restrict__ blob_bound, uint32_t *__restrict__ err) {
    __shared__ uint32_t sh[8];
    if (np2_bid == 0 && threadIdx.x == 0) *blob_bound = 0; // accumulated by k_rech_groups, the next kernel
    const uint32_t bid = lb_block_id(lb, sh);
    uint32_t flag[LB_REG_ITEMS], fsum = 0;
#pragma unroll
    for (uint32_t k = 0; k < LB_REG_ITEMS; ++k) {
        const uint32_t rr = (bid * 256 + threadIdx.x) * LB_REG_ITEMS + k;
        flag[k] = (rr < n_reg && (reg_lable[n_reg - 1 - rr] & LB_RECH)) ? 1u : 0u;
        fsum += flag[k];
    }
    uint32_t cnt, pre, dummy;
    uint32_t lo = block_excl_scan<OpAdd, 4>(fsum, sh, cnt);
    lb_exclusive2(lb, bid, cnt, 0u, sh, err, pre, dummy);
#pragma unroll
    for (uint32_t k = 0; k < LB_REG_ITEMS; ++k) {
        if (flag[k]) rech[pre + lo] = n_reg - 1 - ((bid * 256 + threadIdx.x) * LB_REG_ITEMS + k);
        lo += flag[k];
    }
    if (bid == n_blocks - 1 && threadIdx.x == 0) *n_rech = pre + cnt;
}

struct RechGroup { // one recheck group = 1..6 chained RECH regions
    uint32_t first, n;       // range in rech[]
    uint32_t sl, el, sr, er; // flank index ranges in the consensus
    uint32_t bs[5], be[5];   // consensus ranges between consecutive regions
    uint32_t lens[6];        // kept-candidate counts (product radix)
    uint32_t njobs;
};

// chain grouping (main.rs:1196-1206): natural chains (next.start < prev.end + k) cut every 6 regions.  Two kernels:
// k_rech_groups builds, one thread per RECH region, the group of every region that heads one (the searches for its
// flanks in the consensus are chains of dependent loads: no block waits for another here) into a scratch slot under
// the region's own index; k_rech_compact then moves the groups to their slots and leaves the job offsets (exclusive
// sums of the group / job counts) with a look-back over a few blocks of four regions per thread.  As one kernel the
// look-back chain ran over every block of the heavy threads (120 us per call on the yeast-sized assembly).
__device__ __forceinline__ void k_rech_groups(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ rech,
                                                     const uint32_t *__restrict__ n_rech_p,
                                                     const uint32_t *__restrict__ cns_pos, const uint32_t *__restrict__ M_p,
                                                     const uint32_t *__restrict__ lq_start,
                                                     const uint32_t *__restrict__ lq_end,
                                                     const uint32_t *__restrict__ keep_n, uint32_t ksize,
                                                     const uint32_t *__restrict__ reg_maxlen,
                                                     RechGroup *__restrict__ tmp_groups, uint32_t *__restrict__ head_jobs,
                                                     unsigned long long *__restrict__ blob_bound,
                                                     uint32_t *__restrict__ err) {
    __shared__ unsigned long long s_bound[4];
    unsigned long long bound = 0; // upper bound of the bytes of this group's recheck strings
    const uint32_t e = np2_bid * 256 + threadIdx.x;
    const uint32_t n_rech = *n_rech_p, M = *M_p;
    auto chained = [&](uint32_t x) { return lq_start[rech[x]] < lq_end[rech[x - 1]] + ksize; };
    bool head = false;
    RechGroup G = {};
    uint32_t jobs32 = 0;
    if (e < n_rech) {
        uint32_t back = 0, x = e;
        while (x > 0 && chained(x)) {
            --x;
            ++back;
        }
        head = back % 6 == 0;
    }
    if (head) {
        G.first = e;
        uint32_t n = 1;
        while (n < 6 && e + n < n_rech && chained(e + n)) ++n;
        G.n = n;
        const uint32_t l = ksize - 1;
        // iter_consensus_extend(toleft): first index with pos >= start; the reference indexes [i-1]
        const uint32_t p0 = lq_start[rech[e]];
        const uint32_t i0 = lower_bound_u32(cns_pos, M, p0);
        if (i0 == 0 || i0 >= M) atomicOr(err, 64u);
        G.el = i0;
        G.sl = i0 > l ? i0 - l : 0;
        // iter_consensus_extend(right): last index with pos <= end; the reference indexes [i+1]
        const uint32_t p1 = lq_end[rech[e + n - 1]];
        const uint32_t i1u = upper_bound_u32(cns_pos, M, p1);
        if (i1u == 0 || i1u >= M) atomicOr(err, 64u);
        const uint32_t i1 = i1u ? i1u - 1 : 0;
        G.sr = i1 + 1;
        G.er = (i1 + l < M) ? i1 + l + 1 : M;
        uint64_t jobs = 1;
#pragma unroll // (static indices keep G in registers; a dynamically indexed G lives in scratch memory)
        for (uint32_t x = 0; x < 6; ++x) {
            if (x >= n) break;
            G.lens[x] = keep_n[rech[e + x]];
            jobs *= G.lens[x];
            if (jobs > 0x7FFFFFFFull) {
                atomicOr(err, 128u);
                jobs = 0;
            }
            if (x + 1 < n) { // iter_consensus_region(s, e): indices with s < pos < e
                const uint32_t s = lq_end[rech[e + x]], en = lq_start[rech[e + x + 1]];
                if (s + 1 == en) {
                    G.bs[x] = G.be[x] = 0;
                } else {
                    G.bs[x] = upper_bound_u32(cns_pos, M, s);
                    G.be[x] = lower_bound_u32(cns_pos, M, en);
                    if (G.be[x] < G.bs[x]) G.be[x] = G.bs[x];
                }
            }
        }
        jobs32 = (uint32_t)jobs;
        G.njobs = jobs32;
        // every string of the group: both flanks + per region its longest kept candidate + the stretches in between
        uint64_t len = (uint64_t)(G.el - G.sl) + (G.er - G.sr);
#pragma unroll
        for (uint32_t x = 0; x < 6; ++x) {
            if (x >= n) break;
            len += reg_maxlen[rech[e + x]]; // longest candidate of the region (the kept ones are a subset)
            if (x + 1 < n) len += G.be[x] - G.bs[x];
        }
        bound = (unsigned long long)jobs32 * len;
        tmp_groups[e] = G;
    }
    // head_jobs[e]: 0 = not a group head, else 1 + the group's job count (a head may have 0 jobs)
    if (e < n_rech) head_jobs[e] = head ? jobs32 + 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) bound += __shfl_xor(bound, o);
    if ((threadIdx.x & 63) == 0) s_bound[threadIdx.x >> 6] = bound;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long b = s_bound[0] + s_bound[1] + s_bound[2] + s_bound[3];
        if (b) atomicAdd(blob_bound, b);
    }
}
__device__ __forceinline__ void k_rech_compact(const uint32_t np2_bid, const uint32_t np2_nb, Lookback lb, uint32_t n_blocks,
                                                      const uint32_t *__restrict__ n_rech_p,
                                                      const uint32_t *__restrict__ head_jobs,
                                                      const RechGroup *__restrict__ tmp_groups,
                                                      RechGroup *__restrict__ groups, uint32_t *__restrict__ job_off,
                                                      uint32_t *__restrict__ n_groups, uint32_t *__restrict__ n_jobs,
                                                      uint32_t *__restrict__ err) {
    __shared__ uint32_t sh[8];
    const uint32_t bid = lb_block_id(lb, sh);
    const uint32_t n_rech = *n_rech_p;
    uint32_t hj[LB_REG_ITEMS], hs = 0, js = 0;
#pragma unroll
    for (uint32_t k = 0; k < LB_REG_ITEMS; ++k) {
        const uint32_t e = (bid * 256 + threadIdx.x) * LB_REG_ITEMS + k;
        hj[k] = e < n_rech ? head_jobs[e] : 0u;
        hs += hj[k] ? 1u : 0u;
        js += hj[k] ? hj[k] - 1u : 0u;
    }
    uint32_t nh, nj, pre_h, pre_j;
    uint32_t lh = block_excl_scan<OpAdd, 4>(hs, sh, nh);
    uint32_t lj = block_excl_scan<OpAdd, 4>(js, sh, nj);
    lb_exclusive2(lb, bid, nh, nj, sh, err, pre_h, pre_j);
    if (pre_j + nj < pre_j) atomicOr(err, 128u); // total job count overflows 32 bits
#pragma unroll
    for (uint32_t k = 0; k < LB_REG_ITEMS; ++k) {
        if (!hj[k]) continue;
        const uint32_t e = (bid * 256 + threadIdx.x) * LB_REG_ITEMS + k;
        groups[pre_h + lh] = tmp_groups[e];
        job_off[pre_h + lh] = pre_j + lj;
        ++lh;
        lj += hj[k] - 1u;
    }
    if (bid == n_blocks - 1 && threadIdx.x == 0) {
        *n_groups = pre_h + nh;
        *n_jobs = pre_j + nj;
        job_off[pre_h + nh] = pre_j + nj;
    }
}

struct RechCtx {
    const RechGroup *groups;
    const uint32_t *job_off; // per group
    const uint32_t *rech;
    const uint32_t *cand_off;
    const uint32_t *keep_list;
    const uint32_t *seq_off;
    const uint8_t *seq;
    const uint8_t *cns_base;
    uint32_t n_groups;
};
__device__ __forceinline__ uint32_t job_group(const RechCtx &cx, uint32_t job) {
    uint32_t lo = 0, hi = cx.n_groups; // last group with job_off <= job
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cx.job_off[mid] <= job) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}
// length (WRITE = false) or bytes (WRITE = true) of one recheck string:
// left flank + seq_0 [+ between_0 + seq_1 ...] + right flank  (main.rs:1141-1176, 1224-1230)
template <bool WRITE> __device__ uint32_t rech_string(const RechCtx &cx, uint32_t job, uint8_t *__restrict__ out) {
    const uint32_t gi = job_group(cx, job);
    const RechGroup &G = cx.groups[gi];
    uint32_t rem = job - cx.job_off[gi];
    uint32_t pick[6];
    for (uint32_t x = G.n; x-- > 0;) { // last iterator fastest
        pick[x] = rem % G.lens[x];
        rem /= G.lens[x];
    }
    uint32_t o = 0;
    for (uint32_t i = G.sl; i < G.el; ++i, ++o)
        if (WRITE) out[o] = cx.cns_base[i];
    for (uint32_t x = 0; x < G.n; ++x) {
        const uint32_t g = cx.rech[G.first + x];
        const uint32_t c = cx.keep_list[cx.cand_off[g] + pick[x]];
        const uint32_t so = cx.seq_off[c], len = cx.seq_off[c + 1] - so;
        for (uint32_t t = 0; t < len; ++t, ++o)
            if (WRITE) out[o] = cx.seq[so + t];
        if (x + 1 < G.n)
            for (uint32_t i = G.bs[x]; i < G.be[x]; ++i, ++o)
                if (WRITE) out[o] = cx.cns_base[i];
    }
    for (uint32_t i = G.sr; i < G.er; ++i, ++o)
        if (WRITE) out[o] = cx.cns_base[i];
    return o;
}
__device__ __forceinline__ void k_rech_job_len(const uint32_t np2_bid, const uint32_t np2_nb, RechCtx cx, uint32_t n_jobs, uint32_t *__restrict__ len) {
    uint32_t j = np2_bid * blockDim.x + threadIdx.x;
    if (j < n_jobs) len[j] = rech_string<false>(cx, j, nullptr);
}
__device__ __forceinline__ void k_rech_job_build(const uint32_t np2_bid, const uint32_t np2_nb, RechCtx cx, uint32_t n_jobs, const uint64_t *__restrict__ soff,
                                 uint8_t *__restrict__ blob) {
    uint32_t j = np2_bid * blockDim.x + threadIdx.x;
    if (j < n_jobs) rech_string<true>(cx, j, blob + soff[j]);
}
__device__ __forceinline__ void k_u32_to_u64_off(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ in, uint32_t n, uint64_t *__restrict__ out) {
    uint32_t i = np2_bid * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// apply the scores: single regions take their job's score; chains zero everything, then every product with
// score > 0 stamps its members, later products overwriting earlier ones (main.rs:1358-1366)
__device__ __forceinline__ void k_rech_apply(const uint32_t np2_bid, const uint32_t np2_nb, RechCtx cx, const uint16_t *__restrict__ score, uint16_t *__restrict__ keep_ks) {
    uint32_t gi = np2_bid * blockDim.x + threadIdx.x;
    if (gi >= cx.n_groups) return;
    const RechGroup &G = cx.groups[gi];
    const uint32_t j0 = cx.job_off[gi];
    if (G.n == 1) {
        const uint32_t g = cx.rech[G.first];
        for (uint32_t t = 0; t < G.lens[0]; ++t) keep_ks[cx.cand_off[g] + t] = score[j0 + t];
        return;
    }
    for (uint32_t x = 0; x < G.n; ++x) {
        const uint32_t g = cx.rech[G.first + x];
        for (uint32_t t = 0; t < G.lens[x]; ++t) keep_ks[cx.cand_off[g] + t] = 0;
    }
    for (uint32_t j = 0; j < G.njobs; ++j) {
        const uint16_t ks = score[j0 + j];
        if (!ks) continue;
        uint32_t rem = j;
        for (uint32_t x = G.n; x-- > 0;) {
            const uint32_t p = rem % G.lens[x];
            rem /= G.lens[x];
            keep_ks[cx.cand_off[cx.rech[G.first + x]] + p] = ks;
        }
    }
}

// selection per RECH region (main.rs:1371-1406)
__device__ __forceinline__ void k_rech_select(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ rech, const uint32_t *__restrict__ n_rech_p,
                              const uint32_t *__restrict__ cand_off, const uint32_t *__restrict__ keep_n,
                              const uint32_t *__restrict__ keep_list, const uint16_t *__restrict__ keep_ks,
                              const uint32_t *__restrict__ order, uint32_t first_yak, uint8_t *__restrict__ reg_lable,
                              uint32_t *__restrict__ seed_cand) {
    uint32_t e = np2_bid * blockDim.x + threadIdx.x;
    if (e >= *n_rech_p) return;
    const uint32_t g = rech[e], c0 = cand_off[g], n = keep_n[g];
    uint32_t c = 0, valid = 0;
    for (uint32_t p = 0; p < n; ++p)
        if (keep_ks[c0 + p] != 0) {
            if (c == 0 || order[keep_list[c0 + p]] == 0) c = p + 1;
            ++valid;
        }
    if (valid > 1) reg_lable[g] |= LB_TEMP;
    if (c != 0) {
        seed_cand[g] = keep_list[c0 + c - 1];
    } else if (first_yak) { // keep the contig's own sequence if every candidate is invalid
        uint32_t i = 0;
        for (uint32_t p = 0; p < n; ++p)
            if (order[keep_list[c0 + p]] == 0) {
                i = p;
                break;
            }
        seed_cand[g] = keep_list[c0 + i];
    }
}
__device__ __forceinline__ void k_rech_relabel(const uint32_t np2_bid, const uint32_t np2_nb, uint8_t *__restrict__ reg_lable, uint32_t n_reg) { // main.rs:1411-1417
    uint32_t g = np2_bid * blockDim.x + threadIdx.x;
    if (g >= n_reg) return;
    const uint8_t l = reg_lable[g];
    if (!(l & LB_RECH)) return;
    reg_lable[g] = (l & LB_TEMP) ? (uint8_t)(l ^ LB_TEMP) : (uint8_t)(l ^ LB_RECH);
}

// reads that vote in some HETE region (graph keys) / reads flagged invalid: when both are zero the phasing pass has
// nothing to decide and the host skips the vote read-back altogether
__device__ __forceinline__ void k_vote_counts(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ first_reg, const uint8_t *__restrict__ bad,
                                                      uint32_t R, uint32_t *__restrict__ out) {
    __shared__ uint32_t sk[16], sb[16];
    uint32_t nk = 0, nb = 0;
    for (uint32_t r = threadIdx.x; r < R; r += 1024) {
        nk += first_reg[r] != 0xFFFFFFFFu ? 1u : 0u;
        nb += bad[r] ? 1u : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) {
        nk += __shfl_xor(nk, o);
        nb += __shfl_xor(nb, o);
    }
    if ((threadIdx.x & 63) == 0) {
        sk[threadIdx.x >> 6] = nk;
        sb[threadIdx.x >> 6] = nb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 0;
        for (int i = 0; i < 16; ++i) {
            a += sk[i];
            b += sb[i];
        }
        out[0] = a;
        out[1] = b;
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static inline dim3 g1(uint64_t n, uint32_t bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

void launch_vote_phase(hipStream_t s, const RegionTables &rt, bool asref, bool use_all, const uint32_t *lq_start,
                       uint32_t own_lo, uint32_t own_hi, uint8_t *reg_lable, uint8_t *grp, uint32_t *ecount, int32_t *ref_w,
                       uint8_t *ref_seen, uint8_t *bad, uint32_t *first_reg, uint32_t *err) {
    if (rt.n_reg)
        NP2_LAUNCH(k_vote_phase, g1((uint64_t)((rt.n_reg + 1) / 2) * 64), 256, s, rt, asref ? 1u : 0u, use_all ? 1u : 0u, lq_start, own_lo, own_hi, reg_lable, grp, ecount, ref_w, ref_seen, bad, first_reg, err);
}
void launch_vote_counts(hipStream_t s, const uint32_t *first_reg, const uint8_t *bad, uint32_t R, uint32_t *out) {
    NP2_LAUNCH(k_vote_counts, dim3(1), 1024, s, first_reg, bad, R, out);
}
void launch_edges_write(hipStream_t s, const RegionTables &rt, const uint8_t *reg_lable, const uint8_t *grp,
                        const uint32_t *ecount, const uint32_t *eoff, uint64_t *ekey, uint32_t *eval) {
    if (rt.n_reg)
        NP2_LAUNCH(k_edges_write, g1((uint64_t)rt.n_reg * 64), 256, s, rt, reg_lable, grp, ecount, eoff, ekey, eval);
}
void launch_edges_row(hipStream_t s, const RegionTables &rt, const uint8_t *grp, const uint32_t *ecount, const uint32_t *pj,
                      const uint32_t *pcount, const uint8_t *alive, uint32_t R, uint32_t *band, uint32_t *row_n, uint32_t *ovf) {
    if (R) NP2_LAUNCH(k_edges_row, g1((uint64_t)R * 64), 256, s, rt, grp, ecount, pj, pcount, alive, R, band, row_n, ovf);
}
void launch_band_emit_compact(hipStream_t s, const uint32_t *band, uint32_t R, const uint32_t *row_off, uint32_t *pairs,
                              uint32_t *n_out, uint32_t *ovf) {
    if (R) NP2_LAUNCH(k_band_emit_compact, g1((uint64_t)R * 64), 256, s, band, R, row_off, pairs, n_out, ovf);
}
void launch_band_emit(hipStream_t s, const uint32_t *band, uint32_t R, const uint32_t *row_off, uint64_t *ukey, uint32_t *uw,
                      uint32_t *n_out, uint64_t key_add) {
    if (R) NP2_LAUNCH(k_band_emit, g1((uint64_t)R * 64), 256, s, band, R, row_off, ukey, uw, n_out, key_add);
}
void launch_edge_reduce(hipStream_t s, const uint64_t *ekey, const uint32_t *eval, uint32_t n, uint32_t *flag,
                        uint32_t *wout) {
    if (n) NP2_LAUNCH(k_edge_reduce, g1(n), 256, s, ekey, eval, n, flag, wout);
}
void launch_edge_compact(hipStream_t s, const uint64_t *ekey, const uint32_t *flag, const uint32_t *idx,
                         const uint32_t *wout, uint32_t n, uint64_t *ukey, uint32_t *uw, uint32_t *n_out) {
    if (n) NP2_LAUNCH(k_edge_compact, g1(n), 256, s, ekey, flag, idx, wout, n, ukey, uw, n_out);
}
void launch_seed(hipStream_t s, const RegionTables &rt, int32_t max_indel_len, uint8_t *reg_lable, uint32_t *seed_cand,
                 uint32_t *keep_n, uint32_t *keep_list, uint16_t *keep_ks, uint32_t *err) {
    if (rt.n_reg)
        NP2_LAUNCH(k_seed, g1((uint64_t)((rt.n_reg + 1) / 2) * 64), 256, s, rt, max_indel_len, reg_lable, seed_cand, keep_n, keep_list, keep_ks, err);
}
uint32_t region_lb_blocks(uint32_t n_reg) { return (n_reg + 256 * LB_REG_ITEMS - 1) / (256 * LB_REG_ITEMS); }
void launch_splice_find(hipStream_t s, const uint32_t *cns_pos, const uint32_t *M_p, const uint32_t *lq_start,
                        const uint32_t *lq_end, const uint8_t *reg_lable, uint8_t lable, uint32_t n_reg,
                        uint32_t *idx_s, uint32_t *idx_e, uint32_t *stuck) {
    NP2_LAUNCH(k_splice_find, g1(n_reg), 256, s, cns_pos, M_p, lq_start, lq_end, reg_lable, lable, n_reg, idx_s, idx_e, stuck);
}
void launch_splice_plan(hipStream_t s, const Lookback &lb, const uint8_t *reg_lable, uint8_t lable, uint32_t n_reg,
                        const uint32_t *stuck, const uint32_t *idx_s, const uint32_t *idx_e, const uint32_t *seed_cand,
                        const uint32_t *seq_off, uint32_t *ap_g, uint32_t *ap_s, uint32_t *ap_e, int32_t *ap_delta,
                        int32_t *ap_shift_incl, uint32_t *n_ap, const uint32_t *M_in, uint32_t *M_out, uint32_t out_cap, uint32_t *err) {
    const uint32_t nb = region_lb_blocks(n_reg);
    NP2_LAUNCH(k_splice_plan, dim3(nb), 256, s, lb, nb, reg_lable, lable, n_reg, stuck, idx_s, idx_e, seed_cand, seq_off, ap_g, ap_s, ap_e, ap_delta, ap_shift_incl, n_ap, M_in, M_out, out_cap, err);
}
void launch_splice_write(hipStream_t s, const uint32_t *in_pos, const uint8_t *in_base, const uint32_t *M_p, uint32_t M_cap,
                         const uint32_t *ap_g, const uint32_t *ap_s, const uint32_t *ap_e, const int32_t *ap_delta,
                         const int32_t *ap_shift_incl, const uint32_t *n_ap, uint32_t max_ap, const uint32_t *lq_start,
                         const uint32_t *seed_cand, const uint32_t *seq_off, const uint8_t *seq, uint32_t *out_pos,
                         uint8_t *out_base) {
    NP2_LAUNCH(k_splice_bases, dim3((M_cap + SPLICE_SPAN - 1) / SPLICE_SPAN), 256, s, in_pos, in_base, M_p, ap_s, ap_e, ap_shift_incl, n_ap, out_pos, out_base);
    if (max_ap)
        NP2_LAUNCH(k_splice_seeds, g1((uint64_t)max_ap * 8, 256), 256, s, ap_g, ap_s, ap_delta, ap_shift_incl, n_ap, lq_start, seed_cand, seq_off, seq, out_pos, out_base);
}
void launch_shard_bounds(hipStream_t s, const uint32_t *cns_pos, const uint32_t *M_p, const uint32_t *t, uint32_t *out) {
    NP2_LAUNCH(k_shard_bounds, dim3(1), 64, s, cns_pos, M_p, t[0], t[1], t[2], t[3], t[4], t[5], out);
}
void launch_rech_list(hipStream_t s, const Lookback &lb, const uint8_t *reg_lable, uint32_t n_reg, uint32_t *rech,
                      uint32_t *n_rech, unsigned long long *blob_bound, uint32_t *err) {
    const uint32_t nb = region_lb_blocks(n_reg);
    NP2_LAUNCH(k_rech_list, dim3(nb), 256, s, lb, nb, reg_lable, n_reg, rech, n_rech, blob_bound, err);
}
void launch_rech_groups(hipStream_t s, const Lookback &lb, const uint32_t *rech, const uint32_t *n_rech_p, uint32_t max_rech,
                        const uint32_t *cns_pos, const uint32_t *M_p, const uint32_t *lq_start, const uint32_t *lq_end,
                        const uint32_t *keep_n, uint32_t ksize, const uint32_t *reg_maxlen, void *tmp_groups,
                        uint32_t *head_jobs, void *groups, uint32_t *job_off, uint32_t *n_groups, uint32_t *n_jobs,
                        unsigned long long *blob_bound, uint32_t *err) {
    NP2_LAUNCH(k_rech_groups, dim3((max_rech + 255) / 256), 256, s, rech, n_rech_p, cns_pos, M_p, lq_start, lq_end, keep_n, ksize, reg_maxlen, (RechGroup *)tmp_groups, head_jobs, blob_bound, err);
    const uint32_t nb = region_lb_blocks(max_rech);
    NP2_LAUNCH(k_rech_compact, dim3(nb), 256, s, lb, nb, n_rech_p, head_jobs, (const RechGroup *)tmp_groups, (RechGroup *)groups, job_off, n_groups, n_jobs, err);
}
size_t rech_group_bytes() { return sizeof(RechGroup); }
static RechCtx mk_rech(const RechPtrs &p) {
    return RechCtx{(const RechGroup *)p.groups, p.job_off, p.rech, p.cand_off, p.keep_list, p.seq_off, p.seq,
                   p.cns_base, p.n_groups};
}
void launch_rech_job_len(hipStream_t s, const RechPtrs &p, uint32_t n_jobs, uint32_t *len) {
    if (n_jobs) NP2_LAUNCH(k_rech_job_len, g1(n_jobs, 64), 64, s, mk_rech(p), n_jobs, len);
}
void launch_rech_job_build(hipStream_t s, const RechPtrs &p, uint32_t n_jobs, const uint32_t *soff32, uint64_t *soff64,
                           uint8_t *blob) {
    if (!n_jobs) return;
    NP2_LAUNCH(k_u32_to_u64_off, g1(n_jobs + 1), 256, s, soff32, n_jobs + 1, soff64);
    NP2_LAUNCH(k_rech_job_build, g1(n_jobs, 64), 64, s, mk_rech(p), n_jobs, soff64, blob);
}
void launch_rech_apply(hipStream_t s, const RechPtrs &p, const uint16_t *score, uint16_t *keep_ks) {
    if (p.n_groups) NP2_LAUNCH(k_rech_apply, g1(p.n_groups, 64), 64, s, mk_rech(p), score, keep_ks);
}
void launch_rech_select(hipStream_t s, const uint32_t *rech, const uint32_t *n_rech_p, uint32_t max_rech,
                        const uint32_t *cand_off, const uint32_t *keep_n, const uint32_t *keep_list,
                        const uint16_t *keep_ks, const uint32_t *order, bool first_yak, uint8_t *reg_lable,
                        uint32_t *seed_cand) {
    if (max_rech)
        NP2_LAUNCH(k_rech_select, g1(max_rech, 64), 64, s, rech, n_rech_p, cand_off, keep_n, keep_list, keep_ks, order, first_yak ? 1u : 0u, reg_lable, seed_cand);
}
void launch_rech_relabel(hipStream_t s, uint8_t *reg_lable, uint32_t n_reg) {
    NP2_LAUNCH(k_rech_relabel, g1(n_reg), 256, s, reg_lable, n_reg);
}

} // namespace np2
