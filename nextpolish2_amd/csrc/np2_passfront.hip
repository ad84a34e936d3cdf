// The front of a pass on chip (gfx950, wave64): sorted exception records of a contig tile -> the tile's piece of the raw
// consensus, in one kernel.
//
// One workgroup owns a contig tile (TILE positions).  It stages the tile's records in LDS, groups them into the exception
// nodes of the live reads (Msa::push, main.rs:193-207), orders the nodes of every position (Msa::sort, main.rs:227-229),
// computes the coverage (Msa::coverage, main.rs:232-241) from the tile's read list — and then KEEPS GOING where
// k_tile_write (np2_graph.hip) wrote 13 bytes per position to memory for five more kernels to read: the best-path DP of
// the tile's dirty runs (main.rs:1645-1687, tie rule :1670, dead-end rule :1666-1668), their backtrack and the quality
// class of every consensus base (main.rs:1555-1585: qv = count * 100 / coverage < 95) all run on the nodes in LDS.  What
// leaves the chip is the tile's consensus as 16-bit entries (position inside the tile, base code, class) in a slot of its
// own, the number of entries, the number of low-quality ones and the tile's share of the path score.  One scan over the
// per-tile counts (k_tile_offsets) and one streaming kernel (k_pf_compact) then lay the consensus out contiguously and
// list the low-quality bases for the LQ-region kernels.
//
// A dirty run belongs to the tile it STARTS in.  The DP decomposes exactly at clean positions (every path passes through
// a position that holds a single node), so a tile needs nothing from its left neighbour but one bit (is the
// position before the tile dirty?  then the tile's leading dirty positions are the neighbour's run) and from its right
// neighbour the records and read starts of a HALO of positions, as far as its last run reaches.  A tile that does not fit
// (more records than the LDS variant holds, a run still open at the end of the halo) is listed and redone by the big
// variant (k_pf_tile_big: 3584 records, a whole tile of halo); what that one cannot hold either (or a pass with a position
// covered PF_COV_MAX times or more: scores are 32-bit here) sets PF_REDO and the host runs the pass through the unfused
// kernels (np2_graph.hip / np2_kernels.hip), which remain the general path.
//
// Backtrack without a path buffer: inside one position the nodes of a path come in ascending node order (a node whose
// second column lies at the same position has its predecessor among the EARLIER nodes of that position, and Msa::sort puts
// the node whose second column lies one position back first), so the walk only MARKS the nodes it visits; the write-out is
// position-parallel — every position emits its marked nodes in node order at the offset a block scan gives it.
#include "np2_common.hpp"
#include "np2_kernels.hpp"
#include "np2_blockscan.hpp"

namespace np2 {

static constexpr uint32_t PF_COUNT_BITS = 14;                 // node word: count | best predecessor << 14 | visited << 31
static constexpr uint32_t PF_IDX_MASK = (1u << PF_COUNT_BITS) - 1;
static constexpr uint32_t PF_VISITED = 0x80000000u;
static constexpr uint32_t PF_N0_VISITED = 0x8000u;            // per-position word: best predecessor of N0 | visited << 15
static constexpr uint32_t PF_N0_WEAK = 0x4000u;               // ... before the DP: a dirty position whose exception nodes all lose (| e0)
static constexpr uint32_t PF_N0_STRONG = 0x2000u;             // ... a dirty position that needs the DP
static constexpr int32_t PF_NEG = -(1 << 30);                 // "unreachable" in the 32-bit relative scores

__device__ __forceinline__ uint8_t pf_ref_code(const uint8_t *__restrict__ refnib, uint32_t p) {
    return (refnib[p >> 1] >> (4 * (p & 1))) & 7;
}
// lane I of the caller's quad (lanes 4 m .. 4 m + 3), one DPP move; every lane of the quad must be active
template <int I> __device__ __forceinline__ uint32_t pf_quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, I | (I << 2) | (I << 4) | (I << 6), 0xF, 0xF, true);
}
__device__ __forceinline__ void pf_n0_key(uint32_t p, uint32_t c2, uint32_t c1, uint32_t c0, uint32_t &b, uint32_t &d) {
    if (p >= 2) {
        b = (c2 << 8) | (c1 << 4) | c0, d = 0;
    } else if (p == 1) { // (head(-1,1), c0, c1)
        b = 0x0F00u | (c1 << 4) | c0, d = 1;
    } else { // (head(-1,0), head(-1,1), c0)
        b = 0x4FF0u | c0, d = 0;
    }
}
// slot of tile t: entries [pf_slot_off(t), ...) of the slot array; capacity TILE + 1 + n_t + 2 n_{t+1} (a tile emits at
// most one base per position up to the clean position that closes its last run, plus one per exception node there)
__device__ __forceinline__ uint64_t pf_slot_off(const uint32_t *__restrict__ tile_scan, uint32_t t) {
    return (uint64_t)t * (TILE + 1) + tile_scan[t] + 2ull * (tile_scan[t + 1] - tile_scan[1]);
}

// LEVEL 0: every tile; 1: the tiles level 0 listed (more records); 2: the tiles level 1 listed (more records still, and a
// whole tile of halo); what level 2 cannot hold sets PF_REDO
template <uint32_t CAP, uint32_t HALO, int LEVEL>
__device__ __forceinline__ void pf_tile_body(const uint32_t t, const PfTile &A) {
    constexpr uint32_t E = TILE + HALO;                 // positions a block sees
    constexpr uint32_t RPT = (CAP + 255) / 256;         // records per thread
    constexpr uint32_t HP = HALO >= 256 ? HALO / 256 : 1; // halo positions per thread (HALO < 256: the first HALO threads)
    static_assert(HALO == 64 || HALO % 256 == 0, "halo positions are owned by wave 0 or by every thread alike");
    static_assert(E + CAP < (1u << PF_COUNT_BITS), "best-predecessor indices are 14-bit");
    __shared__ __attribute__((aligned(8))) uint32_t s_cnt[E + 2];  // nodes per position; then s_off / s_n0bi
    __shared__ __attribute__((aligned(8))) int32_t s_dcov[E + 2];  // coverage differences; then s_cov / s_run / s_ref
    __shared__ __attribute__((aligned(8))) uint32_t s_raw[3 * CAP];
    __shared__ uint32_t sh[8];
    __shared__ uint32_t s_wt[4 * RPT];
    __shared__ long long s_gain[4];
    __shared__ uint32_t s_flag[6]; // [0] previous position dirty, [1] tile does not fit, [2] path begin (tile 0), [3] LQ entries, [4] owns the run reaching the contig end
    __shared__ long long s_endrel;
    uint64_t *const s_k = reinterpret_cast<uint64_t *>(s_raw); // record keys             } until the nodes are formed
    uint32_t *const s_v = s_raw + 2 * CAP;                     // record read | live << 31 }
    uint32_t *const s_ncw = s_raw;                             // node: count | besti << 14 | visited << 31
    uint32_t *const s_nmin = s_raw + CAP;                      // node: first live read; then its score
    int32_t *const s_score = reinterpret_cast<int32_t *>(s_raw + CAP);
    uint32_t *const s_nkey = s_raw + 2 * CAP;                  // node: bases | delta1 << 16
    uint16_t *const s_off = reinterpret_cast<uint16_t *>(s_cnt);            // [E + 1] tile-local node offsets
    uint16_t *const s_n0bi = reinterpret_cast<uint16_t *>(s_cnt) + (E + 2); // [E] best predecessor of N0 | visited
    uint16_t *const s_cov = reinterpret_cast<uint16_t *>(s_dcov);           // [E]
    uint16_t *const s_run = reinterpret_cast<uint16_t *>(s_dcov) + E;       // [TILE / 2 + 1] run starts
    uint8_t *const s_ref = reinterpret_cast<uint8_t *>(s_dcov) + 2 * E + 2 * (TILE / 2 + 2); // [E + 4] contig codes of start - 3 ..
    static_assert(2 * E + 2 * (TILE / 2 + 2) + E + 4 <= 4 * (E + 2), "s_cov, s_run and s_ref share the coverage array");

    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t L = A.L, start = t << TILE_SHIFT;
    // (phase timers of a tile, thread 0, for tools/pf_prof.py: NP2_PF_PROF)
    auto stamp = [&](uint32_t i) {
        if (A.prof && tid == 0) A.prof[(size_t)t * 8 + i] = (unsigned long long)clock64();
    };
    stamp(0);
    const uint32_t npos = min((uint32_t)TILE, L - start);          // positions this tile emits clean bases for
    const bool has_next = t + 1 < A.n_tiles;
    const uint32_t n = A.tile_n[t];
    constexpr bool BIG = LEVEL == 2;
    const uint32_t cap = min(CAP, LEVEL == 2 ? A.cap_lim_big : (LEVEL == 1 ? max(A.cap_lim, min(A.cap_lim_big, PF_CAP_MID)) : A.cap_lim));
    // a tile this level cannot hold goes on the next level's list (uniform; thread 0)
    auto hand_on = [&]() {
        A.tile_cnt[t] = 0, A.tile_lq[t] = 0, A.tile_gain[t] = 0; // (an empty slot, should nobody redo the tile)
        if (LEVEL == 2)
            atomicOr(A.flags, PF_REDO);
        else if (LEVEL == 1 && !A.big_enabled)
            atomicOr(A.flags, PF_REDO | PF_NEED_BIG); // (the big variant is not launched until a contig has needed it once)
        else if (LEVEL == 1)
            A.bad_list2[atomicAdd(A.n_bad2, 1u)] = t;
        else
            A.bad_list[atomicAdd(A.n_bad, 1u)] = t;
    };
    // the halo: PF_HALO positions (a test may lower it; a multiple of 16) — the big variant takes as much of the next tile as
    // its record capacity allows (k_tile_sort's index: records before every 16th position)
    uint32_t halo = BIG ? HALO : min(HALO, A.halo_lim), nh = 0; // nh: records of the next tile inside the halo
    if (has_next) {
        const uint32_t n1 = A.tile_n[t + 1];
        const uint16_t *__restrict__ px = A.pidx + (size_t)(t + 1) * (TILE / 16);
        auto upto = [&](uint32_t j) -> uint32_t { return j >= TILE / 16 ? n1 : min(n1, (uint32_t)px[j]); }; // records before position 16 j
        if (n1) {
            if (BIG && n <= cap && n + n1 > cap) {
                uint32_t lo = 0, hi = TILE / 16; // largest j with n + upto(j) <= cap
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1) >> 1;
                    if (n + upto(mid) <= cap) lo = mid; else hi = mid - 1;
                }
                halo = lo * 16;
            }
            nh = upto(halo / 16);
        }
    }
    const uint32_t ext = min(TILE + halo, L - start);              // positions whose nodes and coverage are known here
    const uint32_t nt = n + nh;
    const uint64_t ra = (uint64_t)t * A.bucket_cap, rb = ra + A.bucket_cap; // (bucketed layout)
    if (tid < 6) s_flag[tid] = 0;
    if (tid == 0) s_endrel = 0;
    for (uint32_t i = tid; i < E + 2; i += 256) s_cnt[i] = 0, s_dcov[i] = 0;
    const bool fits = nt <= cap;
    __syncthreads();
    if (!fits) { // (uniform)
        if (tid == 0) hand_on();
        return;
    }
    // ---- level 1 of the dependent loads: records, read lists, the contig codes ----------------------------------------
    uint64_t rk[RPT];
    uint32_t rv[RPT];
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        const uint32_t i = tid + 256 * j;
        rk[j] = 0, rv[j] = 0;
        if (i < nt) {
            const uint64_t x = i < n ? ra + i : rb + (i - n);
            rk[j] = A.keys[x], rv[j] = A.vals[x];
        }
    }
    const uint32_t ro0 = A.tile_rd_off[t], ro1 = A.tile_rd_off[t + 1], ro2 = has_next ? A.tile_rd_off[t + 2] : ro1;
    // (both read lists as one index range: [ro0, ro1) overlap this tile, [ro1, ro2) the next one)
    const uint32_t ci = ro0 + tid;
    uint32_t cr = 0;
    if (ci < ro2) cr = A.tile_rd[ci];
    // contig codes: the thread's four positions (two bytes: start + 4 tid is even), its halo positions, start - 3 .. start - 1
    const uint32_t q0 = tid * 4;
    uint32_t ref4 = 0;
    if (q0 < ext) ref4 = *reinterpret_cast<const uint16_t *>(A.refnib + ((start + q0) >> 1));
    uint32_t refh[HP];
#pragma unroll
    for (uint32_t j = 0; j < HP; ++j) {
        const uint32_t q = TILE + tid * HP + j;
        refh[j] = (tid * HP + j < HALO && q < ext) ? pf_ref_code(A.refnib, start + q) : 0u;
    }
    uint32_t refm = 0;
    if (tid < 3 && start + tid >= 3) refm = pf_ref_code(A.refnib, start + tid - 3);
    // ---- level 2: liveness of the records' reads, spans of the listed reads ------------------------------------------
    uint8_t cal = 0;
    uint32_t cts = 0, cte = 0;
    if (ci < ro2) cal = A.alive[cr], cts = A.reads[cr].aln_t_s, cte = A.reads[cr].aln_t_e;
#pragma unroll
    for (uint32_t j = 0; j < RPT; ++j) {
        const uint32_t i = tid + 256 * j;
        if (i < nt) {
            s_k[i] = rk[j];
            s_v[i] = rv[j] | (A.alive[rv[j]] ? 0x80000000u : 0u);
        }
    }
    if (tid == 0 && start) { // is the position left of the tile dirty?  (then the tile's leading dirty positions are not its own)
        const uint32_t pn = A.tile_n[t - 1];
        const uint64_t pa = (uint64_t)(t - 1) * A.bucket_cap;
        uint32_t d = 0;
        for (uint64_t i = pa + pn; i-- > pa;) {
            if ((uint32_t)(A.keys[i] >> 32) != start - 1) break;
            if (A.alive[A.vals[i]]) {
                d = 1;
                break;
            }
        }
        s_flag[0] = d;
    }
    // coverage as a difference array over [start, start + ext): a read of this tile's list from where it enters the tile,
    // a read of the next tile's list only if it starts there (it is in this tile's list otherwise)
    auto add_read = [&](uint32_t idx, uint8_t al, uint32_t ts, uint32_t te) {
        if (!al) return;
        if (idx >= ro1 && ts < start + TILE) return;
        if (ts >= start + ext) return;
        atomicAdd(&s_dcov[max(ts, start) - start], 1);
        atomicAdd(&s_dcov[min(te, start + ext - 1) - start + 1], -1);
    };
    if (ci < ro2) add_read(ci, cal, cts, cte);
    for (uint32_t i = ci + 256; i < ro2; i += 256) { // (tiles under more than ~128 reads)
        const uint32_t r = A.tile_rd[i];
        add_read(i, A.alive[r], A.reads[r].aln_t_s, A.reads[r].aln_t_e);
    }
    __syncthreads();
    stamp(1);
    // ---- coverage of the thread's positions (four of the tile + its share of the halo) -----------------------------------
    int32_t cv[4], cvh[HP];
    {
        const int32_t d0 = s_dcov[q0], d1 = s_dcov[q0 + 1], d2 = s_dcov[q0 + 2], d3 = s_dcov[q0 + 3];
        int32_t dh[HP], hs = 0;
#pragma unroll
        for (uint32_t j = 0; j < HP; ++j) {
            dh[j] = tid * HP + j < HALO ? s_dcov[TILE + tid * HP + j] : 0;
            hs += dh[j];
        }
        uint32_t tot;
        const int32_t pre = (int32_t)block_excl_scan<OpAdd, 4>((uint32_t)(d0 + d1 + d2 + d3), sh, tot);
        cv[0] = pre + d0, cv[1] = cv[0] + d1, cv[2] = cv[1] + d2, cv[3] = cv[2] + d3;
        int32_t hpre;
        if constexpr (HALO == 64) { // (wave 0 holds the halo: no barrier)
            hpre = (int32_t)(tot + wave_incl_scan<OpAdd>((uint32_t)hs)) - hs;
        } else {
            uint32_t tot2;
            hpre = (int32_t)(tot + block_excl_scan<OpAdd, 4>((uint32_t)hs, sh, tot2));
        }
#pragma unroll
        for (uint32_t j = 0; j < HP; ++j) {
            hpre += dh[j];
            cvh[j] = hpre;
        }
    }
    // ---- nodes in key order: group heads count their live members (k_tile_write's scheme) --------------------------------
    uint32_t nn = 0;
    {
        bool isn[RPT];
        uint32_t gcv[RPT], gmv[RPT], lr[RPT];
        uint64_t kown[RPT];
#pragma unroll
        for (uint32_t j = 0; j < RPT; ++j) {
            const uint32_t i = tid + 256 * j;
            isn[j] = false, gcv[j] = 0, gmv[j] = 0xFFFFFFFFu, kown[j] = 0;
            // A group = the records of one key, neighbours in the sorted list, ascending in read number inside the group
            // (k_tile_sort ranks by (key, read)).  The wavefront's 64 records of this round are consecutive, so a group head
            // counts its live members from two ballots — heads and live records — and takes its first live read from the
            // lane that holds it; only a group that runs past the wavefront's last record is walked, from there on.
            const bool in = i < nt;
            const uint64_t k = in ? s_k[i] : 0ull;
            const uint32_t v = in ? s_v[i] : 0u;
            kown[j] = k;
            const bool head = in && (i == 0 || s_k[i - 1] != k);
            const uint64_t hm = __ballot(head), lm = __ballot(in && (v >> 31));
            const uint64_t above = lane == 63 ? 0ull : hm & ~((2ull << lane) - 1ull); // heads after this lane
            const uint32_t end = above ? (uint32_t)__builtin_ctzll(above) : 64u;       // first lane past the group (64: none in sight)
            const uint64_t run = (end == 64u ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
            const uint64_t lrun = lm & run;
            // (every lane takes part in the cross-lane read; the source lane is only meaningful for heads with a live member)
            const uint32_t firstv = (uint32_t)__shfl((int)v, lrun ? (int)__builtin_ctzll(lrun) : (int)lane);
            if (head) {
                uint32_t gc = (uint32_t)__builtin_popcountll(lrun), gm = lrun ? (firstv & 0x7FFFFFFFu) : 0xFFFFFFFFu;
                if (end == 64u)
                    for (uint32_t jj = i + (64u - lane); jj < nt && s_k[jj] == k; ++jj) { // (the group goes on in the next wavefront's records)
                        const uint32_t vv = s_v[jj];
                        if (vv >> 31) {
                            ++gc;
                            gm = min(gm, vv & 0x7FFFFFFFu);
                        }
                    }
                isn[j] = gc != 0, gcv[j] = gc, gmv[j] = gm;
            }
            const uint64_t bal = __ballot(isn[j]);
            lr[j] = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (lane == 0) s_wt[wv * RPT + j] = (uint32_t)__builtin_popcountll(bal);
        }
        __syncthreads(); // (the staged records are dead from here on: the node arrays take their place)
        uint32_t base = 0;
#pragma unroll
        for (uint32_t j = 0; j < RPT; ++j) {
            uint32_t tot_j = 0, before = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) {
                const uint32_t x = s_wt[w * RPT + j];
                tot_j += x;
                if (w < wv) before += x;
            }
            if (isn[j]) {
                const uint64_t k = kown[j];
                const uint32_t li = base + before + lr[j];
                s_nkey[li] = ((uint32_t)k >> 16) | ((uint32_t)k << 16); // bases | delta1 << 16
                s_ncw[li] = gcv[j];
                s_nmin[li] = gmv[j];
                atomicAdd(&s_cnt[(uint32_t)(k >> 32) - start], 1u);
            }
            base += tot_j;
        }
        nn = base;
    }
    __syncthreads();
    (void)nn;
    stamp(2);
    // ---- node offsets of the positions, coverage and contig codes into their final LDS places ----------------------------
    const uint32_t c0 = s_cnt[q0], c1 = s_cnt[q0 + 1], c2 = s_cnt[q0 + 2], c3 = s_cnt[q0 + 3];
    uint32_t ch[HP], chs = 0;
#pragma unroll
    for (uint32_t j = 0; j < HP; ++j) {
        ch[j] = tid * HP + j < HALO ? s_cnt[TILE + tid * HP + j] : 0u;
        chs += ch[j];
    }
    const uint32_t pdq = q0 ? s_cnt[q0 - 1] : s_flag[0]; // is the position before the thread's first one dirty?
    uint32_t tot;
    const uint32_t l0 = block_excl_scan<OpAdd, 4>(c0 + c1 + c2 + c3, sh, tot);
    uint32_t lh;
    if constexpr (HALO == 64) {
        lh = tot + wave_incl_scan<OpAdd>(chs) - chs;
    } else {
        uint32_t tot2;
        lh = tot + block_excl_scan<OpAdd, 4>(chs, sh, tot2);
    }
    const uint32_t off[5] = {l0, l0 + c0, l0 + c0 + c1, l0 + c0 + c1 + c2, l0 + c0 + c1 + c2 + c3};
    const uint32_t cj[4] = {c0, c1, c2, c3};
    int32_t cmax = max(max(cv[0], cv[1]), max(cv[2], cv[3]));
#pragma unroll
    for (uint32_t j = 0; j < HP; ++j) cmax = max(cmax, cvh[j]);
    if (cmax >= (int32_t)A.cov_max) s_flag[1] = 2; // (scores are 32-bit, counts 14-bit here)
    __syncthreads(); // every thread has read its s_cnt / s_dcov values: the arrays change their meaning
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        s_off[q0 + j] = (uint16_t)off[j];
        s_n0bi[q0 + j] = 0;
        s_cov[q0 + j] = (uint16_t)cv[j];
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) s_ref[3 + q0 + j] = (uint8_t)((ref4 >> (4 * j)) & 7);
    {
        uint32_t o = lh;
#pragma unroll
        for (uint32_t j = 0; j < HP; ++j) {
            if (tid * HP + j < HALO) {
                const uint32_t q = TILE + tid * HP + j;
                s_off[q] = (uint16_t)o;
                s_n0bi[q] = 0;
                s_cov[q] = (uint16_t)cvh[j];
                s_ref[3 + q] = (uint8_t)refh[j];
                o += ch[j];
                if (q == E - 1) s_off[E] = (uint16_t)o;
            }
        }
    }
    if (tid < 3) s_ref[tid] = (uint8_t)refm;
    // ---- order the nodes of each position: (delta3, first read) — Msa::sort over first-seen order --------------------------
    auto sort_pos = [&](uint32_t o0, uint32_t o1) {
        for (uint32_t i = o0 + 1; i < o1; ++i) {
            const uint32_t kk = s_nkey[i], c = s_ncw[i], m = s_nmin[i];
            const uint32_t kd = node_delta3((uint16_t)kk, (uint16_t)(kk >> 16));
            uint32_t x = i;
            while (x > o0) {
                const uint32_t pk = s_nkey[x - 1];
                const uint32_t pd = node_delta3((uint16_t)pk, (uint16_t)(pk >> 16));
                if (pd < kd || (pd == kd && s_nmin[x - 1] < m)) break;
                s_nkey[x] = pk, s_ncw[x] = s_ncw[x - 1], s_nmin[x] = s_nmin[x - 1];
                --x;
            }
            s_nkey[x] = kk, s_ncw[x] = c, s_nmin[x] = m;
        }
    };
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        if (cj[j] > 1) sort_pos(off[j], off[j + 1]);
    {
        uint32_t o = lh;
#pragma unroll
        for (uint32_t j = 0; j < HP; ++j) {
            if (ch[j] > 1) sort_pos(o, o + ch[j]);
            o += ch[j];
        }
    }
    // ---- which dirty positions need the DP at all?  If every exception node whose second column lies one position back has
    // a smaller count than the contig's node, and every node of an insertion column (second column at the same position) a
    // negative weight (10 * count < 4 * coverage), then a path that leaves the contig's nodes anywhere in the run scores
    // strictly less than the one that stays on them — no tie rule gets a say — and the run's consensus is the contig's.
    // That is what a sequencing error in one read out of thirty looks like: ~9 runs of 10.  The word carries e0 (the
    // exception columns that are not insertion columns: count of N0 = coverage - e0) for the gain and the quality class.
    auto weak_word = [&](uint32_t o0, uint32_t cnt, int32_t cov) -> uint16_t {
        uint32_t e0 = 0, maxrep = 0;
        bool ok = true;
        for (uint32_t k = o0; k < o0 + cnt; ++k) {
            const uint32_t key = s_nkey[k], c = s_ncw[k];
            if (key & 0x1000u)
                ok = ok && 10 * c < 4 * (uint32_t)cov;
            else
                e0 += c, maxrep = max(maxrep, c);
        }
        return (uint16_t)((ok && maxrep + e0 < (uint32_t)cov) ? (PF_N0_WEAK | e0) : PF_N0_STRONG);
    };
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        if (cj[j]) s_n0bi[q0 + j] = weak_word(off[j], cj[j], cv[j]);
    {
        uint32_t o = lh;
#pragma unroll
        for (uint32_t j = 0; j < HP; ++j) {
            if (ch[j]) s_n0bi[TILE + tid * HP + j] = weak_word(o, ch[j], cvh[j]);
            o += ch[j];
        }
    }
    // ---- clean positions' share of the path score, dirty-run starts ---------------------------------------------------------
    long long gain = 0;
    {
        bool pdirty = pdq != 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const bool d = cj[j] != 0;
            if (q0 + j < npos && !d && !pdirty) gain += 6LL * cv[j]; // clean after clean: 10 * c0 - 4 * cov with c0 = cov
            pdirty = d;
        }
    }
    const bool s0 = c0 && !pdq, s1 = c1 && !c0, s2 = c2 && !c1, s3 = c3 && !c2; // (positions past npos hold no nodes)
    uint32_t n_runs;
    uint32_t rr = block_excl_scan<OpAdd, 4>((uint32_t)s0 + (uint32_t)s1 + (uint32_t)s2 + (uint32_t)s3, sh, n_runs);
    // (the scan's barriers also order the stores above — offsets, coverage, codes, sorted nodes — before the DP's loads)
    if (s0) s_run[rr++] = (uint16_t)q0;
    if (s1) s_run[rr++] = (uint16_t)(q0 + 1);
    if (s2) s_run[rr++] = (uint16_t)(q0 + 2);
    if (s3) s_run[rr++] = (uint16_t)(q0 + 3);
    __syncthreads();
    if (s_flag[1]) { // (uniform) a position covered too deeply for this kernel's number formats
        if (tid == 0) {
            A.tile_cnt[t] = 0, A.tile_lq[t] = 0, A.tile_gain[t] = 0;
            atomicOr(A.flags, PF_REDO);
        }
        return;
    }
    stamp(3);
    // ---- DP + backtrack of the dirty runs that start in this tile ------------------------------------------------------------
    // Round 5 gave every run a lane of ONE wavefront: a tile's time in this phase was its longest run (8-9 positions in a
    // diploid phasing pass) times ~350 wave instructions a position — all (node, predecessor) pairs of a position one after
    // the other in one lane, 27 k of a tile's 82 k clocks with three wavefronts waiting at the barrier.  Now:
    //  (1) one thread per run CLASSIFIES it: every position weak (above) -> the path stays on the contig's nodes, done;
    //      otherwise the run is listed — for a QUAD of lanes if no position holds more than three exception nodes
    //      (99.3 % of the listed runs of a 30x diploid pass), for a single lane otherwise;
    //  (2) a quad scores a position in one step: lane 0 owns N0(p), lane j exception node j - 1; each tests ITS node against
    //      the previous position's four (keys and scores fetched from the quad's lanes by DPP, in the reference's order,
    //      main.rs:1664-1674), then the nodes whose second column lies at the same position take the earlier nodes of their
    //      own position in three more rounds (node k's predecessors are final by round k); 64 quads a round over all four
    //      wavefronts.  Lane 0 closes the run and walks it back.
    // The two lists take the place of the run starts themselves (quads from the front of s_run, single lanes from its back):
    // every thread has its — at most two — run starts in registers before the first list entry is written.  (A list array of
    // its own was the kilobyte of LDS between eight tiles per CU and seven.)
    __shared__ uint32_t s_nl[2];
    if (tid < 2) s_nl[tid] = 0;
    static_assert(TILE / 2 <= 2 * 256, "a thread keeps at most two run starts");
    uint32_t my_qa[2];
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) my_qa[k] = tid + 256u * k < n_runs ? (uint32_t)s_run[tid + 256u * k] : 0xFFFFFFFFu;
    uint16_t *const s_ql = s_run; // [TILE / 2 + 1]
    __syncthreads();
    constexpr uint32_t PF_QN = 3; // exception nodes a quad holds
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
        const uint32_t qa = my_qa[k];
        if (qa == 0xFFFFFFFFu) continue;
        const uint32_t a = start + qa;
        uint32_t q = qa, mxn = 0;
        int32_t g = 0;
        bool weak = a >= 3, closed = false;
        for (; q < ext; ++q) {
            const uint32_t x = s_n0bi[q];
            if (x == 0) { // the clean position that closes the run
                closed = true;
                break;
            }
            mxn = max(mxn, (uint32_t)s_off[q + 1] - (uint32_t)s_off[q]);
            if (!(x & PF_N0_WEAK)) weak = false;
            const int32_t cov = s_cov[q];
            g += 10 * (cov - (int32_t)(x & 0x1FFFu)) - 4 * cov;
        }
        if (weak && closed) {
            for (uint32_t qq = qa; qq < q; ++qq) s_n0bi[qq] = (uint16_t)(s_n0bi[qq] | PF_N0_VISITED);
            gain += g + 6 * (int32_t)s_cov[q];
        } else if (!closed && start + q != L) { // still open at the end of the halo
            s_flag[1] = 1;
        } else if (mxn <= PF_QN) {
            s_ql[atomicAdd(&s_nl[0], 1u)] = (uint16_t)qa;
        } else {
            s_ql[TILE / 2 - atomicAdd(&s_nl[1], 1u)] = (uint16_t)qa;
        }
    }
    __syncthreads();
    // one (node, predecessor) test: main.rs:1664-1674
    auto pred_test = [](uint32_t vkey, int32_t ps, uint32_t want, uint32_t kd, bool far, int32_t w, uint32_t pi, int32_t &score,
                        uint32_t &besti) {
        const uint32_t vb = vkey & 0xFFFFu, vd = vkey >> 16;
        const uint32_t v2d = (vb & 0x4000u) ? ((vd + 1u) & 0xFFFFu) : 0u;
        const uint32_t v1q = (vb >> 8) & 0xFu;
        const bool ok = (vb & 0x10FFu) == want && v2d == kd && !(far && v1q == 15u); // main.rs:1666-1668
        const int32_t sc = ps + w;
        if (ok && (sc > score || (sc == score && v1q != 4u))) score = sc, besti = pi; // main.rs:1670
    };
    // the end of a run: its gain, where the walk back begins, the walk (marks the nodes of the path)
    auto finish_run = [&](uint32_t qa, uint32_t q, bool closed, int32_t s0_cur, int32_t base, int32_t pv_s0, uint32_t pv_k0, uint32_t pv_n) {
        const uint32_t a = start + qa;
        uint32_t wq, widx; // where the walk back begins
        if (closed) {
            gain += (long long)s0_cur - base;
            wq = q - 1, widx = s_n0bi[q];
            s_n0bi[q] = 0; // (the closing position is written out as a clean position: no mark, no index)
        } else if (start + q == L) {
            // the run reaches the contig end: the best end node (main.rs:1651,1680: the last one of maximal score; that the
            // score is >= 0 is checked by the host against the total of all gains: end_rel + total).
            // A read-start node's score is the absolute 10 count - 4 coverage (main.rs:1659-1660), the others' are relative
            // to the node left of the run (a >= 3): one with a negative score can never be chosen (the choice starts at the
            // default node's 0), one that reaches 0 — a read starting at the very last position with more copies than the
            // pileup is deep there — competes with the path's total, which only k_dp_finish knows: the pass is handed back.
            int32_t best = pv_s0;
            uint32_t bi = 0;
            for (uint32_t k = 0; k < pv_n; ++k) {
                const int32_t sc = s_score[pv_k0 + k];
                if (a >= 3 && ((s_nkey[pv_k0 + k] >> 4) & 0xFu) == 15u) {
                    if (sc >= 0) atomicOr(A.flags, PF_REDO);
                    continue;
                }
                if (sc >= best) best = sc, bi = k + 1;
            }
            gain -= base;
            s_endrel = best <= PF_NEG / 2 ? (long long)SCORE_NEG : (long long)best;
            s_flag[4] = 1;
            wq = q - 1, widx = bi;
        } else { // still open at the end of the halo
            s_flag[1] = 1;
            return;
        }
        for (;;) {
            const uint32_t p = start + wq;
            uint32_t bi;
            bool back;
            if (widx == 0) {
                const uint32_t x = s_n0bi[wq];
                s_n0bi[wq] = (uint16_t)(x | PF_N0_VISITED);
                bi = x & PF_IDX_MASK;
                if (p == 0) break; // N0(0) = (head, head, c0): the path starts here
                back = true;
            } else {
                const uint32_t k = s_off[wq] + widx - 1;
                const uint32_t x = s_ncw[k], kb = s_nkey[k] & 0xFFFFu;
                s_ncw[k] = x | PF_VISITED;
                bi = (x >> PF_COUNT_BITS) & PF_IDX_MASK;
                if (((kb >> 4) & 0xF) == 15) { // a read's start node: the path begins at p (only reachable for p <= 2)
                    if (p > 0) atomicMax(&s_flag[2], p);
                    break;
                }
                back = !(kb & 0x1000);
            }
            if (back) {
                if (wq == qa) break; // left the run: N0(a - 1)
                --wq;
            }
            widx = bi;
        }
    };
    const uint32_t n_quad = s_nl[0], n_one = s_nl[1];
    // (2) quads
    for (uint32_t r0 = 0; r0 < n_quad; r0 += 64) { // (uniform trip count)
        const uint32_t r = r0 + (tid >> 2), j = tid & 3;
        if (r >= n_quad) continue; // (whole quads)
        const uint32_t qa = s_ql[r];
        const uint32_t a = start + qa;
        // scores relative to N0(a - 1); a run starting at position 1 or 2 competes with a read's start node on absolute
        // scores (k_dp_bt_*'s early_run_base): N0(0) is a path start, position 1 is clean when a == 2
        int32_t base = 0;
        if (a == 1 || a == 2) base = 6 * (int32_t)s_cov[0] + (a == 2 ? 6 * (int32_t)s_cov[1] : 0);
        uint32_t cc2 = s_ref[qa + 1], cc1 = s_ref[qa + 2], cc0 = s_ref[qa + 3]; // codes of a - 2, a - 1, a  (s_ref[i] = code of start - 3 + i)
        // this lane's node of the previous position: key, score, valid
        uint32_t pK = 0;
        int32_t pS = PF_NEG;
        uint32_t pV = 0;
        if (a > 0 && j == 0) {
            uint32_t b, d;
            pf_n0_key(a - 1, s_ref[qa], cc2, cc1, b, d);
            pK = b | (d << 16), pS = base, pV = 1;
        }
        int32_t my_score = 0, pv_s0 = base;
        uint32_t pv_k0 = 0, pv_n = 0;
        uint32_t q = qa;
        bool closed = false;
        for (; q < ext; ++q) {
            const uint32_t p = start + q;
            const uint32_t k0 = s_off[q], nq = s_off[q + 1] - k0;
            const int32_t cov = s_cov[q];
            uint32_t b0, d0;
            pf_n0_key(p, cc2, cc1, cc0, b0, d0);
            const bool valid = j == 0 || j - 1 < nq;
            uint32_t key = b0 | (d0 << 16), cnt = 0;
            if (j && valid) key = s_nkey[k0 + j - 1], cnt = s_ncw[k0 + j - 1] & PF_IDX_MASK;
            const uint32_t ce = (j && valid && node_delta3((uint16_t)key, (uint16_t)(key >> 16)) == 0) ? cnt : 0u;
            const uint32_t e0 = pf_quad_bcast<1>(ce) + pf_quad_bcast<2>(ce) + pf_quad_bcast<3>(ce);
            if (j == 0) cnt = (uint32_t)cov - e0;
            const uint32_t kb = key & 0xFFFFu, kd = key >> 16;
            const int32_t w = 10 * (int32_t)cnt - 4 * cov;
            const bool head = ((kb >> 4) & 0xFu) == 15u; // second column is a head sentinel: a path starts here
            const bool same_pos = (kb & 0x1000u) != 0;   // second column at p, else at p - 1
            const uint32_t want = ((kb >> 4) & 0xFFu) | (((kb >> 14) & 1u) << 12);
            int32_t score = head ? w : PF_NEG;
            uint32_t besti = 0;
            { // the previous position's nodes, in the reference's order
                const bool mine = valid && !head && !same_pos, far = p - 1 >= 3;
                uint32_t vk, vv;
                int32_t vs;
                vk = pf_quad_bcast<0>(pK), vs = (int32_t)pf_quad_bcast<0>((uint32_t)pS), vv = pf_quad_bcast<0>(pV);
                if (mine && vv) pred_test(vk, vs, want, kd, far, w, 0u, score, besti);
                vk = pf_quad_bcast<1>(pK), vs = (int32_t)pf_quad_bcast<1>((uint32_t)pS), vv = pf_quad_bcast<1>(pV);
                if (mine && vv) pred_test(vk, vs, want, kd, far, w, 1u, score, besti);
                vk = pf_quad_bcast<2>(pK), vs = (int32_t)pf_quad_bcast<2>((uint32_t)pS), vv = pf_quad_bcast<2>(pV);
                if (mine && vv) pred_test(vk, vs, want, kd, far, w, 2u, score, besti);
                vk = pf_quad_bcast<3>(pK), vs = (int32_t)pf_quad_bcast<3>((uint32_t)pS), vv = pf_quad_bcast<3>(pV);
                if (mine && vv) pred_test(vk, vs, want, kd, far, w, 3u, score, besti);
            }
            { // the earlier nodes of this position (node i is final before round i: its own predecessors are nodes < i)
                const bool mine = valid && !head && same_pos, far = p >= 3;
                uint32_t vk;
                int32_t vs;
                vk = pf_quad_bcast<0>(key), vs = (int32_t)pf_quad_bcast<0>((uint32_t)score);
                if (mine && j > 0) pred_test(vk, vs, want, kd, far, w, 0u, score, besti);
                vk = pf_quad_bcast<1>(key), vs = (int32_t)pf_quad_bcast<1>((uint32_t)score);
                if (mine && j > 1) pred_test(vk, vs, want, kd, far, w, 1u, score, besti);
                vk = pf_quad_bcast<2>(key), vs = (int32_t)pf_quad_bcast<2>((uint32_t)score);
                if (mine && j > 2) pred_test(vk, vs, want, kd, far, w, 2u, score, besti);
            }
            if (j == 0) {
                s_n0bi[q] = (uint16_t)besti;
                my_score = score;
            } else if (valid) {
                s_score[k0 + j - 1] = score;
                s_ncw[k0 + j - 1] = cnt | (besti << PF_COUNT_BITS);
            }
            if (nq == 0) { // the clean position that closes the run
                closed = true;
                break;
            }
            pK = key, pS = score, pV = valid ? 1u : 0u;
            pv_k0 = k0, pv_n = nq, pv_s0 = (int32_t)pf_quad_bcast<0>((uint32_t)score);
            cc2 = cc1, cc1 = cc0, cc0 = s_ref[q + 4];
        }
        if (j == 0) finish_run(qa, q, closed, my_score, base, pv_s0, pv_k0, pv_n);
    }
    // (3) the runs with a deeper position: one lane each, every (node, predecessor) pair out of LDS
    for (uint32_t r = tid; r < n_one; r += 256) {
        const uint32_t qa = s_ql[TILE / 2 - r];
        const uint32_t a = start + qa;
        int32_t base = 0;
        if (a == 1 || a == 2) base = 6 * (int32_t)s_cov[0] + (a == 2 ? 6 * (int32_t)s_cov[1] : 0);
        uint32_t cc2 = s_ref[qa + 1], cc1 = s_ref[qa + 2], cc0 = s_ref[qa + 3];
        uint32_t pv_b0 = 0, pv_d0 = 0;
        bool pv_valid = a > 0;
        if (pv_valid) pf_n0_key(a - 1, s_ref[qa], cc2, cc1, pv_b0, pv_d0);
        int32_t pv_s0 = base, s0_cur = 0;
        uint32_t pv_k0 = 0, pv_n = 0;
        uint32_t q = qa;
        bool closed = false;
        for (; q < ext; ++q) {
            const uint32_t p = start + q;
            const uint32_t k0 = s_off[q], nq = s_off[q + 1] - k0;
            const int32_t cov = s_cov[q];
            uint32_t b0, d0;
            pf_n0_key(p, cc2, cc1, cc0, b0, d0);
            uint32_t e0 = 0;
            for (uint32_t k = 0; k < nq; ++k) {
                const uint32_t key = s_nkey[k0 + k];
                if (node_delta3((uint16_t)key, (uint16_t)(key >> 16)) == 0) e0 += s_ncw[k0 + k] & PF_IDX_MASK;
            }
            const int32_t cn0 = cov - (int32_t)e0;
            for (uint32_t idx = 0; idx <= nq; ++idx) {
                uint32_t kb = b0, kd = d0;
                int32_t cnt = cn0;
                if (idx) {
                    const uint32_t key = s_nkey[k0 + idx - 1];
                    kb = key & 0xFFFFu, kd = key >> 16, cnt = (int32_t)(s_ncw[k0 + idx - 1] & PF_IDX_MASK);
                }
                const int32_t w = 10 * cnt - 4 * cov;
                int32_t score;
                uint32_t besti = 0;
                if (((kb >> 4) & 0xF) == 15) { // second column is a head sentinel: a path starts here
                    score = w;
                } else {
                    score = PF_NEG;
                    const bool same_pos = (kb & 0x1000) != 0; // second column at p, else at p - 1
                    const uint32_t qq = same_pos ? p : p - 1;
                    const uint32_t want = ((kb >> 4) & 0xFFu) | (((kb >> 14) & 1u) << 12);
                    uint32_t qk0 = 0, qn = 0, qb0 = 0, qd0 = 0;
                    int32_t qs0 = 0;
                    bool ok = false;
                    if (same_pos) {
                        qk0 = k0, qn = idx, qb0 = b0, qd0 = d0, qs0 = s0_cur, ok = true;
                    } else if (pv_valid) {
                        qk0 = pv_k0, qn = 1 + pv_n, qb0 = pv_b0, qd0 = pv_d0, qs0 = pv_s0, ok = true;
                    }
                    if (ok) {
                        for (uint32_t pi = 0; pi < qn; ++pi) { // predecessors in the reference's order (main.rs:1664-1674)
                            uint32_t vkey = qb0 | (qd0 << 16);
                            int32_t ps = qs0;
                            if (pi) vkey = s_nkey[qk0 + pi - 1], ps = s_score[qk0 + pi - 1];
                            pred_test(vkey, ps, want, kd, qq >= 3, w, pi, score, besti);
                        }
                    }
                }
                if (idx) {
                    s_score[k0 + idx - 1] = score;
                    s_ncw[k0 + idx - 1] = (uint32_t)cnt | (besti << PF_COUNT_BITS);
                } else {
                    s0_cur = score;
                    s_n0bi[q] = (uint16_t)besti;
                }
            }
            if (nq == 0) { // the clean position that closes the run
                closed = true;
                break;
            }
            pv_k0 = k0, pv_n = nq, pv_b0 = b0, pv_d0 = d0, pv_s0 = s0_cur, pv_valid = true;
            cc2 = cc1, cc1 = cc0, cc0 = s_ref[q + 4];
        }
        finish_run(qa, q, closed, s0_cur, base, pv_s0, pv_k0, pv_n);
    }
    stamp(4);
    for (int o = 32; o > 0; o >>= 1) gain += __shfl_down(gain, o);
    if (lane == 0) s_gain[wv] = gain;
    __syncthreads();
    stamp(5);
    if (s_flag[1]) { // (uniform) a run did not close inside the halo
        if (tid == 0) hand_on();
        return;
    }
    // ---- write-out: every position emits its bases (clean: the contig's; dirty: the marked nodes, in node order) ------------
    const uint32_t pb = s_flag[2]; // positions left of the path's first node emit nothing (tile 0, pb <= 2)
    // entry = position in the tile | base code << 11 | class << 14
    auto pos_entries = [&](uint32_t q, uint32_t cnt_q, uint32_t o0, int32_t cov, uint32_t code, uint16_t *out, uint32_t &n_lq) -> uint32_t {
        const uint32_t p = start + q;
        if (q >= ext || p < pb) return 0;
        if (cnt_q == 0) {
            if (q >= npos || code == 4) return 0;
            if (out) out[0] = (uint16_t)(q | (code << 11) | ((cov < 2 ? CLS_RESET : CLS_HQ) << 14));
            return 1;
        }
        uint32_t m = 0, e0 = 0;
        bool any = (s_n0bi[q] & PF_N0_VISITED) != 0;
        for (uint32_t k = o0; k < o0 + cnt_q; ++k) {
            const uint32_t key = s_nkey[k], x = s_ncw[k];
            if (node_delta3((uint16_t)key, (uint16_t)(key >> 16)) == 0) e0 += x & PF_IDX_MASK;
            any |= (x & PF_VISITED) != 0;
        }
        if (!any) return 0;
        auto put = [&](uint32_t base_code, uint32_t cnt) {
            if (base_code == 4) return;
            const bool lq = (int64_t)cnt * 100 < 95 * (int64_t)cov; // qv = count * 100 / coverage < 95 (main.rs:1572)
            const uint32_t cls = cov < 2 ? CLS_RESET : (lq ? CLS_LQ : CLS_HQ);
            if (out) out[m] = (uint16_t)(q | (base_code << 11) | (cls << 14));
            n_lq += cls == CLS_LQ ? 1u : 0u;
            ++m;
        };
        if (s_n0bi[q] & PF_N0_VISITED) put(code, (uint32_t)cov - e0);
        for (uint32_t k = o0; k < o0 + cnt_q; ++k) {
            const uint32_t x = s_ncw[k];
            if (x & PF_VISITED) put(s_nkey[k] & 0xFu, x & PF_IDX_MASK);
        }
        return m;
    };
    // (offsets, coverage and contig codes come back from LDS here: kept in registers across the DP they cost a third of the
    // kernel's occupancy)
    auto pos_do = [&](uint32_t q, uint16_t *out, uint32_t &n_lq) -> uint32_t {
        const uint32_t o0 = s_off[q];
        return pos_entries(q, s_off[q + 1] - o0, o0, (int32_t)s_cov[q], s_ref[3 + q], out, n_lq);
    };
    uint32_t em[4], emh[HP], ems = 0, emhs = 0, dummy = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        em[j] = pos_do(q0 + j, nullptr, dummy);
        ems += em[j];
    }
#pragma unroll
    for (uint32_t j = 0; j < HP; ++j) {
        emh[j] = 0;
        const uint32_t q = TILE + tid * HP + j;
        if (tid * HP + j < HALO && s_off[q + 1] != s_off[q]) emh[j] = pos_do(q, nullptr, dummy);
        emhs += emh[j];
    }
    uint32_t etot;
    uint32_t eo = block_excl_scan<OpAdd, 4>(ems, sh, etot);
    uint32_t eh, ehtot;
    if constexpr (HALO == 64) {
        const uint32_t inc = wave_incl_scan<OpAdd>(emhs);
        eh = etot + inc - emhs;
        ehtot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63); // (meaningful in wave 0 only)
    } else {
        eh = etot + block_excl_scan<OpAdd, 4>(emhs, sh, ehtot);
    }
    stamp(6);
    uint16_t *const slot = A.slots + pf_slot_off(A.tile_scan, t);
    uint32_t n_lq = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        if (em[j]) pos_do(q0 + j, slot + eo, n_lq);
        eo += em[j];
    }
#pragma unroll
    for (uint32_t j = 0; j < HP; ++j) {
        if (emh[j]) pos_do(TILE + tid * HP + j, slot + eh, n_lq);
        eh += emh[j];
    }
    if (n_lq) atomicAdd(&s_flag[3], n_lq);
    __syncthreads();
    stamp(7);
    if (tid == 0) {
        A.tile_cnt[t] = etot + ehtot;
        A.tile_lq[t] = s_flag[3];
        A.tile_gain[t] = s_gain[0] + s_gain[1] + s_gain[2] + s_gain[3];
        // the best end node's score relative to the total of the gains: by the block that owns the run reaching the last
        // position, or — the last position is clean — by the last tile's block
        const bool last_clean = t + 1 == A.n_tiles && s_off[L - 1 - start + 1] == s_off[L - 1 - start];
        if (s_flag[4] || last_clean) *A.end_rel = s_endrel;
        if (t == 0) *A.gain_total = 0; // (summed by k_tile_offsets, which runs after every block of this kernel)
    }
}

__device__ __forceinline__ void k_pf_tile(const uint32_t np2_bid, const uint32_t np2_nb, PfTile A) {
    pf_tile_body<PF_CAP, PF_HALO, 0>(xcd_order(np2_bid, np2_nb), A); // (neighbouring tiles — shared reads, the halo's records — on one XCD)
}
// the tiles the kernel above listed, with room for 2048 records (a handful of blocks walk the list) ...
__device__ __forceinline__ void k_pf_tile_mid(const uint32_t np2_bid, const uint32_t np2_nb, PfTile A) {
    const uint32_t nb = *A.n_bad;
    for (uint32_t i = np2_bid; i < nb; i += np2_nb) {
        pf_tile_body<PF_CAP_MID, PF_HALO, 1>(A.bad_list[i], A);
        __syncthreads();
    }
}
// ... and what that one listed, with room for 3584 records and a whole tile of halo
__device__ __forceinline__ void k_pf_tile_big(const uint32_t np2_bid, const uint32_t np2_nb, PfTile A) {
    const uint32_t nb = *A.n_bad2;
    for (uint32_t i = np2_bid; i < nb; i += np2_nb) {
        pf_tile_body<PF_CAP_BIG, TILE, 2>(A.bad_list2[i], A);
        __syncthreads();
    }
}

// Per-tile slots -> the contiguous consensus (cns_pos / cns_base / cns_cls, LQ chain flags cleared) and the ascending list
// of the low-quality bases' consensus indices.  tile_coff / tile_lqoff: exclusive scans of the per-tile counts.
__device__ __forceinline__ void k_pf_compact(const uint32_t np2_bid, const uint32_t np2_nb, const uint16_t *__restrict__ slots,
                                             const uint32_t *__restrict__ tile_scan, const uint32_t *__restrict__ tile_cnt,
                                             const uint32_t *__restrict__ tile_coff, const uint32_t *__restrict__ tile_lqoff,
                                             uint32_t *__restrict__ cns_pos, uint8_t *__restrict__ cns_base,
                                             uint8_t *__restrict__ cns_cls, uint8_t *__restrict__ lq_nothead,
                                             uint32_t *__restrict__ lq_list, uint32_t lq_cap, uint32_t *__restrict__ err,
                                             uint32_t *__restrict__ flags, uint32_t *__restrict__ flags_out,
                                             uint32_t *__restrict__ n_bad, uint32_t *__restrict__ n_bad2) {
    __shared__ uint32_t s_w[4];
    const uint32_t t = np2_bid, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (t == 0 && tid == 0) { // the pass's flags to where the host reads them; flags and bad-tile counter ready for the next pass
        *flags_out = *flags;
        *flags = 0;
        *n_bad = 0;
        *n_bad2 = 0;
    }
    const uint32_t n = tile_cnt[t], o0 = tile_coff[t], start = t << TILE_SHIFT;
    const uint16_t *__restrict__ src = slots + pf_slot_off(tile_scan, t);
    uint32_t lqo = tile_lqoff[t];
    for (uint32_t i0 = 0; i0 < n; i0 += 256) { // (uniform trip count)
        const uint32_t i = i0 + tid;
        uint32_t e = 0;
        bool lq = false;
        if (i < n) {
            e = src[i];
            const uint32_t cls = e >> 14;
            cns_pos[o0 + i] = start + (e & 0x7FFu);
            cns_base[o0 + i] = code_to_ascii((uint8_t)((e >> 11) & 7));
            cns_cls[o0 + i] = (uint8_t)cls;
            lq_nothead[o0 + i] = 0;
            lq = cls == CLS_LQ;
        }
        const uint64_t bal = __ballot(lq);
        if (lane == 0) s_w[wv] = (uint32_t)__builtin_popcountll(bal);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) {
            const uint32_t x = s_w[w];
            all += x;
            if (w < wv) before += x;
        }
        if (lq) {
            const uint32_t k = lqo + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (k < lq_cap)
                lq_list[k] = o0 + i;
            else
                atomicOr(err, LQ_LIST_ERR);
        }
        lqo += all;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------
uint64_t pf_slot_entries(uint32_t n_tiles, uint64_t T) { return (uint64_t)n_tiles * (TILE + 1) + 3 * T + 64; }
void launch_pf_tile(hipStream_t s, const PfTile &a) {
    NP2_LAUNCH(k_pf_tile, dim3(a.n_tiles), 256, s, a);
    NP2_LAUNCH(k_pf_tile_mid, dim3(std::min<uint32_t>(a.n_tiles, 256u)), 256, s, a);
    if (a.big_enabled) NP2_LAUNCH(k_pf_tile_big, dim3(std::min<uint32_t>(a.n_tiles, 64u)), 256, s, a);
}
void launch_pf_compact(hipStream_t s, uint32_t n_tiles, const uint16_t *slots, const uint32_t *tile_scan, const uint32_t *tile_cnt,
                       const uint32_t *tile_coff, const uint32_t *tile_lqoff, uint32_t *cns_pos, uint8_t *cns_base,
                       uint8_t *cns_cls, uint8_t *lq_nothead, uint32_t *lq_list, uint32_t lq_cap, uint32_t *err, uint32_t *flags,
                       uint32_t *flags_out, uint32_t *n_bad, uint32_t *n_bad2) {
    NP2_LAUNCH(k_pf_compact, dim3(n_tiles), 256, s, slots, tile_scan, tile_cnt, tile_coff, tile_lqoff, cns_pos, cns_base, cns_cls,
               lq_nothead, lq_list, lq_cap, err, flags, flags_out, n_bad, n_bad2);
}

} // namespace np2
