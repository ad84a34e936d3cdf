// Hand-written HIP kernels (gfx950 / CDNA4, wave64) for the NextPolish2 consensus hot path.
//
// Design: "ref-diff sparse graph" (DESIGN.md).  The reference builds, per contig position, a
// multiset of 3-column-mers from every read column (update_msas, src/main.rs:576-589).
// At HiFi error rates >99 % of those insertions reproduce the node the contig itself
// contributes (read 0).  k_diff_reads streams the packed pileup once (HBM-bound, 0.5 B per
// column), compares it against the nibble-packed contig and emits only *exception* nodes;
// the implicit node N0(p) of the contig has count = coverage(p) - #exceptions(delta3 == 0).
// Everything downstream (DP, backtrack, LQ detection, candidates) runs on that sparse graph
// and is exact: positions without exceptions hold a single node every path passes through,
// so the whole-contig DP (main.rs:1645-1687) decomposes into independent "dirty runs".
#include "np2_common.hpp"
#include "np2_kernels.hpp"
#include "np2_blockscan.hpp"
#include "np2_nib128.hpp"
#include "../../include/np2.h"

namespace np2 {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t ref_code(const uint8_t *__restrict__ refnib, uint32_t p) {
    return (refnib[p >> 1] >> (4 * (p & 1))) & 7;
}

// ------------------------------------------------------------------------------------------
// K0: contig codes from read 0 (the contig aligned to itself, main.rs:1732-1739)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void k_encode_ref(const uint32_t np2_bid, const uint32_t np2_nb, const uint8_t *__restrict__ read0, uint32_t L, uint8_t *__restrict__ refnib,
                             uint32_t nbytes_total, uint32_t stride, uint32_t *__restrict__ err) {
    uint32_t i = np2_bid * blockDim.x + threadIdx.x;
    if (i >= nbytes_total) return;
    uint32_t c0 = 2 * i, c1 = 2 * i + 1;
    uint8_t b = c0 < L ? read0[i] : 0;
    uint8_t hi = c0 < L ? (b >> 4) : 0, lo = c1 < L ? (b & 15) : 0;
    if ((c0 < L && c0 != 0 && (hi & 8)) || (c1 < L && (lo & 8))) atomicOr(err, 1u); // read 0 has an insertion column
    if (c0 < L && (hi & 7) == 7) atomicOr(err, 1u);
    if (c1 < L && (lo & 7) == 7) atomicOr(err, 1u);
    refnib[i] = (uint8_t)((hi & 7) | ((lo & 7) << 4));
    // the dense pass's copies, in the packed streams' nibble order: positions 2 i | 2 i + 1 and 2 i + 1 | 2 i + 2
    const uint8_t nx = c1 + 1 < L ? (uint8_t)((read0[i + 1] >> 4) & 7) : (uint8_t)0;
    refnib[stride + i] = (uint8_t)(((hi & 7) << 4) | (lo & 7));
    refnib[2 * (size_t)stride + i] = (uint8_t)(((lo & 7) << 4) | nx);
}

// gather up to four device-resident counters into scalar slots, then post the whole scalar block to host-mapped memory
// followed by a sequence number the host spins on (no copy command, no stream synchronisation)
__device__ __forceinline__ void k_post(const uint32_t np2_bid, const uint32_t np2_nb, uint32_t *__restrict__ scal, uint32_t n_scal, uint32_t *__restrict__ mbox, uint32_t seq,
                       uint32_t *__restrict__ dst0, const uint32_t *__restrict__ src0, uint32_t *__restrict__ dst1,
                       const uint32_t *__restrict__ src1, uint32_t *__restrict__ dst2, const uint32_t *__restrict__ src2,
                       uint32_t *__restrict__ dst3, const uint32_t *__restrict__ src3,
                       const uint32_t *__restrict__ ends_of, uint32_t *__restrict__ ends_dst) {
    if (threadIdx.x == 0) {
        if (dst0) *dst0 = *src0;
        if (dst1) *dst1 = *src1;
        if (dst2) *dst2 = *src2;
        if (dst3) *dst3 = *src3;
        if (ends_of) { // first and last element of an array whose device-side length was just gathered into *dst0
            const uint32_t n = *dst0;
            ends_dst[0] = n ? ends_of[0] : 0u;
            ends_dst[1] = n ? ends_of[n - 1] : 0u;
        }
    }
    __threadfence();
    if (threadIdx.x < n_scal) // one wavefront: a single store instruction posts the whole block
        mbox[1 + threadIdx.x] = __hip_atomic_load(&scal[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store(&mbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// fills / device-to-device copies as kernels: unlike hipMemsetAsync / hipMemcpyAsync they batch across contigs.
// 16 bytes per thread where the pointers allow it, byte-wise at unaligned ends.
__device__ __forceinline__ void k_fill(const uint32_t np2_bid, const uint32_t np2_nb, uint8_t *__restrict__ p, uint64_t n, uint32_t v4) {
    const uint64_t o = ((uint64_t)np2_bid * 256 + threadIdx.x) * 16;
    if (o >= n) return;
    if ((((uintptr_t)p) & 15) == 0 && o + 16 <= n) {
        *reinterpret_cast<uint4 *>(p + o) = make_uint4(v4, v4, v4, v4);
    } else {
        for (uint64_t i = o; i < min(o + 16, n); ++i) p[i] = (uint8_t)v4;
    }
}
__device__ __forceinline__ void k_copy(const uint32_t np2_bid, const uint32_t np2_nb, uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t n) {
    const uint64_t o = ((uint64_t)np2_bid * 256 + threadIdx.x) * 16;
    if (o >= n) return;
    const uintptr_t al = ((uintptr_t)dst) | ((uintptr_t)src);
    if ((al & 15) == 0 && o + 16 <= n) {
        *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(src + o);
    } else if ((al & 3) == 0 && o + 16 <= n) {
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q)
            *reinterpret_cast<uint32_t *>(dst + o + 4 * q) = *reinterpret_cast<const uint32_t *>(src + o + 4 * q);
    } else {
        for (uint64_t i = o; i < min(o + 16, n); ++i) dst[i] = src[i];
    }
}
// k_copy of min(*n_dev * elem, cap) bytes: the polished sequence leaves for the host at its real length, which only the
// device knows when the copy is recorded (the host's bound is the length plus every splice round's growth allowance:
// 11 % more bytes over the bus on the yeast-sized assembly)
__device__ __forceinline__ void k_copy_len(const uint32_t np2_bid, const uint32_t np2_nb, uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                           const uint32_t *__restrict__ n_dev, uint32_t elem, uint64_t cap) {
    k_copy(np2_bid, np2_nb, dst, src, min((uint64_t)*n_dev * elem, cap));
}
// copy of n 32-bit words where n = min(*n_dev, cap) lives on the device: the grid is sized by the host's bound, the
// threads walk the real length (16 bytes per thread and step).  A read-back whose size is only known on the device
// rides in the same wait as the counters that say how large it is.
__device__ __forceinline__ void k_copy_counted(const uint32_t np2_bid, const uint32_t np2_nb, uint32_t *__restrict__ dst, const uint32_t *__restrict__ src,
                                               const uint32_t *__restrict__ n_dev, uint32_t cap) {
    const uint32_t n = min(*n_dev, cap);
    const bool wide = ((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0;
    for (uint64_t o = ((uint64_t)np2_bid * 256 + threadIdx.x) * 4; o < n; o += (uint64_t)np2_nb * 256 * 4) {
        if (wide && o + 4 <= n) {
            *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(src + o);
        } else {
            for (uint64_t i = o; i < min(o + 4, (uint64_t)n); ++i) dst[i] = src[i];
        }
    }
}
__device__ __forceinline__ void k_init_alive(const uint32_t np2_bid, const uint32_t np2_nb, const np2_read_t *__restrict__ reads, uint32_t R, uint8_t *__restrict__ alive) {
    uint32_t r = np2_bid * blockDim.x + threadIdx.x;
    if (r < R) alive[r] = (reads[r].flags & NP2_READ_DROPPED) ? 0 : 1;
}
// the reads the phasing vote flagged on the device (they disagree with the contig's candidate at a marker, main.rs:977)
__device__ __forceinline__ void k_kill_flagged(const uint32_t np2_bid, const uint32_t np2_nb, const uint8_t *__restrict__ flag, uint32_t n, uint8_t *__restrict__ alive) {
    uint32_t i = np2_bid * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) alive[i] = 0;
}
__device__ __forceinline__ void k_kill_reads(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ ids, uint32_t n, uint8_t *__restrict__ alive) {
    uint32_t i = np2_bid * blockDim.x + threadIdx.x;
    if (i < n) alive[ids[i]] = 0;
}
__device__ __forceinline__ void k_revive_reads(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ ids, uint32_t n, uint8_t *__restrict__ alive) {
    uint32_t i = np2_bid * blockDim.x + threadIdx.x;
    if (i < n) alive[ids[i]] = 1;
}

// ------------------------------------------------------------------------------------------
// sparse-graph accessors.  Node index 0 at a position is the contig's implicit node N0(p);
// index 1+k is exception node node_off[p]+k.
// ------------------------------------------------------------------------------------------
struct Graph {
    const uint8_t *refnib;
    const uint32_t *node_off;
    NodeArrays nd;
    const int32_t *cov;
    uint32_t L;
    const uint2 *nrec; // packed node records {bases | delta << 16, count}
    const uint32_t *deep; // != 0: some position of this pass is covered 65536x or more (no run is "short" then)
};
__device__ __forceinline__ void n0_key(const Graph &g, uint32_t p, uint16_t &bases, uint16_t &delta) {
    const uint8_t c = ref_code(g.refnib, p);
    if (p >= 2) {
        bases = (uint16_t)((ref_code(g.refnib, p - 2) << 8) | (ref_code(g.refnib, p - 1) << 4) | c);
        delta = 0;
    } else if (p == 1) { // (head(-1,1), c0, c1)
        bases = (uint16_t)(0x0F00 | (ref_code(g.refnib, 0) << 4) | c);
        delta = 1;
    } else { // (head(-1,0), head(-1,1), c0)
        bases = (uint16_t)(0x4FF0 | c);
        delta = 0;
    }
}
__device__ __forceinline__ uint32_t n0_count(const Graph &g, uint32_t p) {
    uint32_t e0 = 0;
    for (uint32_t i = g.node_off[p]; i < g.node_off[p + 1]; ++i)
    {
        const uint2 rc = g.nrec[i];
        if (node_delta3((uint16_t)rc.x, (uint16_t)(rc.x >> 16)) == 0) e0 += rc.y;
    }
    return (uint32_t)g.cov[p] - e0;
}

// ------------------------------------------------------------------------------------------
// K5/K6: dirty runs and the per-run DP (get_cns_from_align_tags, main.rs:1645-1687)
// ------------------------------------------------------------------------------------------
// Scores inside a run are relative to its left neighbour N0(a - 1), whose own score (the gains of everything before
// the run) is added back by k_dp_finish.  That is exact as long as every path through the run comes from N0(a - 1) —
// not so for a run starting at position 1 or 2: there a read's head-sentinel start node is a live alternative
// (main.rs:1666-1668 only rejects them from t_pos 3 on) and carries an absolute score, so the run has to know the
// absolute score of N0(a - 1) as well.  N0(1)'s only possible predecessor is N0(0), a path start itself.
__device__ __forceinline__ int64_t early_run_base(const Graph &g, uint32_t a) {
    if (a != 1 && a != 2) return 0;
    int64_t s = 10 * (int64_t)(int32_t)n0_count(g, 0) - 4 * (int64_t)g.cov[0]; // N0(0): a path start
    if (a == 2) s += 6 * (int64_t)g.cov[1]; // position 1 is clean: count == coverage
    return s;
}

// ---- run classes ---------------------------------------------------------------------------------------------------------
// A "short" run (the great majority at HiFi error rates) has at most RW_P - 1 dirty positions followed by a clean one
// (inside the contig) and at most RW_N exception nodes: k_dp_bt_short scores and backtracks it entirely on chip, with
// 16-bit coverages / counts and 32-bit scores relative to the run's left neighbour — which is why a pass that has a
// position covered 65536x or more (Graph::deep, set by k_tile_write) has no short runs at all.  Everything else (long
// or node-rich runs, the run that reaches the contig end) goes to the eight-lane kernel.  The kernels either classify
// for themselves from the node offsets (no hand-over: they can run side by side on two streams) or the short kernel
// lists what it leaves alone (batch driver: one stream).
static constexpr uint32_t RW_P = 13; // positions of a short run kept on chip: up to 12 dirty ones + the closing clean one
static constexpr uint32_t RW_N = 8;  // exception nodes of a short run kept on chip (149 bytes of LDS per run: 16 waves per CU; 12 nodes / 12 waves measured slower)
struct __attribute__((packed, aligned(4))) U32x4 {
    uint32_t x, y, z, w;
};
struct __attribute__((packed, aligned(4))) U32x2 {
    uint32_t x, y;
};
// off[i] = node_off[a + i], i = 0 .. RW_P (the array is padded past L); returns the number of dirty positions, RW_P if
// the run does not close inside the window
__device__ __forceinline__ uint32_t short_run_len(const uint32_t (&off)[RW_P + 1], uint32_t a, uint32_t L) {
    const uint32_t lim = min(RW_P, L - a); // the closing clean position must exist (< L)
    uint32_t len = RW_P;
#pragma unroll
    for (uint32_t i = RW_P; i-- > 0;)
        if (i < lim && off[i + 1] == off[i]) len = i;
    return len;
}
// which kernel owns a run: the on-chip one takes runs of at most SHORT_LEN dirty positions (and RW_N nodes); its own
// duration is set by the longest run it takes (a serial chain per lane), the eight-lane kernel's by the longest run of
// the contig at a third of the cost per position — SHORT_LEN balances the two
static constexpr uint32_t SHORT_LEN = RW_P - 1;
__device__ __forceinline__ bool run_is_short(uint32_t len, uint32_t nn, bool deep) { return len <= SHORT_LEN && nn <= RW_N && !deep; }
__device__ __forceinline__ void load_run_offsets(const uint32_t *__restrict__ node_off, uint32_t a, uint32_t (&off)[RW_P + 1]) {
    static_assert(RW_P + 1 == 14, "three 4-dword loads + one 2-dword load");
    const U32x4 v0 = *reinterpret_cast<const U32x4 *>(node_off + a);
    const U32x4 v1 = *reinterpret_cast<const U32x4 *>(node_off + a + 4);
    const U32x4 v2 = *reinterpret_cast<const U32x4 *>(node_off + a + 8);
    const U32x2 v3 = *reinterpret_cast<const U32x2 *>(node_off + a + 12);
    off[0] = v0.x, off[1] = v0.y, off[2] = v0.z, off[3] = v0.w, off[4] = v1.x, off[5] = v1.y, off[6] = v1.z, off[7] = v1.w;
    off[8] = v2.x, off[9] = v2.y, off[10] = v2.z, off[11] = v2.w, off[12] = v3.x, off[13] = v3.y;
}

__device__ uint32_t bt_walk(const Graph &g, uint32_t a, uint32_t b, uint32_t entry_idx, const uint32_t *__restrict__ nbesti,
                            const uint32_t *__restrict__ n0_besti, uint32_t *__restrict__ path_begin,
                            uint64_t *__restrict__ path);

// One thread per long dirty run.  LDS holds the node records and scores of two positions per thread — the current one and
// its predecessor, in two banks selected by the position's parity (element-major, one 4-byte bank per thread: conflict
// free); a position with more than DP_NR exception nodes falls back to global memory for the excess.
static constexpr uint32_t DP_NR = 8; // exception nodes of one position cached in LDS (per thread)
static constexpr uint32_t DP_BLOCK = 64;
static constexpr uint32_t DP_GRID_CAP = 8192; // workgroups of the per-run kernels (grid-stride beyond)

__device__ __forceinline__ void n0_from_codes(uint32_t p, uint8_t c2, uint8_t c1, uint8_t c0, uint16_t &bases,
                                              uint16_t &delta) {
    if (p >= 2) {
        bases = (uint16_t)((c2 << 8) | (c1 << 4) | c0);
        delta = 0;
    } else if (p == 1) { // (head(-1,1), c0, c1)
        bases = (uint16_t)(0x0F00 | (c1 << 4) | c0);
        delta = 1;
    } else { // (head(-1,0), head(-1,1), c0)
        bases = (uint16_t)(0x4FF0 | c0);
        delta = 0;
    }
}

// The serial DP over a run's positions never waits on memory inside a position: while position p is scored, the
// scalars of p + 1 (node_off, coverage, contig code) and the DP_NR node records that follow p's are in flight; they
// are consumed / parked in the other LDS bank at the end of the step.
__device__ __forceinline__ void dp_bt_long_run(uint32_t r, const uint32_t *__restrict__ run_start, const Graph &g,
                                               const uint2 *__restrict__ nrec, int64_t *__restrict__ nscore,
                                               uint32_t *__restrict__ nbesti, uint32_t *__restrict__ n0_besti,
                                               uint32_t *__restrict__ run_end, int64_t *__restrict__ last_n0_score,
                                               int64_t *__restrict__ run_gain, uint32_t *__restrict__ emit,
                                               uint32_t *__restrict__ path_begin, uint64_t *__restrict__ path,
                                               const uint8_t *__restrict__ run_flag, bool listed,
                                               uint4 (*s_node)[DP_BLOCK]) {
    const uint32_t t = threadIdx.x;
    const uint32_t a = run_start[r], L = g.L;
    uint32_t o0, o1;
    if (listed) { // the run comes from k_dp_bt_short's list of runs it left alone
        if (!run_flag[r]) return; // done by k_dp_bt_oct
        o0 = g.node_off[a], o1 = g.node_off[a + 1];
    } else {
        uint32_t off[RW_P + 1];
        load_run_offsets(g.node_off, a, off);
        const uint32_t len = short_run_len(off, a, L);
        if (len < RW_P && run_is_short(len, off[len] - off[0], *g.deep != 0)) return; // a short run: k_dp_bt_short's
        if (!run_flag[r]) return;                              // done by k_dp_bt_oct
        o0 = off[0], o1 = off[1];
    }
    const uint32_t o_first = o0;
    int64_t cov = g.cov[a];
    uint8_t c2 = a >= 2 ? ref_code(g.refnib, a - 2) : 0, c1 = a >= 1 ? ref_code(g.refnib, a - 1) : 0;
    uint8_t c0 = ref_code(g.refnib, a);
    const uint8_t c3 = a >= 3 ? ref_code(g.refnib, a - 3) : 0;
#pragma unroll
    for (uint32_t k = 0; k < DP_NR; ++k) { // the node arrays are padded by DP_NR entries
        const uint2 rc = nrec[o0 + k];
        s_node[(a & 1) * DP_NR + k][t] = make_uint4(rc.x, rc.y, 0u, 0u);
    }
    // k-th exception node of a position whose nodes start at `base` and live in LDS bank `bank`
    auto node_at = [&](uint32_t bank, uint32_t base, uint32_t k) -> uint4 { // {key, count, score lo, score hi}
        if (k < DP_NR) return s_node[bank * DP_NR + k][t];
        const uint2 rc = nrec[base + k];
        const uint64_t sc = (uint64_t)nscore[base + k];
        return make_uint4(rc.x, rc.y, (uint32_t)sc, (uint32_t)(sc >> 32));
    };
    auto rec_at = [&](uint32_t bank, uint32_t base, uint32_t k) -> uint2 {
        if (k < DP_NR) {
            const uint4 v = s_node[bank * DP_NR + k][t];
            return make_uint2(v.x, v.y);
        }
        return nrec[base + k];
    };
    // previous position (starts as the clean position a-1: only N0, score 0 by convention)
    uint32_t pv_o0 = 0, pv_n = 0; // exception nodes of the previous position
    uint16_t pv_b0 = 0, pv_d0 = 0;
    const int64_t base = early_run_base(g, a);
    int64_t pv_s0 = base;
    bool pv_valid = a > 0;
    if (pv_valid) n0_from_codes(a - 1, c3, c2, c1, pv_b0, pv_d0);
    for (uint32_t p = a; p < L; ++p) {
        const uint32_t bank = p & 1;
        // request the next position's scalars and node records now; they are consumed at the end of this iteration
        uint32_t nx_o1 = o1;
        int64_t nx_cov = 0;
        uint8_t nx_c = 0;
        uint2 nx_rec[DP_NR];
        if (p + 1 < L) {
            nx_o1 = g.node_off[p + 2];
            nx_cov = g.cov[p + 1];
            nx_c = ref_code(g.refnib, p + 1);
        }
        const bool in_run = o1 > o0;
        if (in_run) {
#pragma unroll
            for (uint32_t k = 0; k < DP_NR; ++k) nx_rec[k] = nrec[o1 + k];
        }
        uint16_t b0, d0;
        n0_from_codes(p, c2, c1, c0, b0, d0);
        const uint32_t n = o1 - o0;
        uint32_t e0 = 0;
        for (uint32_t k = 0; k < n; ++k) {
            const uint2 rc = rec_at(bank, o0, k);
            if (node_delta3((uint16_t)rc.x, (uint16_t)(rc.x >> 16)) == 0) e0 += rc.y;
        }
        const int64_t cn0 = cov - (int64_t)e0;
        int64_t s0_cur = 0;
        for (uint32_t idx = 0; idx <= n; ++idx) {
            uint16_t kb = b0, kd = d0;
            int64_t cnt = cn0;
            if (idx) {
                const uint2 rc = rec_at(bank, o0, idx - 1);
                kb = (uint16_t)rc.x, kd = (uint16_t)(rc.x >> 16), cnt = rc.y;
            }
            // Predecessor test without decoding the nodes (Msa::get, main.rs:209-225 + the eq checks of main.rs:1664):
            // V = (v1, v2, v3) precedes K = (b1, b2, b3) iff (v2, v3) == (b1, b2).  V is looked up at t_pos(b2), so the
            // positions agree as soon as "v2, v3 share a position" (V bit 12) equals "b1, b2 share one" (K bit 14); the
            // base codes are the low byte of V against bits 4..11 of K; of the deltas only v2's has to be compared
            // with b1's (= kd): the third column's follows from the shared-position flag (node_decode).
            int64_t score;
            uint32_t besti = 0;
            if (((kb >> 4) & 0xF) == 15) { // b2 is a head sentinel: the path starts here
                score = 10 * cnt - 4 * cov;
            } else {
                score = SCORE_NEG;
                const bool same_pos = (kb & 0x1000) != 0; // b2 sits at p, else at p - 1
                const uint32_t q = same_pos ? p : p - 1;
                const uint32_t want = ((kb >> 4) & 0xFFu) | (((kb >> 14) & 1u) << 12);
                uint32_t qo0 = 0, qn = 0, qbank = bank;
                uint16_t qb0 = 0, qd0 = 0;
                int64_t qs0 = 0;
                bool ok = false;
                if (same_pos) { // same position: only nodes before K can match (their b3.delta = K.b2.delta)
                    qo0 = o0, qn = idx, qb0 = b0, qd0 = d0, qs0 = s0_cur, ok = true;
                } else if (pv_valid) {
                    qo0 = pv_o0, qn = 1 + pv_n, qb0 = pv_b0, qd0 = pv_d0, qs0 = pv_s0, qbank = bank ^ 1, ok = true;
                }
                if (ok) {
                    for (uint32_t pi = 0; pi < qn; ++pi) {
                        uint16_t vb = qb0, vd = qd0;
                        int64_t ps = qs0;
                        if (pi) {
                            const uint4 v = node_at(qbank, qo0, pi - 1);
                            vb = (uint16_t)v.x, vd = (uint16_t)(v.x >> 16);
                            ps = (int64_t)(((uint64_t)v.w << 32) | v.z);
                        }
                        if ((vb & 0x10FFu) != want) continue;
                        const uint16_t v2d = (vb & 0x4000) ? (uint16_t)(vd + 1) : (uint16_t)0;
                        if (v2d != kd) continue;
                        const uint32_t v1q = (vb >> 8) & 0xFu;
                        if (q >= 3 && v1q == 15) continue; // main.rs:1666-1668
                        const int64_t sc = ps + 10 * cnt - 4 * cov;
                        if (sc > score || (sc == score && v1q != 4)) { // main.rs:1670
                            score = sc;
                            besti = pi;
                        }
                    }
                }
            }
            if (idx) {
                const uint32_t k = idx - 1;
                if (k < DP_NR) {
                    s_node[bank * DP_NR + k][t].z = (uint32_t)(uint64_t)score;
                    s_node[bank * DP_NR + k][t].w = (uint32_t)((uint64_t)score >> 32);
                }
                // scores are only read back from memory past the LDS cache and, at the contig's last position, by
                // k_dp_finish: skip the scattered 8-byte store otherwise
                if (k >= DP_NR || p + 1 == L) nscore[o0 + k] = score;
                nbesti[o0 + k] = besti;
            } else {
                s0_cur = score;
                n0_besti[p] = besti;
            }
        }
        if (!in_run) { // p == b+1: the clean position closing the run; its N0 is scored above
            run_end[r] = p - 1;
            run_gain[r] = s0_cur - base; // summed by k_dp_finish (one same-address atomic per run would serialise at L2)
            // backtrack from this closing position's best predecessor (the thread's own stores, read back in order)
            emit[a] = bt_walk(g, a, p - 1, n0_besti[p], nbesti, n0_besti, path_begin, path + (size_t)a + o_first);
            return;
        }
        // park the next position's records in the other bank (its previous content, position p - 1, is dead now)
#pragma unroll
        for (uint32_t k = 0; k < DP_NR; ++k) {
            s_node[(bank ^ 1) * DP_NR + k][t] = make_uint4(nx_rec[k].x, nx_rec[k].y, 0u, 0u);
        }
        pv_o0 = o0, pv_n = n, pv_b0 = b0, pv_d0 = d0, pv_s0 = s0_cur, pv_valid = true;
        o0 = o1, o1 = nx_o1, cov = nx_cov;
        c2 = c1, c1 = c0, c0 = nx_c;
    }
    // the run reaches the contig end
    run_end[r] = L - 1;
    run_gain[r] = -base; // (k_dp_finish adds the total of all gains to this run's scores, which already contain `base`)
    *last_n0_score = pv_s0;
}

// ------------------------------------------------------------------------------------------
// Long runs, eight lanes per run.  A long run is a long serial chain, and the longest one in the contig sets the duration
// of the whole DP stage, so here the chain is made as short as possible instead of as cheap as possible: lane j of the
// octet holds exception node j of the current and of the previous position (key, count, score) in registers, the node
// being scored is broadcast with a sub-wave shuffle, every lane tests "its" predecessor and a three-step butterfly
// merges the candidates with the reference's tie rule.  No LDS, no memory round trip inside a position; the next
// position's scalars and nodes are in flight meanwhile.  A position with more than 8 exception nodes hands the run to
// the per-thread kernel (run_flag = 1).
//
// Candidate merge.  The reference scans predecessors in order and takes one if it scores higher, or equal with a
// first base other than '-' (main.rs:1670).  Over any set of candidates that is: M = the best score, f = the first
// candidate reaching M, l = the last one reaching M with the flag; the winner is l if it comes after f, else f.  The
// scan's initial state (SCORE_NEG, index 0) is a candidate before all others (f = -1).  (M, f, l) triples of disjoint
// sets merge by: higher M wins, equal M -> (min f, max l).
// ------------------------------------------------------------------------------------------
struct Cand3 {
    int64_t m;
    int32_t f, l;
};
// butterfly partners inside an octet, as DPP moves (VALU speed; a general shuffle goes through the LDS crossbar):
// step 0: lane ^ 1 (quad_perm 1,0,3,2), step 1: lane ^ 2 (quad_perm 2,3,0,1), step 2: lane -> 7 - lane (row_half_mirror)
template <int STEP> __device__ __forceinline__ uint32_t oct_partner(uint32_t x) {
    constexpr int CTRL = STEP == 0 ? 0xB1 : (STEP == 1 ? 0x4E : 0x141);
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
}
template <int STEP> __device__ __forceinline__ int64_t oct_partner64(int64_t x) {
    const uint32_t lo = oct_partner<STEP>((uint32_t)(uint64_t)x), hi = oct_partner<STEP>((uint32_t)((uint64_t)x >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void cand_take(Cand3 &c, int64_t sc, int32_t pi, bool flag) {
    if (sc > c.m) {
        c.m = sc, c.f = pi, c.l = flag ? pi : -1;
    } else if (sc == c.m) {
        if (pi < c.f) c.f = pi;
        if (flag && pi > c.l) c.l = pi;
    }
}
__device__ __forceinline__ void cand_merge(Cand3 &a, int64_t m, int32_t f, int32_t l) {
    if (m > a.m) {
        a.m = m, a.f = f, a.l = l;
    } else if (m == a.m) {
        a.f = min(a.f, f);
        a.l = max(a.l, l);
    }
}
__device__ __forceinline__ bool pred_ok(uint16_t vb, uint16_t vd, uint32_t want, uint16_t kd, uint32_t q, bool &flag) {
    if ((vb & 0x10FFu) != want) return false;
    const uint16_t v2d = (vb & 0x4000) ? (uint16_t)(vd + 1) : (uint16_t)0;
    if (v2d != kd) return false;
    const uint32_t v1q = (vb >> 8) & 0xFu;
    if (q >= 3 && v1q == 15) return false; // main.rs:1666-1668
    flag = v1q != 4;
    return true;
}

__device__ __forceinline__ void k_dp_bt_oct(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ run_start,
                                                  const uint32_t *__restrict__ n_runs, Graph g,
                                                  const uint2 *__restrict__ nrec, int64_t *__restrict__ nscore,
                                                  uint32_t *__restrict__ nbesti, uint32_t *__restrict__ n0_besti,
                                                  uint32_t *__restrict__ run_end, int64_t *__restrict__ last_n0_score,
                                                  int64_t *__restrict__ run_gain, uint32_t *__restrict__ emit,
                                                  uint32_t *__restrict__ path_begin, uint64_t *__restrict__ path,
                                                  uint8_t *__restrict__ run_flag,
                                                  const uint32_t *__restrict__ dp_list,
                                                  const uint32_t *__restrict__ n_dp_list) {
    const uint32_t j = threadIdx.x & 7;
    // dp_list: the runs k_dp_bt_short left alone (it ran before this kernel); without it every octet classifies its runs
    // itself from the node offsets (the two kernels then run side by side on two streams)
    const uint32_t nr = dp_list ? *n_dp_list : *n_runs, L = g.L;
    for (uint32_t i = np2_bid * 8 + (threadIdx.x >> 3); i < nr; i += np2_nb * 8) { // (uniform per octet)
        const uint32_t r = dp_list ? dp_list[i] : i;
        const uint32_t a = run_start[r];
        uint32_t o0, o1;
        if (dp_list) {
            o0 = g.node_off[a], o1 = g.node_off[a + 1];
        } else {
            uint32_t off[RW_P + 1];
            load_run_offsets(g.node_off, a, off);
            const uint32_t len = short_run_len(off, a, L);
            if (len < RW_P && run_is_short(len, off[len] - off[0], *g.deep != 0)) continue; // a short run: k_dp_bt_short's
            o0 = off[0], o1 = off[1];
        }
        const uint32_t o_first = o0;
        int64_t cov = g.cov[a];
        uint8_t c2 = a >= 2 ? ref_code(g.refnib, a - 2) : 0, c1 = a >= 1 ? ref_code(g.refnib, a - 1) : 0;
        uint8_t c0 = ref_code(g.refnib, a);
        const uint8_t c3 = a >= 3 ? ref_code(g.refnib, a - 3) : 0;
        uint2 cur = nrec[o0 + j]; // node j of the current position (the array is padded: lanes past n read junk)
        uint32_t prv_key = 0, pv_n = 0;
        int64_t prv_score = 0, cur_score = 0;
        uint16_t pv_b0 = 0, pv_d0 = 0;
        const int64_t base = early_run_base(g, a);
        int64_t pv_s0 = base;
        bool pv_valid = a > 0;
        if (pv_valid) n0_from_codes(a - 1, c3, c2, c1, pv_b0, pv_d0);
        bool done = false, handed_over = false;
        uint32_t p = a;
        for (; p < L; ++p) {
            const uint32_t n = o1 - o0;
            if (n > 8) { // more nodes than lanes: the per-thread kernel redoes this run
                handed_over = true;
                break;
            }
            uint32_t nx_o1 = o1;
            int64_t nx_cov = 0;
            uint8_t nx_c = 0;
            if (p + 1 < L) {
                nx_o1 = g.node_off[p + 2];
                nx_cov = g.cov[p + 1];
                nx_c = ref_code(g.refnib, p + 1);
            }
            const uint2 nxt = nrec[o1 + j];
            uint16_t b0, d0;
            n0_from_codes(p, c2, c1, c0, b0, d0);
            uint32_t e0 = (j < n && node_delta3((uint16_t)cur.x, (uint16_t)(cur.x >> 16)) == 0) ? cur.y : 0u;
            e0 += oct_partner<0>(e0);
            e0 += oct_partner<1>(e0);
            e0 += oct_partner<2>(e0);
            const int64_t cn0 = cov - (int64_t)e0;
            // Every lane scores "its" exception node j and (all of them alike) the contig's node N0(p), scanning the
            // predecessors in the reference's order (main.rs:1664-1674): N0 of the predecessor position first, then its
            // exception nodes, fetched from their lanes one step ahead of their use.  Nodes whose second column lies at
            // p itself (insertion columns) have their predecessors among the earlier nodes of p: they follow, in node
            // order, once the others are done.
            const int64_t cov4 = 4 * cov;
            const uint16_t mkb = (uint16_t)cur.x, mkd = (uint16_t)(cur.x >> 16);
            const bool mine = j < n;
            const bool m_head = ((mkb >> 4) & 0xF) == 15, m_same = (mkb & 0x1000) != 0;
            const int64_t mw = 10 * (int64_t)cur.y - cov4, w0 = 10 * cn0 - cov4;
            const uint32_t mwant = ((mkb >> 4) & 0xFFu) | (((mkb >> 14) & 1u) << 12);
            const uint32_t want0 = ((b0 >> 4) & 0xFFu) | (((b0 >> 14) & 1u) << 12);
            const bool head0 = ((b0 >> 4) & 0xF) == 15; // (N0 is a path start at positions 0 and 1 only)
            int64_t s0_cur = head0 ? w0 : SCORE_NEG, m_score = m_head ? mw : SCORE_NEG;
            uint32_t besti0 = 0, m_besti = 0;
            auto take = [](int64_t sc, bool flag, uint32_t pi, int64_t &score, uint32_t &besti) {
                if (sc > score || (sc == score && flag)) score = sc, besti = pi; // main.rs:1670
            };
            if (pv_valid) {
                bool flag;
                const bool go0 = !head0, gom = mine && !m_head && !m_same;
                if (go0 && pred_ok(pv_b0, pv_d0, want0, d0, p - 1, flag)) take(pv_s0 + w0, flag, 0, s0_cur, besti0);
                if (gom && pred_ok(pv_b0, pv_d0, mwant, mkd, p - 1, flag)) take(pv_s0 + mw, flag, 0, m_score, m_besti);
                uint32_t vk = __shfl(prv_key, 0, 8);
                int64_t vs = __shfl(prv_score, 0, 8);
                for (uint32_t k = 0; k < pv_n; ++k) { // (uniform per octet)
                    const uint32_t nk = __shfl(prv_key, (k + 1) & 7, 8);
                    const int64_t ns = __shfl(prv_score, (k + 1) & 7, 8);
                    const uint16_t vb = (uint16_t)vk, vd = (uint16_t)(vk >> 16);
                    if (go0 && pred_ok(vb, vd, want0, d0, p - 1, flag)) take(vs + w0, flag, k + 1, s0_cur, besti0);
                    if (gom && pred_ok(vb, vd, mwant, mkd, p - 1, flag)) take(vs + mw, flag, k + 1, m_score, m_besti);
                    vk = nk, vs = ns;
                }
            }
            {
                const uint64_t bal = __ballot(mine && m_same && !m_head);
                uint32_t sp = (uint32_t)(bal >> (threadIdx.x & 56u)) & 0xFFu; // this octet's insertion-column nodes
                for (; sp; sp &= sp - 1) { // rare; every lane of the octet computes the same values
                    const uint32_t i = (uint32_t)__builtin_ctz(sp);
                    const uint32_t ikey = __shfl(cur.x, i, 8);
                    const int64_t iw = 10 * (int64_t)__shfl(cur.y, i, 8) - cov4;
                    const uint16_t ikb = (uint16_t)ikey, ikd = (uint16_t)(ikey >> 16);
                    const uint32_t iwant = ((ikb >> 4) & 0xFFu) | (((ikb >> 14) & 1u) << 12);
                    int64_t sc = SCORE_NEG;
                    uint32_t bi = 0;
                    bool flag;
                    if (pred_ok(b0, d0, iwant, ikd, p, flag)) take(s0_cur + iw, flag, 0, sc, bi);
                    for (uint32_t k = 0; k < i; ++k) {
                        const uint32_t vk = __shfl(cur.x, k, 8);
                        const int64_t vs = __shfl(m_score, k, 8);
                        if (pred_ok((uint16_t)vk, (uint16_t)(vk >> 16), iwant, ikd, p, flag)) take(vs + iw, flag, k + 1, sc, bi);
                    }
                    if (j == i) m_score = sc, m_besti = bi;
                }
            }
            cur_score = m_score;
            if (mine) {
                nbesti[o0 + j] = m_besti;
                if (p + 1 == L) nscore[o0 + j] = m_score; // read by k_dp_finish
            }
            if (j == 0) n0_besti[p] = besti0;
            if (n == 0) { // the clean position closing the run
                done = true;
                if (j == 0) {
                    run_end[r] = p - 1;
                    run_gain[r] = s0_cur - base;
                }
                break;
            }
            pv_n = n, pv_b0 = b0, pv_d0 = d0, pv_s0 = s0_cur, pv_valid = true;
            prv_key = cur.x, prv_score = cur_score;
            cur = nxt;
            o0 = o1, o1 = nx_o1, cov = nx_cov;
            c2 = c1, c1 = c0, c0 = nx_c;
        }
        if (j == 0) run_flag[r] = handed_over ? 1 : 0;
        if (handed_over) continue;
        if (!done) { // the run reaches the contig end
            if (j == 0) {
                run_end[r] = L - 1;
                run_gain[r] = -base;
                *last_n0_score = pv_s0;
            }
            continue;
        }
        __threadfence_block(); // the octet's besti stores, read back by the walk
        if (j == 0) emit[a] = bt_walk(g, a, p - 1, n0_besti[p], nbesti, n0_besti, path_begin, path + (size_t)a + o_first);
    }
}

// (grid-stride over a capped grid, see k_dp_bt_short)
__device__ __forceinline__ void k_dp_bt_long(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ run_start,
                                                      const uint32_t *__restrict__ n_runs, Graph g,
                                                      const uint2 *__restrict__ nrec, int64_t *__restrict__ nscore,
                                                      uint32_t *__restrict__ nbesti, uint32_t *__restrict__ n0_besti,
                                                      uint32_t *__restrict__ run_end,
                                                      int64_t *__restrict__ last_n0_score,
                                                      int64_t *__restrict__ run_gain, uint32_t *__restrict__ emit,
                                                      uint32_t *__restrict__ path_begin, uint64_t *__restrict__ path,
                                                      const uint8_t *__restrict__ run_flag,
                                                      const uint32_t *__restrict__ dp_list,
                                                      const uint32_t *__restrict__ n_dp_list) {
    // one 16-byte LDS word per cached node: {key, count, score lo, score hi} (a node is read as a whole: the DP chain
    // is bound by LDS round trips, not by bytes)
    __shared__ uint4 s_node[2 * DP_NR][DP_BLOCK];
    const uint32_t nr = dp_list ? *n_dp_list : *n_runs;
    for (uint32_t i = np2_bid * DP_BLOCK + threadIdx.x; i < nr; i += np2_nb * DP_BLOCK)
        dp_bt_long_run(dp_list ? dp_list[i] : i, run_start, g, nrec, nscore, nbesti, n0_besti, run_end, last_n0_score,
                       run_gain, emit, path_begin, path, run_flag, dp_list != nullptr, s_node);
}

// ------------------------------------------------------------------------------------------
// Short runs: DP and backtrack of a run in one go, entirely out of LDS.
//
// The per-run kernels are latency chains (one lane walks a run position by position; measured ~2.5 us per position with
// everything in LDS, VALU utilisation ~20 %), so what counts is how much of a run's work needs no memory round trip
// at all.  A short run is fetched once with a handful of wide per-lane loads (node offsets, coverage, contig codes,
// node records), scored, walked back and written out as its path slice; scores and best predecessors never leave
// the chip.  Long runs and the run that reaches the contig end belong to k_dp_bt_long.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool dp_bt_short_run(uint32_t r, const uint32_t *__restrict__ run_start, const Graph &g,
                                                    const uint32_t *__restrict__ refw32, uint32_t *__restrict__ run_end,
                                                    int64_t *__restrict__ run_gain, uint32_t *__restrict__ emit, uint32_t *__restrict__ path_begin,
                                                    uint64_t *__restrict__ path, bool deep, uint8_t (*s_off)[64],
                                                    uint16_t (*s_cov)[64], uint2 (*s_ks)[64], uint32_t (*s_cb)[64],
                                                    uint8_t (*s_n0bi)[64]) {
    const uint32_t t = threadIdx.x;
    const uint32_t a = run_start[r], L = g.L;
    // ---- node offsets of positions a .. a + RW_P, run class -----------------------------------------------------------
    uint32_t len, nn, o_base;
    {
        uint32_t off[RW_P + 1];
        load_run_offsets(g.node_off, a, off);
        len = short_run_len(off, a, L);
        o_base = off[0];
        nn = len < RW_P ? off[len] - o_base : 0xFFFFFFFFu;
        if (!run_is_short(len, nn, deep)) return false; // the long-run kernels'
#pragma unroll
        for (uint32_t i = 0; i <= RW_P; ++i) s_off[i][t] = (uint8_t)min(off[i] - o_base, 255u); // (only [0, len] are used)
    }
    // ---- coverage of positions a .. a + len (< 65536: !deep), contig codes of a - 3 .. a + len, the run's node records ---
    {
        const U32x4 v0 = *reinterpret_cast<const U32x4 *>(g.cov + a);
        s_cov[0][t] = (uint16_t)v0.x, s_cov[1][t] = (uint16_t)v0.y, s_cov[2][t] = (uint16_t)v0.z, s_cov[3][t] = (uint16_t)v0.w;
        if (len >= 4) {
            const U32x4 v1 = *reinterpret_cast<const U32x4 *>(g.cov + a + 4);
            s_cov[4][t] = (uint16_t)v1.x, s_cov[5][t] = (uint16_t)v1.y, s_cov[6][t] = (uint16_t)v1.z, s_cov[7][t] = (uint16_t)v1.w;
        }
        if (len >= 8) {
            const U32x4 v2 = *reinterpret_cast<const U32x4 *>(g.cov + a + 8);
            s_cov[8][t] = (uint16_t)v2.x, s_cov[9][t] = (uint16_t)v2.y, s_cov[10][t] = (uint16_t)v2.z, s_cov[11][t] = (uint16_t)v2.w;
            if (len >= 12) s_cov[12][t] = (uint16_t)g.cov[a + 12];
        }
    }
    // contig codes of positions a - 3 .. a + 12 as one 64-bit word (nibble i = position a - 3 + i): three dwords of the
    // nibble-packed contig, shifted into place (plain shifts: a selected-by-index temporary would live in scratch)
    uint64_t cpk;
    {
        const uint32_t cbase = a >= 3 ? ((a - 3) >> 3) : 0;
        const uint64_t lo = (uint64_t)refw32[cbase] | ((uint64_t)refw32[cbase + 1] << 32);
        const uint64_t hi = refw32[cbase + 2];
        if (a >= 3) {
            const uint32_t sh = ((a - 3) & 7) * 4;
            cpk = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
        } else {
            cpk = lo << ((3 - a) * 4);
        }
    }
    auto code_at = [&](uint32_t p) -> uint8_t { return (uint8_t)((cpk >> ((p + 3 - a) * 4)) & 7); }; // a - 3 <= p <= a + 12
    for (uint32_t k = 0; k < nn; k += 2) { // two 8-byte records per load (the node array is padded)
        const U32x4 v = *reinterpret_cast<const U32x4 *>(g.nrec + o_base + k);
        s_ks[k][t] = make_uint2(v.x, 0u);
        s_cb[k][t] = v.y; // count (< 65536), best predecessor << 16 later
        if (k + 1 < RW_N) {
            s_ks[k + 1][t] = make_uint2(v.z, 0u);
            s_cb[k + 1][t] = v.w;
        }
    }
    auto n0_key_at = [&](uint32_t p, uint16_t &b, uint16_t &d) {
        n0_from_codes(p, p >= 2 ? code_at(p - 2) : 0, p >= 1 ? code_at(p - 1) : 0, code_at(p), b, d);
    };
    // ---- DP over positions a .. a + len (the last one is clean: only its N0) ----------------------------------------------
    // Scores are 32-bit and relative to N0(a - 1) (+ early_run_base): the reference's i64 scores (main.rs:1661-1677)
    // minus that node's own score for everything reachable from it, NEG32 + the same sums for what is not — an
    // order-preserving image of the i64 values, since a path inside the run has at most RW_N + RW_P nodes of
    // |10 * count - 4 * coverage| < 2^20 each.
    constexpr int32_t NEG32 = -(1 << 30);
    uint16_t pv_b0 = 0, pv_d0 = 0;
    const int32_t base = (int32_t)early_run_base(g, a); // (0 unless the run starts at position 1 or 2; coverages < 65536)
    int32_t pv_s0 = base;
    bool pv_valid = a > 0;
    if (pv_valid) n0_key_at(a - 1, pv_b0, pv_d0);
    uint32_t pv_k0 = 0, pv_n = 0;
    int32_t s0_cur = 0;
    for (uint32_t st = 0; st <= len; ++st) {
        const uint32_t p = a + st;
        const uint32_t k0 = s_off[st][t], n = s_off[st + 1][t] - k0;
        const int32_t cov = s_cov[st][t];
        uint16_t b0, d0;
        n0_key_at(p, b0, d0);
        uint32_t e0 = 0;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t key = s_ks[k0 + k][t].x;
            if (node_delta3((uint16_t)key, (uint16_t)(key >> 16)) == 0) e0 += s_cb[k0 + k][t] & 0xFFFFu;
        }
        const int32_t cn0 = cov - (int32_t)e0;
        for (uint32_t idx = 0; idx <= n; ++idx) {
            uint16_t kb = b0, kd = d0;
            int32_t cnt = cn0;
            if (idx) {
                const uint32_t key = s_ks[k0 + idx - 1][t].x;
                kb = (uint16_t)key, kd = (uint16_t)(key >> 16), cnt = (int32_t)(s_cb[k0 + idx - 1][t] & 0xFFFFu);
            }
            const int32_t w = 10 * cnt - 4 * cov;
            int32_t score;
            uint32_t besti = 0;
            if (((kb >> 4) & 0xF) == 15) { // (see k_dp_bt_long for the predecessor test)
                score = w;
            } else {
                score = NEG32;
                const bool same_pos = (kb & 0x1000) != 0;
                const uint32_t q = same_pos ? p : p - 1;
                const uint32_t want = ((kb >> 4) & 0xFFu) | (((kb >> 14) & 1u) << 12);
                uint32_t qk0 = 0, qn = 0;
                uint16_t qb0 = 0, qd0 = 0;
                int32_t qs0 = 0;
                bool ok = false;
                if (same_pos) {
                    qk0 = k0, qn = idx, qb0 = b0, qd0 = d0, qs0 = s0_cur, ok = true;
                } else if (pv_valid) {
                    qk0 = pv_k0, qn = 1 + pv_n, qb0 = pv_b0, qd0 = pv_d0, qs0 = pv_s0, ok = true;
                }
                if (ok) {
                    for (uint32_t pi = 0; pi < qn; ++pi) {
                        uint16_t vb = qb0, vd = qd0;
                        int32_t ps = qs0;
                        if (pi) {
                            const uint2 v = s_ks[qk0 + pi - 1][t];
                            vb = (uint16_t)v.x, vd = (uint16_t)(v.x >> 16);
                            ps = (int32_t)v.y;
                        }
                        if ((vb & 0x10FFu) != want) continue;
                        const uint16_t v2d = (vb & 0x4000) ? (uint16_t)(vd + 1) : (uint16_t)0;
                        if (v2d != kd) continue;
                        const uint32_t v1q = (vb >> 8) & 0xFu;
                        if (q >= 3 && v1q == 15) continue; // main.rs:1666-1668
                        const int32_t sc = ps + w;
                        if (sc > score || (sc == score && v1q != 4)) { // main.rs:1670
                            score = sc;
                            besti = pi;
                        }
                    }
                }
            }
            if (idx) {
                s_ks[k0 + idx - 1][t].y = (uint32_t)score;
                s_cb[k0 + idx - 1][t] = (uint32_t)cnt | (besti << 16);
            } else {
                s0_cur = score;
                s_n0bi[st][t] = (uint8_t)besti;
            }
        }
        pv_k0 = k0, pv_n = n, pv_b0 = b0, pv_d0 = d0, pv_s0 = s0_cur, pv_valid = true;
    }
    run_end[r] = a + len - 1;
    run_gain[r] = (int64_t)s0_cur - base; // N0 of the closing clean position, relative to N0(a - 1)
    // ---- backtrack from the closing position's best predecessor (bt_walk) --------------------------------------------------
    uint64_t *out = path + (size_t)a + o_base;
    uint32_t st = len - 1, idx = s_n0bi[len][t], n_out = 0;
    for (;;) {
        const uint32_t p = a + st;
        const uint32_t k0 = s_off[st][t];
        uint16_t kb, kd;
        uint32_t cnt, bi;
        if (idx == 0) {
            n0_key_at(p, kb, kd);
            uint32_t e0 = 0; // count of the contig's own node: coverage minus the exception nodes ending like it
            for (uint32_t k = k0; k < s_off[st + 1][t]; ++k) {
                const uint32_t key = s_ks[k][t].x;
                if (node_delta3((uint16_t)key, (uint16_t)(key >> 16)) == 0) e0 += s_cb[k][t] & 0xFFFFu;
            }
            cnt = (uint32_t)s_cov[st][t] - e0;
            bi = s_n0bi[st][t];
        } else {
            const uint32_t key = s_ks[k0 + idx - 1][t].x, cb = s_cb[k0 + idx - 1][t];
            kb = (uint16_t)key, kd = (uint16_t)(key >> 16);
            cnt = cb & 0xFFFFu;
            bi = cb >> 16;
        }
        const uint8_t k3q = kb & 0xF;
        if (k3q != 4) {
            const int64_t cov = s_cov[st][t];
            const bool lq = (int64_t)cnt * 100 < 95 * cov;
            const uint32_t cls = cov < 2 ? CLS_RESET : (lq ? CLS_LQ : CLS_HQ);
            out[n_out++] = ((uint64_t)p << 32) | ((uint32_t)code_to_ascii(k3q) << 8) | cls;
        }
        if (((kb >> 4) & 0xF) == 15) {
            if (p > 0) atomicMax(path_begin, p);
            break;
        }
        if (!(kb & 0x1000)) { // second column at p - 1
            if (st == 0) break; // left the run: N0(a - 1)
            --st;
        }
        idx = bi;
    }
    emit[a] = n_out;
    return true;
}

// One thread per short run, grid-stride: the launch covers a host-side bound on the number of runs (the record count)
// that is ~4x the real number, and every workgroup — also one that finds nothing to do — costs a dispatch slot with its
// LDS allocation, so the grid is capped and the threads loop instead.
__device__ __forceinline__ void k_dp_bt_short(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ run_start,
                                                    const uint32_t *__restrict__ n_runs, Graph g,
                                                    const uint32_t *__restrict__ refw32, uint32_t *__restrict__ run_end,
                                                    int64_t *__restrict__ run_gain, uint32_t *__restrict__ emit,
                                                    uint32_t *__restrict__ path_begin, uint64_t *__restrict__ path,
                                                    uint32_t *__restrict__ dp_list, uint32_t *__restrict__ n_dp_list) {
    __shared__ uint8_t s_off[RW_P + 1][64]; // node offsets relative to the run's first node (<= RW_N)
    __shared__ uint16_t s_cov[RW_P][64];
    __shared__ uint2 s_ks[RW_N][64];    // {key, score}
    __shared__ uint32_t s_cb[RW_N][64]; // count | best predecessor << 16
    __shared__ uint8_t s_n0bi[RW_P][64];
    const uint32_t nr = *n_runs, lane = threadIdx.x;
    const bool deep = *g.deep != 0;
    for (uint32_t r0 = np2_bid * 64; r0 < nr; r0 += np2_nb * 64) { // (uniform trip count: the ballot below)
        const uint32_t r = r0 + lane;
        const bool left = r < nr && !dp_bt_short_run(r, run_start, g, refw32, run_end, run_gain, emit, path_begin, path, deep,
                                                     s_off, s_cov, s_ks, s_cb, s_n0bi);
        if (dp_list) { // the runs left to the long-run kernels, in no particular order: one reservation per wave
            const uint64_t m = __ballot(left);
            if (m) {
                const uint32_t lead = (uint32_t)__builtin_ctzll(m);
                uint32_t base = 0;
                if (lane == lead) base = atomicAdd(n_dp_list, (uint32_t)__builtin_popcountll(m));
                base = __shfl(base, lead);
                if (left) dp_list[base + (uint32_t)__builtin_popcountll(m & ((1ULL << lane) - 1ULL))] = r;
            }
        }
    }
}

// global best node at L-1 (main.rs:1651,1680): later node wins ties, must reach score >= 0
// (run_a: first position of the run that reaches the contig end.  From position 3 on the run's scores are relative to the
// node left of it — except those of READ-START nodes, whose score is the absolute 10 count - 4 coverage, main.rs:1659-1660:
// a read that starts at the very last position competes with that, not with the path's total added on top)
__device__ uint32_t pick_best(const Graph &g, const int64_t *__restrict__ nscore, int64_t last_n0_score, int64_t total, uint32_t run_a) {
    const uint32_t p = g.L - 1;
    const uint32_t o0 = g.node_off[p], o1 = g.node_off[p + 1];
    if (o1 == o0) return total >= 0 ? 0u : 0xFFFFFFFFu;
    int64_t best = 0;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t idx = 0; idx < 1 + (o1 - o0); ++idx) {
        const int64_t rel = idx ? nscore[o0 + idx - 1] : last_n0_score;
        const bool start_node = idx && run_a >= 3 && ((g.nrec[o0 + idx - 1].x >> 4) & 0xFu) == 15u;
        const int64_t s = (rel <= (SCORE_NEG / 2) || start_node) ? rel : total + rel;
        if (s >= best) {
            best = s;
            bi = idx;
        }
    }
    return bi;
}

// ------------------------------------------------------------------------------------------
// K7: backtrack + consensus emission (generate_cns_from_best_score_lq, main.rs:1555-1637)
// ------------------------------------------------------------------------------------------
// Walks right -> left from node (b, entry_idx) until it leaves [a, b]; returns the number of emitted bases and records
// them, in walk order, as t_pos << 32 | base << 8 | class at path[0..n).  The run's slice of the path buffer starts at
// a + node_off[a]: runs are disjoint in positions and in nodes, and a run emits at most one base per position plus one
// per exception node, so the slices never overlap.
__device__ uint32_t bt_walk(const Graph &g, uint32_t a, uint32_t b, uint32_t entry_idx,
                            const uint32_t *__restrict__ nbesti, const uint32_t *__restrict__ n0_besti,
                            uint32_t *__restrict__ path_begin, uint64_t *__restrict__ path) {
    uint32_t pos = b, idx = entry_idx, n = 0;
    for (;;) {
        uint16_t kb, kd;
        uint32_t cnt, bi;
        const uint32_t o0 = g.node_off[pos];
        if (idx == 0) {
            n0_key(g, pos, kb, kd);
            cnt = n0_count(g, pos);
            bi = n0_besti[pos];
        } else {
            const uint2 rc = g.nrec[o0 + idx - 1];
            kb = (uint16_t)rc.x;
            kd = (uint16_t)(rc.x >> 16);
            cnt = rc.y;
            bi = nbesti[o0 + idx - 1];
        }
        // third column: always at `pos`; second column: at `pos` iff bit 12 (node_decode)
        const uint8_t k3q = kb & 0xF;
        if (k3q != 4) {
            const int64_t cov = g.cov[pos];
            // qv = count * 100 / coverage (integer division); qv < 95 <=> count * 100 < 95 * coverage
            const bool lq = (int64_t)cnt * 100 < 95 * cov;
            const uint32_t cls = cov < 2 ? CLS_RESET : (lq ? CLS_LQ : CLS_HQ);
            path[n] = ((uint64_t)pos << 32) | ((uint32_t)code_to_ascii(k3q) << 8) | cls;
            ++n;
        }
        if (((kb >> 4) & 0xF) == 15) { // second column is a head sentinel: the path starts here
            if (pos > 0) atomicMax(path_begin, pos);
            break;
        }
        const uint32_t np_ = (kb & 0x1000) ? pos : pos - 1;
        if (np_ < a || np_ > pos) break; // left the run (np_ == a-1 -> N0(a-1)); np_ > pos: wrapped
        pos = np_;
        idx = bi;
    }
    return n;
}

// End of the DP stage in one launch.  Every block adds its share of the gains (clean-position partials per contig tile
// from the graph build + the dirty runs') to the absolute best-path score; the last block to finish then does the three
// things that need that total, in order, on one thread:
//  * the best end node at the last position;
//  * the backtrack of the run that reaches the contig end (it exists iff the last position is dirty, and is the last
//    run): it is entered at that best node, so it could not be walked with the others;
//  * positions left of the path's first node emit nothing (start nodes are only accepted at t_pos < 3,
//    main.rs:1666-1668, so at most positions 0..1 are affected); emit[L] = 0 terminates the offset scan.
__device__ __forceinline__ void k_dp_finish(const uint32_t np2_bid, const uint32_t np2_nb, const int64_t *__restrict__ run_gain, const uint32_t *__restrict__ n_runs,
                                                   const long long *__restrict__ tile_gain, uint32_t n_tiles,
                                                   unsigned long long *__restrict__ total_gain,
                                                   uint32_t *__restrict__ blocks_done, Graph g,
                                                   const int64_t *__restrict__ nscore,
                                                   const int64_t *__restrict__ last_n0_score,
                                                   const uint32_t *__restrict__ run_start,
                                                   const uint32_t *__restrict__ nbesti,
                                                   const uint32_t *__restrict__ n0_besti, uint32_t *__restrict__ best_idx,
                                                   uint32_t *__restrict__ emit, uint32_t *__restrict__ path_begin,
                                                   uint64_t *__restrict__ path) {
    long long v = 0;
    const uint32_t stride = np2_nb * blockDim.x, t0 = np2_bid * blockDim.x + threadIdx.x;
    const uint32_t nr = *n_runs;
    for (uint32_t r = t0; r < nr; r += stride) v += run_gain[r];
    for (uint32_t t = t0; t < n_tiles; t += stride) v += tile_gain[t];
    __shared__ long long sm[4];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x) return;
    const long long sum = sm[0] + sm[1] + sm[2] + sm[3];
    if (sum) atomicAdd(total_gain, (unsigned long long)sum);
    __threadfence();
    if (atomicAdd(blocks_done, 1u) != np2_nb - 1) return;
    __threadfence();
    const int64_t total = (int64_t)atomicAdd(total_gain, 0ULL); // (the device-coherent value)
    const uint32_t L = g.L;
    const uint32_t best = pick_best(g, nscore, *last_n0_score, total, nr ? run_start[nr - 1] : 0u);
    *best_idx = best;
    if (best == 0xFFFFFFFFu) {
        // No node at L - 1 reaches a score >= 0: the reference walks back from its DEFAULT Kmer (main.rs:1651: bases = 0,
        // count = 0, besti = 0; main.rs:1564, 1569-1570: an 'A' at L - 1 whose qv is 0 — low quality unless the coverage
        // is below 2 —, then node 0 of position L - 2, i.e. N0(L - 2), and on along the recorded best predecessors).
        const uint64_t tail = ((uint64_t)(L - 1) << 32) | ((uint32_t)'A' << 8) | (g.cov[L - 1] < 2 ? CLS_RESET : CLS_LQ);
        if (nr != 0 && g.node_off[L] != g.node_off[L - 1]) { // the last position is dirty: its run's path starts with the 'A'
            const uint32_t a = run_start[nr - 1];
            uint64_t *sl = path + (size_t)a + g.node_off[a];
            sl[0] = tail;
            emit[a] = 1u + (a + 2 <= L ? bt_walk(g, a, L - 2, 0u, nbesti, n0_besti, path_begin, sl + 1) : 0u);
        } else if (nr != 0 && g.node_off[L - 1] != g.node_off[L - 2]) {
            // L - 1 is clean (k_default_tail rewrites its base), the run ending at L - 2 was entered at N0(L - 1)'s best
            // predecessor: walked again from N0(L - 2).  The earlier walk may have left a path start behind (positions
            // 1 and 2 only, main.rs:1666-1668): the few runs that can set one are walked again as well.
            const uint32_t a = run_start[nr - 1];
            if (a <= 2) {
                *path_begin = 0;
                for (uint32_t r = 0; r + 1 < nr && run_start[r] <= 2; ++r) {
                    const uint32_t ar = run_start[r];
                    uint32_t e = ar;
                    while (g.node_off[e + 2] != g.node_off[e + 1]) ++e; // (the last run follows: e + 1 < L)
                    emit[ar] = bt_walk(g, ar, e, n0_besti[e + 1], nbesti, n0_besti, path_begin, path + (size_t)ar + g.node_off[ar]);
                }
            }
            emit[a] = bt_walk(g, a, L - 2, 0u, nbesti, n0_besti, path_begin, path + (size_t)a + g.node_off[a]);
        }
    } else if (nr != 0 && g.node_off[L] != g.node_off[L - 1]) {
        const uint32_t a = run_start[nr - 1];
        emit[a] = bt_walk(g, a, L - 1, best, nbesti, n0_besti, path_begin, path + (size_t)a + g.node_off[a]);
    }
    const uint32_t pb = atomicMax(path_begin, 0u);
    uint32_t p = 0;
    while (p < pb && p < L) {
        if (!(g.node_off[p + 1] > g.node_off[p])) {
            emit[p] = 0; // clean position before the path start
            ++p;
        } else {
            uint32_t e = p;
            while (e + 1 < L && g.node_off[e + 2] > g.node_off[e + 1]) ++e;
            if (e < pb) emit[p] = 0; // whole dirty run lies before the path start
            p = e + 1;
        }
    }
    emit[L] = 0;
}

// Consensus write-out in two kernels.  k_cns_write: one thread per contig position, clean positions only — each emits
// the contig base at its scanned offset (neighbouring threads write neighbouring consensus indices: a pure stream).
// k_cns_runs: one thread per dirty run copies the run's recorded path (the walk went right -> left) and counts the
// low-quality bases in it (they only come out of dirty runs).  One kernel for both had nearly every wave of 64
// positions wait for its one run-start lane's serial copy (98 us per call on the yeast-sized assembly); split it is
// 46 + 55 us — no faster by itself (the per-run copy is a chain of dependent loads and byte stores), but the runs'
// counts then come in run order, which halves the list kernel.  The scanned per-run counts place the consensus indices of the low-quality
// bases in an ordered list (k_lq_list), which the LQ-region kernels walk instead of keeping one mostly idle thread
// per consensus base.  (A single device counter bumped once per block would serialise: tried.)
__device__ __forceinline__ void k_cns_write(const uint32_t np2_bid, const uint32_t np2_nb, const uint8_t *__restrict__ refnib,
                            const uint8_t *__restrict__ pflag, const uint32_t *__restrict__ emit,
                            const uint32_t *__restrict__ eoff, uint32_t L,
                            uint32_t *__restrict__ cns_pos, uint8_t *__restrict__ cns_base,
                            uint8_t *__restrict__ cns_cls, uint8_t *__restrict__ lq_nothead) {
    const uint32_t p = np2_bid * blockDim.x + threadIdx.x;
    if (p >= L || !emit[p]) return;
    const uint8_t pf = pflag[p];
    if (pf & 1) return; // first position of a dirty run: k_cns_runs
    const uint32_t o0 = eoff[p];
    lq_nothead[o0] = 0; // every consensus index is written exactly once: clears the LQ chain flags for k_lq_scan
    cns_pos[o0] = p;
    cns_base[o0] = code_to_ascii(ref_code(refnib, p));
    cns_cls[o0] = (pf & 2) ? CLS_RESET : CLS_HQ; // count == coverage -> qv = 100
}
// (launched over the host-side bound on the number of runs; slots past the device-side count are cleared for the scan)
__device__ __forceinline__ void k_cns_runs(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ run_start,
                           const uint32_t *__restrict__ n_runs, uint32_t bound, const uint32_t *__restrict__ node_off,
                           const uint32_t *__restrict__ emit, const uint32_t *__restrict__ eoff,
                           const uint64_t *__restrict__ path, uint32_t *__restrict__ cns_pos,
                           uint8_t *__restrict__ cns_base, uint8_t *__restrict__ cns_cls,
                           uint8_t *__restrict__ lq_nothead, uint32_t *__restrict__ lqc) {
    // four lanes per run (a run's path has a handful of entries: one round of loads and stores instead of a chain)
    const uint32_t nr = min(*n_runs, bound), q = threadIdx.x & 3;
    for (uint32_t r = (np2_bid * blockDim.x + threadIdx.x) >> 2; r <= bound; r += (np2_nb * blockDim.x) >> 2) {
        uint32_t n_lq = 0;
        if (r < nr) {
            const uint32_t a = run_start[r];
            const uint32_t e = emit[a];
            if (e) {
                const uint32_t o0 = eoff[a];
                const uint64_t *src = path + (size_t)a + node_off[a];
                for (uint32_t n = q; n < e; n += 4) {
                    const uint64_t w = src[n];
                    const uint32_t o = o0 + e - 1 - n;
                    cns_pos[o] = (uint32_t)(w >> 32);
                    cns_base[o] = (uint8_t)(w >> 8);
                    cns_cls[o] = (uint8_t)w;
                    lq_nothead[o] = 0;
                    n_lq += (uint8_t)w == CLS_LQ ? 1u : 0u;
                }
            }
        }
        n_lq += __shfl_xor(n_lq, 1);
        n_lq += __shfl_xor(n_lq, 2);
        if (q == 0) lqc[r] = n_lq;
    }
}
// The reference's default node at a CLEAN last position (k_dp_finish has the rest): the base written there becomes 'A'
// with qv 0 (main.rs:1564-1575) — the last consensus entry, and the last of the low-quality list if coverage >= 2.
__device__ __forceinline__ void k_default_tail(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ best_idx,
                                               const uint32_t *__restrict__ node_off, const uint8_t *__restrict__ pflag, uint32_t L,
                                               const uint32_t *__restrict__ M_p, uint8_t *__restrict__ cns_base,
                                               uint8_t *__restrict__ cns_cls, uint32_t *__restrict__ lq_list,
                                               uint32_t *__restrict__ n_lq, uint32_t cap, uint32_t *__restrict__ err) {
    if (threadIdx.x || np2_bid) return;
    if (*best_idx != 0xFFFFFFFFu || node_off[L] != node_off[L - 1]) return;
    const uint32_t M = *M_p;
    if (!M) return;
    cns_base[M - 1] = (uint8_t)'A';
    const bool lq = !(pflag[L - 1] & 2);
    cns_cls[M - 1] = lq ? CLS_LQ : CLS_RESET;
    if (lq) {
        const uint32_t n = *n_lq;
        if (n >= cap) atomicOr(err, LQ_LIST_ERR);
        else lq_list[n] = M - 1, *n_lq = n + 1;
    }
}
// the consensus indices of the low-quality bases, ascending (runs are in position order): one thread per run that has any
__device__ __forceinline__ void k_lq_list(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ run_start,
                          const uint32_t *__restrict__ n_runs, uint32_t bound, const uint32_t *__restrict__ node_off,
                          const uint32_t *__restrict__ emit, const uint32_t *__restrict__ eoff,
                          const uint64_t *__restrict__ path, const uint32_t *__restrict__ lqoff, uint32_t cap,
                          uint32_t *__restrict__ lq_list, uint32_t *__restrict__ err) {
    const uint32_t nr = min(*n_runs, bound);
    for (uint32_t r = np2_bid * blockDim.x + threadIdx.x; r < nr; r += np2_nb * blockDim.x) {
        uint32_t k = lqoff[r];
        const uint32_t c = lqoff[r + 1] - k;
        if (!c) continue;
        if (k + c > cap) {
            atomicOr(err, LQ_LIST_ERR);
            continue;
        }
        const uint32_t a = run_start[r];
        const uint32_t e = emit[a], o0 = eoff[a];
        const uint64_t *src = path + (size_t)a + node_off[a];
        for (uint32_t n = e; n-- > 0;) // path entry n sits at consensus index o0 + e - 1 - n: ascending indices
            if ((uint8_t)src[n] == CLS_LQ) lq_list[k++] = o0 + e - 1 - n;
    }
}

// ------------------------------------------------------------------------------------------
// K8: LQ-region detection — the right-to-left state machine of main.rs:1586-1625, evaluated in
// parallel.  Indices `p` below are the reference's emission indices (p = M-1-i).
// ------------------------------------------------------------------------------------------
enum : uint8_t { LQK_LINK = 0, LQK_CLOSE = 1, LQK_RESET = 2, LQK_OPEN = 3 };

// one thread per low-quality base (lq_list: their consensus indices, ascending; the count lives on the device)
__device__ __forceinline__ void k_lq_scan(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ cns_pos, const uint8_t *__restrict__ cns_base,
                          const uint8_t *__restrict__ cns_cls, const uint32_t *__restrict__ M_p,
                          const uint32_t *__restrict__ lq_list, const uint32_t *__restrict__ n_lq_p,
                          uint32_t lq_cap, uint8_t *__restrict__ lq_kind, uint32_t *__restrict__ lq_next,
                          uint8_t *__restrict__ lq_nothead, uint32_t *__restrict__ hbits, uint32_t n_hwords) {
    const uint32_t M = *M_p, n_lq = min(*n_lq_p, lq_cap);
    // (the head bitmap k_lq_region marks in, cleared here: a fill launch of its own less per pass)
    for (uint32_t w = np2_bid * blockDim.x + threadIdx.x; w < n_hwords; w += np2_nb * blockDim.x) hbits[w] = 0;
    for (uint32_t j = np2_bid * blockDim.x + threadIdx.x; j < n_lq; j += np2_nb * blockDim.x) {
        const uint32_t i = lq_list[j];
        const uint32_t p = M - 1 - i;
        uint8_t kind = LQK_OPEN;
        uint32_t pp = p + 1;
        bool resolved = false;
        if (i >= 8) {
            // the next eight emission indices (consensus indices i-1 .. i-8) out of one round of loads: the walk below
            // usually ends within six steps, each of which would otherwise be a dependent load or three
            uint64_t c8, b8;
            uint32_t ps[8];
            __builtin_memcpy(&c8, cns_cls + i - 8, 8); // byte k <-> consensus index i - 8 + k
            __builtin_memcpy(&b8, cns_base + i - 8, 8);
            __builtin_memcpy(ps, cns_pos + i - 8, 32);
#pragma unroll
            for (uint32_t st = 1; st <= 8; ++st) {
                if (resolved) continue;
                const uint32_t w = 8 - st; // window slot of consensus index i - st
                const uint8_t cl = (uint8_t)(c8 >> (8 * w));
                if (cl == CLS_RESET) {
                    kind = LQK_RESET, pp = p + st, resolved = true;
                } else if (cl == CLS_LQ) {
                    kind = LQK_LINK, pp = p + st, resolved = true;
                } else if (st > 4) {
                    const uint32_t w1 = w + 1 <= 7 ? w + 1 : 7, w2 = w + 2 <= 7 ? w + 2 : 7; // (st > 4: w + 2 <= 5)
                    if (ps[w1] != ps[w2] && (uint8_t)(b8 >> (8 * w1)) != (uint8_t)(b8 >> (8 * w2)))
                        kind = LQK_CLOSE, pp = p + st, resolved = true;
                }
            }
            if (!resolved) pp = p + 9;
        }
        for (; !resolved && pp < M; ++pp) {
            const uint32_t ii = M - 1 - pp;
            const uint8_t cl = cns_cls[ii];
            if (cl == CLS_RESET) {
                kind = LQK_RESET;
                break;
            }
            if (cl == CLS_LQ) {
                kind = LQK_LINK;
                break;
            }
            if (pp - p > 4) { // p - lq_e > 2*lq_min_length, c(pp-1) vs c(pp-2) (main.rs:1596-1598)
                const uint32_t i1 = ii + 1, i2 = ii + 2;
                if (cns_pos[i1] != cns_pos[i2] && cns_base[i1] != cns_base[i2]) {
                    kind = LQK_CLOSE;
                    break;
                }
            }
        }
        lq_kind[p] = kind;
        lq_next[p] = pp;
        if (kind == LQK_LINK) lq_nothead[pp] = 1;
    }
}

// raw regions, one per chain head: bounds stored under the head's emission index p, the head marked in a bitmap over
// the emission indices — the regions are numbered by ascending p (right -> left), i.e. by the rank of their bit
__device__ __forceinline__ void k_lq_region(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ cns_pos, const uint8_t *__restrict__ cns_base,
                            const uint32_t *__restrict__ M_p, const uint32_t *__restrict__ lq_list,
                            const uint32_t *__restrict__ n_lq_p, uint32_t lq_cap,
                            const uint8_t *__restrict__ lq_kind, const uint32_t *__restrict__ lq_next,
                            const uint8_t *__restrict__ lq_nothead, uint32_t *__restrict__ hbits,
                            uint32_t *__restrict__ rstart, uint32_t *__restrict__ rend) {
    const uint32_t M = *M_p, n_lq = min(*n_lq_p, lq_cap);
    for (uint32_t j = np2_bid * blockDim.x + threadIdx.x; j < n_lq; j += np2_nb * blockDim.x) {
        const uint32_t p = M - 1 - lq_list[j];
        if (lq_nothead[p]) continue;
        uint32_t q = p;
        while (lq_kind[q] == LQK_LINK) q = lq_next[q];
        if (lq_kind[q] != LQK_CLOSE) continue;
        const uint32_t pe = lq_next[q];
        const uint32_t lq_e = pe - 2;
        uint32_t lq_s = p > 2 ? p - 2 : 1;
#define CP(x) cns_pos[M - 1 - (x)]
#define CB(x) cns_base[M - 1 - (x)]
        // (the walk over a homopolymer / an insertion column's bases: eight steps out of one round of loads instead of a
        // dependent load or two per step — emission index x is consensus index M - 1 - x, the walk goes up the consensus)
        for (;;) {
            const uint32_t i = M - 1 - lq_s; // consensus index of lq_s; lq_s - k is i + k
            if (lq_s <= 1) break;
            if (lq_s >= 9 && i + 8 < M) {
                uint32_t ps[9];
                uint8_t bs[9];
                __builtin_memcpy(ps, cns_pos + i, 36);
                __builtin_memcpy(bs, cns_base + i, 9);
                uint32_t k = 0;
                while (k < 8 && (ps[k + 1] == ps[k] || bs[k + 1] == bs[k])) ++k; // (lq_s - k > 1 throughout: lq_s >= 9)
                lq_s -= k;
                if (k < 8) break;
            } else {
                if (CP(lq_s - 1) == CP(lq_s) || CB(lq_s - 1) == CB(lq_s)) --lq_s; else break;
            }
        }
        rend[p] = CP(lq_s);
        rstart[p] = CP(lq_e);
#undef CP
#undef CB
        atomicOr(&hbits[p >> 5], 1u << (p & 31));
    }
}
// heads per bitmap word (scanned next), and the regions written out in bit order
__device__ __forceinline__ void k_lq_bits_count(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ hbits, uint32_t n_words,
                                uint32_t *__restrict__ wcnt) {
    const uint32_t w = np2_bid * blockDim.x + threadIdx.x;
    if (w <= n_words) wcnt[w] = w < n_words ? (uint32_t)__builtin_popcount(hbits[w]) : 0u;
}
__device__ __forceinline__ void k_scatter_regions(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ hbits, uint32_t n_words,
                                  const uint32_t *__restrict__ woff, const uint32_t *__restrict__ rstart,
                                  const uint32_t *__restrict__ rend, uint32_t *__restrict__ raw_start,
                                  uint32_t *__restrict__ raw_end, uint32_t *__restrict__ n_raw) {
    const uint32_t w = np2_bid * blockDim.x + threadIdx.x;
    if (w == 0) *n_raw = woff[n_words];
    if (w >= n_words) return;
    uint32_t bits = hbits[w], k = woff[w];
    for (; bits; bits &= bits - 1, ++k) {
        const uint32_t p = (w << 5) + (uint32_t)__builtin_ctz(bits);
        raw_start[k] = rstart[p];
        raw_end[k] = rend[p];
    }
}

// merge rule main.rs:1613-1615: region j merges into j-1 iff end_j >= start_{j-1}
__device__ __forceinline__ void k_lq_merge_flag(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ raw_start, const uint32_t *__restrict__ raw_end,
                                const uint32_t *__restrict__ n_raw, uint32_t *__restrict__ headflag) {
    const uint32_t n = *n_raw; // on the device: grid-stride over whatever it turns out to be
    for (uint32_t j = np2_bid * blockDim.x + threadIdx.x; j < n; j += np2_nb * blockDim.x)
        headflag[j] = !(j >= 1 && raw_end[j] >= raw_start[j - 1]);
}
// ... the flags and their exclusive scan in one single-block kernel (head j's merged region is number hidx[j])
__device__ __forceinline__ void k_lq_merge_scan(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ raw_start, const uint32_t *__restrict__ raw_end,
                                                const uint32_t *__restrict__ n_raw, uint32_t n_host, uint32_t *__restrict__ headflag,
                                                uint32_t *__restrict__ hidx) {
    __shared__ uint32_t sh[16];
    const uint32_t n = min(*n_raw, n_host);
    block_scan_array<OpAdd>(
        n, sh, [&](uint32_t j) { return (j >= 1 && raw_end[j] >= raw_start[j - 1]) ? 0u : 1u; },
        [&](uint32_t j, uint32_t pre, uint32_t v) {
            hidx[j] = pre;
            headflag[j] = v;
        });
}
__device__ __forceinline__ void k_lq_merge_write(const uint32_t np2_bid, const uint32_t np2_nb, const uint32_t *__restrict__ raw_start, const uint32_t *__restrict__ raw_end,
                                 const uint32_t *__restrict__ n_raw, const uint32_t *__restrict__ headflag,
                                 const uint32_t *__restrict__ hidx, uint32_t *__restrict__ lq_start,
                                 uint32_t *__restrict__ lq_end, uint32_t *__restrict__ n_reg) {
    const uint32_t n = *n_raw;
    for (uint32_t j = np2_bid * blockDim.x + threadIdx.x; j < n; j += np2_nb * blockDim.x) {
        if (headflag[j]) {
            uint32_t l = j;
            while (l + 1 < n && !headflag[l + 1]) ++l;
            lq_end[hidx[j]] = raw_end[j];
            lq_start[hidx[j]] = raw_start[l];
        }
        if (j == n - 1) *n_reg = hidx[j] + headflag[j];
    }
}

// ------------------------------------------------------------------------------------------
// K9: candidate extraction (generate_lqseqs_from_tags_kmer part 1, main.rs:1439-1523)
// regions are indexed right -> left (index 0 = rightmost), starts/ends strictly decreasing.
// ------------------------------------------------------------------------------------------
// number of entries of a descending array that are >= v  /  > v
__device__ __forceinline__ uint32_t count_ge_desc(const uint32_t *a, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (a[mid] >= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint32_t count_gt_desc(const uint32_t *a, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (a[mid] > v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// the cursor `s` of main.rs:1446-1448 is a running minimum over live reads of
// m(r) = max(0, #regions with start >= aln_t_s - 1)
__device__ __forceinline__ void k_read_m(const uint32_t np2_bid, const uint32_t np2_nb, const np2_read_t *__restrict__ reads, uint32_t R, const uint8_t *__restrict__ alive,
                         const uint32_t *__restrict__ lq_start, uint32_t n_reg, int32_t *__restrict__ mval) {
    uint32_t r = np2_bid * blockDim.x + threadIdx.x;
    if (r >= R) return;
    if (!alive[r]) {
        mval[r] = 0x7FFFFFFF;
        return;
    }
    const uint32_t c = count_ge_desc(lq_start, n_reg, reads[r].aln_t_s);
    mval[r] = c ? (int32_t)(c - 1) : 0;
}

__device__ __forceinline__ void k_pair_count(const uint32_t np2_bid, const uint32_t np2_nb, const np2_read_t *__restrict__ reads, uint32_t R, const uint8_t *__restrict__ alive,
                             const uint32_t *__restrict__ lq_start, const uint32_t *__restrict__ lq_end,
                             uint32_t n_reg, const int32_t *__restrict__ smin, uint32_t *__restrict__ pj,
                             uint32_t *__restrict__ pcount, const uint64_t *__restrict__ ck_off,
                             ReadInfo *__restrict__ rinfo) {
    uint32_t r = np2_bid * blockDim.x + threadIdx.x;
    if (r >= R) return;
    uint32_t cnt = 0, j = 0;
    const np2_read_t rd = reads[r];
    if (alive[r]) {
        const uint32_t s = min((uint32_t)smin[r], n_reg - 1);
        const uint32_t ts = rd.aln_t_s, te = rd.aln_t_e;
        if (!(lq_start[s] < ts || lq_end[s] > te)) {
            j = count_gt_desc(lq_end, n_reg, te); // main.rs:1454-1460
            cnt = s - j + 1;
        }
    }
    pj[r] = j;
    pcount[r] = cnt;
    rinfo[r] = ReadInfo{rd.aln_t_s, rd.n_cols, (uint32_t)(rd.nib_off >> 4), (uint32_t)ck_off[r], rd.aln_t_e, j, cnt, 0u};
}

// ------------------------------------------------------------------------------------------
// K10/K11: HBM-resident yak table.  1024 sub-tables (one per file bucket, x & 1023), each
// open-addressed with linear probing on (x >> 10); the slot holds the file word verbatim
// ((x >> 10) << 10 | count, kmer.rs:52-58).  EMPTY = ~0 (a file word never has its top bits set).
// ------------------------------------------------------------------------------------------
static constexpr uint64_t YAK_EMPTY = ~0ULL;

__global__ void k_yak_insert(const uint64_t *__restrict__ words, const uint64_t *__restrict__ bucket_off,
                             uint32_t n_buckets, uint64_t *__restrict__ table, uint32_t cap_log2,
                             uint32_t *__restrict__ dup_flag, uint32_t gap) {
    // gap = words between the end of one bucket's words and the start of the next one's: 0 for the boundary form
    // (np2_yak_t), 1 when `words` is the dump file itself (each bucket preceded by its 8-byte header, kmer.rs:143-147)
    const uint32_t b = blockIdx.y;
    if (b >= n_buckets) return;
    const uint64_t n = bucket_off[b + 1] - bucket_off[b] - gap;
    const uint64_t capm = (1ULL << cap_log2) - 1;
    uint64_t *tb = table + ((uint64_t)b << cap_log2);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w = words[bucket_off[b] + i];
        const uint64_t key = w >> 10;
        uint64_t s = key & capm;
        for (;;) {
            const unsigned long long old = atomicCAS((unsigned long long *)&tb[s], (unsigned long long)YAK_EMPTY,
                                                     (unsigned long long)w);
            if (old == YAK_EMPTY) break;
            if ((old >> 10) == key) {
                atomicOr(dup_flag, 1u);
                break;
            }
            s = (s + 1) & capm;
        }
    }
}

// A dump that repeats a key inside a bucket (yak never writes one): every word keeps a slot of its own and `ord` holds
// its index in the bucket's file order.  retrieve_kmers (kmer.rs:148-167) streams the file and REPLACES the candidate's
// entry with every word that passes `count >= min_count`, so the last such word in file order is what get() returns.
__global__ void k_yak_insert_dup(const uint64_t *__restrict__ words, const uint64_t *__restrict__ bucket_off,
                                 uint32_t n_buckets, uint64_t *__restrict__ table, uint32_t cap_log2,
                                 uint32_t *__restrict__ ord, uint32_t gap) {
    const uint32_t b = blockIdx.y;
    if (b >= n_buckets) return;
    const uint64_t n = bucket_off[b + 1] - bucket_off[b] - gap;
    const uint64_t capm = (1ULL << cap_log2) - 1;
    uint64_t *tb = table + ((uint64_t)b << cap_log2);
    uint32_t *ob = ord + ((uint64_t)b << cap_log2);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w = words[bucket_off[b] + i];
        uint64_t s = (w >> 10) & capm;
        while (atomicCAS((unsigned long long *)&tb[s], (unsigned long long)YAK_EMPTY, (unsigned long long)w) != YAK_EMPTY)
            s = (s + 1) & capm;
        ob[s] = (uint32_t)i; // (read by later launches only)
    }
}

__device__ __forceinline__ uint16_t yak_get(const YakDev &y, uint64_t x, uint16_t min_count) {
    // KmerInfo::get after retrieve_kmers(min_count) (kmer.rs:123-125,160-166), unwrap_or(0)
    const uint64_t capm = (1ULL << y.cap_log2) - 1;
    const uint64_t *tb = y.table + ((x & 1023) << y.cap_log2);
    const uint64_t key = x >> 10;
    uint64_t s = key & capm;
    if (y.ord) { // repeated keys: the whole probe cluster, last passing word in file order
        const uint32_t *ob = y.ord + ((x & 1023) << y.cap_log2);
        uint16_t c = 0;
        int64_t at = -1;
        for (;;) {
            const uint64_t w = tb[s];
            if (w == YAK_EMPTY) return c;
            if ((w >> 10) == key && (uint16_t)(w & 1023) >= min_count && (int64_t)ob[s] > at) at = ob[s], c = (uint16_t)(w & 1023);
            s = (s + 1) & capm;
        }
    }
    for (;;) {
        const uint64_t w = tb[s];
        if (w == YAK_EMPTY) return 0;
        if ((w >> 10) == key) {
            const uint16_t c = (uint16_t)(w & 1023);
            return c >= min_count ? c : 0;
        }
        s = (s + 1) & capm;
    }
}

__device__ __forceinline__ void k_lookup(const uint32_t np2_bid, const uint32_t np2_nb, YakDev y, const uint64_t *__restrict__ hashes, uint64_t n, uint16_t min_count,
                         uint16_t *__restrict__ out) {
    uint64_t i = (uint64_t)np2_bid * blockDim.x + threadIdx.x;
    if (i < n) out[i] = yak_get(y, hashes[i], min_count);
}

// one wavefront per string: lanes own k-mer start offsets; min-count over all valid k-mers
// (iter2kmer kmer.rs:255-287: a k-mer exists where the last k characters are all ACGT)
// min count over the k-mers of a string of arbitrary ASCII (np2_score_strings: the caller's bytes go through SEQ_NUM like
// the reference's, kmer.rs:11-22), one lane per k-mer start, a k-step byte loop per lane
__device__ __forceinline__ uint16_t wave_score_string_any(const YakDev &y, const uint8_t *__restrict__ s, uint32_t len,
                                                          uint16_t min_count) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t k = y.k;
    const uint64_t mask = (1ULL << (2 * (uint64_t)k)) - 1, shift = 2 * ((uint64_t)k - 1);
    uint32_t mn = 0xFFFFFFFFu;
    if (len >= k) {
        for (uint32_t st = lane; st + k <= len; st += 64) {
            uint64_t fw = 0, rv = 0;
            bool ok = true;
            for (uint32_t j = 0; j < k; ++j) {
                const uint64_t c = ascii_to_code(s[st + j]);
                if (c >= 4) {
                    ok = false;
                    break;
                }
                fw = ((fw << 2) | c) & mask;
                rv = (rv >> 2) | ((3ULL ^ c) << shift);
            }
            if (ok) mn = min(mn, (uint32_t)yak_get(y, yak_hash64(fw < rv ? fw : rv, mask), min_count));
        }
    }
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor(mn, o));
    return mn == 0xFFFFFFFFu ? (uint16_t)0 : (uint16_t)mn;
}
// min count over the k-mers of a string, one lane per k-mer start (iter2kmer, kmer.rs:255-314).  The strings scored here
// are written by this library from 3-bit codes (code_to_ascii: "ACGT-NM", upper case), so a lane builds its k-mer
// from five unaligned 8-byte loads instead of a k-step byte loop: bit 3 of a byte singles out '-', 'N' and 'M' (the
// k-mer is skipped: code >= 4 resets the rolling k-mer), ((c >> 1) ^ (c >> 2)) & 3 is the 2-bit code of A / C / G / T.
__device__ __forceinline__ uint16_t wave_score_string(const YakDev &y, const uint8_t *__restrict__ s, uint32_t len,
                                                      uint16_t min_count) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t k = y.k;
    const uint64_t mask = (1ULL << (2 * (uint64_t)k)) - 1;
    uint32_t mn = 0xFFFFFFFFu;
    if (len >= k) {
        for (uint32_t st = lane; st + k <= len; st += 64) {
            uint64_t codes = 0, bad = 0; // 2-bit code of byte j at bits 2j; bit j of `bad`: byte j is not a base
#pragma unroll
            for (uint32_t wd = 0; wd < 4; ++wd) { // 32 bytes cover k <= 31 (the pool is padded past its end)
                uint64_t x;
                __builtin_memcpy(&x, s + st + 8 * wd, 8);
                uint64_t c = ((x >> 1) ^ (x >> 2)) & 0x0303030303030303ULL;
                c = (c | (c >> 6)) & 0x000F000F000F000FULL;
                c = (c | (c >> 12)) & 0x000000FF000000FFULL;
                c = (c | (c >> 24)) & 0xFFFFULL;
                codes |= c << (16 * wd);
                uint64_t b = (x >> 3) & 0x0101010101010101ULL;
                b = (b | (b >> 7)) & 0x0003000300030003ULL;
                b = (b | (b >> 14)) & 0x0000000F0000000FULL;
                b = (b | (b >> 28)) & 0xFFULL;
                bad |= b << (8 * wd);
            }
            if ((bad & ((1ULL << k) - 1ULL)) == 0) {
                codes &= mask;
                uint64_t rb = __builtin_bitreverse64(codes);
                rb = ((rb >> 1) & 0x5555555555555555ULL) | ((rb & 0x5555555555555555ULL) << 1); // pairs back in bit order
                const uint64_t fw = rb >> (64 - 2 * k), rv = ~codes & mask;
                mn = min(mn, (uint32_t)yak_get(y, yak_hash64(fw < rv ? fw : rv, mask), min_count));
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor(mn, o));
    return mn == 0xFFFFFFFFu ? (uint16_t)0 : (uint16_t)mn;
}

__device__ __forceinline__ void k_score_strings(const uint32_t np2_bid, const uint32_t np2_nb, YakDev y, const uint8_t *__restrict__ strs, const uint64_t *__restrict__ off,
                                uint64_t n, uint16_t min_count, uint16_t *__restrict__ out, uint32_t own_strings) {
    // (one string per wavefront: the index is uniform, said so the loads of its offsets are scalar)
    const uint64_t w = ((uint64_t)np2_bid * (blockDim.x >> 6)) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (w >= n) return;
    // own_strings: built by this library from 3-bit codes (and padded by 32 bytes), not handed in by a caller
    const uint16_t sc = own_strings ? wave_score_string(y, strs + off[w], (uint32_t)(off[w + 1] - off[w]), min_count)
                                    : wave_score_string_any(y, strs + off[w], (uint32_t)(off[w + 1] - off[w]), min_count);
    if ((threadIdx.x & 63) == 0) out[w] = sc;
}

// retrieve_kmer_count (main.rs:740-778): len > k -> min over the candidate's own k-mers (rare: one wave
// each, second kernel), else the pre-hashed first k-mer (one lookup, thread per candidate), else 0
__device__ __forceinline__ void k_cand_score(const uint32_t np2_bid, const uint32_t np2_nb, YakDev y, const uint32_t *__restrict__ cand_seq_off, const uint64_t *__restrict__ cand_kmer,
                             const uint32_t *__restrict__ n_cand_p, uint16_t min_count,
                             uint16_t *__restrict__ kscore, uint32_t *__restrict__ long_list,
                             uint32_t *__restrict__ n_long) {
    const uint32_t c = np2_bid * blockDim.x + threadIdx.x;
    const bool live = c < *n_cand_p; // the candidate count lives on the device; the launch covers its bound
    const uint32_t len = live ? cand_seq_off[c + 1] - cand_seq_off[c] : 0u;
    const bool is_long = live && len > y.k;
    // one reservation per wave in the list of long candidates (a same-address atomic per candidate serialises at ~10 ns each)
    const uint64_t m = __ballot(is_long);
    if (m) {
        const uint32_t lane = threadIdx.x & 63, lead = (uint32_t)__builtin_ctzll(m);
        uint32_t base = 0;
        if (lane == lead) base = atomicAdd(n_long, (uint32_t)__builtin_popcountll(m));
        base = __shfl(base, lead);
        if (is_long) long_list[base + (uint32_t)__builtin_popcountll(m & ((1ULL << lane) - 1ULL))] = c;
    }
    if (!live) return;
    uint16_t sc = 0;
    if (!is_long) {
        const uint64_t km = cand_kmer[c];
        if (km != INVALID_KMER) sc = yak_get(y, km, min_count);
    }
    kscore[c] = sc;
}
__device__ __forceinline__ void k_cand_score_long(const uint32_t np2_bid, const uint32_t np2_nb, YakDev y, const uint32_t *__restrict__ cand_seq_off,
                                  const uint8_t *__restrict__ cand_seq, const uint32_t *__restrict__ long_list,
                                  const uint32_t *__restrict__ n_long, uint16_t min_count,
                                  uint16_t *__restrict__ kscore) {
    const uint32_t nl = *n_long;
    for (uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)((np2_bid * blockDim.x + threadIdx.x) >> 6)); w < nl; w += (np2_nb * blockDim.x) >> 6) {
        const uint32_t c = long_list[w];
        const uint16_t sc = wave_score_string(y, cand_seq + cand_seq_off[c], cand_seq_off[c + 1] - cand_seq_off[c],
                                              min_count);
        if ((threadIdx.x & 63) == 0) kscore[c] = sc;
    }
}

} // namespace np2

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
namespace np2 {
static inline dim3 grid1(uint64_t n, uint32_t bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

void launch_encode_ref(hipStream_t s, const uint8_t *read0, uint32_t L, uint8_t *refnib, uint32_t nbytes,
                       uint32_t stride, uint32_t *err) {
    NP2_LAUNCH(k_encode_ref, grid1(nbytes), 256, s, read0, L, refnib, nbytes, stride, err);
}
void launch_post(hipStream_t s, uint32_t *scal, uint32_t n_scal, uint32_t *mbox, uint32_t seq, uint32_t *d0,
                 const uint32_t *s0, uint32_t *d1, const uint32_t *s1, uint32_t *d2, const uint32_t *s2, uint32_t *d3,
                 const uint32_t *s3, const uint32_t *ends_of, uint32_t *ends_dst) {
    NP2_LAUNCH(k_post, dim3(1), 64, s, scal, n_scal, mbox, seq, d0, s0, d1, s1, d2, s2, d3, s3, ends_of, ends_dst);
}
void launch_fill(hipStream_t s, uint8_t *p, uint64_t bytes, uint8_t byte) {
    if (bytes) NP2_LAUNCH(k_fill, grid1((bytes + 15) / 16), 256, s, p, bytes, 0x01010101u * byte);
}
void launch_copy(hipStream_t s, uint8_t *dst, const uint8_t *src, uint64_t bytes) {
    if (bytes) NP2_LAUNCH(k_copy, grid1((bytes + 15) / 16), 256, s, dst, src, bytes);
}
void launch_copy_len(hipStream_t s, uint8_t *dst, const uint8_t *src, const uint32_t *n_dev, uint32_t elem, uint64_t cap_bytes) {
    if (cap_bytes) NP2_LAUNCH(k_copy_len, grid1((cap_bytes + 15) / 16), 256, s, dst, src, n_dev, elem, cap_bytes);
}
void launch_copy_counted(hipStream_t s, uint32_t *dst, const uint32_t *src, const uint32_t *n_dev, uint32_t cap) {
    if (cap) NP2_LAUNCH(k_copy_counted, dim3(std::max<uint32_t>(1, std::min<uint32_t>((cap + 1023) / 1024, 2048))), 256, s, dst, src, n_dev, cap);
}
void launch_init_alive(hipStream_t s, const np2_read_t *reads, uint32_t R, uint8_t *alive) {
    NP2_LAUNCH(k_init_alive, grid1(R), 256, s, reads, R, alive);
}
void launch_kill_flagged(hipStream_t s, const uint8_t *flag, uint32_t n, uint8_t *alive) {
    if (n) NP2_LAUNCH(k_kill_flagged, grid1(n), 256, s, flag, n, alive);
}
void launch_kill_reads(hipStream_t s, const uint32_t *ids, uint32_t n, uint8_t *alive) {
    if (n) NP2_LAUNCH(k_kill_reads, grid1(n), 256, s, ids, n, alive);
}
void launch_revive_reads(hipStream_t s, const uint32_t *ids, uint32_t n, uint8_t *alive) {
    if (n) NP2_LAUNCH(k_revive_reads, grid1(n), 256, s, ids, n, alive);
}
static Graph mk_graph(const GraphPtrs &gp) { return Graph{gp.refnib, gp.node_off, gp.nd, gp.cov, gp.L, gp.nrec, gp.deep}; } // (pflag: write-out only)

void launch_dp_short(hipStream_t s, const GraphPtrs &gp, const void *refw, const uint32_t *run_start,
                     const uint32_t *n_runs, uint32_t max_runs, uint32_t *run_end, int64_t *run_gain, uint32_t *emit,
                     uint32_t *path_begin, uint64_t *path, uint32_t *dp_list, uint32_t *n_dp_list) {
    if (max_runs)
        NP2_LAUNCH(k_dp_bt_short, dim3(std::min<uint32_t>((max_runs + 63) / 64, DP_GRID_CAP)), 64, s, run_start, n_runs, mk_graph(gp), (const uint32_t *)refw, run_end, run_gain, emit, path_begin, path, dp_list, n_dp_list);
}
void launch_dp_long(hipStream_t s, const GraphPtrs &gp, const uint32_t *run_start, const uint32_t *n_runs,
                    uint32_t max_runs, const uint2 *nrec, int64_t *nscore, uint32_t *nbesti, uint32_t *n0_besti,
                    uint32_t *run_end, int64_t *last_n0_score, int64_t *run_gain, uint32_t *emit, uint32_t *path_begin,
                    uint64_t *path, uint8_t *run_flag, const uint32_t *dp_list, const uint32_t *n_dp_list) {
    if (!max_runs) return;
    // with the short kernel's list the grids only have to keep the chip busy (grid-stride over the listed runs)
    const uint32_t oct_cap = dp_list ? 2048u : 4 * DP_GRID_CAP, long_cap = dp_list ? 512u : DP_GRID_CAP;
    NP2_LAUNCH(k_dp_bt_oct, dim3(std::min<uint32_t>((max_runs + 7) / 8, oct_cap)), 64, s, run_start, n_runs, mk_graph(gp), nrec, nscore, nbesti, n0_besti, run_end, last_n0_score, run_gain, emit, path_begin, path, run_flag, dp_list, n_dp_list);
    // runs with a position of more than 8 exception nodes (deep pileups): the per-thread kernel
    NP2_LAUNCH(k_dp_bt_long, dim3(std::min<uint32_t>((max_runs + DP_BLOCK - 1) / DP_BLOCK, long_cap)), DP_BLOCK, s, run_start, n_runs, mk_graph(gp), nrec, nscore, nbesti, n0_besti, run_end, last_n0_score, run_gain, emit, path_begin, path, run_flag, dp_list, n_dp_list);
}
void launch_dp_finish(hipStream_t s, const GraphPtrs &gp, const uint32_t *run_start, const uint32_t *n_runs,
                      const int64_t *nscore, const uint32_t *nbesti, const uint32_t *n0_besti, const int64_t *last_n0_score,
                      unsigned long long *total_gain, uint32_t *blocks_done, uint32_t *best_idx, const int64_t *run_gain,
                      const long long *tile_gain, uint32_t n_tiles, uint32_t *emit, uint32_t *path_begin, uint64_t *path) {
    NP2_LAUNCH(k_dp_finish, dim3(64), 256, s, run_gain, n_runs, tile_gain, n_tiles, total_gain, blocks_done, mk_graph(gp), nscore, last_n0_score, run_start, nbesti, n0_besti, best_idx, emit, path_begin, path);
}
static inline dim3 run_grid(uint32_t bound) { return dim3(std::max<uint32_t>(1, std::min<uint32_t>((bound + 256) / 256, 4096))); }
void launch_bt_write(hipStream_t s, const GraphPtrs &gp, const uint32_t *run_start, const uint32_t *n_runs,
                     uint32_t run_bound, const uint32_t *emit, const uint32_t *eoff, const uint64_t *path,
                     uint32_t *cns_pos, uint8_t *cns_base, uint8_t *cns_cls, uint8_t *lq_nothead, uint32_t *lqc) {
    NP2_LAUNCH(k_cns_write, grid1(gp.L), 256, s, gp.refnib, gp.pflag, emit, eoff, gp.L, cns_pos, cns_base, cns_cls, lq_nothead);
    NP2_LAUNCH(k_cns_runs, run_grid(4 * (uint64_t)run_bound > 0xFFFFFFF0ull ? 0xFFFFFFF0u : 4 * run_bound), 256, s, run_start, n_runs, run_bound, gp.node_off, emit, eoff, path, cns_pos, cns_base, cns_cls, lq_nothead, lqc);
}
void launch_lq_list(hipStream_t s, const GraphPtrs &gp, const uint32_t *run_start, const uint32_t *n_runs,
                    uint32_t run_bound, const uint32_t *emit, const uint32_t *eoff, const uint64_t *path,
                    const uint32_t *lqoff, uint32_t cap, uint32_t *lq_list, uint32_t *err) {
    NP2_LAUNCH(k_lq_list, run_grid(run_bound), 256, s, run_start, n_runs, run_bound, gp.node_off, emit, eoff, path, lqoff, cap, lq_list, err);
}
void launch_default_tail(hipStream_t s, const GraphPtrs &gp, const uint32_t *best_idx, const uint32_t *M_p, uint8_t *cns_base,
                         uint8_t *cns_cls, uint32_t *lq_list, uint32_t *n_lq, uint32_t cap, uint32_t *err) {
    NP2_LAUNCH(k_default_tail, 1, 64, s, best_idx, gp.node_off, gp.pflag, gp.L, M_p, cns_base, cns_cls, lq_list, n_lq, cap, err);
}
// (grid-stride kernels over a list whose length lives on the device: `cap` is the bound of its buffer — two entries per
// exception record —, the list itself a twentieth of that; a block per 256 entries of the BOUND was 17 k blocks per
// yeast-sized batch of which 500 found work, and an empty block still costs its two scalar loads: a block per 2048)
static inline dim3 lq_grid(uint32_t cap) { return dim3(std::max<uint32_t>(1, std::min<uint32_t>((cap + 2047) / 2048, 2048))); }
void launch_lq_scan(hipStream_t s, const uint32_t *cns_pos, const uint8_t *cns_base, const uint8_t *cns_cls,
                    const uint32_t *M_p, const uint32_t *lq_list, const uint32_t *n_lq, uint32_t lq_cap, uint8_t *lq_kind,
                    uint32_t *lq_next, uint8_t *lq_nothead, uint32_t *hbits, uint32_t n_hwords, uint32_t *rstart, uint32_t *rend) {
    NP2_LAUNCH(k_lq_scan, lq_grid(std::max(lq_cap, n_hwords / 4)), 256, s, cns_pos, cns_base, cns_cls, M_p, lq_list, n_lq, lq_cap, lq_kind, lq_next, lq_nothead, hbits, n_hwords);
    NP2_LAUNCH(k_lq_region, lq_grid(lq_cap), 256, s, cns_pos, cns_base, M_p, lq_list, n_lq, lq_cap, lq_kind, lq_next, lq_nothead, hbits, rstart, rend);
}
void launch_lq_bits_count(hipStream_t s, const uint32_t *hbits, uint32_t n_words, uint32_t *wcnt) {
    NP2_LAUNCH(k_lq_bits_count, grid1((uint64_t)n_words + 1), 256, s, hbits, n_words, wcnt);
}
void launch_scatter_regions(hipStream_t s, const uint32_t *hbits, uint32_t n_words, const uint32_t *woff,
                            const uint32_t *rstart, const uint32_t *rend, uint32_t *raw_start, uint32_t *raw_end,
                            uint32_t *n_raw) {
    NP2_LAUNCH(k_scatter_regions, grid1(std::max<uint32_t>(n_words, 1)), 256, s, hbits, n_words, woff, rstart, rend, raw_start, raw_end, n_raw);
}
void launch_lq_merge_flag(hipStream_t s, const uint32_t *raw_start, const uint32_t *raw_end, const uint32_t *n_raw,
                          uint32_t *headflag) {
    NP2_LAUNCH(k_lq_merge_flag, dim3(256), 256, s, raw_start, raw_end, n_raw, headflag);
}
void launch_lq_merge_scan(hipStream_t s, const uint32_t *raw_start, const uint32_t *raw_end, const uint32_t *n_raw, uint32_t n_host,
                          uint32_t *headflag, uint32_t *hidx) {
    NP2_LAUNCH(k_lq_merge_scan, dim3(1), 1024, s, raw_start, raw_end, n_raw, n_host, headflag, hidx);
}
void launch_lq_merge_write(hipStream_t s, const uint32_t *raw_start, const uint32_t *raw_end, const uint32_t *n_raw,
                           const uint32_t *headflag, const uint32_t *hidx, uint32_t *lq_start, uint32_t *lq_end,
                           uint32_t *n_reg) {
    NP2_LAUNCH(k_lq_merge_write, dim3(256), 256, s, raw_start, raw_end, n_raw, headflag, hidx, lq_start, lq_end, n_reg);
}
void launch_read_m(hipStream_t s, const np2_read_t *reads, uint32_t R, const uint8_t *alive, const uint32_t *lq_start,
                   uint32_t n_reg, int32_t *mval) {
    NP2_LAUNCH(k_read_m, grid1(R), 256, s, reads, R, alive, lq_start, n_reg, mval);
}
void launch_pair_count(hipStream_t s, const np2_read_t *reads, uint32_t R, const uint8_t *alive,
                       const uint32_t *lq_start, const uint32_t *lq_end, uint32_t n_reg, const int32_t *smin,
                       uint32_t *pj, uint32_t *pcount, const uint64_t *ck_off, ReadInfo *rinfo) {
    NP2_LAUNCH(k_pair_count, grid1(R), 256, s, reads, R, alive, lq_start, lq_end, n_reg, smin, pj, pcount, ck_off, rinfo);
}
void launch_yak_insert(hipStream_t s, const uint64_t *words, const uint64_t *bucket_off, uint32_t n_buckets,
                       uint64_t max_bucket, uint64_t *table, uint32_t cap_log2, uint32_t *dup_flag, uint32_t gap) {
    uint32_t gx = (uint32_t)((max_bucket + 255) / 256);
    if (gx == 0) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_yak_insert, dim3(gx, n_buckets), dim3(256), 0, s, words, bucket_off, n_buckets, table,
                       cap_log2, dup_flag, gap);
}
void launch_yak_insert_dup(hipStream_t s, const uint64_t *words, const uint64_t *bucket_off, uint32_t n_buckets,
                           uint64_t max_bucket, uint64_t *table, uint32_t cap_log2, uint32_t *ord, uint32_t gap) {
    uint32_t gx = (uint32_t)((max_bucket + 255) / 256);
    gx = std::min<uint32_t>(64, std::max<uint32_t>(1, gx));
    hipLaunchKernelGGL(k_yak_insert_dup, dim3(gx, n_buckets), dim3(256), 0, s, words, bucket_off, n_buckets, table,
                       cap_log2, ord, gap);
}
void launch_lookup(hipStream_t s, const YakDev &y, const uint64_t *hashes, uint64_t n, uint16_t min_count,
                   uint16_t *out) {
    if (n) NP2_LAUNCH(k_lookup, grid1(n), 256, s, y, hashes, n, min_count, out);
}
void launch_score_strings(hipStream_t s, const YakDev &y, const uint8_t *strs, const uint64_t *off, uint64_t n,
                          uint16_t min_count, uint16_t *out, bool own_strings) {
    if (n) NP2_LAUNCH(k_score_strings, grid1(n * 64), 256, s, y, strs, off, n, min_count, out, own_strings ? 1u : 0u);
}
void launch_cand_score(hipStream_t s, const YakDev &y, const uint32_t *cand_seq_off, const uint8_t *cand_seq,
                       const uint64_t *cand_kmer, const uint32_t *n_cand_p, uint32_t cand_cap, uint16_t min_count,
                       uint16_t *kscore, uint32_t *long_list, uint32_t *n_long) {
    if (!cand_cap) return;
    NP2_LAUNCH(k_cand_score, grid1(cand_cap), 256, s, y, cand_seq_off, cand_kmer, n_cand_p, min_count, kscore, long_list, n_long);
    // (a wavefront per listed candidate, grid-stride: the list — candidates longer than k — is a small part of the candidates;
    // a fixed 1024 blocks per contig were 17 k blocks per yeast-sized batch, nearly all of them empty)
    NP2_LAUNCH(k_cand_score_long, dim3(std::max<uint32_t>(16, std::min<uint32_t>(1024, cand_cap / 4096))), 256, s, y, cand_seq_off, cand_seq, long_list, n_long, min_count, kscore);
}
} // namespace np2
