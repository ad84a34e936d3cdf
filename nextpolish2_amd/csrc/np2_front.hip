// GPU columnariser: BAM record (CIGAR + 4-bit SEQ) x contig -> the reference's packed
// AlignSeq nibble stream, one wavefront per record, 32 alignment columns per lane per iteration.
// Reproduces Alignment::fill_with_cigar (main.rs:386-440, column expansion), Alignment::trim(8)
// (main.rs:447-513, first/last run of 8 byte-equal columns; case sensitive) and AlignSeq::new
// (main.rs:279-312, nibble packing + 0xF terminator).  Per-op column/query/target prefix sums are
// computed by the host while it parses the record (it walks the CIGAR anyway for the admission filter).
#include "np2_common.hpp"
#include "np2_kernels.hpp"

namespace np2 {

__device__ __forceinline__ uint8_t seq4_char(uint8_t nib) { // BAM "=ACMGRSVTWYHKDBN"
    switch (nib & 15) {
    case 0: return '=';
    case 1: return 'A';
    case 2: return 'C';
    case 3: return 'M';
    case 4: return 'G';
    case 5: return 'R';
    case 6: return 'S';
    case 7: return 'V';
    case 8: return 'T';
    case 9: return 'W';
    case 10: return 'Y';
    case 11: return 'H';
    case 12: return 'K';
    case 13: return 'D';
    case 14: return 'B';
    default: return 'N';
    }
}

struct Seg32 {
    uint32_t eq;      // bit j: target byte == query byte (trim's match test)
    uint64_t lo, hi;  // nibble j at bits 4j: SEQ_NUM code | 8 for insertion columns
    uint32_t t_first; // target position of the first / last column (valid if that column consumes target)
    uint32_t t_last;
};

// expand columns [c0, c0 + nv) of one record
__device__ Seg32 walk32(const FrontRec &rc, const FrontOp *__restrict__ ops, const uint8_t *__restrict__ ref,
                        const uint8_t *__restrict__ seq4, uint32_t c0, uint32_t nv) {
    Seg32 s;
    s.eq = 0;
    s.lo = s.hi = 0;
    s.t_first = s.t_last = 0;
    if (nv == 0) return s;
    const FrontOp *o = ops + rc.op_off;
    uint32_t lo = 0, hi = rc.n_ops; // last op with col0 <= c0
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (o[mid].col0 <= c0) lo = mid; else hi = mid;
    }
    uint32_t i = lo;
    FrontOp op = o[i];
    uint32_t off = c0 - op.col0;
    const uint8_t *sq = seq4 + rc.seq_off;
    for (uint32_t j = 0; j < nv; ++j) {
        while (off >= (op.len_type >> 4)) {
            ++i;
            op = o[i];
            off = 0;
        }
        const uint32_t ty = op.len_type & 15;
        uint8_t t = '-', q = '-';
        uint32_t tp = 0;
        if (ty != 1) { // M = X D consume the contig
            tp = rc.pos + op.t0 + off;
            t = ref[tp];
        }
        if (ty != 2) { // M = X I consume the read
            const uint32_t qi = op.q0 + off;
            const uint8_t b = sq[qi >> 1];
            q = seq4_char((qi & 1) ? (b & 15) : (b >> 4));
        }
        if (t == q) s.eq |= 1u << j;
        const uint64_t code = (uint64_t)(ascii_to_code(q) | (t == '-' ? 8 : 0));
        if (j < 16) s.lo |= code << (4 * j); else s.hi |= code << (4 * (j - 16));
        if (j == 0) s.t_first = tp;
        if (j == nv - 1) s.t_last = tp;
        ++off;
    }
    return s;
}

// per lane: bit e set <=> an 8-run of equal columns ends at lane-local column e (may start in the previous lane)
__device__ __forceinline__ uint32_t run8_ends(uint32_t m, uint32_t prev_m) {
    uint64_t y = ((uint64_t)m << 7) | (prev_m >> 25);
    y &= y >> 1;
    y &= y >> 2;
    y &= y >> 4; // bit b: y[b..b+7] all ones -> run ends at local column b
    return (uint32_t)y;
}

__device__ __forceinline__ void k_columnarise(const uint32_t np2_bid, const uint32_t np2_nb, const FrontRec *__restrict__ recs, uint32_t n_recs,
                                                     const FrontOp *__restrict__ ops, const uint8_t *__restrict__ ref,
                                                     const uint8_t *__restrict__ seq4, uint8_t *__restrict__ nib,
                                                     FrontOut *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)((np2_bid * blockDim.x + threadIdx.x) >> 6)); // (one record per wavefront)
    if (r >= n_recs) return;
    const FrontRec rc = recs[r];
    const uint32_t N = rc.n_cols;
    // ---- forward: first run of 8 equal columns -> shift -------------------------------------------
    uint32_t shift = 0xFFFFFFFFu, tail = 0;
    for (uint32_t c0 = 0; c0 < N && shift == 0xFFFFFFFFu; c0 += 2048) {
        const uint32_t lc0 = c0 + lane * 32;
        const uint32_t nv = lc0 < N ? min(32u, N - lc0) : 0u;
        const Seg32 s = walk32(rc, ops, ref, seq4, lc0, nv);
        uint32_t pm = __shfl_up(s.eq, 1);
        if (lane == 0) pm = tail;
        const uint32_t ends = run8_ends(s.eq, pm);
        const uint64_t any = __ballot(ends != 0);
        if (any) {
            const uint32_t fl = __builtin_ctzll(any);
            const uint32_t e = __shfl(ends ? (uint32_t)__builtin_ctz(ends) : 0u, fl);
            shift = c0 + fl * 32 + e - 7;
        }
        tail = __shfl(s.eq, 63);
    }
    if (shift == 0xFFFFFFFFu) { // no anchor: shift = len -> zero columns (main.rs:510-512)
        if (lane == 0) {
            out[r] = FrontOut{rc.pos, rc.pos, 0, 0};
            nib[rc.out_off] = 0xFF;
        }
        return;
    }
    // ---- backward: last run of 8 equal columns -> new_len ------------------------------------------
    uint32_t new_len = 0;
    {
        const uint32_t nchunk = (N + 2047) / 2048;
        for (uint32_t k = nchunk; k-- > 0 && new_len == 0;) {
            const uint32_t c0 = k * 2048;
            const uint32_t lc0 = c0 + lane * 32;
            const uint32_t nv = lc0 < N ? min(32u, N - lc0) : 0u;
            const Seg32 s = walk32(rc, ops, ref, seq4, lc0, nv);
            uint32_t pm = __shfl_up(s.eq, 1);
            if (lane == 0) pm = c0 ? walk32(rc, ops, ref, seq4, c0 - 32, 32).eq : 0u;
            const uint32_t ends = run8_ends(s.eq, pm);
            const uint64_t any = __ballot(ends != 0);
            if (any) {
                const uint32_t ll = 63 - __builtin_clzll(any);
                const uint32_t e = __shfl(ends ? 31u - (uint32_t)__builtin_clz(ends) : 0u, ll);
                new_len = c0 + ll * 32 + e + 1;
            }
        }
    }
    // ---- pack columns [shift, new_len) ----------------------------------------------------------------
    const uint32_t n_out = new_len - shift;
    uint8_t *dst = nib + rc.out_off;
    uint32_t t_s = 0, t_e = 0;
    // the wave owns its whole output slot (roundup16((N+1)/2 + 1) bytes): chunks past the terminator are zeroed here,
    // so no pre-clearing of the buffer is needed
    const uint32_t slot_cols = (uint32_t)((((((uint64_t)N + 1) >> 1) + 1 + 15) & ~15ull) * 2);
    for (uint32_t oc = 0; oc < slot_cols; oc += 2048) {
        const uint32_t lo0 = oc + lane * 32;
        const uint32_t nv = lo0 < n_out ? min(32u, n_out - lo0) : 0u;
        Seg32 s = walk32(rc, ops, ref, seq4, shift + lo0, nv);
        const bool has_term = n_out >= lo0 && n_out < lo0 + 32;
        if (has_term) { // AlignSeq::new terminator (main.rs:306-310)
            const uint32_t j = n_out - lo0;
            if (j < 16) s.lo |= 0xFULL << (4 * j); else s.hi |= 0xFULL << (4 * (j - 16));
            if ((n_out & 1) == 0) { // even: the whole byte becomes 0xFF (j is even, j + 1 <= 31)
                const uint32_t j1 = j + 1;
                if (j1 < 16) s.lo |= 0xFULL << (4 * j1); else s.hi |= 0xFULL << (4 * (j1 - 16));
            }
        }
        if (lo0 < slot_cols) {
            auto sw = [](uint32_t w) { return ((w & 0x0F0F0F0Fu) << 4) | ((w >> 4) & 0x0F0F0F0Fu); };
            uint4 v;
            v.x = sw((uint32_t)s.lo);
            v.y = sw((uint32_t)(s.lo >> 32));
            v.z = sw((uint32_t)s.hi);
            v.w = sw((uint32_t)(s.hi >> 32));
            *reinterpret_cast<uint4 *>(dst + (lo0 >> 1)) = v;
        }
        if (lo0 == 0 && nv) t_s = s.t_first;          // first kept column is a match column
        if (nv && lo0 + nv == n_out) t_e = s.t_last;  // so is the last one
    }
    // gather t_s (lane 0 of the first iteration) and t_e (the lane holding the last column)
    const uint32_t last_lane = ((n_out - 1) % 2048) / 32;
    t_e = __shfl(t_e, last_lane);
    t_s = __shfl(t_s, 0);
    if (lane == 0) out[r] = FrontOut{t_s, t_e, n_out, 0};
}

// reads[0]: the contig aligned to itself (main.rs:1732-1739), packed like any other AlignSeq
__device__ __forceinline__ void k_pack_ref(const uint32_t np2_bid, const uint32_t np2_nb, const uint8_t *__restrict__ ref, uint32_t L, uint8_t *__restrict__ dst) {
    const uint32_t b = np2_bid * blockDim.x + threadIdx.x; // output byte
    const uint32_t nbytes = ((L + 1) >> 1) + 1;
    if (b >= nbytes) return;
    const uint32_t c0 = 2 * b, c1 = 2 * b + 1;
    uint8_t hi = c0 < L ? ascii_to_code(ref[c0]) : (c0 == L ? 15 : 0);
    uint8_t lo = c1 < L ? ascii_to_code(ref[c1]) : ((c1 == L || (c0 == L && (L & 1) == 0)) ? 15 : 0);
    dst[b] = (uint8_t)((hi << 4) | lo);
}

void launch_columnarise(hipStream_t s, const FrontRec *recs, uint32_t n_recs, const FrontOp *ops, const uint8_t *ref,
                        const uint8_t *seq4, uint8_t *nib, FrontOut *out) {
    if (n_recs)
        NP2_LAUNCH(k_columnarise, dim3((n_recs + 3) / 4), 256, s, recs, n_recs, ops, ref, seq4, nib, out);
}
void launch_pack_ref(hipStream_t s, const uint8_t *ref, uint32_t L, uint8_t *dst) {
    const uint32_t nbytes = ((L + 1) >> 1) + 1;
    NP2_LAUNCH(k_pack_ref, dim3((nbytes + 255) / 256), 256, s, ref, L, dst);
}

} // namespace np2
