// 128-bit nibble vectors for the dense pass (gfx950): the 32 columns of a 16-byte piece of a packed read, column j in
// nibble j (bits 4j..4j+3; lo = columns 0-15).  All column masks carry their flag at bit 3 of the nibble (NF3).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace np2 {

struct N128 {
    uint64_t lo, hi;
};
static constexpr uint64_t NF3 = 0x8888888888888888ULL; // bit 3 of every nibble: the flag position of all column masks

__device__ __forceinline__ uint32_t swap_nib(uint32_t w) { // the packed stream holds the even column in the high nibble
    return ((w & 0x0F0F0F0Fu) << 4) | ((w >> 4) & 0x0F0F0F0Fu);
}
__device__ __forceinline__ N128 n_below(uint32_t p) { // all bits of nibbles [0, p)
    N128 m;
    m.lo = p >= 16 ? ~0ULL : ((1ULL << (4 * p)) - 1ULL);
    m.hi = p <= 16 ? 0ULL : (p >= 32 ? ~0ULL : ((1ULL << (4 * (p - 16))) - 1ULL));
    return m;
}
__device__ __forceinline__ uint32_t n_ctz(const N128 &x) { // index of the first set bit, 128 if none
    return x.lo ? (uint32_t)__builtin_ctzll(x.lo) : (x.hi ? 64u + (uint32_t)__builtin_ctzll(x.hi) : 128u);
}
__device__ __forceinline__ uint32_t n_popc(const N128 &x) {
    return (uint32_t)__builtin_popcountll(x.lo) + (uint32_t)__builtin_popcountll(x.hi);
}
// x holds nibble flags (bit 3) and is not zero: all bits below the nibble of its lowest flag / up to and including it
__device__ __forceinline__ N128 n_mask_before_first(const N128 &x) {
    N128 m;
    if (x.lo) {
        m.lo = ((x.lo & (0ULL - x.lo)) >> 3) - 1ULL;
        m.hi = 0;
    } else {
        m.lo = ~0ULL;
        m.hi = ((x.hi & (0ULL - x.hi)) >> 3) - 1ULL;
    }
    return m;
}
__device__ __forceinline__ N128 n_mask_through_first(const N128 &x) {
    N128 m;
    if (x.lo) {
        m.lo = x.lo ^ (x.lo - 1ULL);
        m.hi = 0;
    } else {
        m.lo = ~0ULL;
        m.hi = x.hi ^ (x.hi - 1ULL);
    }
    return m;
}
__device__ __forceinline__ N128 n_shl(const N128 &x, uint32_t s) { // 0 < s < 128
    N128 r;
    if (s < 64) {
        r.hi = (x.hi << s) | (x.lo >> (64 - s));
        r.lo = x.lo << s;
    } else {
        r.hi = x.lo << (s - 64);
        r.lo = 0;
    }
    return r;
}
// column (0..31) of the k-th (1-based, must exist) flag of a column mask: binary descent over popcounts
__device__ __forceinline__ uint32_t n_kth_flag(const N128 &x, uint32_t k) {
    uint32_t col = 0, m8;
    const uint32_t cA = __builtin_popcount((uint32_t)x.lo), cB = __builtin_popcount((uint32_t)(x.lo >> 32));
    const uint32_t cC = __builtin_popcount((uint32_t)x.hi);
    if (k <= cA) {
        m8 = (uint32_t)x.lo;
    } else if (k <= cA + cB) {
        k -= cA, col = 8, m8 = (uint32_t)(x.lo >> 32);
    } else if (k <= cA + cB + cC) {
        k -= cA + cB, col = 16, m8 = (uint32_t)x.hi;
    } else {
        k -= cA + cB + cC, col = 24, m8 = (uint32_t)(x.hi >> 32);
    }
    uint32_t c = __builtin_popcount(m8 & 0xFFFFu);
    if (k > c) k -= c, col += 4, m8 >>= 16;
    c = __builtin_popcount(m8 & 0xFFu);
    if (k > c) k -= c, col += 2, m8 >>= 8;
    c = __builtin_popcount(m8 & 0xFu);
    if (k > c) col += 1;
    return col;
}

} // namespace np2
