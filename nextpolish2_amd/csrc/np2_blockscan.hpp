// Single-workgroup scans over short device arrays (1024 threads).  A thread owns 4 consecutive elements of each
// 4096-element tile (coalesced 16-byte accesses); the loads of 8 tiles are issued before the first block-wide scan,
// so 32 K elements cost one memory round trip plus 8 short scans.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace np2 {

static constexpr uint32_t BS_THREADS = 1024;
static constexpr uint32_t BS_ITEMS = 4;
static constexpr uint32_t BS_BATCH = 8; // tiles loaded ahead
static constexpr uint32_t BS_TILE = BS_THREADS * BS_ITEMS;

struct OpAdd {
    static __device__ __forceinline__ uint32_t ident() { return 0u; }
    static __device__ __forceinline__ uint32_t apply(uint32_t a, uint32_t b) { return a + b; }
};
struct OpMinI32 {
    static __device__ __forceinline__ uint32_t ident() { return 0x7FFFFFFFu; }
    static __device__ __forceinline__ uint32_t apply(uint32_t a, uint32_t b) {
        return (uint32_t)min((int32_t)a, (int32_t)b);
    }
};
struct OpMaxU32 {
    static __device__ __forceinline__ uint32_t ident() { return 0u; }
    static __device__ __forceinline__ uint32_t apply(uint32_t a, uint32_t b) { return max(a, b); }
};

// exclusive prefix (under Op) of one value per thread across the 1024-thread block; `total` = reduction of all.
// sh: 16 words of LDS.  Contains barriers: every thread of the block must call it.
template <class Op> __device__ __forceinline__ uint32_t block_excl_1024(uint32_t v, uint32_t *sh, uint32_t &total) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(x, o);
        if (lane >= (uint32_t)o) x = Op::apply(t, x);
    }
    __syncthreads(); // sh may still be read by a previous call
    if (lane == 63) sh[w] = x;
    __syncthreads();
    uint32_t base = Op::ident(), tot = Op::ident();
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t s = sh[i];
        if (i < w) base = Op::apply(base, s);
        tot = Op::apply(tot, s);
    }
    total = tot;
    uint32_t excl = __shfl_up(x, 1);
    if (lane == 0) excl = Op::ident();
    return Op::apply(base, excl);
}

// Scan n elements with a single block.  load(i) -> value, store(i, exclusive_prefix, value).  Returns the total.
template <class Op, class Load, class Store>
__device__ __forceinline__ uint32_t block_scan_array(uint32_t n, uint32_t *sh, Load load, Store store) {
    uint32_t carry = Op::ident();
    for (uint32_t s0 = 0; s0 < n; s0 += BS_TILE * BS_BATCH) {
        uint32_t v[BS_BATCH][BS_ITEMS];
#pragma unroll
        for (uint32_t b = 0; b < BS_BATCH; ++b) {
            const uint32_t i0 = s0 + b * BS_TILE + threadIdx.x * BS_ITEMS;
#pragma unroll
            for (uint32_t k = 0; k < BS_ITEMS; ++k) v[b][k] = i0 + k < n ? load(i0 + k) : Op::ident();
        }
#pragma unroll
        for (uint32_t b = 0; b < BS_BATCH; ++b) {
            const uint32_t t0 = s0 + b * BS_TILE;
            if (t0 >= n) break; // uniform
            const uint32_t i0 = t0 + threadIdx.x * BS_ITEMS;
            uint32_t acc = Op::ident();
#pragma unroll
            for (uint32_t k = 0; k < BS_ITEMS; ++k) acc = Op::apply(acc, v[b][k]);
            uint32_t tot;
            uint32_t run = Op::apply(carry, block_excl_1024<Op>(acc, sh, tot));
#pragma unroll
            for (uint32_t k = 0; k < BS_ITEMS; ++k) {
                if (i0 + k < n) store(i0 + k, run, v[b][k]);
                run = Op::apply(run, v[b][k]);
            }
            carry = Op::apply(carry, tot);
        }
    }
    return carry;
}

} // namespace np2
