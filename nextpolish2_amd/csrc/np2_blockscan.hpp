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

// DPP cross-lane moves (GFX9 encodings): row_shr:n = 0x110 + n, wave_shr:1 = 0x138, row_bcast:15 = 0x142,
// row_bcast:31 = 0x143.  Lanes whose source lies outside the row (or whose row is masked off) receive `ident`.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_get(uint32_t ident, uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)x, CTRL, ROW_MASK, 0xF, false);
}
// inclusive scan over the 64 lanes of a wavefront: 4 row-shift steps + 2 row broadcasts, one DPP VALU op each
template <class Op> __device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    x = Op::apply(dpp_get<0x111, 0xF>(Op::ident(), x), x);
    x = Op::apply(dpp_get<0x112, 0xF>(Op::ident(), x), x);
    x = Op::apply(dpp_get<0x114, 0xF>(Op::ident(), x), x);
    x = Op::apply(dpp_get<0x118, 0xF>(Op::ident(), x), x);
    x = Op::apply(dpp_get<0x142, 0xA>(Op::ident(), x), x); // lane 15 of rows 0 / 2 -> rows 1 / 3
    x = Op::apply(dpp_get<0x143, 0xC>(Op::ident(), x), x); // lane 31 -> rows 2, 3
    return x;
}
// value of the previous lane (lane 0 receives ident)
__device__ __forceinline__ uint32_t wave_prev_lane(uint32_t ident, uint32_t x) { return dpp_get<0x138, 0xF>(ident, x); }

// exclusive prefix (under Op) of one value per thread across a block of NW wavefronts (NW <= 16); `total` =
// reduction of all.  sh: NW words of LDS.  Contains barriers: every thread of the block must call it.
template <class Op, uint32_t NW>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *sh, uint32_t &total) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t x = wave_incl_scan<Op>(v);
    __syncthreads(); // sh may still be read by a previous call
    if (lane == 63) sh[w] = x;
    __syncthreads();
    // every wave scans the NW wave totals inside its first row
    uint32_t t = lane < NW ? sh[lane] : Op::ident();
    t = Op::apply(dpp_get<0x111, 0xF>(Op::ident(), t), t);
    t = Op::apply(dpp_get<0x112, 0xF>(Op::ident(), t), t);
    t = Op::apply(dpp_get<0x114, 0xF>(Op::ident(), t), t);
    t = Op::apply(dpp_get<0x118, 0xF>(Op::ident(), t), t);
    total = (uint32_t)__builtin_amdgcn_readlane((int)t, NW - 1);
    const uint32_t base = w ? (uint32_t)__builtin_amdgcn_readlane((int)t, (int)w - 1) : Op::ident();
    return Op::apply(base, wave_prev_lane(Op::ident(), x));
}
template <class Op> __device__ __forceinline__ uint32_t block_excl_1024(uint32_t v, uint32_t *sh, uint32_t &total) {
    return block_excl_scan<Op, 16>(v, sh, total);
}

// Scan n elements with a single block.  load(i) -> value, store(i, exclusive_prefix, value).  Returns the total.
// Per batch of 8 tiles: all loads first, then the block-wide scans (their barriers would otherwise wait for
// outstanding stores), then all stores.
template <class Op, class Load, class Store>
__device__ __forceinline__ uint32_t block_scan_array(uint32_t n, uint32_t *sh, Load load, Store store) {
    uint32_t carry = Op::ident();
    for (uint32_t s0 = 0; s0 < n; s0 += BS_TILE * BS_BATCH) {
        uint32_t v[BS_BATCH][BS_ITEMS], pre[BS_BATCH];
#pragma unroll
        for (uint32_t b = 0; b < BS_BATCH; ++b) {
            const uint32_t i0 = s0 + b * BS_TILE + threadIdx.x * BS_ITEMS;
            const bool tile_live = s0 + b * BS_TILE < n; // (uniform: a short array — a contig's few thousand regions — has one live tile of the eight)
#pragma unroll
            for (uint32_t k = 0; k < BS_ITEMS; ++k) { // clamped, unconditional loads: all of them are in flight at once
                uint32_t x = Op::ident();
                if (tile_live) x = load(min(i0 + k, n - 1));
                v[b][k] = i0 + k < n ? x : Op::ident();
            }
        }
#pragma unroll
        for (uint32_t b = 0; b < BS_BATCH; ++b) {
            pre[b] = carry;
            if (s0 + b * BS_TILE < n) { // uniform
                uint32_t acc = Op::ident();
#pragma unroll
                for (uint32_t k = 0; k < BS_ITEMS; ++k) acc = Op::apply(acc, v[b][k]);
                uint32_t tot;
                pre[b] = Op::apply(carry, block_excl_1024<Op>(acc, sh, tot));
                carry = Op::apply(carry, tot);
            }
        }
#pragma unroll
        for (uint32_t b = 0; b < BS_BATCH; ++b) {
            const uint32_t i0 = s0 + b * BS_TILE + threadIdx.x * BS_ITEMS;
            uint32_t run = pre[b];
#pragma unroll
            for (uint32_t k = 0; k < BS_ITEMS; ++k) {
                if (i0 + k < n) store(i0 + k, run, v[b][k]);
                run = Op::apply(run, v[b][k]);
            }
        }
    }
    return carry;
}

} // namespace np2
