// Device-wide sort/scan plumbing on rocPRIM (library primitives, not the hot kernels).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "np2_kernels.hpp"

namespace np2 {

size_t prim_temp_bytes(size_t n) {
    size_t best = 0, b = 0;
    if (n == 0) n = 1;
    (void)rocprim::radix_sort_pairs(nullptr, b, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr,
                              (uint32_t *)nullptr, n, 0, 64, (hipStream_t)0);
    best = b > best ? b : best;
    (void)rocprim::exclusive_scan(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, n,
                            rocprim::plus<uint32_t>(), (hipStream_t)0);
    best = b > best ? b : best;
    (void)rocprim::inclusive_scan(nullptr, b, (const int32_t *)nullptr, (int32_t *)nullptr, n, rocprim::minimum<int32_t>(),
                            (hipStream_t)0);
    best = b > best ? b : best;
    return best + 256;
}
int prim_sort_pairs_u64_u32(hipStream_t s, void *tmp, size_t tmp_bytes, const uint64_t *kin, uint64_t *kout,
                            const uint32_t *vin, uint32_t *vout, size_t n, unsigned end_bit) {
    if (n == 0) return 0;
    return (int)rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, end_bit, s);
}
int prim_exclusive_sum_u32(hipStream_t s, void *tmp, size_t tmp_bytes, const uint32_t *in, uint32_t *out, size_t n) {
    if (n == 0) return 0;
    return (int)rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s);
}
int prim_inclusive_sum_i32(hipStream_t s, void *tmp, size_t tmp_bytes, const int32_t *in, int32_t *out, size_t n) {
    if (n == 0) return 0;
    return (int)rocprim::inclusive_scan(tmp, tmp_bytes, in, out, n, rocprim::plus<int32_t>(), s);
}
int prim_inclusive_min_i32(hipStream_t s, void *tmp, size_t tmp_bytes, const int32_t *in, int32_t *out, size_t n) {
    if (n == 0) return 0;
    return (int)rocprim::inclusive_scan(tmp, tmp_bytes, in, out, n, rocprim::minimum<int32_t>(), s);
}

} // namespace np2
