// Device -> host transfer rate of copy KERNELS writing host-mapped pinned memory (what the batch driver's result and vote
// read-backs are) against hipMemcpyAsync, for the sizes a batch group moves per flush (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_d2h.hip -o tools/bin/ubench_d2h ; tools/bin/ubench_d2h
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_one(uint4 *dst, const uint4 *src, size_t n) { // one 16-byte element per thread (csrc k_copy)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
template <int U> __global__ void k_stride(uint4 *dst, const uint4 *src, size_t n) { // grid-stride, U loads in flight
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * step < n) v[u] = src[i + u * step];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * step < n) dst[i + u * step] = v[u];
    }
}
template <int U> __global__ void k_stride_nt(uint4 *dst, const uint4 *src, size_t n) { // the same with non-temporal stores
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * step < n) v[u] = src[i + u * step];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * step < n) {
                __builtin_nontemporal_store(v[u].x, &dst[i + u * step].x);
                __builtin_nontemporal_store(v[u].y, &dst[i + u * step].y);
                __builtin_nontemporal_store(v[u].z, &dst[i + u * step].z);
                __builtin_nontemporal_store(v[u].w, &dst[i + u * step].w);
            }
    }
}

int main() {
    const size_t MAXB = 16u << 20;
    uint4 *d = nullptr, *h = nullptr, *hd = nullptr;
    CHK(hipMalloc(&d, MAXB));
    CHK(hipMemset(d, 1, MAXB));
    CHK(hipHostMalloc((void **)&h, MAXB, hipHostMallocMapped));
    CHK(hipHostGetDevicePointer((void **)&hd, h, 0));
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const size_t sizes[] = {256u << 10, 1u << 20, 3u << 20, 12u << 20};
    for (size_t bytes : sizes) {
        const size_t n = bytes / 16;
        auto run = [&](const char *name, auto launch) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                (void)hipEventRecord(e0, s);
                launch();
                (void)hipEventRecord(e1, s);
                (void)hipEventSynchronize(e1);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep) best = ms < best ? ms : best;
            }
            printf("%8zu KiB  %-28s %8.1f us  %6.1f GB/s\n", bytes >> 10, name, best * 1e3, bytes / (best * 1e-3) / 1e9);
        };
        run("hipMemcpyAsync", [&] { (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s); });
        run("one element per thread", [&] { k_one<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(hd, d, n); });
        for (unsigned g : {32u, 64u, 128u, 256u, 512u}) {
            char nm[64];
            snprintf(nm, sizeof nm, "stride x4, %u blocks", g);
            run(nm, [&] { k_stride<4><<<g, 256, 0, s>>>(hd, d, n); });
        }
        run("stride x8, 128 blocks", [&] { k_stride<8><<<128, 256, 0, s>>>(hd, d, n); });
        run("stride x4 nt, 128 blocks", [&] { k_stride_nt<4><<<128, 256, 0, s>>>(hd, d, n); });
        run("stride x4 nt, 256 blocks", [&] { k_stride_nt<4><<<256, 256, 0, s>>>(hd, d, n); });
    }
    return 0;
}
