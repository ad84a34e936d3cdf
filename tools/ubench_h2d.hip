// Host -> device transfer rate of a caller's PAGEABLE buffer (what np2_contig_upload is handed) on gfx950: one
// hipMemcpyAsync (the runtime locks the pages and copies in place), the same buffer pinned, and a hand-made pipeline —
// pieces copied by T threads into a ring of pinned blocks, every thread queueing its slot's DMA itself (built as
// upload_pageable in round 5, measured here slower than the plain call, and taken out again).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_h2d.hip -o tools/bin/ubench_h2d -lpthread ; tools/bin/ubench_h2d
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void piped(void *dst, const void *src, size_t bytes, size_t piece, unsigned T, hipStream_t s, std::vector<void *> &pin, std::vector<hipEvent_t> &ev) {
    const size_t n_pieces = (bytes + piece - 1) / piece;
    std::atomic<size_t> next{0};
    std::vector<char> used(2 * T, 0);
    auto work = [&](unsigned t) {
        for (unsigned round = 0;; ++round) {
            const size_t i = next.fetch_add(1);
            if (i >= n_pieces) return;
            const unsigned sl = t + T * (round & 1);
            if (used[sl]) CHK(hipEventSynchronize(ev[sl]));
            const size_t off = i * piece, n = std::min(piece, bytes - off);
            memcpy(pin[sl], (const uint8_t *)src + off, n);
            CHK(hipMemcpyAsync((uint8_t *)dst + off, pin[sl], n, hipMemcpyHostToDevice, s));
            CHK(hipEventRecord(ev[sl], s));
            used[sl] = 1;
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    CHK(hipStreamSynchronize(s));
}

int main() {
    hipStream_t s;
    CHK(hipStreamCreate(&s));
    const size_t sizes[] = {(size_t)32 << 20, (size_t)256 << 20, (size_t)1 << 30};
    for (size_t bytes : sizes) {
        uint8_t *src = (uint8_t *)malloc(bytes), *pinned = nullptr, *dst = nullptr;
        memset(src, 0x5A, bytes);
        CHK(hipHostMalloc((void **)&pinned, bytes, hipHostMallocDefault));
        memcpy(pinned, src, bytes);
        CHK(hipMalloc((void **)&dst, bytes));
        auto best = [&](auto f) {
            double b = 1e9;
            for (int r = 0; r < 4; ++r) {
                const double t0 = now();
                f();
                b = std::min(b, now() - t0);
            }
            return bytes / b / 1e9;
        };
        printf("%5zu MiB: pageable hipMemcpyAsync %.1f GB/s, pinned %.1f GB/s", bytes >> 20,
               best([&] { CHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s)); CHK(hipStreamSynchronize(s)); }),
               best([&] { CHK(hipMemcpyAsync(dst, pinned, bytes, hipMemcpyHostToDevice, s)); CHK(hipStreamSynchronize(s)); }));
        { // one thread's plain memcpy, for scale
            const double t0 = now();
            memcpy(pinned, src, bytes);
            printf(", host memcpy %.1f GB/s\n", bytes / (now() - t0) / 1e9);
        }
        for (size_t piece_mib : {2, 8, 32}) {
            for (unsigned T : {1u, 2u, 4u, 8u}) {
                const size_t piece = piece_mib << 20;
                std::vector<void *> pin(2 * T);
                std::vector<hipEvent_t> ev(2 * T);
                for (unsigned i = 0; i < 2 * T; ++i) {
                    CHK(hipHostMalloc(&pin[i], piece, hipHostMallocDefault));
                    memset(pin[i], 0, piece);
                    CHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
                }
                printf("    pieces of %2zu MiB, %u thread(s): %.1f GB/s\n", piece_mib, T, best([&] { piped(dst, src, bytes, piece, T, s, pin, ev); }));
                for (unsigned i = 0; i < 2 * T; ++i) {
                    CHK(hipHostFree(pin[i]));
                    CHK(hipEventDestroy(ev[i]));
                }
            }
        }
        CHK(hipFree(dst));
        CHK(hipHostFree(pinned));
        free(src);
    }
    return 0;
}
