// Kernel launch layer of the np2 hot path: every kernel body is a __device__ function
//     void k_xxx(uint32_t np2_bid, uint32_t np2_nb, <parameters>)
// (np2_bid / np2_nb = index of the workgroup inside ITS contig's grid / size of that grid) and is launched through
// NP2_LAUNCH.  One generic __global__ template runs the body for up to MAXB independent argument packs ("slots") in a
// single grid: workgroup b finds its slot in the prefix sums of the per-slot grid sizes and calls the body with the
// slot-local workgroup index.  A plain launch is the one-slot case.
//
// Why: a contig of a few hundred kb keeps the MI355X busy for a few microseconds per kernel, so a many-contig assembly
// (yeast: 17 contigs, ~140 launches per contig and pass) is bound by launch count, not by work.  The batch driver
// (np2_batch.cpp) runs the unchanged per-contig host pipeline on one host thread per contig, *records* its launches
// instead of issuing them (thread-local Recorder), and at every point where the pipelines wait for the device merges
// the recorded queues: launches of the same kernel at the heads of several queues become ONE batched launch.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <utility>
#include <vector>
#include <hip/hip_runtime.h>

namespace np2 {

static constexpr int MAXB = 32;                 // slots per batched launch (further limited by the kernarg segment)
static constexpr size_t KERNARG_LIMIT = 4096;   // bytes of kernel arguments HIP accepts

// ---- trivially copyable argument pack --------------------------------------------------------------------------
template <class... A> struct Pack;
template <> struct Pack<> {};
template <class H, class... T> struct Pack<H, T...> {
    H h;
    Pack<T...> t;
};
template <size_t I, class P> struct PackGet;
template <class H, class... T> struct PackGet<0, Pack<H, T...>> {
    static __host__ __device__ __forceinline__ const H &get(const Pack<H, T...> &p) { return p.h; }
};
template <size_t I, class H, class... T> struct PackGet<I, Pack<H, T...>> {
    static __host__ __device__ __forceinline__ auto &get(const Pack<H, T...> &p) {
        return PackGet<I - 1, Pack<T...>>::get(p.t);
    }
};
template <class... A> struct PackMaker;
template <> struct PackMaker<> {
    static Pack<> make() { return Pack<>{}; }
};
template <class H, class... T> struct PackMaker<H, T...> {
    static Pack<H, T...> make(H h, T... t) { return Pack<H, T...>{h, PackMaker<T...>::make(t...)}; }
};

template <int N, class... A> struct Batch {
    uint32_t n;
    uint32_t off[MAXB + 1]; // prefix sums of the per-slot grid sizes
    Pack<A...> a[N];
};
template <class... A> constexpr int batch_slots() {
    constexpr size_t hdr = sizeof(uint32_t) * (MAXB + 2) + 16;
    constexpr size_t per = sizeof(Pack<A...>) > 0 ? sizeof(Pack<A...>) : 1;
    constexpr size_t fit = (KERNARG_LIMIT - hdr) / per;
    return fit >= (size_t)MAXB ? MAXB : (fit < 1 ? 1 : (int)fit);
}

template <auto Body, class... A, size_t... I>
__device__ __forceinline__ void call_body(uint32_t bid, uint32_t nb, const Pack<A...> &p, std::index_sequence<I...>) {
    Body(bid, nb, PackGet<I, Pack<A...>>::get(p)...);
}

template <int BLOCK, auto Body, int N, class... A>
__global__ __launch_bounds__(BLOCK) void k_np2_batched(const Batch<N, A...> B) {
    uint32_t slot = 0;
    if (B.n > 1)
        while (slot + 1 < B.n && blockIdx.x >= B.off[slot + 1]) ++slot; // (scalar: blockIdx is uniform)
    call_body<Body, A...>(blockIdx.x - B.off[slot], B.off[slot + 1] - B.off[slot], B.a[slot],
                          std::index_sequence_for<A...>{});
}
// The same with a floor on the waves per SIMD the register allocator has to leave room for (NP2_LAUNCH_WAVES): a kernel
// that is a chain of dependent loads and sits one or two registers above an occupancy step.
template <int BLOCK, int WAVES, auto Body, int N, class... A>
__global__ __launch_bounds__(BLOCK, WAVES) void k_np2_batched_w(const Batch<N, A...> B) {
    uint32_t slot = 0;
    if (B.n > 1)
        while (slot + 1 < B.n && blockIdx.x >= B.off[slot + 1]) ++slot;
    call_body<Body, A...>(blockIdx.x - B.off[slot], B.off[slot + 1] - B.off[slot], B.a[slot],
                          std::index_sequence_for<A...>{});
}

// ---- recorder ----------------------------------------------------------------------------------------------------
struct KernelDesc {
    const char *name;
    uint32_t arg_bytes;
    int max_batch;
    // launch n slots (n <= max_batch) of this kernel as one grid on stream s
    void (*launch)(hipStream_t s, int n, const uint32_t *grids, const void *const *args);
};

struct Recorder {
    struct Cmd {
        const KernelDesc *kd = nullptr; // nullptr: generic stream operation `fn`
        uint32_t grid = 0;
        size_t arg_off = 0;
        std::function<void(hipStream_t)> fn;
    };
    std::vector<Cmd> q;
    std::vector<uint64_t> arena; // argument packs (8-byte aligned)
    void *group = nullptr;       // owned by the batch driver
    int slot = 0;
    // flush every queue of the group; need_done: wait for the device too (false: the commands only have to be on their
    // way — a pipeline that goes on with host work the device is not needed for)
    void (*sync_fn)(Recorder *, bool need_done) = nullptr;
    void push_kernel(const KernelDesc *kd, uint32_t grid, const void *args, size_t bytes) {
        Cmd c;
        c.kd = kd;
        c.grid = grid;
        c.arg_off = arena.size();
        arena.resize(arena.size() + (bytes + 7) / 8);
        memcpy(arena.data() + c.arg_off, args, bytes);
        q.push_back(std::move(c));
    }
    void push_fn(std::function<void(hipStream_t)> f) {
        Cmd c;
        c.fn = std::move(f);
        q.push_back(std::move(c));
    }
    // device buffers released while recorded commands may still name them: freed after the next flush
    std::vector<std::pair<void *, size_t>> graveyard; // (block, its slab size — or, bit 63 set, its cache size class)
    void clear() {
        q.clear();
        arena.clear();
    }
};
Recorder *&tl_recorder(); // this thread's recorder (nullptr = launches go straight to their stream); np2_host.cpp

template <class S> struct Sig;
template <class... A> struct Sig<void (*)(uint32_t, uint32_t, A...)> {
    using pack_t = Pack<A...>;
    static constexpr int N = batch_slots<A...>();
    static pack_t make(A... a) { return PackMaker<A...>::make(a...); }
    template <int BLOCK, auto Body, int WAVES = 0> static void launch_n(hipStream_t s, int n, const uint32_t *grids, const void *const *args) {
        Batch<N, A...> B;
        B.n = (uint32_t)n;
        B.off[0] = 0;
        for (int i = 0; i < n; ++i) {
            B.off[i + 1] = B.off[i] + grids[i];
            memcpy(&B.a[i], args[i], sizeof(pack_t));
        }
        for (int i = n; i < MAXB; ++i) B.off[i + 1] = B.off[n];
        if constexpr (WAVES == 0) hipLaunchKernelGGL((k_np2_batched<BLOCK, Body, N, A...>), dim3(B.off[n]), dim3(BLOCK), 0, s, B);
        else hipLaunchKernelGGL((k_np2_batched_w<BLOCK, WAVES, Body, N, A...>), dim3(B.off[n]), dim3(BLOCK), 0, s, B);
    }
    template <int BLOCK, auto Body, int WAVES = 0> static const KernelDesc *desc(const char *name) {
        static const KernelDesc d{name, (uint32_t)sizeof(pack_t), N, &launch_n<BLOCK, Body, WAVES>};
        return &d;
    }
};

inline uint32_t grid_x(dim3 g) { return g.x; }
inline uint32_t grid_x(uint32_t g) { return g; }
inline uint32_t grid_x(int g) { return (uint32_t)g; }
inline uint32_t grid_x(uint64_t g) { return (uint32_t)g; }
inline uint32_t grid_x(long g) { return (uint32_t)g; }

template <int BLOCK, auto Body, int WAVES = 0, class... P> inline void launch(const char *name, hipStream_t s, uint32_t grid, P &&...p) {
    using S = Sig<decltype(Body)>;
    if (grid == 0) return;
    const typename S::pack_t pk = S::make(std::forward<P>(p)...);
    if (Recorder *r = tl_recorder()) {
        r->push_kernel(S::template desc<BLOCK, Body, WAVES>(name), grid, &pk, sizeof pk);
        return;
    }
    const void *ap = &pk;
    S::template launch_n<BLOCK, Body, WAVES>(s, 1, &grid, &ap);
}

} // namespace np2

// NP2_LAUNCH(kernel body, grid (dim3 or integer; x only), block size (compile-time constant), stream, args...)
#define NP2_LAUNCH(kernel, grid, block, s, ...) \
    ::np2::launch<(int)(block), &kernel>(#kernel, s, ::np2::grid_x(grid), __VA_ARGS__)
// ... with at least `waves` resident waves per SIMD (the register allocator may spill to get there)
#define NP2_LAUNCH_WAVES(kernel, waves, grid, block, s, ...) \
    ::np2::launch<(int)(block), &kernel, (int)(waves)>(#kernel, s, ::np2::grid_x(grid), __VA_ARGS__)
