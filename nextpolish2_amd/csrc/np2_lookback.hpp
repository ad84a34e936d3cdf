// Single-pass device-wide exclusive scan of per-block aggregates ("decoupled look-back"), used to fuse
// flag -> scan -> scatter sequences into one multi-block kernel.
//
// Every block takes a ticket (its logical index in the scan order), publishes its aggregate, and wave 0 walks back
// over the predecessors' status words, 64 at a time, until it meets one that already carries an inclusive prefix.
// Status word: epoch (30 bits) | state (2 bits: 1 = aggregate, 2 = inclusive prefix) | value (32 bits); a launch
// uses a fresh epoch, so the status arrays never need clearing.  Predecessors hold lower tickets, i.e. they are
// already running, which guarantees progress; a spin limit turns a would-be hang into an error flag.
//
// The status words are RELAXED agent-scope atomics: everything a successor needs is inside the word, nothing else is
// published through it.  (Release / acquire at agent scope make every block write its XCD's L2 back on gfx942/950 —
// eight L2s, no coherence between them short of memory: measured 71 vs 30 us for k_splice_plan, 505 vs 40 us for a
// chained scan over 6 k blocks.)
//
// Use it for kernels with few, light, uniform blocks (e.g. one block per 256 regions).  A block can only retire after
// every predecessor has published, so with thousands of blocks of uneven cost (one per contig tile, one per four
// regions) the grid degenerates to in-order retirement: measured 10-20x slower than count -> scan -> write there.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace np2 {

struct Lookback {
    uint64_t *status_a; // one word per block and scanned quantity
    uint64_t *status_b;
    uint32_t *ticket;   // monotonically increasing block counter (never reset)
    uint32_t ticket_base;
    uint32_t epoch;     // 1 .. 2^30-1
    uint32_t n_blocks;  // grid size this descriptor was issued for
    uint32_t *err;      // device error word (LB_ERR is or-ed in on a ticket outside the grid)
};

static constexpr uint32_t LB_AGG = 1, LB_PREFIX = 2;
// A wait gives up after LB_WAIT_TICKS of the constant 100 MHz wall clock (4 s), looked at every 1024 spins (a count of
// spins is anything from a fraction of a second to seconds, depending on the sleep and the memory latency).
// TWO PROCESSES on one device (ranks rehearsing on one GPU) are the one case in which a wait of the DENSE PASS can hang:
// its chunks wait for status words of lower-numbered blocks, which is safe while blocks are dispatched in order, but a
// queue that is being preempted by draining stops dispatching while blocks already resident on other XCDs still wait
// for ones that now never start (measured: tools/e2e_chr1_probe.py, two 124 Mb shards; small contigs' kernels are too
// short to collide).  A process that knows it shares its device takes the chunks' counts from a pre-pass instead
// (NP2_DENSE_PRECOUNT, dist.note_ranks_per_device: no chunk waits for another), and a dense pass whose wait gave up is
// repeated that way (run_diff).  The ticket-ordered look-backs below wait only for blocks that are already resident.
static constexpr uint64_t LB_WAIT_TICKS = 400000000ull;
static constexpr uint32_t LB_ERR = 0x400u; // or-ed into the error word on a wait that gave up
__device__ __forceinline__ bool lb_gave_up(uint32_t &spins, uint64_t &t0) {
    if ((++spins & 1023u) != 0) return false;
    const uint64_t now = wall_clock64();
    if (t0 == 0) t0 = now;
    return now - t0 > LB_WAIT_TICKS;
}

__device__ __forceinline__ uint64_t lb_pack(uint32_t epoch, uint32_t state, uint32_t value) {
    return ((uint64_t)epoch << 34) | ((uint64_t)state << 32) | value;
}
__device__ __forceinline__ uint64_t lb_wait(const uint64_t *p, uint32_t epoch, bool &timeout) {
    uint64_t w, t0 = 0;
    uint32_t spins = 0;
    for (;;) {
        w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(w >> 34) == epoch && ((uint32_t)(w >> 32) & 3u) != 0) break;
        if (lb_gave_up(spins, t0)) {
            timeout = true;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return w;
}

// logical block index of this block in the scan order (uniform across the block); sh: 1 word of LDS
__device__ __forceinline__ uint32_t lb_block_id(const Lookback &lb, uint32_t *sh) {
    if (threadIdx.x == 0) {
        uint32_t id = atomicAdd(lb.ticket, 1u) - lb.ticket_base;
        if (id >= lb.n_blocks) { // host-side ticket accounting out of step with the launches: never index past the grid
            atomicOr(lb.err, LB_ERR);
            id = lb.n_blocks - 1;
        }
        sh[0] = id;
    }
    __syncthreads();
    const uint32_t id = sh[0];
    __syncthreads();
    return id;
}

// Exclusive prefixes over the blocks of two block aggregates (agg_a, agg_b: uniform across the block).
// Called by every thread of the block; returns the prefixes in pre_a / pre_b.  sh: 2 words of LDS.
__device__ __forceinline__ void lb_exclusive2(const Lookback &lb, uint32_t bid, uint32_t agg_a, uint32_t agg_b,
                                              uint32_t *sh, uint32_t *err, uint32_t &pre_a, uint32_t &pre_b) {
    // callers hand over the LDS words of the block scan that produced agg_a / agg_b: its last step has every wave read
    // them, so nobody may overwrite sh[0..1] (wave 0 below, at once when bid == 0) before all waves are through
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t lane = threadIdx.x;
        if (lane == 0) {
            const uint32_t st = bid == 0 ? LB_PREFIX : LB_AGG;
            __hip_atomic_store(&lb.status_a[bid], lb_pack(lb.epoch, st, agg_a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&lb.status_b[bid], lb_pack(lb.epoch, st, agg_b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t ea = 0, eb = 0;
        if (bid > 0) {
            int64_t j0 = (int64_t)bid - 1;
            bool timeout = false;
            for (;;) {
                const int64_t j = j0 - lane;
                uint64_t wa = 0, wb = 0;
                if (j >= 0) {
                    wa = lb_wait(&lb.status_a[j], lb.epoch, timeout);
                    wb = lb_wait(&lb.status_b[j], lb.epoch, timeout);
                }
                if (__ballot(timeout)) {
                    if (lane == 0) atomicOr(err, LB_ERR);
                    break;
                }
                // both words of a block are published together, but read one after the other: a block counts as
                // "prefix" only if both words say so; otherwise use its aggregates... which are only valid if both
                // words are aggregates.  Re-read until the pair is consistent.
                while (j >= 0 && (((uint32_t)(wa >> 32) ^ (uint32_t)(wb >> 32)) & 3u)) {
                    wa = lb_wait(&lb.status_a[j], lb.epoch, timeout);
                    wb = lb_wait(&lb.status_b[j], lb.epoch, timeout);
                    if (timeout) break;
                }
                const bool is_prefix = j >= 0 && ((uint32_t)(wa >> 32) & 3u) == LB_PREFIX;
                const uint64_t pm = __ballot(is_prefix);
                uint32_t va = j >= 0 ? (uint32_t)wa : 0u, vb = j >= 0 ? (uint32_t)wb : 0u;
                if (pm) {
                    const uint32_t first = (uint32_t)__builtin_ctzll(pm); // nearest predecessor with a prefix
                    if (lane > first) va = 0, vb = 0;
                }
                for (int o = 32; o > 0; o >>= 1) {
                    va += __shfl_xor(va, o);
                    vb += __shfl_xor(vb, o);
                }
                ea += va;
                eb += vb;
                if (pm) break;
                j0 -= 64;
                if (j0 < 0) break;
            }
            if (lane == 0) {
                __hip_atomic_store(&lb.status_a[bid], lb_pack(lb.epoch, LB_PREFIX, ea + agg_a), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&lb.status_b[bid], lb_pack(lb.epoch, LB_PREFIX, eb + agg_b), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (lane == 0) {
            sh[0] = ea;
            sh[1] = eb;
        }
    }
    __syncthreads();
    pre_a = sh[0];
    pre_b = sh[1];
    __syncthreads();
}

} // namespace np2
