// Internal shared definitions of the np2 host driver (context, device buffers, transfer helpers).
#pragma once
#include "../../include/np2.h"
#include "np2_common.hpp"
#include "np2_kernels.hpp"
#include "np2_hostcpu.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace np2h {

using namespace np2;


struct Np2Error : std::runtime_error {
    int code;
    Np2Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
#define HIPCHK(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess)                                                                      \
            throw Np2Error(NP2_E_DEVICE, std::string(#x) + ": " + hipGetErrorString(e_));          \
    } while (0)
#define REFPANIC_IF(c, m)                                                                          \
    do {                                                                                           \
        if (c) throw Np2Error(NP2_E_REFPANIC, std::string("reference would panic: ") + (m));        \
    } while (0)

// NP2_ALLOC_PROFILE: driver calls that allocate, release or drain — the ones that take the runtime's process-wide locks —
// reported on stderr when they take a millisecond or more (a tool's switch: tools/cli_probe.py)
struct SlowCall {
    static bool on() {
        static const bool v = getenv("NP2_ALLOC_PROFILE") != nullptr;
        return v;
    }
    const char *what;
    size_t bytes;
    std::chrono::steady_clock::time_point t0;
    SlowCall(const char *w, size_t b = 0) : what(w), bytes(b) {
        if (on()) t0 = std::chrono::steady_clock::now();
    }
    ~SlowCall() {
        if (!on()) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms >= 1.0) fprintf(stderr, "[np2 alloc] %s %.1f MB: %.1f ms\n", what, (double)bytes / 1e6, ms);
    }
};

// Device blocks of short-lived objects — the resident pileup of a contig, the front end's staging buffers — are cached
// per device by size class instead of going back to the driver: hipMalloc / hipFree cost 0.1 - 1 ms each, hipFree
// synchronises the device, and a contig of a many-contig assembly brings a dozen of each.  Only for memory whose owner
// releases it after its last use has COMPLETED (a contig is freed by the caller after the polish calls returned, the
// front end's temporaries after the read-back of their kernel's results): a cached block may be handed to any stream.
struct DevCache {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void *>> free_; // (device, bytes) -> blocks
    size_t cached = 0;
    // idle bytes kept; beyond that blocks go back to the driver: NP2_DEV_CACHE_GB, else a twelfth of the device's memory
    // (24 GB of an MI355X's 288), found when the first block comes back
    size_t LIMIT = 0;
    void set_limit() {
        if (LIMIT) return;
        if (const char *e = getenv("NP2_DEV_CACHE_GB")) LIMIT = (size_t)(atof(e) * (double)(1ull << 30));
        size_t fr = 0, tot = 0;
        if (!LIMIT && hipMemGetInfo(&fr, &tot) == hipSuccess) LIMIT = tot / 12;
        if (!LIMIT) LIMIT = 24ull << 30;
    }
    static size_t size_class(size_t bytes) { // four classes per octave (<= 25 % slack), at least 4 KiB
        size_t b = std::max<size_t>(bytes, 4096);
        unsigned lg = 63 - (unsigned)__builtin_clzll(b);
        const size_t step = (size_t)1 << (lg > 2 ? lg - 2 : 0);
        return (b + step - 1) & ~(step - 1);
    }
    void *get(size_t &bytes) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        bytes = size_class(bytes);
        {
            std::lock_guard<std::mutex> l(mu);
            auto it = free_.find({dev, bytes});
            if (it != free_.end() && !it->second.empty()) {
                void *p = it->second.back();
                it->second.pop_back();
                cached -= bytes;
                return p;
            }
        }
        void *p = nullptr;
        SlowCall sc("hipMalloc (block)", bytes);
        if (hipMalloc(&p, bytes) != hipSuccess) {
            trim(0); // (the idle blocks may be what is missing)
            HIPCHK(hipMalloc(&p, bytes));
        }
        return p;
    }
    void put(void *p, size_t bytes) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        bool over;
        {
            std::lock_guard<std::mutex> l(mu);
            set_limit();
            free_[{dev, bytes}].push_back(p);
            cached += bytes;
            over = cached > LIMIT;
        }
        if (over) trim(LIMIT / 2);
    }
    void trim(size_t keep) {
        std::vector<std::pair<int, void *>> drop;
        {
            std::lock_guard<std::mutex> l(mu);
            for (auto it = free_.rbegin(); it != free_.rend() && cached > keep; ++it) // largest classes first
                while (!it->second.empty() && cached > keep) {
                    drop.emplace_back(it->first.first, it->second.back());
                    it->second.pop_back();
                    cached -= it->first.second;
                }
        }
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (auto &d : drop) {
            (void)hipSetDevice(d.first);
            SlowCall sc("hipFree (cache trim)");
            (void)hipFree(d.second);
        }
        (void)hipSetDevice(cur);
    }
};
inline DevCache &dev_cache() {
    static DevCache *c = new DevCache(); // leaked on purpose: outlives every context
    return *c;
}

// Small scratch blocks (a context holds ~130 buffers, most of them a few KB to a few MB) are carved out of 64 MiB slabs:
// a cold context then costs a handful of hipMalloc calls instead of one per buffer (0.1 - 0.3 ms each, serialised by
// the driver across the contexts of a process: four fresh contexts spent ~100 ms of a 0.3 s command-line run there).
// A released block goes to a per-size free list after the device has drained (a buffer grows while kernels that use
// its old storage may still be queued; hipFree, which it replaces, synchronises the device as well).
struct DevSlabs {
    static constexpr size_t SLAB = 64ull << 20, MAX_BLOCK = 4ull << 20;
    std::mutex mu;
    struct Slab {
        int dev;
        uint8_t *base;
        size_t used;
    };
    std::vector<Slab> slabs;
    std::map<std::pair<int, size_t>, std::vector<void *>> free_;
    static size_t round(size_t bytes) { return DevCache::size_class(std::max<size_t>(bytes, 256)); }
    void *get(size_t &bytes) { // bytes <= MAX_BLOCK
        int dev = 0;
        (void)hipGetDevice(&dev);
        bytes = (std::max<size_t>(bytes, 256) <= 4096) ? ((bytes + 255) & ~(size_t)255) : round(bytes);
        std::lock_guard<std::mutex> l(mu);
        auto it = free_.find({dev, bytes});
        if (it != free_.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            return p;
        }
        for (auto &sl : slabs)
            if (sl.dev == dev && SLAB - sl.used >= bytes) {
                void *p = sl.base + sl.used;
                sl.used += bytes;
                return p;
            }
        void *b = nullptr;
        SlowCall sc("hipMalloc (slab)", SLAB);
        if (hipMalloc(&b, SLAB) != hipSuccess) {
            (void)hipGetLastError();
            dev_cache().trim(0); // (the idle blocks of the size-class cache may be what is missing)
            HIPCHK(hipMalloc(&b, SLAB));
        }
        slabs.push_back(Slab{dev, (uint8_t *)b, bytes});
        return b;
    }
    bool owns(const void *p) {
        std::lock_guard<std::mutex> l(mu);
        for (auto &sl : slabs)
            if ((const uint8_t *)p >= sl.base && (const uint8_t *)p < sl.base + SLAB) return true;
        return false;
    }
    void put(void *p, size_t bytes) { // the caller guarantees the block is no longer in use on the device
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> l(mu);
        free_[{dev, bytes}].push_back(p);
    }
};
inline DevSlabs &dev_slabs() {
    static DevSlabs *s = new DevSlabs(); // leaked on purpose: outlives every context
    return *s;
}
// Inside such a scope the device has been drained once and stays unused by the releasing object: its buffers go back
// to the slabs / the cache without a synchronisation each (a context frees ~130 of them when it is destroyed).
inline int &dev_sync_depth() {
    static thread_local int d = 0;
    return d;
}
struct DevSyncScope {
    DevSyncScope() {
        if (dev_sync_depth() == 0) { // (an enclosing scope has drained the device already: a batch driver releasing its slots)
            SlowCall sc("hipDeviceSynchronize (release scope)");
            (void)hipDeviceSynchronize();
        }
        ++dev_sync_depth();
    }
    ~DevSyncScope() { --dev_sync_depth(); }
};
// release of a block whose last use on the device has completed: slab pieces to their free list, hipMalloc blocks to the
// cache (handed to the next buffer of that size class — of a later context, too — instead of back to the driver)
inline void dev_release_idle(void *p, size_t slab_bytes, size_t big_bytes) {
    if (slab_bytes)
        dev_slabs().put(p, slab_bytes);
    else if (big_bytes)
        dev_cache().put(p, big_bytes);
    else
        (void)hipFree(p);
}

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    bool cached = false;   // blocks come from / go back to dev_cache() (see there for when that is allowed)
    size_t cache_bytes = 0;
    size_t slab_bytes = 0; // != 0: the block is a piece of a slab (DevSlabs)
    ~DevBuf() { release(); }
    void release() {
        if (p) {
            if (cached) {
                // (normally the owner has finished its device work; when an exception unwinds through it, a transfer or a
                // kernel of its stream may still be in flight and the next taker of the block would be corrupted)
                if (std::uncaught_exceptions() > 0 && dev_sync_depth() == 0) (void)hipDeviceSynchronize();
                dev_cache().put(p, cache_bytes);
            } else if (Recorder *r = tl_recorder()) {
                // recorded, not yet issued commands may name it: released after the next flush
                r->graveyard.push_back({p, slab_bytes ? slab_bytes : (cache_bytes | (1ull << 63))});
            } else {
                // a buffer grows while kernels that use its old storage may still be queued: drain the device first (what
                // hipFree, which this replaces, does implicitly) unless the owner already has
                if (dev_sync_depth() == 0) {
                    SlowCall sc("hipDeviceSynchronize (buffer regrown)", cap * sizeof(T));
                    (void)hipDeviceSynchronize();
                }
                dev_release_idle(p, slab_bytes, cache_bytes);
            }
        }
        p = nullptr;
        cap = 0;
        slab_bytes = 0;
        cache_bytes = 0;
    }
    T *ensure(size_t n) {
        if (n > cap) {
            release();
            size_t want = n + n / 8 + 64;
            if (cached) {
                cache_bytes = want * sizeof(T);
                p = (T *)dev_cache().get(cache_bytes);
                cap = cache_bytes / sizeof(T);
            } else if (want * sizeof(T) <= DevSlabs::MAX_BLOCK) {
                slab_bytes = want * sizeof(T);
                p = (T *)dev_slabs().get(slab_bytes);
                cap = slab_bytes / sizeof(T);
            } else { // a block of its own, from the cache if an idle one of its size class is there
                cache_bytes = want * sizeof(T);
                p = (T *)dev_cache().get(cache_bytes);
                cap = cache_bytes / sizeof(T);
            }
        }
        return p;
    }
};

// Pinned host memory pool for result buffers: np2_free() returns blocks here (no ctx needed).
// Pageable D2H makes the ROCm runtime pin/unpin user pages lazily (multi-ms stalls on the next copy).
struct PinnedPool {
    std::mutex mu;
    std::map<void *, size_t> live;                 // handed out
    std::vector<std::pair<size_t, void *>> free_;  // (capacity, ptr)
    void *get(size_t bytes) {
        std::lock_guard<std::mutex> l(mu);
        size_t best = free_.size();
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].first >= bytes && free_[i].first <= bytes * 4 + (1u << 20) && // (a fitting block, not a huge one)
                (best == free_.size() || free_[i].first < free_[best].first))
                best = i;
        void *p = nullptr;
        size_t cap = 0;
        if (best != free_.size()) {
            p = free_[best].second;
            cap = free_[best].first;
            free_.erase(free_.begin() + (long)best);
        } else {
            cap = bytes + bytes / 8 + 4096;
            SlowCall sc("hipHostMalloc (pool)", cap);
            if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
        }
        live[p] = cap;
        return p;
    }
    bool put(void *p) {
        std::lock_guard<std::mutex> l(mu);
        auto it = live.find(p);
        if (it == live.end()) return false;
        free_.emplace_back(it->second, p);
        live.erase(it);
        // keep what a batch of contigs hands back between two steps (a pinned allocation costs ~1 ms per MB), but bound
        // the idle memory: beyond 2 GiB (or 4096 blocks) the oldest go.  (Every context's staging lives here too since
        // round 3: a count limit of 64 blocks had the 17-contig bench free and re-pin blocks inside every step.)
        size_t held = 0;
        for (auto &f : free_) held += f.first;
        while (free_.size() > 4096 || (held > (2ull << 30) && free_.size() > 8)) {
            held -= free_.front().first;
            SlowCall sc("hipHostFree (pool)", free_.front().first);
            (void)hipHostFree(free_.front().second);
            free_.erase(free_.begin());
        }
        return true;
    }
};
inline PinnedPool &pinned_pool() {
    static PinnedPool *p = new PinnedPool(); // leaked on purpose: outlives every context
    return *p;
}

// A context's pinned staging: blocks of the process-wide pool (pinning and unpinning cost milliseconds per block: a
// context that comes and goes with every command-line run must not pay them each time)
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    ~PinnedBuf() {
        if (p) pinned_pool().put(p);
    }
    void *ensure(size_t n) {
        if (n > cap) {
            if (p) pinned_pool().put(p);
            p = nullptr;
            cap = 0;
            const size_t want = n + n / 4 + 65536;
            p = pinned_pool().get(want);
            if (!p) throw Np2Error(NP2_E_NOMEM, "hipHostMalloc failed");
            cap = want;
        }
        return p;
    }
};

// Streams, events and the host-mapped mailbox of a context, kept when the context goes: creating three streams is
// ~9 ms (more when several threads do it at once), destroying them ~6 ms, and the command line makes a handful of
// contexts per run.  A released set has been synchronised; at most 32 idle sets are kept per device.
struct CtxDeviceState {
    hipStream_t stream = nullptr, stream2 = nullptr, stream_out = nullptr;
    hipEvent_t ev_out = nullptr, ev_fork = nullptr, ev_join = nullptr;
    uint32_t *mbox_host = nullptr, *mbox_dev = nullptr;
    void destroy() {
        if (ev_out) (void)hipEventDestroy(ev_out);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (stream_out) (void)hipStreamDestroy(stream_out);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream) (void)hipStreamDestroy(stream);
        if (mbox_host) (void)hipHostFree(mbox_host);
        *this = CtxDeviceState();
    }
};
struct CtxStatePool {
    std::mutex mu;
    std::map<int, std::vector<CtxDeviceState>> idle; // by device
    bool get(int device, CtxDeviceState &out) {
        std::lock_guard<std::mutex> l(mu);
        auto &v = idle[device];
        if (v.empty()) return false;
        out = v.back();
        v.pop_back();
        return true;
    }
    void put(int device, CtxDeviceState &st) {
        {
            std::lock_guard<std::mutex> l(mu);
            auto &v = idle[device];
            if (v.size() < 32) {
                v.push_back(st);
                st = CtxDeviceState();
                return;
            }
        }
        st.destroy();
    }
};
inline CtxStatePool &ctx_state_pool() {
    static CtxStatePool *p = new CtxStatePool(); // leaked on purpose: outlives every context
    return *p;
}

struct YakTable {
    uint32_t k = 0, cap_log2 = 0;
    std::shared_ptr<DevBuf<uint64_t>> table; // shared by the contexts of one device (np2_ctx_create_shared)
    std::shared_ptr<DevBuf<uint32_t>> ord;   // only for a dump that repeated a key (k_yak_insert_dup)
    YakDev dev() const { return YakDev{table->p, cap_log2, k, ord ? ord->p : nullptr}; }
};

struct Timing {
    std::vector<std::string> names;
    std::vector<float> ms;
    std::string joined;
    std::vector<std::pair<std::string, float>> host; // host wall-clock sections (ms)
};
inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}


} // namespace np2h

using namespace np2h;

struct np2_contig {
    np2_contig() { // (a contig's blocks are cached: see DevCache)
        reads.cached = nib.cached = refnib.cached = ck_off.cached = ckpt.cached = descs.cached = true;
        tile_rd_off.cached = tile_rd.cached = true;
    }
    uint32_t L = 0, R = 0;
    uint64_t nib_bytes = 0, n_cols = 0, n_ckpt = 0;
    DevBuf<np2_read_t> reads;
    DevBuf<uint8_t> nib;
    DevBuf<uint8_t> refnib; // nibble-packed contig codes (+ padding), also viewed as uint64_t words; then the dense pass's copies
    uint32_t ref_stride = 0; // bytes of each of the three copies
    DevBuf<uint64_t> ck_off;
    DevBuf<uint32_t> ckpt;
    // DENSE_COLS-column chunks of the streamed reads (read 0 and dropped reads have none)
    uint32_t n_chunks = 0;
    DevBuf<ChunkDesc> descs;
    // reads overlapping each contig tile (ascending read index), CSR
    uint32_t n_tiles = 0;
    DevBuf<uint32_t> tile_rd_off, tile_rd;
    // records a contig tile's bucket holds: sized by the deepest tile of this contig (a tile of 1024 positions under d
    // reads is 1024 * d columns, of which about one per cent become exception records: room for 4 %), between 1024 and
    // what a tile can sort inside LDS.  24 B per contig position at 30x instead of a flat 48; fuller tiles spill.
    uint32_t tile_cap = TILE_CAP;
};

struct np2_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr; // side stream for kernels that can overlap the main one (fork / join by events)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // deferred output (np2_result_fetch_begin / _end): device snapshot, two pinned host buffers, their own stream
    hipStream_t stream_out = nullptr;
    hipEvent_t ev_out = nullptr;
    uint8_t *out_host[2] = {nullptr, nullptr};
    size_t out_host_cap[2] = {0, 0};
    int out_slot = 0;
    bool out_pending = false;
    uint64_t out_len = 0;
    std::vector<YakTable> yaks;
    std::string err;
    bool trace = false;
    std::map<std::string, std::vector<uint8_t>> trace_items;
    Timing timing;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending_events;

    PinnedBuf pin_d2h, pin_h2d;
    uint32_t *mbox_host = nullptr, *mbox_dev = nullptr; // host-mapped scalar mailbox: [0] = sequence, [1..] = scal
    uint32_t mbox_seq = 0;
    // a batch driver's slot context: its three stream handles are the batch's ONE stream and its mailbox a piece of the
    // batch's host-mapped block (a slot only records commands; making three streams and a pinned block per slot cost
    // 10 - 16 ms each in a fresh process: 250 ms for a 16-slot driver)
    bool borrowed_state = false;
    // np2_batch_set_sink: where a copy of the polished bases goes on the device (a rank's gather buffer), 0 = nowhere
    uint8_t *sink_dst = nullptr;
    uint64_t sink_cap = 0;
    uint32_t last_first_pos = 0, last_last_pos = 0;
    const uint8_t *last_dbase = nullptr; // device copy of the last polished sequence (valid until the next call)
    const uint32_t *last_dpos = nullptr; // ... and of its positions
    DevBuf<uint32_t> shard_bounds;       // np2_shard_final_device: slice / strip indices
    uint64_t last_len = 0;
    bool reuse_identical_pass = true;
    bool h2d_inflight = false; // pin_h2d holds data of a copy that may not have completed yet
    bool stage_timing = false; // arm every stage timer (np2_ctx_set_timing)
    // scratch (reused across contigs)
    DevBuf<uint8_t> tmp;
    DevBuf<uint64_t> keys_raw, keys;
    DevBuf<uint32_t> vals_raw, vals;
    DevBuf<uint32_t> npos, ncount, nminr, nbesti, node_off, run_start, run_end, n0_besti, emit, eoff;
    DevBuf<uint16_t> nbases, ndelta;
    DevBuf<int64_t> nscore;
    DevBuf<int32_t> cov, mval, smin;
    DevBuf<uint8_t> alive, cns_base, cns_cls, lq_kind, lq_nothead;
    DevBuf<uint32_t> cns_pos, lq_next, rflag, rstart, rend, ridx, raw_start, raw_end, headflag, hidx, lq_start,
        lq_end;
    DevBuf<uint32_t> pj, pcount, reg_ncand, reg_bytes, reg_soff, reg_maxlen, blk_sum, blk_coff, blk_soff, kept_read, kept_len, kept_col, cand_off, cand_order, cand_seq_off, kill_ids;
    DevBuf<uint64_t> cand_kmer;
    DevBuf<uint8_t> cand_seq;
    DevBuf<uint16_t> kscore;
    DevBuf<uint32_t> scal; // device scalars: see enum below
    DevBuf<uint32_t> scan_part, scan_poff; // tile sums / offsets of the long scans
    // decoupled look-back state (np2_lookback.hpp): status words per block, ticket counter, launch epoch
    DevBuf<uint64_t> lb_status;
    DevBuf<uint32_t> lb_ticket;
    uint32_t lb_ticket_total = 0, lb_epoch = 0;
    bool lb_dirty = false; // recorded commands of this context were dropped (a failed batch wave): the host-side ticket
                           // base / epoch may be ahead of the device's counters — start the look-back state over
    static constexpr size_t LB_MAX_BLOCKS = 1u << 20;
    DevBuf<uint32_t> mlen; // consensus length after each splice round of the final pass (device-side chain)
    // region logic
    DevBuf<uint8_t> reg_lable, grp, cns_base2, rech_groups, rech_groups_tmp;
    DevBuf<uint8_t> votepack;  // the vote's device-to-host payload in one piece: per-read arrays, row offsets, compact pair words
    DevBuf<uint8_t> votepack2; // ... of the wide form (shards, sort fallback), gathered by copy kernels (batch driver)
    DevBuf<uint32_t> rech_headjobs; // per RECH region: 0 or 1 + the job count of the group it heads
    DevBuf<uint32_t> ecount, eval, eval_s, eflag, eidx, seed_cand, keep_n, keep_list, cns_pos2, sp_idx_s,
        sp_idx_e, ap_g, ap_s, ap_e, rech, rech_joboff, job_len,
        job_off32;
    DevBuf<int32_t> ew, ap_delta, ap_shift;
    DevBuf<uint64_t> ekey, ekey_s;
    DevBuf<uint16_t> keep_ks;
    DevBuf<uint32_t> long_list;
    DevBuf<uint64_t> bt_path;  // recorded backtrack paths of the dirty runs (t_pos << 32 | base << 8 | class)
    DevBuf<uint64_t> chunk_st; // per chunk: launch epoch | non-insertion columns (k_diff_reads)
    uint32_t chunk_epoch = 0;
    DevBuf<uint32_t> tile_cur, tile_n, tile_scan, tile_scanb, tile_nn, tile_nr, tile_noff, tile_roff;
    uint32_t tile_cap = TILE_CAP; // records per tile bucket (tests lower it to force the spill path)
    uint32_t bucket_cap = 0;      // layout of the sorted records of the current contig (0 = compact)
    DevBuf<uint2> nrec;
    DevBuf<uint8_t> votebuf;
    DevBuf<uint32_t> band, band_n, band_off; // banded read-pair accumulator of the phasing vote
    DevBuf<int64_t> run_gain, tile_gain;
    DevBuf<uint8_t> out_snap;
    DevBuf<uint8_t> run_flag; // long runs handed from the eight-lane DP kernel to the per-thread one
    uint32_t deep_min = 65536; // coverage from which the on-chip DP of short runs is off (NP2_TEST_DEEP_COV lowers it: tests)
    DevBuf<uint32_t> lq_list, hbits; // consensus indices of the low-quality bases; bitmap of the raw regions' head indices
    DevBuf<uint16_t> tile_pidx;  // per tile: first record at or beyond every 16th position (k_tile_sort)
    bool pidx_valid = false;     // false after the device-wide sort fallback
    DevBuf<ReadInfo> rinfo;      // per read and pass: descriptor + checkpoint offset + region interval in one line
    DevBuf<uint32_t> lqc, lqoff; // low-quality bases written per dirty run, and their exclusive scan
    DevBuf<uint8_t> pflag;               // per contig position: has exception nodes | coverage below 2
    DevBuf<uint32_t> dp_list; // runs the short-run DP kernel left to the long-run kernels (batch driver: one stream)
    DevBuf<uint16_t> pf_slots; // fused pass front: per-tile consensus entries
    DevBuf<uint32_t> pf_bad, pf_bad2; // ... tiles listed for the middle / the big variant
    DevBuf<uint64_t> pf_prof;  // ... phase timers (NP2_PF_PROF)
    bool front_fused = false;  // the pass front under way went through the fused kernels (np2_passfront.hip)
    // Test and tool switches, read ONCE when the context is created (getenv on every pass of the hot path,
    // racing with setenv elsewhere in the process; a stray variable silently forcing a slow branch for good).  A test sets
    // its variables before it creates its context.
    struct Hooks {
        bool front_unfused = false, pf_prof = false, cand_decode_all = false, edge_sort = false, dp_fork = false;
        bool test_misspeculate = false, shard_spec_log = false, phase_profile = false, exact_grow = false, no_speculate = false;
        bool has_grow_guess = false;
        uint32_t grow_guess = 0;
        bool has_pf_cap = false, has_pf_cap_big = false, has_pf_halo = false, has_pf_cov_max = false;
        uint32_t pf_cap = 0, pf_cap_big = 0, pf_halo = 0, pf_cov_max = 0;
        void read() {
            auto on = [](const char *n) { return getenv(n) != nullptr; };
            auto u32 = [](const char *n, bool &has, uint32_t &v) {
                if (const char *e = getenv(n)) has = true, v = (uint32_t)atol(e);
            };
            front_unfused = on("NP2_FRONT_UNFUSED"), pf_prof = on("NP2_PF_PROF"), cand_decode_all = on("NP2_CAND_DECODE_ALL");
            edge_sort = on("NP2_EDGE_SORT"), dp_fork = on("NP2_DP_FORK"), test_misspeculate = on("NP2_TEST_MISSPECULATE");
            shard_spec_log = on("NP2_SHARD_SPEC_LOG"), phase_profile = on("NP2_PHASE_PROFILE"), exact_grow = on("NP2_EXACT_GROW");
            no_speculate = on("NP2_NO_SPECULATE");
            u32("NP2_TEST_GROW_GUESS", has_grow_guess, grow_guess);
            u32("NP2_PF_CAP", has_pf_cap, pf_cap), u32("NP2_PF_CAP_BIG", has_pf_cap_big, pf_cap_big);
            u32("NP2_PF_HALO", has_pf_halo, pf_halo), u32("NP2_PF_COV_MAX", has_pf_cov_max, pf_cov_max);
        }
    } hooks;
    bool pf_big = getenv("NP2_PF_BIG") != nullptr; // k_pf_tile_big is launched (from the first pass that needed it on; NP2_PF_BIG: always)
    uint32_t front_redos = 0;  // passes the fused front handed back to the unfused kernels (tests read it through the timings)
    DevBuf<uint16_t> kscore_saved;
    DevBuf<uint8_t> sstr;
    DevBuf<uint64_t> soff;
    DevBuf<uint16_t> sscore;
};


namespace np2h {


enum Scal { S_ERR = 0, S_NNODES, S_NRUNS, S_BEST, S_PATHBEGIN, S_NRAW, S_NREG, S_DUP, S_LAST0, S_LAST1, S_GAIN0,
            S_GAIN1, S_DEEP, S_NDPLIST, S_NRECH, S_NGROUPS, S_NLONG, S_NLQ, S_PF, S_M0, S_M1, S_M2, S_M3, S_NC, S_SB, S_GROW,
            // the fused pass front (np2_passfront.hip): S_PF = flag word of the pass under way (above), tiles listed for the big
            // variant, the flags as the host reads them, total of the path-score gains, best end node's relative score
            S_NBAD, S_PFOUT, S_PFGAIN0, S_PFGAIN1, S_PFEND0, S_PFEND1, S_NBAD2, S_PFPAD, S_COUNT = 34 };
static_assert(S_M1 % 2 == 0 && S_LAST0 % 2 == 0 && S_GAIN0 % 2 == 0 && S_PFGAIN0 % 2 == 0 && S_PFEND0 % 2 == 0,
              "64-bit device counters live in these slot pairs");

static constexpr int NP2_MAX_YAK = 15; // splice rounds 0 .. n_yak index mlen[16] and the counters behind S_COUNT
static constexpr uint32_t SCAL_TOTAL = 96; // posted block (S_COUNT) + per-splice-round counters behind it
static_assert(S_COUNT + 2 * (NP2_MAX_YAK + 2) <= SCAL_TOTAL && S_COUNT < 64, "round counters behind the posted block; the mailbox holds 64 words");

inline double thread_cpu_ms() { // CPU time of the calling thread
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
struct WallTimer { // host clock of a stage, and the CPU time the calling thread spent in it ("cpu" + name without "wall")
    np2_ctx *cx;
    const char *name;
    double t0, c0 = 0;
    WallTimer(np2_ctx *c, const char *n);
    ~WallTimer();
};

// HIP-event stage timer.  Event records are packets on the stream, so by default only the timers marked `always`
// (the dense pass, whose duration the benchmark's roofline needs) are armed; np2_ctx_set_timing(ctx, 1) arms all.
struct EventTimer {
    np2_ctx *cx;
    hipEvent_t a = nullptr, b = nullptr;
    bool on;
    EventTimer(np2_ctx *c, const char *name, bool always = false)
        : cx(c), on((always || c->stage_timing) && !tl_recorder()) { // (batched launches are timed by the batch driver)
        if (!on) return;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, cx->stream);
        cx->pending_events.push_back({name, {a, b}});
    }
    ~EventTimer() {
        if (on) (void)hipEventRecord(b, cx->stream);
    }
};

inline WallTimer::WallTimer(np2_ctx *c, const char *n) : cx(c), name(n), t0(c->stage_timing ? now_ms() : 0.0), c0(c->stage_timing ? thread_cpu_ms() : 0.0) {}
inline WallTimer::~WallTimer() {
    if (cx->stage_timing) {
        cx->timing.host.push_back({name, (float)(now_ms() - t0)});
        if (strncmp(name, "wall", 4) == 0) cx->timing.host.push_back({std::string("cpu") + (name + 4), (float)(thread_cpu_ms() - c0)});
    }
}

inline void flush_timings(np2_ctx *cx) {
    std::map<std::string, float> acc;
    std::vector<std::string> order;
    for (auto &e : cx->pending_events) {
        float ms = 0;
        (void)hipEventSynchronize(e.second.second);
        (void)hipEventElapsedTime(&ms, e.second.first, e.second.second);
        if (!acc.count(e.first)) order.push_back(e.first);
        acc[e.first] += ms;
        (void)hipEventDestroy(e.second.first);
        (void)hipEventDestroy(e.second.second);
    }
    cx->pending_events.clear();
    for (auto &h : cx->timing.host) {
        if (!acc.count(h.first)) order.push_back(h.first);
        acc[h.first] += h.second;
    }
    cx->timing.host.clear();
    cx->timing.names = order;
    cx->timing.ms.clear();
    cx->timing.joined.clear();
    for (auto &n : order) {
        cx->timing.ms.push_back(acc[n]);
        cx->timing.joined += n;
        cx->timing.joined.push_back('\0');
    }
    cx->timing.joined.push_back('\0');
}

// ---- stream operations of the per-contig pipeline ------------------------------------------------------------------
// Issued on the context's stream, or recorded when this thread runs under the batch driver (np2_launch.hpp): fills and
// device-to-device copies are kernels (so that they batch across contigs), host transfers are generic recorded
// operations, a synchronisation flushes the whole group.
inline void recorder_sync(Recorder *r) {
    r->sync_fn(r, true);
    for (auto &g : r->graveyard) // (the flush has waited for the device)
        dev_release_idle(g.first, (g.second >> 63) ? 0 : g.second, (g.second >> 63) ? (g.second & ~(1ull << 63)) : 0);
    r->graveyard.clear();
}
// The commands recorded so far are to be issued (with the rest of the batch group's), but the caller goes on with host
// work that does not need their results: no wait for the device, nothing released.  Without a recorder the launches
// went straight to the stream: nothing to do.
np2_ctx *ctx_create_slot(np2_ctx *parent, hipStream_t s, uint32_t *mbox_host, uint32_t *mbox_dev); // (np2_host.cpp)
void ctx_slot_set_stream(np2_ctx *cx, hipStream_t s);
inline void op_submit(np2_ctx *cx) {
    (void)cx;
    if (Recorder *r = tl_recorder()) r->sync_fn(r, false);
}
inline void op_sync(np2_ctx *cx) {
    if (Recorder *r = tl_recorder())
        recorder_sync(r);
    else
        HIPCHK(hipStreamSynchronize(cx->stream));
    cx->h2d_inflight = false;
}
inline void op_fill(np2_ctx *cx, void *p, uint8_t byte, size_t bytes) {
    if (bytes) launch_fill(cx->stream, (uint8_t *)p, bytes, byte);
}
inline void op_copy_d2d(np2_ctx *cx, void *dst, const void *src, size_t bytes) {
    if (bytes) launch_copy(cx->stream, (uint8_t *)dst, (const uint8_t *)src, bytes);
}
// host transfers.  Pinned host memory (hipHostMalloc) is mapped into the device's address space, so under the batch
// driver a transfer is a copy KERNEL reading / writing host memory over the bus: the copies of all contigs of a batch
// go out as one launch instead of one blit per contig.  Large transfers keep the DMA path.
static constexpr size_t KERNEL_COPY_MAX = 1u << 20;
// (device -> host: round 2 sent everything beyond 64 KiB through hipMemcpyAsync — which on this stack is a blit KERNEL
// per transfer anyway, with a 12 us submission gap between two of them: 37 transfers per assembly step.  One merged copy
// kernel for all contigs of a batch has no gaps: +6 % with one batch group, +3 % with four.  NP2_KERNEL_D2H_MAX for A/B.)
static const size_t KERNEL_D2H_MAX = getenv("NP2_KERNEL_D2H_MAX") ? (size_t)atol(getenv("NP2_KERNEL_D2H_MAX")) : (size_t)(4u << 20);
inline void op_d2h(np2_ctx *cx, void *pinned_dst, const void *src, size_t bytes) {
    if (!bytes) return;
    if (Recorder *r = tl_recorder()) {
        if (bytes <= KERNEL_D2H_MAX)
            launch_copy(cx->stream, (uint8_t *)pinned_dst, (const uint8_t *)src, bytes);
        else
            r->push_fn([=](hipStream_t s) { HIPCHK(hipMemcpyAsync(pinned_dst, src, bytes, hipMemcpyDeviceToHost, s)); });
    } else
        HIPCHK(hipMemcpyAsync(pinned_dst, src, bytes, hipMemcpyDeviceToHost, cx->stream));
}
// device -> host of min(*n_dev * elem, cap_bytes) bytes (true: done; false: not this way — the caller copies cap_bytes)
inline bool op_d2h_len(np2_ctx *cx, void *pinned_dst, const void *src, const uint32_t *n_dev, uint32_t elem, size_t cap_bytes) {
    if (!tl_recorder() || cap_bytes > KERNEL_D2H_MAX) return false;
    launch_copy_len(cx->stream, (uint8_t *)pinned_dst, (const uint8_t *)src, n_dev, elem, cap_bytes);
    return true;
}
inline void op_h2d(np2_ctx *cx, void *dst, const void *pinned_src, size_t bytes) {
    if (!bytes) return;
    if (Recorder *r = tl_recorder()) {
        if (bytes <= KERNEL_COPY_MAX)
            launch_copy(cx->stream, (uint8_t *)dst, (const uint8_t *)pinned_src, bytes);
        else
            r->push_fn([=](hipStream_t s) { HIPCHK(hipMemcpyAsync(dst, pinned_src, bytes, hipMemcpyHostToDevice, s)); });
    } else
        HIPCHK(hipMemcpyAsync(dst, pinned_src, bytes, hipMemcpyHostToDevice, cx->stream));
}

template <class T> std::vector<T> d2h(np2_ctx *cx, const T *d, size_t n) {
    std::vector<T> v(n);
    if (n) {
        void *pin = cx->pin_d2h.ensure(n * sizeof(T));
        op_d2h(cx, pin, d, n * sizeof(T));
        op_sync(cx);
        memcpy(v.data(), pin, n * sizeof(T));
    }
    return v;
}
// Read the device scalar block (enum Scal) with low latency: a one-thread kernel first gathers up to four device
// counters into scal slots (dst = scal + S_x), then posts the block to the host-mapped mailbox; the host spins on the
// sequence word.  Everything queued on the stream before the call has completed when it returns.
inline std::vector<uint32_t> fetch_scal(np2_ctx *cx, uint32_t *d0 = nullptr, const uint32_t *s0 = nullptr,
                                        uint32_t *d1 = nullptr, const uint32_t *s1 = nullptr, uint32_t *d2 = nullptr,
                                        const uint32_t *s2 = nullptr, uint32_t *d3 = nullptr,
                                        const uint32_t *s3 = nullptr, const uint32_t *ends_of = nullptr,
                                        uint32_t *ends_dst = nullptr);

// host -> device through the pinned staging buffer (valid until the next h2d_staged or an explicit sync)
inline void h2d_staged(np2_ctx *cx, void *dst, const void *src, size_t bytes) {
    if (!bytes) return;
    if (cx->h2d_inflight) op_sync(cx); // the staging buffer may still be in flight (cleared by every synchronisation)
    void *pin = cx->pin_h2d.ensure(bytes);
    memcpy(pin, src, bytes);
    op_h2d(cx, dst, pin, bytes);
    cx->h2d_inflight = true;
}
template <class T> void trace_put(np2_ctx *cx, int pass, const std::string &name, const std::vector<T> &v) {
    if (!cx->trace) return;
    auto &dst = cx->trace_items[std::to_string(pass) + ":" + name];
    dst.resize(v.size() * sizeof(T));
    if (!v.empty()) memcpy(dst.data(), v.data(), dst.size());
}

inline std::vector<uint32_t> fetch_scal(np2_ctx *cx, uint32_t *d0, const uint32_t *s0, uint32_t *d1, const uint32_t *s1,
                                        uint32_t *d2, const uint32_t *s2, uint32_t *d3, const uint32_t *s3,
                                        const uint32_t *ends_of, uint32_t *ends_dst) {
    const uint32_t seq = ++cx->mbox_seq;
    launch_post(cx->stream, cx->scal.p, S_COUNT, cx->mbox_dev, seq, d0, s0, d1, s1, d2, s2, d3, s3, ends_of, ends_dst);
    cx->h2d_inflight = false;
    if (Recorder *r = tl_recorder()) { // batch driver: flush the group's queues and wait for the device
        recorder_sync(r);
        if (__atomic_load_n(&cx->mbox_host[0], __ATOMIC_ACQUIRE) != seq)
            throw Np2Error(NP2_E_DEVICE, "mailbox not posted after a batch flush");
        return std::vector<uint32_t>(cx->mbox_host + 1, cx->mbox_host + 1 + S_COUNT);
    }
    uint64_t spins = 0;
    HostWait hw;
    while (__atomic_load_n(&cx->mbox_host[0], __ATOMIC_ACQUIRE) != seq) {
        hw.pause();
        if ((++spins & (wait_naps() ? 0xFFu : 0xFFFFu)) == 0) { // a failed launch / device fault would never post: surface it
            hipError_t e = hipStreamQuery(cx->stream);
            if (e != hipSuccess && e != hipErrorNotReady)
                throw Np2Error(NP2_E_DEVICE, std::string("device error while waiting for the mailbox: ") + hipGetErrorString(e));
            if (e == hipSuccess && __atomic_load_n(&cx->mbox_host[0], __ATOMIC_ACQUIRE) != seq)
                throw Np2Error(NP2_E_DEVICE, "mailbox kernel completed without posting");
        }
    }
    return std::vector<uint32_t>(cx->mbox_host + 1, cx->mbox_host + 1 + S_COUNT);
}

// short arrays: one single-block kernel (no temp storage, no init launch); long ones: rocPRIM
static constexpr size_t SCAN_SMALL_MAX = 1u << 16;
// look-back descriptor for one launch of n_blocks blocks (fresh epoch, ticket base advanced)
inline Lookback next_lookback(np2_ctx *cx, uint32_t n_blocks) {
    if (n_blocks > np2_ctx::LB_MAX_BLOCKS) throw Np2Error(NP2_E_UNSUPPORTED, "look-back scan over too many blocks");
    if (!cx->lb_status.p) {
        cx->lb_status.ensure(2 * np2_ctx::LB_MAX_BLOCKS);
        cx->lb_ticket.ensure(4);
        op_fill(cx, cx->lb_status.p, 0, 2 * np2_ctx::LB_MAX_BLOCKS * sizeof(uint64_t));
        op_fill(cx, cx->lb_ticket.p, 0, 16);
        cx->lb_ticket_total = 0;
        cx->lb_epoch = 0;
    }
    if (cx->lb_dirty) { // look-back launches were recorded and never issued: device ticket counter != host ticket base
        op_fill(cx, cx->lb_status.p, 0, 2 * np2_ctx::LB_MAX_BLOCKS * sizeof(uint64_t));
        op_fill(cx, cx->lb_ticket.p, 0, 16);
        cx->lb_ticket_total = 0;
        cx->lb_epoch = 0;
        cx->lb_dirty = false;
    }
    if (++cx->lb_epoch >= (1u << 30)) { // epochs exhausted: start over with cleared status words
        op_fill(cx, cx->lb_status.p, 0, 2 * np2_ctx::LB_MAX_BLOCKS * sizeof(uint64_t));
        cx->lb_epoch = 1;
    }
    Lookback lb{cx->lb_status.p, cx->lb_status.p + np2_ctx::LB_MAX_BLOCKS, cx->lb_ticket.p, cx->lb_ticket_total,
                cx->lb_epoch, n_blocks, cx->scal.p + S_ERR};
    cx->lb_ticket_total += n_blocks;
    return lb;
}

// a rocPRIM call (device-wide sort / long signed scans): runs on the context's stream, or is recorded as one
// un-batched operation of this contig's queue
template <class F> inline void prim_op(np2_ctx *cx, F f) {
    if (Recorder *r = tl_recorder())
        r->push_fn(f);
    else
        f(cx->stream);
}
// Exclusive sums of host-known length, any length: ONE kernel, blocks of 8192 elements chained by the decoupled
// look-back (k_scan_lb_excl, np2_cand.hip).  Not a choice by length: a threshold had the contigs of a batch pick
// different kernels for the same step, their queues fell out of step and every later stage went out in more, smaller
// launches.  NP2_SCAN3=1 brings back the three-launch reduce-then-scan for long arrays (A/B measurements).
inline void scan_large_excl(np2_ctx *cx, const uint32_t *in, uint32_t *out, size_t n, bool write_end = false) {
    if (n >= 0xFFFFF000ull) throw Np2Error(NP2_E_NOMEM, "scan over more than 2^32 elements");
    static const bool scan3 = getenv("NP2_SCAN3") != nullptr;
    if (!scan3 || n <= SCAN_SMALL_MAX) {
        launch_scan_lb_excl(cx->stream, next_lookback(cx, scan_lb_blocks(n)), in, out, (uint32_t)n, write_end,
                            cx->scal.p + S_ERR);
        return;
    }
    const uint32_t nt = scan3_tiles((uint32_t)n);
    cx->scan_part.ensure((size_t)nt + 2);
    cx->scan_poff.ensure((size_t)nt + 2);
    launch_scan3_excl(cx->stream, in, out, (uint32_t)n, cx->scan_part.p, cx->scan_poff.p, write_end);
}

inline uint32_t exclusive_total(np2_ctx *cx, const uint32_t *in, uint32_t *out, size_t n_plus1) {
    // scans n_plus1 elements (caller guarantees in[n_plus1-1] == 0); out[n_plus1-1] = the total, on the device
    scan_large_excl(cx, in, out, n_plus1);
    return 0;
}
inline void scan_incl_min(np2_ctx *cx, const int32_t *in, int32_t *out, size_t n) {
    if (n <= SCAN_SMALL_MAX) {
        launch_scan_small_min(cx->stream, in, out, (uint32_t)n, nullptr);
    } else {
        prim_op(cx, [=](hipStream_t s) {
            if (prim_inclusive_min_i32(s, cx->tmp.p, cx->tmp.cap, in, out, n)) throw Np2Error(NP2_E_DEVICE, "rocprim min-scan failed");
        });
    }
}
inline void scan_incl_sum(np2_ctx *cx, const int32_t *in, int32_t *out, size_t n) {
    if (n <= SCAN_SMALL_MAX) {
        launch_scan_small_incl(cx->stream, in, out, (uint32_t)n, nullptr);
    } else {
        prim_op(cx, [=](hipStream_t s) {
            if (prim_inclusive_sum_i32(s, cx->tmp.p, cx->tmp.cap, in, out, n)) throw Np2Error(NP2_E_DEVICE, "rocprim inclusive_scan failed");
        });
    }
}
inline void zero32(np2_ctx *cx, void *p, size_t n_elems, size_t elem = 4) {
    op_fill(cx, p, 0, n_elems * elem);
}
// exclusive sums of in[0..n) into out[0..n], out[n] = total
inline void exclusive_total_n(np2_ctx *cx, uint32_t *in, uint32_t *out, size_t n) {
    scan_large_excl(cx, in, out, n, true);
}

// validate the read descriptors, build checkpoint offsets / chunk tables / contig codes for a contig whose
// reads (host copy given) and nibble buffer are already resident
void finish_contig(np2_ctx *cx, np2_contig *c, const np2_read_t *reads, uint32_t n_reads, uint32_t L,
                   uint64_t nib_bytes);
int fail(np2_ctx *cx, const Np2Error &e);

} // namespace np2h
