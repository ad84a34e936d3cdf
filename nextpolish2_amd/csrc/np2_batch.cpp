// Batch driver: polishes several contigs of one assembly at once on one GPU, with one kernel launch per pipeline step
// for the whole batch instead of one per contig.
//
// The reference hands one contig to each worker thread (src/main.rs:1717-1843) and the per-contig loop
// (main.rs:1819-1836) shares nothing between contigs.  On the GPU a contig of a few hundred kb is a few microseconds
// of work per kernel, so a many-contig assembly is bound by the number of launches.  Here every contig of a batch runs
// the unchanged per-contig host pipeline (np2_polish_resident) on its own host thread and its own scratch context,
// but the threads *record* their device commands (np2_launch.hpp).  Whenever every pipeline of the batch waits for the
// device (a read-back of counters, a result copy), the last thread to arrive merges the recorded queues: commands
// keep their per-contig order, and launches of the same kernel at the heads of several queues go out as ONE grid
// (k_np2_batched).  Host-side phases — the Louvain phasing vote above all — run concurrently on the contigs' threads.
#include "../../include/np2.h"
#include "np2_ctx.hpp"

#include <atomic>
#include <climits>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <pthread.h>

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

using namespace np2;

namespace {

struct Job {
    np2_contig *contig = nullptr;
    const np2_opts_t *opts = nullptr;
    uint8_t **out_bases = nullptr;
    uint32_t **out_pos = nullptr;
    uint64_t *out_len = nullptr;
    uint32_t *out_span = nullptr;
    int *rc = nullptr;
};

} // namespace

struct np2_batch {
    np2_ctx *parent = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<np2_ctx *> slots;
    std::vector<Recorder> recs;
    std::vector<std::thread> workers;
    std::string err;

    // job hand-off (one generation = one wave of at most slots.size() contigs)
    std::mutex mu;
    std::condition_variable cv_start, cv_done;
    uint64_t job_gen = 0;
    std::vector<Job> jobs; // jobs[slot]
    int n_running = 0;
    bool quit = false;

    // device synchronisation of the wave: counts are guarded by `sync_mu`, waiters spin on `flush_gen`
    std::mutex sync_mu;
    int n_active = 0, n_waiting = 0;
    std::atomic<uint32_t> flush_gen{0}; // (32 bits: waiters sleep on it with futex(2) after a short spin)
    std::atomic<bool> failed{false};
    std::string fail_msg;

    // completion word of a flush: host-mapped, written by the last kernel of the flush
    uint32_t *done_host = nullptr, *done_dev = nullptr;
    uint32_t done_seq = 0;
    std::atomic<uint32_t> issued_seq{0}; // completion sequence of the last flush that has been issued
    std::atomic<uint32_t> done_pub{0};   // ... of the last flush known to have completed (the waiters sleep on it)
    std::atomic<bool> spinner{false};    // one pipeline at a time polls the device's completion word for all of them
    double t_issued_last = 0;            // when it was
    uint32_t wait_logged_seq = 0;        // the flush whose device wait has been entered in the log (once per flush)

    // HIP-event timing of the batched dense kernel (bench roofline): pairs recorded around its launches
    bool time_diff = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> diff_events; // created once, reused by every call
    size_t diff_used = 0;
    float last_diff_ms = 0;
    int last_diff_launches = 0;
    uint64_t stat_launches = 0, stat_cmds = 0, stat_flushes = 0;
    // where a wave's wall time goes (ms, cumulative): host phases between flushes, issuing commands, waiting for the GPU
    double t_last_end = 0, ms_host = 0, ms_issue = 0, ms_wait = 0;
    std::vector<double> flush_log; // per flush of the last polish call: host, issue, wait (ms)
    double call_ms = 0, tail_ms = 0; // the last polish call, measured inside it: whole / after its last flush
};

namespace {

// Recorder::sync_fn: wait until every running pipeline of the wave has reached a synchronisation point; the last one
// to arrive flushes for all.
// A pipeline that reached its synchronisation point before the others waits for the group's flush: a short spin (the
// common case — the pipelines of a wave arrive within microseconds of each other and a flush of a few kernels is over
// in tens of microseconds), then it sleeps on the generation word.  (Spinning for the whole wait had every waiting
// pipeline of every batch group hold a core: 60+ busy host threads for one GPU.)
inline void wait_generation(std::atomic<uint32_t> &gen, uint32_t seen) {
    // (30 us until round 4: 1.5 CPUs of a rank's 6.9 went into it for no measurable gain — profiles/r04_host_cpu.txt)
    static const double spin_us = getenv("NP2_BATCH_SPIN_US") ? atof(getenv("NP2_BATCH_SPIN_US")) : (wait_naps() ? 0.0 : 5.0);
    const double t_end = now_ms() + spin_us * 1e-3;
    for (;;) {
        for (int i = 0; i < 64; ++i) {
            if (gen.load(std::memory_order_acquire) != seen) return;
            __builtin_ia32_pause();
        }
        if (now_ms() >= t_end) break;
    }
    while (gen.load(std::memory_order_acquire) == seen)
        (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(&gen), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
}
inline void publish_generation(std::atomic<uint32_t> &gen, uint32_t next) {
    gen.store(next, std::memory_order_release);
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(&gen), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}

// one-thread kernel that marks the end of a flush in host-mapped memory
__device__ __forceinline__ void k_flush_done(const uint32_t np2_bid, const uint32_t np2_nb, uint32_t *__restrict__ word, uint32_t seq) {
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Issue every recorded command of the active slots on the batch stream, merging equal kernels at the queue heads; the
// completion word of the flush (b->issued_seq) is posted by its last kernel.  Called with sync_mu held by the last
// thread that arrived; the WAIT for the device is the business of the pipelines that need it (wait_done): a pipeline
// that only wanted its commands on their way goes on at once.
void flush(np2_batch *b) {
    const int n = (int)b->recs.size();
    const double t_begin = now_ms();
    double t_issued = t_begin;
    std::vector<size_t> idx(n, 0);
    hipStream_t s = b->stream;
    Recorder *saved = tl_recorder();
    tl_recorder() = nullptr; // the launches below are real
    try {
        for (;;) {
            // generic operations go out as soon as they reach the head of their queue
            bool any = false;
            for (int i = 0; i < n; ++i) {
                auto &q = b->recs[i].q;
                while (idx[i] < q.size() && !q[idx[i]].kd) {
                    q[idx[i]].fn(s);
                    ++idx[i];
                    ++b->stat_cmds;
                }
                any |= idx[i] < q.size();
            }
            if (!any) break;
            // Which of the kernels at the queue heads goes out now?  One that no other queue is about to reach: if kernel K
            // heads some queues and lies a few commands ahead in others, launching it now would launch it again for
            // those others a moment later — and they would stay one step behind for the rest of the flush, every later
            // stage going out twice (a contig that takes an extra kernel, or another variant of one, is enough).  So the
            // head that nobody else has in its near future goes first (the stragglers catch up); ties: the kernel most
            // queues are waiting for, then the lowest slot.
            const KernelDesc *best = nullptr;
            int best_n = 0, best_defer = 0;
            static constexpr size_t LOOKAHEAD = 8;
            for (int i = 0; i < n; ++i) {
                auto &q = b->recs[i].q;
                if (idx[i] >= q.size()) continue;
                const KernelDesc *kd = q[idx[i]].kd;
                bool seen = false; // (counted when its first queue came by)
                for (int j = 0; j < i && !seen; ++j)
                    seen = idx[j] < b->recs[j].q.size() && b->recs[j].q[idx[j]].kd == kd;
                if (seen) continue;
                int c = 0, defer = 0;
                for (int j = 0; j < n; ++j) {
                    auto &qj = b->recs[j].q;
                    if (idx[j] >= qj.size()) continue;
                    if (qj[idx[j]].kd == kd) {
                        ++c;
                        continue;
                    }
                    for (size_t k = idx[j] + 1; k < qj.size() && k <= idx[j] + LOOKAHEAD; ++k)
                        if (qj[k].kd == kd) {
                            ++defer;
                            break;
                        }
                }
                if (!best || defer < best_defer || (defer == best_defer && c > best_n)) best = kd, best_n = c, best_defer = defer;
            }
            uint32_t grids[MAXB];
            const void *args[MAXB];
            int m = 0;
            const bool timed = b->time_diff && strcmp(best->name, "k_diff_reads") == 0;
            auto issue = [&]() {
                if (!m) return;
                if (timed) {
                    if (b->diff_used == b->diff_events.size()) {
                        hipEvent_t e0 = nullptr, e1 = nullptr;
                        HIPCHK(hipEventCreate(&e0));
                        HIPCHK(hipEventCreate(&e1));
                        b->diff_events.push_back({e0, e1});
                    }
                    HIPCHK(hipEventRecord(b->diff_events[b->diff_used].first, s));
                }
                best->launch(s, m, grids, args);
                if (timed) {
                    HIPCHK(hipEventRecord(b->diff_events[b->diff_used].second, s));
                    ++b->diff_used;
                }
                ++b->stat_launches;
                m = 0;
            };
            for (int i = 0; i < n; ++i) {
                auto &r = b->recs[i];
                if (idx[i] >= r.q.size() || r.q[idx[i]].kd != best) continue;
                grids[m] = r.q[idx[i]].grid;
                args[m] = r.arena.data() + r.q[idx[i]].arg_off;
                ++m;
                ++idx[i];
                ++b->stat_cmds;
                if (m == best->max_batch) issue();
            }
            issue();
        }
        HIPCHK(hipGetLastError());
        // completion: a last one-thread kernel posts a sequence number the host spins on (a stream synchronisation
        // costs a few tens of microseconds more per flush)
        const uint32_t seq = ++b->done_seq;
        NP2_LAUNCH(k_flush_done, 1, 64, s, b->done_dev, seq);
        t_issued = now_ms();
        b->issued_seq.store(seq, std::memory_order_release);
    } catch (const std::exception &ex) {
        b->fail_msg = ex.what();
        b->failed.store(true);
        (void)hipStreamSynchronize(s);
        // whatever was not issued is dropped below: the slots' host-side look-back state no longer matches the device
        for (np2_ctx *cx : b->slots) cx->lb_dirty = true, cx->h2d_inflight = false;
    }
    tl_recorder() = saved;
    for (auto &r : b->recs) r.clear();
    ++b->stat_flushes;
    b->ms_host += t_begin - b->t_last_end;
    b->ms_issue += t_issued - t_begin;
    b->flush_log.push_back(t_begin - b->t_last_end);
    b->flush_log.push_back(t_issued - t_begin);
    b->flush_log.push_back(0.0); // (device wait: entered by the first pipeline that waits for this flush, wait_done)
    b->t_last_end = t_issued;
    b->t_issued_last = t_issued;
}

// wait until the flush with completion sequence `seq` (or a later one) has run on the device.  The device posts the
// sequence number in host-mapped memory, which nothing can sleep on: ONE of the waiting pipelines polls it and publishes
// what it saw, the others sleep on the published word (every waiter polling for itself had 17 threads spinning through
// the other pipelines' host phases: 13 of the container's 16 CPUs busy).
void wait_done(np2_batch *b, uint32_t seq) {
    for (;;) {
        const uint32_t pub = b->done_pub.load(std::memory_order_acquire);
        if ((int32_t)(pub - seq) >= 0 || b->failed.load()) break;
        bool expected = false;
        if (b->spinner.compare_exchange_strong(expected, true)) {
            uint64_t spins = 0;
            uint32_t cur;
            HostWait hw; // (spins, or naps when the process is short of CPUs: np2_ctx.hpp)
            while ((int32_t)((cur = __atomic_load_n(b->done_host, __ATOMIC_ACQUIRE)) - seq) < 0) {
                if (b->failed.load()) break;
                if ((++spins & (wait_naps() ? 0xFFu : 0xFFFFu)) == 0) {
                    hipError_t e = hipStreamQuery(b->stream);
                    if (e != hipSuccess && e != hipErrorNotReady) {
                        b->fail_msg = std::string("device error during a batch flush: ") + hipGetErrorString(e);
                        b->failed.store(true);
                        break;
                    }
                    if (e == hipSuccess && (int32_t)(__atomic_load_n(b->done_host, __ATOMIC_ACQUIRE) - seq) < 0) {
                        b->fail_msg = "batch flush completed without posting";
                        b->failed.store(true);
                        break;
                    }
                }
                hw.pause();
            }
            b->spinner.store(false);
            publish_generation(b->done_pub, b->failed.load() ? pub + 1 : cur); // (wakes the sleepers either way)
            if (b->failed.load()) return;
            continue;
        }
        wait_generation(b->done_pub, pub);
    }
    if (b->failed.load()) return;
    std::lock_guard<std::mutex> l(b->sync_mu);
    if (b->wait_logged_seq != seq && b->issued_seq.load() == seq && !b->flush_log.empty()) { // first waiter of the newest flush
        const double t = now_ms();
        b->wait_logged_seq = seq;
        b->ms_wait += t - b->t_issued_last;
        b->flush_log.back() = t - b->t_issued_last;
        b->t_last_end = t;
    }
}

void group_sync(Recorder *r, bool need_done) {
    np2_batch *b = (np2_batch *)r->group;
    uint32_t my_gen;
    bool flushed = false;
    {
        std::lock_guard<std::mutex> l(b->sync_mu);
        my_gen = b->flush_gen.load(std::memory_order_relaxed);
        if (++b->n_waiting == b->n_active) {
            flush(b);
            b->n_waiting = 0;
            publish_generation(b->flush_gen, my_gen + 1);
            flushed = true;
        }
    }
    if (!flushed) wait_generation(b->flush_gen, my_gen);
    // (a later flush may have been issued by now: its sequence number is larger and its completion implies ours)
    if (need_done && !b->failed.load()) wait_done(b, b->issued_seq.load(std::memory_order_acquire));
    if (b->failed.load()) throw Np2Error(NP2_E_DEVICE, "batch flush failed: " + b->fail_msg);
}

// a pipeline left the wave (finished or failed): the others must not wait for it
void leave_wave(np2_batch *b, Recorder *r) {
    std::lock_guard<std::mutex> l(b->sync_mu);
    if (!r->q.empty()) { // commands recorded after the last synchronisation of a failed pipeline are dropped: its
                         // context's look-back tickets / epochs advanced at record time and must start over
        np2_ctx *cx = b->slots[r->slot];
        cx->lb_dirty = true;
        cx->h2d_inflight = false;
    }
    r->clear();
    --b->n_active;
    if (b->n_active > 0 && b->n_waiting == b->n_active) {
        const uint32_t g = b->flush_gen.load(std::memory_order_relaxed);
        flush(b);
        b->n_waiting = 0;
        publish_generation(b->flush_gen, g + 1);
    }
}

void worker_main(np2_batch *b, int slot) {
    (void)hipSetDevice(b->device);
    (void)pthread_setname_np(pthread_self(), "np2-slot"); // (bench.py's per-thread CPU accounting reads the names)
    uint64_t seen = 0;
    for (;;) {
        Job job;
        {
            std::unique_lock<std::mutex> l(b->mu);
            b->cv_start.wait(l, [&] { return b->quit || b->job_gen != seen; });
            if (b->quit) return;
            seen = b->job_gen;
            job = b->jobs[slot];
        }
        if (!job.contig) continue; // this slot has no contig in this wave
        Recorder *r = &b->recs[slot];
        tl_recorder() = r;
        int rc = np2_polish_resident(b->slots[slot], job.contig, job.opts, job.out_bases, job.out_pos, job.out_len);
        tl_recorder() = nullptr;
        for (auto &g : r->graveyard) // (the pipeline's last flush has completed)
            dev_release_idle(g.first, (g.second >> 63) ? 0 : g.second, (g.second >> 63) ? (g.second & ~(1ull << 63)) : 0);
        r->graveyard.clear();
        leave_wave(b, r);
        if (rc == NP2_OK && job.out_span) (void)np2_last_span(b->slots[slot], &job.out_span[0], &job.out_span[1]);
        *job.rc = rc;
        {
            std::lock_guard<std::mutex> l(b->mu);
            if (--b->n_running == 0) b->cv_done.notify_all();
        }
    }
}

} // namespace

extern "C" {

int np2_batch_create(np2_batch_t **out, np2_ctx_t *parent, int n_slots) {
    if (!out || !parent || n_slots < 1 || n_slots > 256) return NP2_E_ARG;
    *out = nullptr;
    np2_batch *b = new np2_batch();
    b->parent = parent;
    b->device = parent->device;
    try {
        HIPCHK(hipSetDevice(b->device));
        HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
        // one host-mapped block: the completion word of the flushes, then a 64-word mailbox per slot
        HIPCHK(hipHostMalloc((void **)&b->done_host, 256 * (size_t)(n_slots + 1), hipHostMallocMapped | hipHostMallocCoherent));
        memset(b->done_host, 0, 256 * (size_t)(n_slots + 1));
        HIPCHK(hipHostGetDevicePointer((void **)&b->done_dev, b->done_host, 0));
        b->recs.resize(n_slots);
        b->jobs.resize(n_slots);
        for (int i = 0; i < n_slots; ++i) {
            // (a slot only records commands: it takes the batch's stream and a piece of its mailbox block instead of three
            // streams and a pinned block of its own)
            np2_ctx *cx = ctx_create_slot(parent, b->stream, b->done_host + 64 * (size_t)(i + 1), b->done_dev + 64 * (size_t)(i + 1));
            b->slots.push_back(cx);
            b->recs[i].group = b;
            b->recs[i].slot = i;
            b->recs[i].sync_fn = &group_sync;
        }
        for (int i = 0; i < n_slots; ++i) b->workers.emplace_back(worker_main, b, i);
    } catch (const Np2Error &e) {
        parent->err = e.what();
        int code = e.code;
        np2_batch_destroy(b);
        return code;
    }
    *out = b;
    return NP2_OK;
}

void np2_batch_destroy(np2_batch_t *b) {
    if (!b) return;
    {
        std::lock_guard<std::mutex> l(b->mu);
        b->quit = true;
    }
    b->cv_start.notify_all();
    for (auto &t : b->workers) t.join();
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    {
        DevSyncScope idle; // (one wait for the device, not one per slot)
        for (np2_ctx *cx : b->slots) np2_ctx_destroy(cx);
    }
    for (auto &e : b->diff_events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (b->done_host) (void)hipHostFree(b->done_host);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

// Several batches on one device (one host thread each) fill each other's host phases — the phasing vote of a group of
// contigs is a millisecond of Louvain on the CPU — but equal-priority streams tend to advance in lockstep and end up
// in their host phases together.  Alternating priorities keep the groups out of step.
int np2_batch_set_priority(np2_batch_t *b, int high) {
    if (!b) return NP2_E_ARG;
    try {
        HIPCHK(hipSetDevice(b->device));
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamSynchronize(b->stream)); // (no call in flight: the caller's contract)
        hipStream_t s = nullptr;
        HIPCHK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high ? greatest : least));
        (void)hipStreamDestroy(b->stream);
        b->stream = s;
        for (np2_ctx *cx : b->slots) ctx_slot_set_stream(cx, s);
    } catch (const Np2Error &e) {
        b->err = e.what();
        return e.code;
    }
    return NP2_OK;
}

int np2_batch_slots(np2_batch_t *b) { return b ? (int)b->slots.size() : 0; }
np2_ctx_t *np2_batch_slot_ctx(np2_batch_t *b, int slot) {
    return (b && slot >= 0 && slot < (int)b->slots.size()) ? b->slots[slot] : nullptr;
}
const char *np2_batch_last_error(np2_batch_t *b) { return b ? b->err.c_str() : "null batch"; }
int np2_batch_set_sink(np2_batch_t *b, int slot, void *device_ptr, uint64_t cap) {
    if (!b || slot < 0 || slot >= (int)b->slots.size()) return NP2_E_ARG;
    b->slots[slot]->sink_dst = (uint8_t *)device_ptr;
    b->slots[slot]->sink_cap = device_ptr ? cap : 0;
    return NP2_OK;
}

int np2_batch_polish(np2_batch_t *b, np2_contig_t *const *contigs, int n, const np2_opts_t *opts, uint8_t **out_bases,
                     uint32_t **out_pos, uint64_t *out_len, uint32_t *out_span, int *rcs) {
    if (!b || !contigs || n < 0 || !opts || !out_len || !rcs) return NP2_E_ARG;
    const int S = (int)b->slots.size();
    int worst = NP2_OK;
    b->diff_used = 0;
    b->flush_log.clear();
    b->t_last_end = now_ms();
    const double t_call0 = b->t_last_end;
    for (int w0 = 0; w0 < n; w0 += S) { // waves of at most S contigs; contig w0 + i runs on slot i
        const int m = std::min(S, n - w0);
        {
            std::lock_guard<std::mutex> l(b->mu);
            for (int i = 0; i < S; ++i) {
                Job j;
                if (i < m) {
                    j.contig = contigs[w0 + i];
                    j.opts = opts;
                    j.out_bases = out_bases ? &out_bases[w0 + i] : nullptr;
                    j.out_pos = out_pos ? &out_pos[w0 + i] : nullptr;
                    j.out_len = &out_len[w0 + i];
                    j.out_span = out_span ? &out_span[2 * (w0 + i)] : nullptr;
                    j.rc = &rcs[w0 + i];
                }
                b->jobs[i] = j;
            }
            b->n_running = m;
            {
                std::lock_guard<std::mutex> l2(b->sync_mu);
                b->n_active = m;
                b->n_waiting = 0;
                b->failed.store(false);
            }
            ++b->job_gen;
        }
        b->cv_start.notify_all();
        {
            std::unique_lock<std::mutex> l(b->mu);
            b->cv_done.wait(l, [&] { return b->n_running == 0; });
        }
        for (int i = 0; i < m; ++i)
            if (rcs[w0 + i] != NP2_OK) {
                worst = rcs[w0 + i];
                b->err = std::string("contig ") + std::to_string(w0 + i) + ": " + np2_last_error(b->slots[i]);
            }
    }
    b->call_ms = now_ms() - t_call0;
    b->tail_ms = now_ms() - b->t_last_end;
    if (b->time_diff) {
        b->last_diff_ms = 0;
        b->last_diff_launches = (int)b->diff_used;
        for (size_t i = 0; i < b->diff_used; ++i) {
            float ms = 0;
            (void)hipEventSynchronize(b->diff_events[i].second);
            (void)hipEventElapsedTime(&ms, b->diff_events[i].first, b->diff_events[i].second);
            b->last_diff_ms += ms;
        }
    }
    return worst;
}

void np2_batch_set_timing(np2_batch_t *b, int enable) {
    if (b) b->time_diff = enable != 0;
}
int np2_batch_last_diff_ms(np2_batch_t *b, float *ms, int *launches) {
    if (!b || !ms || !launches) return NP2_E_ARG;
    *ms = b->last_diff_ms;
    *launches = b->last_diff_launches;
    return NP2_OK;
}
int np2_batch_last_call_ms(np2_batch_t *b, double *total_ms, double *tail_ms) {
    if (!b || !total_ms || !tail_ms) return NP2_E_ARG;
    *total_ms = b->call_ms;
    *tail_ms = b->tail_ms;
    return NP2_OK;
}
// per flush of the last np2_batch_polish: (host phase before it, command issue, device wait) in ms; returns the count
int np2_batch_flush_log(np2_batch_t *b, const double **log) {
    if (!b || !log) return 0;
    *log = b->flush_log.data();
    return (int)(b->flush_log.size() / 3);
}
int np2_batch_stats(np2_batch_t *b, uint64_t *launches, uint64_t *commands, uint64_t *flushes) {
    if (!b || !launches || !commands || !flushes) return NP2_E_ARG;
    *launches = b->stat_launches;
    *commands = b->stat_cmds;
    *flushes = b->stat_flushes;
    return NP2_OK;
}
}
