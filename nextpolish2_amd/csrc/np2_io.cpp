// Input side of the hot path (include/np2_io.h): FASTA[.gz], yak v2 dumps, indexed BAM (BGZF + BAI on zlib),
// record admission (src/main.rs:1758-1817) and the GPU columnariser orchestration.
#include "../../include/np2_io.h"
#include "np2_ctx.hpp"

#include <algorithm>
#include <zlib.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <unordered_set>

#include "np2_inflate.hpp"
#include "np2_inflate_core.hpp"
namespace {

// Raw-deflate decoder for BGZF blocks (each block is one complete deflate stream of known inflated size): libdeflate's
// whole-buffer decoder when the host has its runtime library (2-3 x zlib's rate on BAM payloads; htslib makes the same
// choice at build time), zlib otherwise or with NP2_INFLATE=zlib.  Resolved once, at run time: the image ships
// libdeflate.so.0 without headers, and a host without it must still work.
struct Inflater {
    typedef void *(*alloc_fn)(void);
    typedef int (*dec_fn)(void *, const void *, size_t, void *, size_t, size_t *);
    typedef void (*free_fn)(void *);
    alloc_fn alloc = nullptr;
    dec_fn dec = nullptr;
    free_fn fre = nullptr;
    Inflater() {
        const char *e = getenv("NP2_INFLATE");
        if (e && !strcmp(e, "zlib")) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (alloc_fn)dlsym(h, "libdeflate_alloc_decompressor");
        dec = (dec_fn)dlsym(h, "libdeflate_deflate_decompress");
        fre = (free_fn)dlsym(h, "libdeflate_free_decompressor");
        if (!alloc || !dec || !fre) alloc = nullptr, dec = nullptr, fre = nullptr;
    }
    static Inflater &get() {
        static Inflater *i = new Inflater();
        return *i;
    }
    const char *name() const { return dec ? "libdeflate" : "zlib"; }
    struct PerThread { // one decoder per thread, released with the thread
        void *d = nullptr;
        free_fn fre = nullptr;
        ~PerThread() {
            if (d && fre) fre(d);
        }
    };
    // inflates `clen` bytes at `in` into exactly `isize` bytes at `out`
    bool run(const uint8_t *in, size_t clen, uint8_t *out, size_t isize) const {
        if (dec) {
            static thread_local PerThread t;
            if (!t.d) t.d = alloc(), t.fre = fre;
            if (t.d) {
                size_t got = 0;
                return dec(t.d, in, clen, out, isize, &got) == 0 && got == isize;
            }
        }
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return false;
        zs.next_in = const_cast<Bytef *>(in);
        zs.avail_in = (uInt)clen;
        zs.next_out = out;
        zs.avail_out = (uInt)isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        return rc == Z_STREAM_END && zs.avail_out == 0;
    }
};

thread_local std::string g_io_err;
int io_fail(int code, const std::string &m) {
    g_io_err = m;
    return code;
}

// ---------------------------------------------------------------------------------------------
// FASTA[.gz] (kseq semantics: name = header up to the first whitespace, sequence lines concatenated)
// ---------------------------------------------------------------------------------------------
struct Fasta {
    gzFile f = nullptr;
    std::string name, seq, pending; // pending = next header line
    std::vector<char> buf;
    bool getline(std::string &out) {
        out.clear();
        for (;;) {
            if (!gzgets(f, buf.data(), (int)buf.size())) return !out.empty();
            size_t n = strlen(buf.data());
            out.append(buf.data(), n);
            if (n && buf[n - 1] == '\n') break;
        }
        while (!out.empty() && (out.back() == '\n' || out.back() == '\r')) out.pop_back();
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// BGZF stream
// ---------------------------------------------------------------------------------------------
struct Bgzf {
    FILE *f = nullptr;
    std::vector<uint8_t> block, cdata;
    size_t bpos = 0;
    uint64_t block_off = 0; // file offset of the current block
    // virtual offset of the next byte to be read
    uint64_t tell() {
        if (bpos >= block.size()) return (uint64_t)ftello(f) << 16;
        return (block_off << 16) | bpos;
    }
    bool read_block() { // returns false at EOF
        block.clear();
        bpos = 0;
        block_off = (uint64_t)ftello(f);
        uint8_t hd[18];
        size_t n = fread(hd, 1, 18, f);
        if (n == 0) return false;
        if (n != 18 || hd[0] != 31 || hd[1] != 139 || hd[2] != 8 || !(hd[3] & 4))
            throw np2h::Np2Error(NP2_E_ARG, "not a BGZF block");
        const uint32_t xlen = hd[10] | (hd[11] << 8);
        // the BC subfield is normally the first (and only) extra field
        std::vector<uint8_t> extra(xlen);
        memcpy(extra.data(), hd + 12, std::min<size_t>(6, xlen));
        if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, f) != xlen - 6)
            throw np2h::Np2Error(NP2_E_ARG, "truncated BGZF header");
        uint32_t bsize = 0;
        for (size_t p = 0; p + 4 <= xlen;) {
            const uint32_t slen = extra[p + 2] | (extra[p + 3] << 8);
            if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2) bsize = (extra[p + 4] | (extra[p + 5] << 8)) + 1;
            p += 4 + slen;
        }
        if (!bsize) throw np2h::Np2Error(NP2_E_ARG, "BGZF block without BC field");
        const size_t clen = bsize - 12 - xlen - 8;
        cdata.resize(clen + 8);
        if (fread(cdata.data(), 1, clen + 8, f) != clen + 8) throw np2h::Np2Error(NP2_E_ARG, "truncated BGZF block");
        const uint32_t isize = cdata[clen + 4] | (cdata[clen + 5] << 8) | (cdata[clen + 6] << 16) | ((uint32_t)cdata[clen + 7] << 24);
        block.resize(isize);
        if (isize && !Inflater::get().run(cdata.data(), clen, block.data(), isize))
            throw np2h::Np2Error(NP2_E_ARG, "BGZF inflate failed");
        return true;
    }
    void seek(uint64_t voffset) {
        fseeko(f, (off_t)(voffset >> 16), SEEK_SET);
        if (!read_block()) {
            block.clear();
            bpos = 0;
            return;
        }
        bpos = voffset & 0xFFFF;
    }
    // read exactly n bytes; returns false on clean EOF before the first byte
    bool read(void *dst, size_t n) {
        uint8_t *d = (uint8_t *)dst;
        size_t got = 0;
        while (got < n) {
            if (bpos >= block.size()) {
                if (!read_block()) {
                    if (got == 0) return false;
                    throw np2h::Np2Error(NP2_E_ARG, "truncated BAM");
                }
                continue;
            }
            const size_t k = std::min(n - got, block.size() - bpos);
            memcpy(d + got, block.data() + bpos, k);
            got += k;
            bpos += k;
        }
        return true;
    }
};

// Small persistent pool for the input side: BGZF blocks are independent deflate streams and BAM records independent
// byte ranges, so inflate and record copy are plain parallel loops.  Work items are handed out by an atomic counter;
// the calling thread works too.  One pool per process, sized to the host (at most 64 workers).  Several loops may be
// in flight at once (the command line keeps a few contigs' front ends going side by side): a worker takes items from
// whichever open loop still has some, so a single caller gets the whole pool and concurrent callers share it.
class IoPool {
  public:
    static IoPool &get() {
        static IoPool *p = new IoPool(); // leaked on purpose: workers outlive static destruction
        return *p;
    }
    unsigned size() const { return (unsigned)workers_.size() + 1; }
    // run fn(i) for i in [0, n), at most `max_threads` threads including the caller
    template <class F> void parallel_for(size_t n, unsigned max_threads, F fn) {
        if (n == 0) return;
        const unsigned want = (unsigned)std::min<size_t>(std::min<size_t>(max_threads, size()), n);
        if (want <= 1) {
            for (size_t i = 0; i < n; ++i) fn(i);
            return;
        }
        Job job;
        job.n = n;
        job.slots = want - 1; // helpers wanted besides the caller
        std::function<void(size_t)> body = fn;
        job.fn = &body;
        {
            std::lock_guard<std::mutex> l(mu_);
            jobs_.push_back(&job);
        }
        cv_.notify_all();
        run(job);
        std::unique_lock<std::mutex> l(mu_);
        jobs_.erase(std::find(jobs_.begin(), jobs_.end(), &job)); // no new helper can pick it up from here on
        done_cv_.wait(l, [&] { return job.helpers == 0; });
    }

    // the same loop with the CALLER doing `during()` first — work that consumes the items' results as they appear (it
    // must only wait for items in index order: they are handed out in that order) — and joining the loop afterwards
    template <class F, class G> void parallel_for_during(size_t n, unsigned max_threads, F fn, G during) {
        const unsigned want = (unsigned)std::min<size_t>(std::min<size_t>(max_threads, size()), n);
        if (want <= 1) { // nobody to wait for: items first
            for (size_t i = 0; i < n; ++i) fn(i);
            during();
            return;
        }
        Job job;
        job.n = n;
        job.slots = want - 1;
        std::function<void(size_t)> body = fn;
        job.fn = &body;
        {
            std::lock_guard<std::mutex> l(mu_);
            jobs_.push_back(&job);
        }
        cv_.notify_all();
        std::exception_ptr ep;
        try {
            during();
        } catch (...) {
            ep = std::current_exception();
        }
        run(job);
        {
            std::unique_lock<std::mutex> l(mu_);
            jobs_.erase(std::find(jobs_.begin(), jobs_.end(), &job));
            done_cv_.wait(l, [&] { return job.helpers == 0; });
        }
        if (ep) std::rethrow_exception(ep);
    }

  private:
    struct Job {
        size_t n = 0;
        std::atomic<size_t> next{0};
        unsigned slots = 0;   // helpers that may still join (guarded by mu_)
        unsigned helpers = 0; // helpers currently inside (guarded by mu_)
        std::function<void(size_t)> *fn = nullptr;
    };
    static void run(Job &j) {
        for (;;) {
            const size_t i = j.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j.n) break;
            (*j.fn)(i);
        }
    }
    IoPool() {
        // sized by the hardware, not by the quota: the pool works in bursts of a few milliseconds (one contig's inflate),
        // which a CFS quota does not throttle — measured on a box with 256 hardware threads and a quota of 16 CPUs: an
        // E. coli-sized contig's records arrive in 8 ms with 64 workers and in 18 ms with 16
        unsigned hw = std::thread::hardware_concurrency();
        unsigned n = std::min<unsigned>(64, std::max<unsigned>(2, hw / 2));
        if (const char *e = getenv("NP2_IO_THREADS")) n = (unsigned)std::max(1, atoi(e));
        for (unsigned i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
        for (auto &t : workers_) t.detach();
    }
    Job *pick() { // mu_ held: an open loop with items left and a free helper slot
        for (Job *j : jobs_)
            if (j->slots && j->next.load(std::memory_order_relaxed) < j->n) return j;
        return nullptr;
    }
    void loop() {
        std::unique_lock<std::mutex> l(mu_);
        for (;;) {
            Job *j = nullptr;
            cv_.wait(l, [&] { return (j = pick()) != nullptr; });
            --j->slots;
            ++j->helpers;
            l.unlock();
            run(*j);
            l.lock();
            if (--j->helpers == 0) done_cv_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<Job *> jobs_;
};

// growable byte buffer without value-initialisation (a std::vector would zero 100+ MiB per refill just to have inflate
// overwrite it); kept by the BAM handle, so its pages are faulted in once
// Large host blocks kept across BAM handles (the command line opens one handle per front-end thread and run): a block
// of this size goes back to the kernel when freed, and the next handle's inflate threads then fault 200 MB of fresh
// pages in again (11-14 ms of an E. coli-sized contig's first front end).  At most 8 idle blocks / 1 GiB are kept.
struct HostBlockPool {
    std::mutex mu;
    std::vector<std::pair<size_t, uint8_t *>> idle;
    static HostBlockPool &get() {
        static HostBlockPool *p = new HostBlockPool(); // leaked on purpose
        return *p;
    }
    uint8_t *take(size_t want, size_t &cap) {
        {
            std::lock_guard<std::mutex> l(mu);
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); ++i)
                if (idle[i].first >= want && (best == idle.size() || idle[i].first < idle[best].first)) best = i;
            if (best != idle.size()) {
                uint8_t *p = idle[best].second;
                cap = idle[best].first;
                idle.erase(idle.begin() + (long)best);
                return p;
            }
        }
        cap = want;
        return (uint8_t *)malloc(want);
    }
    void give(uint8_t *p, size_t cap) {
        if (!p) return;
        if (cap >= ((size_t)4 << 20)) {
            std::lock_guard<std::mutex> l(mu);
            size_t held = cap;
            for (auto &b : idle) held += b.first;
            if (idle.size() < 8 && held <= ((size_t)1 << 30)) {
                idle.emplace_back(cap, p);
                return;
            }
        }
        free(p);
    }
};
struct RawBuf {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { HostBlockPool::get().give(p, cap); }
    size_t size() const { return n; }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    void clear() { n = 0; }
    void resize(size_t m) {
        if (m > cap) {
            const size_t want = std::max(m, cap + cap / 2 + (1u << 20));
            size_t got = 0;
            uint8_t *q = HostBlockPool::get().take(want, got);
            if (!q) throw std::bad_alloc();
            if (n) memcpy(q, p, n);
            HostBlockPool::get().give(p, cap);
            p = q;
            cap = got;
        }
        n = m;
    }
    void drop_front(size_t k) { // discard the first k bytes
        if (k >= n) {
            n = 0;
            return;
        }
        memmove(p, p + k, n - k);
        n -= k;
    }
};

// Reads BGZF blocks in batches and inflates each batch with several host threads (blocks are independent).
struct BgzfBatch {
    FILE *f = nullptr;
    // the file mapped read-only: a block's deflate payload is inflated straight out of the page cache (reading 35 MB of
    // blocks with two freads each was 9 of the 19 ms the records of an E. coli-sized contig took to arrive); the pages
    // are first touched by the inflating threads, in parallel
    const uint8_t *map = nullptr;
    size_t map_len = 0, fpos = 0;
    struct Blk {
        const uint8_t *c = nullptr; // raw deflate payload (inside the mapping)
        uint32_t clen = 0, isize = 0;
        size_t out_off = 0;
        uint64_t file_off = 0; // offset of the block in the file
    };
    // (buffer position of a block's first inflated byte, file offset of the block) for the blocks in `buf`: a record's
    // BGZF virtual offset = file offset << 16 | offset inside the block
    // (the front block may begin before the buffer once consumed bytes were dropped: signed positions)
    std::vector<std::pair<int64_t, uint64_t>> blk_index;
    uint64_t voffset_at(size_t p) const {
        size_t lo = 0, hi = blk_index.size();
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) / 2;
            if (blk_index[mid].first <= (int64_t)p) lo = mid; else hi = mid;
        }
        return (blk_index[lo].second << 16) | (uint64_t)((int64_t)p - blk_index[lo].first);
    }
    RawBuf buf; // inflated bytes not yet consumed (+ the current batch)
    size_t pos = 0, skip = 0; // (skip: offset inside the first block after a seek, applied by the first fill)
    // A HINT for where the wanted records end in the file (the index's last chunk end of the reference): a refill reads
    // blocks up to it instead of a full batch — a 0.75 Mb contig's records are 250 blocks, a batch 2048 —, and once past
    // it only a few at a time (8, 16, ...: an index that understates the end costs time, never records).
    size_t hint_fpos = SIZE_MAX, past_hint = 8;
    // the batch being inflated: its blocks, a completion flag per block, the prefix of `buf` known to be inflated
    std::vector<Blk> cur_blks;
    std::unique_ptr<std::atomic<uint8_t>[]> blk_done;
    size_t blk_done_cap = 0, cur_base = 0, ready_bi = 0, ready_end = 0;
    bool eof = false;
    double ms_read = 0, ms_inflate = 0, ms_drop = 0; // where the refills' time goes (NP2_IO_PROFILE)
    double ms_walk = 0, ms_size = 0, ms_copy = 0;   // ... and the record pass over each refill
    size_t batch_blocks = 2048; // 64 KiB blocks per refill: 128 MiB of inflated BAM, all inflated in parallel
    bool read_raw(Blk &b) {
        if (fpos >= map_len) return false;
        if (map_len - fpos < 18) throw np2h::Np2Error(NP2_E_ARG, "not a BGZF block");
        const uint8_t *hd = map + fpos;
        b.file_off = fpos;
        if (hd[0] != 31 || hd[1] != 139 || hd[2] != 8 || !(hd[3] & 4)) throw np2h::Np2Error(NP2_E_ARG, "not a BGZF block");
        const uint32_t xlen = hd[10] | (hd[11] << 8);
        if (map_len - fpos < 12 + (size_t)xlen) throw np2h::Np2Error(NP2_E_ARG, "truncated BGZF header");
        const uint8_t *extra = hd + 12;
        uint32_t bsize = 0;
        for (size_t p = 0; p + 4 <= xlen;) {
            const uint32_t slen = extra[p + 2] | (extra[p + 3] << 8);
            if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= xlen) bsize = (extra[p + 4] | (extra[p + 5] << 8)) + 1;
            p += 4 + slen;
        }
        if (!bsize) throw np2h::Np2Error(NP2_E_ARG, "BGZF block without BC field");
        if (bsize < 12 + xlen + 8 || map_len - fpos < bsize) throw np2h::Np2Error(NP2_E_ARG, "truncated BGZF block");
        const size_t clen = bsize - 12 - xlen - 8;
        b.c = hd + 12 + xlen;
        b.clen = (uint32_t)clen;
        const uint8_t *tail = b.c + clen;
        b.isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        fpos += bsize;
        return true;
    }
    // start at a virtual offset
    void seek(uint64_t voffset, bool prefill = true) {
        fpos = (size_t)(voffset >> 16);
        buf.clear();
        blk_index.clear();
        pos = 0;
        eof = false;
        skip = voffset & 0xFFFF;
        hint_fpos = SIZE_MAX, past_hint = 8;
        if (prefill) fill(8);
    }
    // Bytes [0, end) of `buf` inflated?  Blocks until they are while a fill is in flight (called from the fill's `during`
    // on the filling thread: blocks are handed to the pool in buffer order, so waiting for them in order cannot starve);
    // false when `end` lies beyond what the buffer will hold.
    bool wait_ready(size_t end) {
        if (end > buf.size()) return false;
        while (ready_end < end) {
            if (ready_bi >= cur_blks.size()) {
                ready_end = buf.size();
                break;
            }
            for (unsigned spin = 0; !blk_done[ready_bi].load(std::memory_order_acquire); ++spin)
                if (spin > 64) sched_yield();
            ready_end = cur_base + cur_blks[ready_bi].out_off + cur_blks[ready_bi].isize;
            ++ready_bi;
        }
        return true;
    }
    // Appends up to n_blocks inflated blocks to `buf`.  `during` (optional) runs on this thread while the pool inflates:
    // it may read the buffer through wait_ready() as the blocks complete.
    void fill(size_t n_blocks, const std::function<void()> *during = nullptr) {
        if (eof) {
            if (during) (*during)();
            return;
        }
        const double t_f0 = np2h::now_ms();
        if (pos) { // drop consumed bytes (blocks that lie entirely before the new front leave the index)
            size_t keep = 0;
            while (keep + 1 < blk_index.size() && blk_index[keep + 1].first <= (int64_t)pos) ++keep;
            blk_index.erase(blk_index.begin(), blk_index.begin() + (long)keep);
            for (auto &e : blk_index) e.first -= (int64_t)pos;
            buf.drop_front(pos);
            pos = 0;
        }
        const double t_f1 = np2h::now_ms();
        std::vector<Blk> &blks = cur_blks;
        blks.clear();
        size_t total = 0;
        const bool past = fpos > hint_fpos;
        if (past) n_blocks = std::min(n_blocks, past_hint), past_hint *= 2;
        for (size_t i = 0; i < n_blocks; ++i) {
            if (!past && fpos > hint_fpos) break;
            Blk b;
            if (!read_raw(b)) {
                eof = true;
                break;
            }
            b.out_off = total;
            total += b.isize;
            blks.push_back(b);
        }
        const size_t base = buf.size();
        buf.resize(base + total);
        for (auto &bk : blks) blk_index.emplace_back((int64_t)(base + bk.out_off), bk.file_off);
        if (blks.size() > blk_done_cap) {
            blk_done_cap = blks.size() + blks.size() / 2;
            blk_done.reset(new std::atomic<uint8_t>[blk_done_cap]);
        }
        for (size_t i = 0; i < blks.size(); ++i) blk_done[i].store(0, std::memory_order_relaxed);
        cur_base = base, ready_bi = 0, ready_end = base;
        if (skip) pos = std::min<size_t>(skip, buf.size()), skip = 0;
        const double t_f2 = np2h::now_ms();
        std::atomic<int> bad{0};
        const Inflater &inf = Inflater::get();
        auto one = [&](size_t i) {
            if (blks[i].isize && !inf.run(blks[i].c, blks[i].clen, buf.data() + base + blks[i].out_off, blks[i].isize)) bad.store(1);
            blk_done[i].store(1, std::memory_order_release); // (also after a failure: a reader must not wait forever)
        };
        const unsigned nt = (unsigned)std::max<size_t>(1, blks.size() / 2);
        try {
            if (during)
                IoPool::get().parallel_for_during(blks.size(), nt, one, *during);
            else
                IoPool::get().parallel_for(blks.size(), nt, one);
        } catch (...) {
            ready_end = buf.size();
            if (bad.load()) throw np2h::Np2Error(NP2_E_ARG, "BGZF inflate failed"); // (what the reader tripped over)
            throw;
        }
        ready_bi = blks.size(), ready_end = buf.size();
        if (bad.load()) throw np2h::Np2Error(NP2_E_ARG, "BGZF inflate failed");
        ms_drop += t_f1 - t_f0, ms_read += t_f2 - t_f1, ms_inflate += np2h::now_ms() - t_f2;
    }
    // pointer to n contiguous bytes (nullptr on clean EOF before the first byte)
    const uint8_t *take(size_t n) {
        while (buf.size() - pos < n) {
            if (eof) {
                if (buf.size() == pos) return nullptr;
                throw np2h::Np2Error(NP2_E_ARG, "truncated BAM");
            }
            fill(batch_blocks);
        }
        const uint8_t *p = buf.data() + pos;
        pos += n;
        return p;
    }
};

uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

} // namespace

// std::vector storage in pinned host memory: the SEQ bytes gathered from the BAM go to the GPU by DMA straight from
// here (a pageable source is staged through bounce buffers at a fraction of the bus rate)
// growable byte buffer in pinned host memory WITHOUT value-initialisation (std::vector::resize zero-fills what the
// record copies overwrite a moment later: 69 MB of SEQ per E. coli-sized contig)
struct PinnedBytes {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    PinnedBytes() = default;
    PinnedBytes(const PinnedBytes &) = delete;
    PinnedBytes &operator=(const PinnedBytes &) = delete;
    ~PinnedBytes() {
        if (p) np2h::pinned_pool().put(p);
    }
    size_t size() const { return n; }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    void clear() { n = 0; }
    void reserve(size_t m) {
        if (m <= cap) return;
        const size_t want = std::max(m, cap + cap / 2 + (1u << 20));
        void *q = np2h::pinned_pool().get(want); // (blocks of the process-wide pool: pinning 69 MB costs 50 ms)
        if (!q) throw std::bad_alloc();
        if (n) memcpy(q, p, n);
        if (p) np2h::pinned_pool().put(p);
        p = (uint8_t *)q;
        cap = want;
    }
    void resize(size_t m) { // (new bytes are NOT initialised)
        reserve(m);
        n = m;
    }
    void resize(size_t m, uint8_t fill) {
        const size_t old = n;
        resize(m);
        if (m > old) memset(p + old, fill, m - old);
    }
    void push_back(uint8_t b) {
        if (n == cap) reserve(n + 1);
        p[n++] = b;
    }
    void append(const uint8_t *src, size_t k) {
        reserve(n + k);
        memcpy(p + n, src, k);
        n += k;
    }
    uint8_t &back() { return p[n - 1]; }
    uint8_t &operator[](size_t i) { return p[i]; }
};

struct np2_fasta {
    Fasta f;
};
struct SecSeq { // SEQ of a primary alignment in read orientation (4-bit BAM codes, high nibble first)
    std::vector<uint8_t> seq4;
    uint32_t len = 0;
};
// The SEQ bytes of a contig's records on their way to the device: batch by batch (one refill of the inflater: <= 128 MiB
// of BAM, ~40 MB of SEQ) through two alternating pinned pieces, each uploaded as soon as its records are copied.  The
// whole contig's SEQ as ONE pinned array (round 3) is 3.7 GB for a 248 Mb chromosome at 30x: page-locking that much takes
// ~1 s per 4 GB, grown by reallocation it was 1.5 - 5 s — and the driver serialises it with every other allocation of
// the process (the k-mer tables' hipMalloc waited 4.8 s behind it: tools/alloc_probe.py, profiles/r04_e2e_chr1_*).
struct SeqStream {
    hipStream_t s = nullptr;
    np2h::DevBuf<uint8_t> dev; // the contig's SEQ bytes, contiguous (capacity kept from contig to contig)
    uint64_t n = 0;            // bytes uploaded
    void *pin[2] = {nullptr, nullptr};
    size_t pin_cap[2] = {0, 0};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int turn = 0;
    SeqStream() { dev.cached = true; }
    ~SeqStream() {
        if (s) (void)hipStreamSynchronize(s); // (the device block goes back to the cache: nothing of ours may be in flight)
        for (int i = 0; i < 2; ++i) {
            if (pin[i]) np2h::pinned_pool().put(pin[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
    }
    void begin(hipStream_t stream, uint64_t expect_bytes) {
        s = stream;
        n = 0;
        if (dev.cap < expect_bytes + 64) {
            HIPCHK(hipStreamSynchronize(s));
            dev.ensure(expect_bytes + 64);
        }
    }
    uint8_t *piece(size_t bytes) { // staging for the next batch
        const int i = turn;
        if (busy[i]) {
            HIPCHK(hipEventSynchronize(ev[i]));
            busy[i] = false;
        }
        if (pin_cap[i] < bytes) {
            if (pin[i]) np2h::pinned_pool().put(pin[i]);
            pin[i] = nullptr, pin_cap[i] = 0;
            const size_t want = bytes + bytes / 4 + (1u << 20);
            pin[i] = np2h::pinned_pool().get(want);
            if (!pin[i]) throw np2h::Np2Error(NP2_E_NOMEM, "hipHostMalloc failed");
            pin_cap[i] = want;
        }
        if (!ev[i]) HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        return (uint8_t *)pin[i];
    }
    void push(size_t bytes) { // the piece handed out last holds `bytes` bytes that follow the n uploaded so far
        if (dev.cap < n + bytes + 64) { // the estimate fell short: a larger block, what is there copied over on the device
            np2h::DevBuf<uint8_t> bigger;
            bigger.cached = true;
            bigger.ensure((n + bytes) * 3 / 2 + 64);
            if (n) HIPCHK(hipMemcpyAsync(bigger.p, dev.p, n, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
            std::swap(bigger.p, dev.p);
            std::swap(bigger.cap, dev.cap);
            std::swap(bigger.cache_bytes, dev.cache_bytes);
            std::swap(bigger.slab_bytes, dev.slab_bytes);
        }
        const int i = turn;
        if (bytes) {
            HIPCHK(hipMemcpyAsync(dev.p + n, pin[i], bytes, hipMemcpyHostToDevice, s));
            HIPCHK(hipEventRecord(ev[i], s));
            busy[i] = true;
        }
        n += bytes;
        turn ^= 1;
    }
    void finish() { // 16 readable zero bytes behind the last record's SEQ (the columnariser loads whole words)
        if (dev.cap < n + 64) push(0);
        HIPCHK(hipMemsetAsync(dev.p + n, 0, 16, s));
        n += 16;
    }
};

struct np2_bam {
    Bgzf z;
    std::vector<std::string> ref_names;
    std::vector<uint32_t> ref_lens;
    std::vector<uint64_t> ref_start; // virtual offset of the first record of each reference (~0 = none)
    std::vector<uint64_t> ref_end;   // ... and of the end of its last record, as far as the index's chunks say (0 = unknown)
    std::vector<std::vector<uint64_t>> lin; // .bai linear index per reference: smallest virtual offset of a record
                                            // overlapping each 16 kb window (0 = none)
    uint64_t first_rec = 0;          // virtual offset of the first alignment record
    // -S: secondary alignments carry no SEQ; recovered from the primary record of the same read (secondary.rs:82-148)
    bool sec_loaded = false;
    std::unordered_map<std::string, SecSeq> sec;
    PinnedBytes seq4; // SEQ staging of the contig being read (-S, and callers without a context; capacity kept across contigs)
    SeqStream seqs;   // ... streamed to the device batch by batch (the usual path)
    BgzfBatch batch;  // batch inflater (its buffer is reused from contig to contig)
    struct GpuFetch *gpu = nullptr; // read extraction on the device (NP2_INFLATE=gpu / auto: fetch_records_gpu); made on first use
    std::shared_ptr<struct ResidentBam> resident; // ... from the WHOLE file inflated on the device once (a many-reference BAM of moderate size)
    bool resident_tried = false;
    ~np2_bam();
    const uint8_t *map = nullptr; // the whole file, mapped read-only (BgzfBatch inflates out of it)
    size_t map_len = 0;
};

namespace {

// ---- record admission + GPU columnarisation (main.rs:1758-1817) -----------------------------------
struct Admitted {
    uint32_t rec;     // input record index
    bool is_clip;
    uint32_t n_cols;  // untrimmed columns
};

// 4-bit SEQ nibble i (BAM: high nibble first)
inline uint8_t seq4_at(const uint8_t *s, uint32_t i) { return (i & 1) ? (s[i >> 1] & 15) : (s[i >> 1] >> 4); }
// reverse complement in the 4-bit domain: A(1)<->T(8), C(2)<->G(4), every other code unchanged
// (reverse_complement_seq_u8, secondary.rs:66-80, over the decoded letters "=ACMGRSVTWYHKDBN")
template <class V> void append_seq4(V &dst, const uint8_t *src, uint32_t len, bool revcomp) {
    const size_t base = dst.size();
    dst.resize(base + (len + 1) / 2, 0);
    for (uint32_t i = 0; i < len; ++i) {
        uint8_t c = seq4_at(src, revcomp ? len - 1 - i : i);
        if (revcomp) c = c == 1 ? 8 : (c == 8 ? 1 : (c == 2 ? 4 : (c == 4 ? 2 : c)));
        dst[base + (i >> 1)] |= (i & 1) ? c : (uint8_t)(c << 4);
    }
}

// retrieve_secondary_seq_from_bam (secondary.rs:8-148): pass 1 collects the names of all secondary records, pass 2
// the SEQ (read orientation) of the primary record (neither secondary nor supplementary) of each of those reads.
// Two sequential passes over the whole file, once per BAM handle.
void load_secondary_seqs(np2_bam *bam) {
    if (bam->sec_loaded) return;
    std::unordered_set<std::string> ids;
    for (int pass = 0; pass < 2; ++pass) {
        BgzfBatch z;
        z.f = bam->z.f;
        z.map = bam->map, z.map_len = bam->map_len;
        z.seek(bam->first_rec);
        for (;;) {
            const uint8_t *h4 = z.take(4);
            if (!h4) break;
            const uint32_t bs = le32(h4);
            const uint8_t *rec = z.take(bs);
            if (!rec || bs < 32) throw np2h::Np2Error(NP2_E_ARG, "BAM/SAM parsing failed!");
            const uint32_t l_read_name = rec[8];
            const uint32_t n_cigar = rec[12] | (rec[13] << 8), flag = rec[14] | (rec[15] << 8);
            const uint32_t l_seq = le32(rec + 16);
            const uint8_t *ps = rec + 32 + l_read_name + (size_t)n_cigar * 4;
            if ((size_t)(ps - rec) + (l_seq + 1) / 2 > bs) throw np2h::Np2Error(NP2_E_ARG, "BAM/SAM parsing failed!");
            const std::string name((const char *)rec + 32, l_read_name ? l_read_name - 1 : 0); // qname without the NUL
            if (pass == 0) {
                if (flag & 0x100) ids.insert(name);
            } else if (!(flag & 0x900) && ids.count(name)) {
                SecSeq sq;
                sq.len = l_seq;
                append_seq4(sq.seq4, ps, l_seq, (flag & 0x10) != 0);
                if (!bam->sec.emplace(name, std::move(sq)).second) // assert!(seqs.insert(..).is_none()), secondary.rs:130
                    throw np2h::Np2Error(NP2_E_REFPANIC, "reference would panic: two primary records for one read name");
            }
        }
        if (ids.empty()) break;
    }
    bam->sec_loaded = true;
}

// A reference-interval shard built straight from its BAM records (np2_shard_bam_*): the pileup covers the sub-contig
// [sub_lo, sub_hi) of a contig of L_glob positions; `renumber` turns the pushed records (local order, with the index of
// their input record) into the shard's read list in contig-wide numbering (holes for the contig's reads the shard does
// not hold) once the ranks have exchanged their counts.
struct ShardSpec {
    uint32_t sub_lo = 0, sub_hi = 0, zone_lo = 0, zone_hi = 0;
};
// state between the two halves of the front end: admission + GPU columnariser + keep / drop (front_begin), then the
// clip filter and the contig bookkeeping (front_finish).  A shard renumbers `reads` in between (contig-wide numbers,
// holes for the reads it does not hold) once the ranks have exchanged their records' file offsets.
struct FrontWork {
    np2_contig *c = nullptr;
    std::vector<np2_read_t> reads; // [0] = the (sub-)contig; then the pushed records in file order
    std::vector<uint8_t> lable;
    std::vector<uint32_t> rec_of;  // input record of reads[i]
    uint64_t nib_bytes = 0;
    uint32_t L = 0, L_glob = 0, sub_lo = 0;
    ~FrontWork() { delete c; }
};

// seq4: the records' SEQ bytes on the host (uploaded here) — or, with d_seq_in, already on the device (SeqStream, on
// this context's stream)
void front_begin(np2_ctx *cx, const uint8_t *ref_glob, uint32_t L_in, const np2_bamrec_t *recs, uint32_t n_recs,
                 const uint32_t *cigar, const uint8_t *seq4, uint64_t seq4_bytes, const np2_front_opts_t *o,
                 const ShardSpec *sp, FrontWork &fw, const uint8_t *d_seq_in = nullptr) {
    const double t_a0 = np2h::now_ms();
    HIPCHK(hipSetDevice(cx->device));
    hipStream_t s = cx->stream;
    const uint32_t L_glob = L_in;                        // the contig (clip policy, contig-end checks)
    const uint32_t sub_lo = sp ? sp->sub_lo : 0u;
    const uint32_t L = sp ? sp->sub_hi - sp->sub_lo : L_in; // the (sub-)contig the pileup is built on
    const uint8_t *ref = ref_glob + sub_lo;
    if (L < 3) throw np2h::Np2Error(NP2_E_ARG, "contig too short");
    // the SEQ bytes are complete before anything else is: on their way to the device at once, next to the host work below
    // (69 MB of an E. coli-sized contig are 1.4 ms of bus time)
    np2h::DevBuf<uint8_t> d_seq;
    d_seq.cached = true;
    struct StreamIdleOnExit { // (d_seq goes back to the block cache when this function ends, however it ends)
        hipStream_t s;
        ~StreamIdleOnExit() { (void)hipStreamSynchronize(s); }
    } seq_guard{s};
    if (seq4_bytes && !d_seq_in) {
        d_seq.ensure(seq4_bytes + 16);
        HIPCHK(hipMemcpyAsync(d_seq.p, seq4, seq4_bytes, hipMemcpyHostToDevice, s));
    }
    // Admission + fill_with_cigar bookkeeping, in three steps over the host pool: (1) every record on its own — the
    // admission predicate (main.rs:1758-1771), the number of column-producing CIGAR ops, alignment columns, clipping;
    // (2) a prefix sum over the admitted records places their ops and their nibble slots; (3) the ops are written.  (One
    // thread walking 2 M CIGAR ops of an E. coli-sized contig was 3 of the 6.5 ms between the records and the pileup.)
    struct Pre {
        uint8_t admit = 0, is_clip = 0;
        int err = 0; // first offending condition of this record (NP2_E_*), message below
        const char *msg = nullptr;
        uint32_t n_ops = 0, col = 0;
    };
    std::vector<Pre> pre(n_recs);
    IoPool::get().parallel_for(((size_t)n_recs + 255) / 256, 64, [&](size_t blk) {
        for (uint32_t i = (uint32_t)blk * 256; i < std::min<uint64_t>(n_recs, ((uint64_t)blk + 1) * 256); ++i) {
            const np2_bamrec_t &r = recs[i];
            const uint32_t *cg = cigar + r.cigar_off;
            Pre &q = pre[i];
            uint64_t rlen = 0;
            int64_t span = 0;
            for (uint32_t k = 0; k < r.n_cigar; ++k) { // seq_len_from_cigar(true) / reference_end - reference_start
                const uint32_t l = cg[k] >> 4, op = cg[k] & 15;
                if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8 || op == 5) rlen += l;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += l;
            }
            const bool secondary = r.flag & 0x100, supplementary = r.flag & 0x800;
            const int64_t need = std::max<int64_t>((int64_t)o->min_map_len, (int64_t)((float)rlen * o->min_map_fra));
            if ((r.flag & 0x404) != 0 || (int16_t)r.mapq <= o->min_map_qual || rlen <= o->min_read_len ||
                (secondary && !o->use_secondary) || (supplementary && !o->use_supplementary) || span < need)
                continue;
            // (-S: the record's SEQ must already be the one recovered from the read's primary alignment, main.rs:1775-1789;
            // np2_contig_from_bam does that, a caller of np2_contig_from_records passes it in)
            auto fail = [&](int code, const char *m) { q.err = code, q.msg = m; };
            if (r.pos < 0 || (uint32_t)r.pos > L_glob) { fail(NP2_E_REFPANIC, "reference would panic: record start outside the contig"); continue; }
            if (sp && (uint32_t)r.pos < sub_lo) { fail(NP2_E_ARG, "shard: record starts before the sub-contig"); continue; }
            // fill_with_cigar bookkeeping (main.rs:390-439): query clipping, columns
            uint32_t qs = 0, ts = 0, col = 0, aln_q_s = 0, aln_q_e = 0, n_ops = 0;
            bool is_first = true, bad_op = false;
            for (uint32_t k = 0; k < r.n_cigar; ++k) {
                const uint32_t l = cg[k] >> 4, op = cg[k] & 15;
                switch (op) {
                case 4:
                    qs += l;
                    if (is_first) aln_q_s = qs; else aln_q_e = qs - l;
                    break;
                case 0: case 7: case 8:
                    n_ops += l ? 1u : 0u;
                    col += l, qs += l, ts += l;
                    break;
                case 1:
                    n_ops += l ? 1u : 0u;
                    col += l, qs += l;
                    break;
                case 2:
                    n_ops += l ? 1u : 0u;
                    col += l, ts += l;
                    break;
                case 5:
                    break;
                default:
                    bad_op = true;
                }
                if (bad_op) break;
                is_first = false;
            }
            if (bad_op) { fail(NP2_E_REFPANIC, "reference would panic: Unknown cigar"); continue; }
            if (aln_q_e == 0) aln_q_e = qs;
            if (qs > r.l_seq) {
                fail(NP2_E_REFPANIC, secondary ? "reference would panic: no (or too short a) primary SEQ for a secondary alignment"
                                               : "reference would panic: SEQ shorter than CIGAR");
                continue;
            }
            if ((uint64_t)r.pos + ts > L_glob) { fail(NP2_E_REFPANIC, "reference would panic: alignment runs past the contig end"); continue; }
            if (sp && (uint64_t)r.pos + ts > sp->sub_hi) { fail(NP2_E_ARG, "shard: record ends behind the sub-contig"); continue; }
            if (r.seq_off + ((uint64_t)r.l_seq + 1) / 2 > seq4_bytes) { fail(NP2_E_ARG, "SEQ outside the buffer"); continue; }
            q.admit = 1;
            q.n_ops = n_ops, q.col = col;
            q.is_clip = aln_q_e - aln_q_s + o->max_clip_len < (uint32_t)rlen; // main.rs:1796-1797
        }
    });
    std::vector<FrontRec> frec;
    std::vector<Admitted> adm;
    uint64_t out_off = ((((uint64_t)L + 1) >> 1) + 1 + 15) & ~15ull; // slot 0 = the contig itself
    uint64_t n_fops = 0;
    for (uint32_t i = 0; i < n_recs; ++i) { // (the first offending record in file order speaks, as in a sequential walk)
        const Pre &q = pre[i];
        if (q.err) throw np2h::Np2Error(q.err, q.msg);
        if (!q.admit) continue;
        FrontRec fr;
        fr.pos = (uint32_t)recs[i].pos - sub_lo;
        fr.op_off = n_fops;
        fr.seq_off = recs[i].seq_off;
        fr.n_ops = q.n_ops;
        fr.n_cols = q.col;
        fr.pad = 0;
        fr.out_off = out_off;
        n_fops += q.n_ops;
        out_off += ((((uint64_t)q.col + 1) >> 1) + 1 + 15) & ~15ull;
        adm.push_back(Admitted{i, q.is_clip != 0, q.col});
        frec.push_back(fr);
    }
    std::vector<FrontOp> fops(n_fops);
    IoPool::get().parallel_for((frec.size() + 63) / 64, 64, [&](size_t blk) {
        for (size_t a = blk * 64; a < std::min(frec.size(), (blk + 1) * 64); ++a) {
            const np2_bamrec_t &r = recs[adm[a].rec];
            const uint32_t *cg = cigar + r.cigar_off;
            FrontOp *dst = fops.data() + frec[a].op_off;
            uint32_t qs = 0, ts = 0, col = 0;
            for (uint32_t k = 0; k < r.n_cigar; ++k) {
                const uint32_t l = cg[k] >> 4, op = cg[k] & 15;
                switch (op) {
                case 4:
                    qs += l;
                    break;
                case 0: case 7: case 8:
                    if (l) *dst++ = FrontOp{col, qs, ts, (l << 4) | op};
                    col += l, qs += l, ts += l;
                    break;
                case 1:
                    if (l) *dst++ = FrontOp{col, qs, ts, (l << 4) | 1u};
                    col += l, qs += l;
                    break;
                case 2:
                    if (l) *dst++ = FrontOp{col, qs, ts, (l << 4) | 2u};
                    col += l, ts += l;
                    break;
                default:
                    break;
                }
            }
        }
    });
    const uint32_t n = (uint32_t)frec.size();
    const uint64_t nib_bytes = out_off + 64;
    const bool prof = getenv("NP2_IO_PROFILE") != nullptr;
    const double t_b0 = np2h::now_ms();
    double t_b1 = t_b0, t_b2 = t_b0;
    if (prof) fprintf(stderr, "  front_begin: admission + CIGAR prefix sums %.2f ms\n", t_b0 - t_a0);

    np2_contig *c = fw.c = new np2_contig();
    {
        c->nib.ensure(nib_bytes + 64); // every slot is written completely by its producer kernel
        np2h::DevBuf<uint8_t> d_ref;
        np2h::DevBuf<FrontRec> d_rec;
        np2h::DevBuf<FrontOp> d_ops;
        np2h::DevBuf<FrontOut> d_out;
        d_ref.cached = d_rec.cached = d_ops.cached = d_out.cached = true; // (released after the read-back below)
        d_ref.ensure(L + 16);
        HIPCHK(hipMemcpyAsync(d_ref.p, ref, L, hipMemcpyHostToDevice, s));
        launch_pack_ref(s, d_ref.p, L, c->nib.p);
        std::vector<FrontOut> fout(n);
        if (n) {
            if (!d_seq_in) d_seq.ensure(seq4_bytes + 16); // (already there unless the records carry no SEQ at all)
            d_rec.ensure(n);
            d_ops.ensure(fops.size() + 1);
            d_out.ensure(n);
            t_b1 = np2h::now_ms();
            HIPCHK(hipMemcpyAsync(d_rec.p, frec.data(), (size_t)n * sizeof(FrontRec), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(d_ops.p, fops.data(), fops.size() * sizeof(FrontOp), hipMemcpyHostToDevice, s));
            {
                np2h::EventTimer t(cx, "columnarise");
                launch_columnarise(s, d_rec.p, n, d_ops.p, d_ref.p, d_seq_in ? d_seq_in : d_seq.p, c->nib.p, d_out.p);
            }
            fout = np2h::d2h(cx, d_out.p, n);
            t_b2 = np2h::now_ms();
        } else {
            HIPCHK(hipStreamSynchronize(s));
        }
        // keep / drop / label (main.rs:1798-1813), then filter_alignseqs_by_clip (531-574)
        std::vector<np2_read_t> &reads = fw.reads;
        std::vector<uint8_t> &lable = fw.lable;
        np2_read_t r0;
        memset(&r0, 0, sizeof r0);
        r0.aln_t_s = 0, r0.aln_t_e = L - 1, r0.nib_off = 0, r0.n_cols = L;
        reads.push_back(r0);
        lable.push_back(0);
        std::vector<uint8_t> pushed(n_recs, 0);
        std::vector<uint32_t> &rec_of = fw.rec_of;
        rec_of.assign(1, 0);
        for (uint32_t i = 0; i < n; ++i) {
            if (fout[i].n_cols <= o->min_map_len) continue; // aln_len() <= min_map_len
            if (adm[i].is_clip && L_glob < 500000) continue;
            pushed[adm[i].rec] = 1;
            rec_of.push_back(adm[i].rec);
            np2_read_t rd;
            memset(&rd, 0, sizeof rd);
            rd.aln_t_s = fout[i].aln_t_s;
            rd.aln_t_e = fout[i].aln_t_e;
            rd.n_cols = fout[i].n_cols;
            rd.nib_off = frec[i].out_off;
            reads.push_back(rd);
            lable.push_back(adm[i].is_clip ? 1 : 0);
        }
        {
            // the sortedness assertion compares every record with the start of the last record that was *pushed*
            // (pre_pos only advances at main.rs:1814-1815, after the trim-length and short-contig clip skips)
            int64_t pre_pos = 0;
            for (uint32_t i = 0; i < n_recs; ++i) {
                if (recs[i].pos < pre_pos) throw np2h::Np2Error(NP2_E_REFPANIC, "reference would panic: Unsorted input file!");
                if (pushed[i]) pre_pos = recs[i].pos;
            }
        }
        if (sp) {
            // reads that do not reach the shard's zone keep their number but are not held (like np2_shard_upload)
            for (size_t i = 1; i < reads.size(); ++i) {
                const uint32_t gs = reads[i].aln_t_s + sub_lo, ge = reads[i].aln_t_e + sub_lo;
                if (ge < sp->zone_lo || gs >= sp->zone_hi) {
                    reads[i].flags |= NP2_READ_DROPPED;
                    reads[i].n_cols = 0;
                    lable[i] = 0;
                }
            }
        }
        fw.nib_bytes = nib_bytes;
        fw.L = L, fw.L_glob = L_glob, fw.sub_lo = sub_lo;
    }
    if (prof)
        fprintf(stderr, "  front_begin: device allocations %.2f ms, H2D + columnarise + read-back %.2f ms, keep/drop + temp frees %.2f ms\n",
                t_b1 - t_b0, t_b2 - t_b1, np2h::now_ms() - t_b2);
}

void front_finish(np2_ctx *cx, FrontWork &fw, np2_contig **out) {
    const double t_c0 = np2h::now_ms();
    np2_contig *c = fw.c;
    std::vector<np2_read_t> &reads = fw.reads;
    std::vector<uint8_t> &lable = fw.lable;
    const uint32_t L = fw.L, L_glob = fw.L_glob, sub_lo = fw.sub_lo;
    const uint64_t nib_bytes = fw.nib_bytes;
    {
        {
            // (a shard applies the clip filter in contig coordinates: entry 0 is the whole contig there)
            const uint32_t offset = 50;
            std::vector<std::pair<uint32_t, uint32_t>> ranges;
            std::vector<size_t> clip_dropped;
            uint32_t rs = 0, re = 0;
            for (size_t i = 0; i < reads.size(); ++i) {
                if (lable[i] || (reads[i].flags & NP2_READ_DROPPED)) continue;
                const uint32_t ts = (i == 0 ? 0u : reads[i].aln_t_s + sub_lo) + offset;
                const uint32_t te = (i == 0 ? L_glob - 1 : reads[i].aln_t_e + sub_lo) - offset;
                if (rs == re) {
                    rs = ts, re = te;
                } else if (ts > re) {
                    ranges.emplace_back(rs, re);
                    rs = ts, re = te;
                } else if (re < te) {
                    re = te;
                }
            }
            if (rs != re) ranges.emplace_back(rs, re);
            for (size_t i = 0; i < reads.size(); ++i) {
                if (!lable[i]) continue;
                const uint32_t gs = reads[i].aln_t_s + sub_lo, ge = reads[i].aln_t_e + sub_lo;
                for (auto &rg : ranges) {
                    if (rg.first <= gs && ge <= rg.second) {
                        reads[i].flags |= NP2_READ_DROPPED; // align_bases = Vec::new(), index retained
                        reads[i].n_cols = 0;
                        clip_dropped.push_back(i);
                        break;
                    } else if (ge < rg.first) {
                        break;
                    }
                }
            }
            // a slot emptied by the clip filter still needs a terminator in its (unused) stream
            for (size_t i : clip_dropped) {
                const uint8_t ff = 0xFF;
                np2h::h2d_staged(cx, c->nib.p + reads[i].nib_off, &ff, 1);
            }
        }
        const double t_c1 = np2h::now_ms();
        np2h::finish_contig(cx, c, reads.data(), (uint32_t)reads.size(), L, nib_bytes);
        if (getenv("NP2_IO_PROFILE"))
            fprintf(stderr, "  front_finish: clip filter %.2f ms, finish_contig (descriptors, tile read lists, uploads) %.2f ms\n",
                    t_c1 - t_c0, np2h::now_ms() - t_c1);
    }
    fw.c = nullptr; // handed over
    *out = c;
}

void contig_from_records(np2_ctx *cx, const uint8_t *ref, uint32_t L, const np2_bamrec_t *recs, uint32_t n_recs,
                         const uint32_t *cigar, const uint8_t *seq4, uint64_t seq4_bytes,
                         const np2_front_opts_t *o, np2_contig **out) {
    FrontWork fw;
    front_begin(cx, ref, L, recs, n_recs, cigar, seq4, seq4_bytes, o, nullptr, fw);
    front_finish(cx, fw, out);
}

// The records of reference `tid` that overlap [zone_lo, zone_hi) (the whole contig: [0, L)), in file order, as
// np2_bamrec_t + CIGAR words + SEQ bytes (pinned staging of the handle); optionally their BGZF virtual offsets.
// With `up_stream` (and without -S) the SEQ bytes do not stay on the host: they go to bam->seqs.dev batch by batch
// (SeqStream), *seq_bytes is their total, bam->seq4 stays empty.
void fetch_records(np2_bam *bam, int tid, uint32_t L, uint32_t zone_lo, uint32_t zone_hi, const np2_front_opts_t *opts,
                   std::vector<np2_bamrec_t> &recs, std::vector<uint32_t> &cigar, std::vector<uint64_t> *voffs,
                   hipStream_t up_stream = nullptr, uint64_t *seq_bytes = nullptr) {
        PinnedBytes &seq4 = bam->seq4;
        seq4.clear();
        const bool streamed = up_stream != nullptr && !opts->use_secondary;
        uint64_t seq_total = 0; // (streamed: the running SEQ offset that seq4.size() is otherwise)
        if (opts->use_secondary) load_secondary_seqs(bam);
        // where to start: the whole contig from its first record; a zone from the linear index (the smallest offset of a
        // record overlapping the 16 kb window of zone_lo; an empty window takes the next one's, like htslib)
        uint64_t start_off = bam->ref_start[tid];
        if (start_off != ~0ull && zone_lo > 0 && !bam->lin[tid].empty()) {
            size_t w = std::min<size_t>(zone_lo >> 14, bam->lin[tid].size() - 1);
            while (w + 1 < bam->lin[tid].size() && bam->lin[tid][w] == 0) ++w;
            if (bam->lin[tid][w] != 0) start_off = bam->lin[tid][w];
        }
        if (start_off != ~0ull) {
            BgzfBatch &z = bam->batch;
            z.f = bam->z.f;
            z.map = bam->map, z.map_len = bam->map_len;
            z.seek(start_off, false);
            if (bam->ref_end[tid]) z.hint_fpos = (size_t)(bam->ref_end[tid] >> 16);
            if (streamed) {
                // SEQ is about a third of a record with qualities (half without), BAM deflates 3 - 6x: 2.5 bytes of SEQ per
                // compressed byte is rarely short (and a shortfall only costs one device-side copy)
                const uint64_t c_lo = start_off >> 16, c_hi = bam->ref_end[tid] ? (bam->ref_end[tid] >> 16) : (uint64_t)bam->map_len;
                double frac = 1.0;
                if (zone_lo > 0 || zone_hi < L) frac = std::min(1.0, ((double)zone_hi - zone_lo + 65536.0) / std::max<double>(1.0, L));
                bam->seqs.begin(up_stream, (uint64_t)((c_hi > c_lo ? (double)(c_hi - c_lo) : 0.0) * 2.5 * frac) + (32u << 20));
            }
            // Per refill (up to 128 MiB of inflated BAM, inflated in parallel): one light sequential walk over the record
            // length fields finds this contig's records — on this thread, WHILE the pool inflates, trailing the blocks as
            // they complete (10 k records = 10 k cache misses into lines other cores just wrote: 1.6 ms of an E. coli-sized
            // contig's 7.4 ms when it ran after the inflate) —, a prefix sum places their CIGAR words and SEQ bytes, and
            // the copies run in parallel (records are independent byte ranges).
            struct RecRef {
                size_t off; // first byte after the record's block_size field, inside z.buf
                uint32_t bs, n_cigar, l_seq;
                uint64_t cigar_off, seq_off, voff;
            };
            std::vector<RecRef> rr;
            bool stop = false;
            while (!stop) {
                rr.clear();
                size_t p = 0;
                uint64_t co = cigar.size(), so = streamed ? seq_total : seq4.size();
                const uint64_t so0 = so;
                double t_walk = 0;
                const std::function<void()> walk = [&]() {
                const double t_w0 = np2h::now_ms();
                p = z.pos;
                while (z.wait_ready(p + 4)) {
                    const uint32_t bs = le32(z.buf.data() + p);
                    if (bs < 32) throw np2h::Np2Error(NP2_E_ARG, "BAM/SAM parsing failed!");
                    if (!z.wait_ready(p + 4 + (size_t)bs)) break; // the record continues in the next refill
                    const uint8_t *rec = z.buf.data() + p + 4;
                    const int32_t refID = (int32_t)le32(rec);
                    if (refID != tid) {
                        if (refID > tid || refID < 0) {
                            stop = true;
                            break;
                        }
                        p += 4 + (size_t)bs;
                        continue;
                    }
                    const int32_t pos = (int32_t)le32(rec + 4);
                    if (pos >= 0 && (uint32_t)pos >= zone_hi) { // coordinate-sorted: nothing further overlaps the zone
                        stop = true;
                        break;
                    }
                    bool take = (uint32_t)pos < L; // (fetch(tid, 0, len): records starting beyond the region are skipped)
                    if (take && zone_lo > 0) { // reference end from the CIGAR: the record must reach the zone
                        const uint32_t nc = rec[12] | (rec[13] << 8);
                        const uint8_t *pc0 = rec + 32 + rec[8];
                        if ((size_t)32 + rec[8] + (size_t)nc * 4 > bs) throw np2h::Np2Error(NP2_E_ARG, "BAM/SAM parsing failed!");
                        uint64_t span = 0;
                        for (uint32_t k = 0; k < nc; ++k) {
                            const uint32_t w = le32(pc0 + 4 * k), op = w & 15;
                            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += w >> 4;
                        }
                        take = (uint64_t)pos + span > zone_lo;
                    }
                    if (take) {
                        RecRef r;
                        r.voff = z.voffset_at(p);
                        r.off = p + 4;
                        r.bs = bs;
                        r.n_cigar = rec[12] | (rec[13] << 8);
                        r.l_seq = le32(rec + 16);
                        const uint32_t l_read_name = rec[8];
                        if ((size_t)32 + l_read_name + (size_t)r.n_cigar * 4 + ((size_t)r.l_seq + 1) / 2 > bs)
                            throw np2h::Np2Error(NP2_E_ARG, "BAM/SAM parsing failed!");
                        r.cigar_off = co;
                        r.seq_off = so;
                        co += r.n_cigar;
                        const uint32_t flag = rec[14] | (rec[15] << 8);
                        if (!(opts->use_secondary && (flag & 0x100))) so += ((uint64_t)r.l_seq + 1) / 2;
                        rr.push_back(r);
                    }
                    p += 4 + (size_t)bs;
                }
                t_walk = np2h::now_ms() - t_w0;
                };
                z.fill(z.batch_blocks, &walk);
                const double t_w1 = np2h::now_ms();
                const size_t r0 = recs.size();
                recs.resize(r0 + rr.size());
                if (voffs)
                    for (auto &q : rr) voffs->push_back(q.voff);
                cigar.resize(co);
                uint8_t *seq_dst = nullptr; // where SEQ byte offset so0 of this batch goes
                if (streamed) seq_dst = bam->seqs.piece((size_t)(so - so0));
                else if (!opts->use_secondary) {
                    seq4.resize(so);
                    seq_dst = seq4.data() + so0;
                }
                const double t_w2 = np2h::now_ms();
                z.ms_walk += t_walk, z.ms_size += t_w2 - t_w1;
                if (!opts->use_secondary) {
                    IoPool::get().parallel_for(rr.size(), 64, [&](size_t i) {
                        const RecRef &q = rr[i];
                        const uint8_t *rec = z.buf.data() + q.off;
                        const uint32_t l_read_name = rec[8];
                        const uint8_t *pc = rec + 32 + l_read_name;
                        np2_bamrec_t r;
                        memset(&r, 0, sizeof r);
                        r.pos = (int32_t)le32(rec + 4);
                        r.flag = (uint16_t)(rec[14] | (rec[15] << 8));
                        r.mapq = rec[9];
                        r.n_cigar = q.n_cigar;
                        r.cigar_off = q.cigar_off;
                        r.l_seq = q.l_seq;
                        r.seq_off = q.seq_off;
                        for (uint32_t k = 0; k < q.n_cigar; ++k) cigar[q.cigar_off + k] = le32(pc + 4 * k);
                        memcpy(seq_dst + (q.seq_off - so0), pc + (size_t)q.n_cigar * 4, ((size_t)q.l_seq + 1) / 2);
                        recs[r0 + i] = r;
                    });
                    if (streamed) {
                        bam->seqs.push((size_t)(so - so0));
                        seq_total = so;
                    }
                } else {
                    for (size_t i = 0; i < rr.size(); ++i) { // -S: secondary records take their SEQ from the primary's
                        const RecRef &q = rr[i];
                        const uint8_t *rec = z.buf.data() + q.off;
                        const uint32_t l_read_name = rec[8], flag = rec[14] | (rec[15] << 8);
                        const uint8_t *pc = rec + 32 + l_read_name;
                        const uint8_t *ps = pc + (size_t)q.n_cigar * 4;
                        np2_bamrec_t r;
                        memset(&r, 0, sizeof r);
                        r.pos = (int32_t)le32(rec + 4);
                        r.flag = (uint16_t)flag;
                        r.mapq = rec[9];
                        r.n_cigar = q.n_cigar;
                        r.cigar_off = q.cigar_off;
                        r.l_seq = q.l_seq;
                        r.seq_off = seq4.size();
                        for (uint32_t k = 0; k < q.n_cigar; ++k) cigar[q.cigar_off + k] = le32(pc + 4 * k);
                        if (flag & 0x100) {
                            // SEQ of the read's primary alignment, reverse-complemented again if this record is on the
                            // reverse strand (main.rs:1775-1784).  A missing name leaves l_seq = 0: the reference would only
                            // panic if the record passes the admission filters, and so do we (contig_from_records).
                            const std::string qname((const char *)rec + 32, l_read_name ? l_read_name - 1 : 0);
                            const auto it = bam->sec.find(qname);
                            r.l_seq = 0;
                            if (it != bam->sec.end()) {
                                r.l_seq = it->second.len;
                                append_seq4(seq4, it->second.seq4.data(), it->second.len, (flag & 0x10) != 0);
                            }
                        } else {
                            seq4.append(ps, ((size_t)q.l_seq + 1) / 2);
                        }
                        recs[r0 + i] = r;
                    }
                }
                z.ms_copy += np2h::now_ms() - t_w2;
                z.pos = p;
                if (stop) break;
                if (z.eof) {
                    if (z.pos != z.buf.size()) throw np2h::Np2Error(NP2_E_ARG, "truncated BAM");
                    break;
                }
            }
        }
    if (streamed) {
        if (start_off == ~0ull) bam->seqs.begin(up_stream, 0);
        bam->seqs.finish();
        if (seq_bytes) *seq_bytes = bam->seqs.n;
    } else {
        seq4.resize(seq4.size() + 16, 0);
        if (seq_bytes) *seq_bytes = seq4.size();
    }
}

} // namespace

// NP2_SEGV_TRACE=1: print the native stack of a crashing thread (field debugging; off by default)
#include <execinfo.h>
#include <signal.h>
namespace {
void np2_segv_handler(int sig) {
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "np2: fatal signal, native stack:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
struct SegvTraceInit {
    SegvTraceInit() {
        if (getenv("NP2_SEGV_TRACE")) {
            signal(SIGSEGV, np2_segv_handler);
            signal(SIGBUS, np2_segv_handler);
            signal(SIGABRT, np2_segv_handler);
        }
    }
} g_segv_trace_init;
} // namespace

extern "C" {

const char *np2_io_last_error(void) { return g_io_err.c_str(); }

// ---- FASTA ---------------------------------------------------------------------------------------
int np2_fasta_open(const char *path, np2_fasta_t **out) {
    np2_fasta *h = new np2_fasta();
    h->f.f = gzopen(path, "rb");
    if (!h->f.f) {
        delete h;
        return io_fail(NP2_E_ARG, std::string("cannot open ") + path);
    }
    gzbuffer(h->f.f, 1 << 20);
    h->f.buf.resize(1 << 16);
    std::string line;
    while (h->f.getline(line)) // skip to the first header
        if (!line.empty() && line[0] == '>') {
            h->f.pending = line;
            break;
        }
    *out = h;
    return NP2_OK;
}
int np2_fasta_next(np2_fasta_t *h, const char **name, const uint8_t **seq, uint64_t *len) try {
    Fasta &f = h->f;
    if (f.pending.empty()) return 0;
    size_t e = 1;
    while (e < f.pending.size() && !isspace((unsigned char)f.pending[e])) ++e;
    f.name = f.pending.substr(1, e - 1);
    f.pending.clear();
    f.seq.clear();
    std::string line;
    while (f.getline(line)) {
        if (!line.empty() && line[0] == '>') {
            f.pending = line;
            break;
        }
        f.seq += line;
    }
    *name = f.name.c_str();
    *seq = (const uint8_t *)f.seq.data();
    *len = f.seq.size();
    return 1;
} catch (const std::exception &ex) {
    return io_fail(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what());
}
void np2_fasta_close(np2_fasta_t *h) {
    if (!h) return;
    if (h->f.f) gzclose(h->f.f);
    delete h;
}

// ---- yak v2 (kmer.rs:72-170) -----------------------------------------------------------------------
int np2_yak_load(const char *path, np2_yak_t *out) try {
    FILE *f = fopen(path, "rb");
    if (!f) return io_fail(NP2_E_ARG, std::string("cannot open ") + path);
    uint8_t hd[16];
    if (fread(hd, 1, 16, f) != 16 || memcmp(hd, "YAK\2", 4) != 0) {
        fclose(f);
        return io_fail(NP2_E_ARG, "The input binary k-mer dump file is incompatible.");
    }
    const uint32_t k = le32(hd + 4), pre = le32(hd + 8), cbits = le32(hd + 12);
    if (cbits != 10) {
        fclose(f);
        return io_fail(NP2_E_ARG, "different YAK_COUNTER_BITS");
    }
    if (pre > 20) {
        fclose(f);
        return io_fail(NP2_E_UNSUPPORTED, "yak prefix bits too large");
    }
    const size_t nb = (size_t)1 << pre;
    // every word of the file is read ONCE, straight into the array that is handed out (the file's size bounds it): a
    // human-scale dump is tens of GB, and the command line loads its dumps while the first contigs are being read
    struct stat st;
    if (fstat(fileno(f), &st) != 0) {
        fclose(f);
        return io_fail(NP2_E_ARG, std::string("cannot stat ") + path);
    }
    const size_t max_words = (size_t)st.st_size / 8 + 1;
    uint64_t *w = (uint64_t *)malloc(max_words * 8);
    uint64_t *off = (uint64_t *)malloc((nb + 1) * 8);
    if (!w || !off) {
        fclose(f);
        free(w);
        free(off);
        return io_fail(NP2_E_NOMEM, "out of memory loading the k-mer dump");
    }
    off[0] = 0;
    size_t have = 0;
    for (size_t b = 0; b < nb; ++b) {
        uint8_t h8[8];
        if (fread(h8, 1, 8, f) != 8) {
            fclose(f);
            free(off);
            free(w);
            return io_fail(NP2_E_ARG, "Failed to parse the dump file");
        }
        const uint32_t n = le32(h8 + 4); // first u32 (capacity bits) is ignored like the reference (kmer.rs:143-147)
        const size_t want = std::min<size_t>(n, max_words - have);
        const size_t got = fread(w + have, 8, want, f);
        have += got; // UnexpectedEof ends the bucket (kmer.rs:151-155)
        off[b + 1] = have;
    }
    fclose(f);
    out->k = k;
    out->pre = pre;
    out->n_words = have;
    out->words = w;
    out->bucket_off = off;
    return NP2_OK;
} catch (const std::exception &ex) {
    return io_fail(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what());
}
void np2_yak_free(np2_yak_t *y) {
    if (!y) return;
    free((void *)y->words);
    free((void *)y->bucket_off);
    y->words = nullptr;
    y->bucket_off = nullptr;
}

// The dumps straight into HBM tables (what np2_yak_load + np2_ctx_create do through host arrays): each file is read
// once, in 8 MiB pieces, into pinned staging and copied to the device while the next piece is being read; the insert
// kernel then builds the table from the file image itself (bucket headers skipped by offset).  One host thread, one
// stream per dump.  A 12 Mb genome's two dumps (2 x 104 MB) are tables 0.1 s sooner than through pageable host
// arrays; a human-scale dump (tens of GB) never needs its host copy at all.
namespace {
struct YakFile {
    std::string path;
    int fd = -1;
    uint32_t k = 0, pre = 0;
    size_t size = 0;
    np2h::YakTable table;
    int code = NP2_OK;
    std::string msg;
    ~YakFile() {
        if (fd >= 0) close(fd);
    }
};
void yak_file_to_table(YakFile &yf, int device, hipStream_t given) {
    hipStream_t st = given; // (one of the new context's idle streams, or our own)
    hipEvent_t ev[2] = {nullptr, nullptr};
    uint8_t *pin[2] = {nullptr, nullptr};
    auto fail = [&](int code, const std::string &m) { yf.code = code, yf.msg = m; };
    const bool prof = getenv("NP2_IO_PROFILE") != nullptr;
    auto body = [&]() {
        const double t0 = np2h::now_ms();
        HIPCHK(hipSetDevice(device));
        const size_t nb = (size_t)1 << yf.pre;
        // bucket headers: 1024 small reads at positions that depend on one another (page cache)
        std::vector<uint64_t> off(nb + 1);
        uint64_t mx = 0;
        size_t at = 16; // file offset of the next bucket header
        const size_t body_words = (yf.size - 16) / 8; // the device image: the file from byte 16 on, whole words
        for (size_t b = 0; b < nb; ++b) {
            uint8_t h8[8];
            if (at + 8 > yf.size || pread(yf.fd, h8, 8, (off_t)at) != 8) return fail(NP2_E_ARG, "Failed to parse the dump file");
            const uint64_t n = le32(h8 + 4); // (first u32, the capacity bits, is ignored like the reference, kmer.rs:143-147)
            const uint64_t take = std::min<uint64_t>(n, (yf.size - at - 8) / 8); // UnexpectedEof ends the bucket (kmer.rs:151-155)
            off[b] = (at - 16) / 8 + 1;
            mx = std::max(mx, take);
            at += 8 + take * 8;
            if (b + 1 == nb) off[nb] = off[b] + take + 1;
        }
        uint32_t cl = 4;
        while ((1ull << cl) < mx * 2 + 2) ++cl;
        const size_t slots = nb << cl;
        const double t1 = np2h::now_ms();
        double ta[6] = {t1, t1, t1, t1, t1, t1};
        if (!st) HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        ta[0] = np2h::now_ms();
        yf.table.k = yf.k;
        yf.table.cap_log2 = cl;
        yf.table.table = std::make_shared<np2h::DevBuf<uint64_t>>();
        yf.table.table->ensure(slots);
        ta[1] = np2h::now_ms();
        HIPCHK(hipMemsetAsync(yf.table.table->p, 0xFF, slots * 8, st));
        ta[2] = np2h::now_ms();
        np2h::DevBuf<uint64_t> d_raw, d_off;
        np2h::DevBuf<uint32_t> d_dup;
        d_raw.ensure(body_words + 1);
        ta[3] = np2h::now_ms();
        d_off.ensure(nb + 1);
        d_dup.ensure(1);
        HIPCHK(hipMemsetAsync(d_dup.p, 0, 4, st));
        ta[4] = np2h::now_ms();
        const size_t PIECE = (size_t)8 << 20;
        for (int i = 0; i < 2; ++i) {
            pin[i] = (uint8_t *)np2h::pinned_pool().get(PIECE);
            if (!pin[i]) throw np2h::Np2Error(NP2_E_NOMEM, "hipHostMalloc failed");
            HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        const double t2 = np2h::now_ms();
        if (prof)
            fprintf(stderr, "  yak k=%u allocations: stream %.1f ms, table (%.1f GB) %.1f ms, memset call %.1f ms, file image (%.1f GB) %.1f ms, "
                            "small buffers %.1f ms, pinned pieces + events %.1f ms\n", yf.k, ta[0] - t1, slots * 8 / 1e9, ta[1] - ta[0],
                    ta[2] - ta[1], body_words * 8 / 1e9, ta[3] - ta[2], ta[4] - ta[3], t2 - ta[4]);
        const size_t body = body_words * 8;
        size_t piece = 0;
        for (size_t o = 0; o < body; o += PIECE, ++piece) {
            const int sl = (int)(piece & 1);
            if (piece >= 2) HIPCHK(hipEventSynchronize(ev[sl])); // the copy that last used this staging piece
            const size_t want = std::min(PIECE, body - o);
            // (the page cache is copied from at ~10 GB/s per thread, the bus takes 50: the piece is read by several of the pool's)
            const size_t SUB = (size_t)1 << 20;
            std::atomic<int> bad{0};
            uint8_t *const stage = pin[sl];
            IoPool::get().parallel_for((want + SUB - 1) / SUB, 8, [&](size_t k) {
                size_t got = 0;
                const size_t n = std::min(SUB, want - k * SUB);
                while (got < n) {
                    const ssize_t r = pread(yf.fd, stage + k * SUB + got, n - got, (off_t)(16 + o + k * SUB + got));
                    if (r <= 0) {
                        bad.store(1);
                        return;
                    }
                    got += (size_t)r;
                }
            });
            if (bad.load()) return fail(NP2_E_ARG, "Failed to parse the dump file");
            HIPCHK(hipMemcpyAsync((uint8_t *)d_raw.p + o, pin[sl], want, hipMemcpyHostToDevice, st));
            HIPCHK(hipEventRecord(ev[sl], st));
        }
        const double t3 = np2h::now_ms();
        // (the offsets are tiny: through the first staging piece once its last copy has drained)
        HIPCHK(hipStreamSynchronize(st));
        memcpy(pin[0], off.data(), (nb + 1) * 8);
        HIPCHK(hipMemcpyAsync(d_off.p, pin[0], (nb + 1) * 8, hipMemcpyHostToDevice, st));
        np2::launch_yak_insert(st, d_raw.p, d_off.p, (uint32_t)nb, mx, yf.table.table->p, cl, d_dup.p, 1);
        uint32_t *dup = (uint32_t *)pin[1];
        HIPCHK(hipMemcpyAsync(dup, d_dup.p, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (*dup) { // a repeated key (yak writes none): a slot per word, the winner chosen at lookup (kmer.rs:148-167)
            yf.table.ord = std::make_shared<np2h::DevBuf<uint32_t>>();
            yf.table.ord->ensure(slots);
            HIPCHK(hipMemsetAsync(yf.table.table->p, 0xFF, slots * 8, st));
            np2::launch_yak_insert_dup(st, d_raw.p, d_off.p, (uint32_t)nb, mx, yf.table.table->p, cl, yf.table.ord->p, 1);
            HIPCHK(hipStreamSynchronize(st));
        }
        if (prof)
            fprintf(stderr, "yak dump -> table (k=%u, %.0f MB): bucket headers %.2f ms, device + staging allocations %.2f ms, read + copy %.2f ms, "
                            "insert %.2f ms\n", yf.k, yf.size / 1e6, t1 - t0, t2 - t1, t3 - t2, np2h::now_ms() - t3);
    };
    try {
        body();
    } catch (const np2h::Np2Error &e) {
        fail(e.code, e.what());
    } catch (const std::exception &e) {
        fail(NP2_E_NOMEM, e.what());
    }
    if (st) (void)hipStreamSynchronize(st);
    for (int i = 0; i < 2; ++i) {
        if (pin[i]) np2h::pinned_pool().put(pin[i]);
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    }
    if (st && st != given) (void)hipStreamDestroy(st);
}
} // namespace

int np2_ctx_create_from_files(np2_ctx_t **out, int device, const char *const *paths, int n_paths) try {
    if (!out) return NP2_E_ARG;
    *out = nullptr;
    if (n_paths < 0 || n_paths > NP2_MAX_YAK || (n_paths && !paths)) return io_fail(NP2_E_ARG, "n_yak must be in [0, 15]");
    std::vector<std::unique_ptr<YakFile>> files;
    for (int i = 0; i < n_paths; ++i) {
        std::unique_ptr<YakFile> yf(new YakFile());
        yf->path = paths[i];
        yf->fd = open(paths[i], O_RDONLY);
        if (yf->fd < 0) return io_fail(NP2_E_ARG, std::string("cannot open ") + paths[i]);
        struct stat st;
        uint8_t hd[16];
        if (fstat(yf->fd, &st) != 0) return io_fail(NP2_E_ARG, std::string("cannot stat ") + paths[i]);
        yf->size = (size_t)st.st_size;
        if (pread(yf->fd, hd, 16, 0) != 16 || memcmp(hd, "YAK\2", 4) != 0)
            return io_fail(NP2_E_ARG, "The input binary k-mer dump file is incompatible.");
        yf->k = le32(hd + 4), yf->pre = le32(hd + 8);
        if (le32(hd + 12) != 10) return io_fail(NP2_E_ARG, "different YAK_COUNTER_BITS");
        if (yf->k >= 32 || yf->k < 2) return io_fail(NP2_E_UNSUPPORTED, "yak k must be in [2, 32) (main.rs:1433-1434)");
        if (yf->pre != 10) return io_fail(NP2_E_UNSUPPORTED, "yak pre must be 10 (kmer.rs:52-54,123-125)");
        files.push_back(std::move(yf));
    }
    std::stable_sort(files.begin(), files.end(), [](const std::unique_ptr<YakFile> &a, const std::unique_ptr<YakFile> &b) { return a->k < b->k; }); // option.rs:238
    np2_ctx_t *cx = nullptr;
    const int rc = np2_ctx_create(&cx, device, nullptr, 0);
    if (rc != NP2_OK) return io_fail(rc, "np2_ctx_create failed (see stderr)");
    {
        std::vector<std::thread> th;
        hipStream_t own[3] = {cx->stream, cx->stream2, cx->stream_out}; // idle until the context's first call
        for (size_t i = 1; i < files.size(); ++i)
            th.emplace_back([&, i] { yak_file_to_table(*files[i], device, i < 3 ? own[i] : nullptr); });
        if (!files.empty()) yak_file_to_table(*files[0], device, own[0]);
        for (auto &t : th) t.join();
    }
    for (auto &yf : files)
        if (yf->code != NP2_OK) {
            const int code = yf->code;
            const std::string msg = yf->path + ": " + yf->msg;
            files.clear(); // (tables released before the context's device state goes)
            np2_ctx_destroy(cx);
            return io_fail(code, msg);
        }
    for (auto &yf : files) cx->yaks.push_back(yf->table);
    *out = cx;
    return NP2_OK;
} catch (const std::exception &ex) {
    return io_fail(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what());
}

// ---- BAM ---------------------------------------------------------------------------------------------
int np2_bam_open(const char *path, np2_bam_t **out) {
    np2_bam *b = new np2_bam();
    b->z.f = fopen(path, "rb");
    try {
        if (!b->z.f) throw np2h::Np2Error(NP2_E_ARG, std::string("cannot open ") + path);
        fseeko(b->z.f, 0, SEEK_SET);
        b->z.block.clear();
        b->z.bpos = 0;
        uint8_t h8[8];
        if (!b->z.read(h8, 8) || memcmp(h8, "BAM\1", 4) != 0) throw np2h::Np2Error(NP2_E_ARG, "not a BAM file");
        const uint32_t l_text = le32(h8 + 4);
        std::vector<uint8_t> text(l_text + 1);
        if (l_text) b->z.read(text.data(), l_text);
        uint8_t h4[4];
        b->z.read(h4, 4);
        const uint32_t n_ref = le32(h4);
        for (uint32_t i = 0; i < n_ref; ++i) {
            b->z.read(h4, 4);
            const uint32_t l_name = le32(h4);
            std::vector<char> nm(l_name + 1, 0);
            b->z.read(nm.data(), l_name);
            b->z.read(h4, 4);
            b->ref_names.emplace_back(nm.data());
            b->ref_lens.push_back(le32(h4));
        }
        b->ref_start.assign(n_ref, ~0ull);
        b->ref_end.assign(n_ref, 0);
        b->lin.assign(n_ref, {});
        b->first_rec = b->z.tell();
        {
            struct stat st;
            if (fstat(fileno(b->z.f), &st) != 0) throw np2h::Np2Error(NP2_E_ARG, std::string("cannot stat ") + path);
            b->map_len = (size_t)st.st_size;
            void *m = b->map_len ? mmap(nullptr, b->map_len, PROT_READ, MAP_SHARED, fileno(b->z.f), 0) : nullptr;
            if (m == MAP_FAILED) throw np2h::Np2Error(NP2_E_NOMEM, std::string("cannot map ") + path);
            b->map = (const uint8_t *)m;
        }
        // index: <path>.bai or <stem>.bai
        std::string p1 = std::string(path) + ".bai", p2 = path;
        if (p2.size() > 4 && p2.substr(p2.size() - 4) == ".bam") p2 = p2.substr(0, p2.size() - 4) + ".bai";
        FILE *fi = fopen(p1.c_str(), "rb");
        if (!fi) fi = fopen(p2.c_str(), "rb");
        if (!fi) throw np2h::Np2Error(NP2_E_ARG, "Faield random access BAM/SAM! (no .bai index)");
        std::vector<uint8_t> idx;
        {
            fseeko(fi, 0, SEEK_END);
            const size_t sz = (size_t)ftello(fi);
            fseeko(fi, 0, SEEK_SET);
            idx.resize(sz);
            if (fread(idx.data(), 1, sz, fi) != sz) {
                fclose(fi);
                throw np2h::Np2Error(NP2_E_ARG, "cannot read the .bai index");
            }
            fclose(fi);
        }
        if (idx.size() < 8 || memcmp(idx.data(), "BAI\1", 4) != 0) throw np2h::Np2Error(NP2_E_ARG, "bad .bai magic");
        size_t p = 4;
        const uint32_t n_ref_i = le32(idx.data() + p);
        p += 4;
        for (uint32_t r = 0; r < n_ref_i && r < n_ref; ++r) {
            if (p + 4 > idx.size()) throw np2h::Np2Error(NP2_E_ARG, "truncated .bai");
            const uint32_t n_bin = le32(idx.data() + p);
            p += 4;
            uint64_t best = ~0ull, last = 0;
            for (uint32_t bi = 0; bi < n_bin; ++bi) {
                const uint32_t bin = le32(idx.data() + p), n_chunk = le32(idx.data() + p + 4);
                p += 8;
                for (uint32_t ci = 0; ci < n_chunk; ++ci) {
                    const uint64_t beg = le64(idx.data() + p), end = le64(idx.data() + p + 8);
                    p += 16;
                    if (bin != 37450 && beg < best) best = beg; // 37450 = metadata pseudo-bin
                    if (bin != 37450 && end > last) last = end;
                }
            }
            b->ref_end[r] = last;
            const uint32_t n_intv = le32(idx.data() + p);
            p += 4;
            if (p + (size_t)n_intv * 8 > idx.size()) throw np2h::Np2Error(NP2_E_ARG, "truncated .bai");
            b->lin[r].resize(n_intv);
            for (uint32_t w = 0; w < n_intv; ++w) b->lin[r][w] = le64(idx.data() + p + (size_t)w * 8);
            p += (size_t)n_intv * 8;
            b->ref_start[r] = best;
        }
    } catch (const np2h::Np2Error &e) {
        if (b->map) munmap(const_cast<uint8_t *>(b->map), b->map_len);
        if (b->z.f) fclose(b->z.f);
        delete b;
        return io_fail(e.code, e.what());
    } catch (const std::exception &ex) {
        if (b->map) munmap(const_cast<uint8_t *>(b->map), b->map_len);
        if (b->z.f) fclose(b->z.f);
        delete b;
        return io_fail(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what());
    }
    *out = b;
    return NP2_OK;
}
void np2_bam_close(np2_bam_t *b) {
    if (!b) return;
    if (b->map) munmap(const_cast<uint8_t *>(b->map), b->map_len);
    if (b->z.f) fclose(b->z.f);
    delete b;
}
int np2_bam_n_refs(np2_bam_t *b) { return b ? (int)b->ref_names.size() : 0; }
const char *np2_bam_ref_name(np2_bam_t *b, int tid, uint32_t *len) {
    if (!b || tid < 0 || (size_t)tid >= b->ref_names.size()) return nullptr;
    if (len) *len = b->ref_lens[tid];
    return b->ref_names[tid].c_str();
}

int np2_contig_from_records(np2_ctx_t *cx, const uint8_t *ref, uint32_t L, const np2_bamrec_t *recs, uint32_t n_recs,
                            const uint32_t *cigar, const uint8_t *seq4, const np2_front_opts_t *opts,
                            np2_contig_t **out) {
    if (!cx || !ref || !opts || !out) return NP2_E_ARG;
    *out = nullptr;
    try {
        uint64_t sbytes = 0;
        for (uint32_t i = 0; i < n_recs; ++i) sbytes = std::max<uint64_t>(sbytes, recs[i].seq_off + ((uint64_t)recs[i].l_seq + 1) / 2);
        contig_from_records(cx, ref, L, recs, n_recs, cigar, seq4, sbytes, opts, out);
        np2h::flush_timings(cx);
    } catch (const np2h::Np2Error &e) {
        (void)hipStreamSynchronize(cx->stream);
        np2h::flush_timings(cx);
        return np2h::fail(cx, e);
    } catch (const std::exception &ex) {
        (void)hipStreamSynchronize(cx->stream);
        np2h::flush_timings(cx);
        return np2h::fail(cx, np2h::Np2Error(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what()));
    }
    return NP2_OK;
}

struct GBlk { // a BGZF block of the file
    uint64_t file_off; // of the block
    uint32_t hdr_len;  // 12 + XLEN: the raw DEFLATE payload starts there
    uint32_t clen, isize;
    uint32_t bsize;    // the whole block
};
// The whole file inflated on the device, once per process and device, for BAMs of many references and moderate size (an
// assembly's: yeast 96 MB -> 600 MB): ONE inflate launch over every block — a block's decode latency, 2 - 3 ms, is paid once
// instead of once per reference, and 10^4 blocks fill the device where a reference's few hundred leave it idle —, after which
// a reference's front end is the record walk over its stretch of the resident stream.  Shared by the handles a process
// opens on the file (the command line: one per front-end thread); released with the last of them.
struct ResidentBam {
    std::mutex mu;
    bool built = false, failed = false;
    std::vector<GBlk> blks;        // every block from the first record's on, the end-of-file marker included
    std::vector<uint64_t> out_off; // blks.size() + 1
    np2h::DevBuf<uint8_t> d_inf;
    ResidentBam() { d_inf.cached = true; }
};
// ---- read extraction on the device ---------------------------------------------------------------------------------------
// The contig's BGZF blocks go to the device as they lie in the file, are inflated there (k_bgzf_inflate: one wavefront per
// block), and the records are found by walking the inflated stream along the .bai linear index (k_bam_chain_*).  The host
// parses the 18-byte block headers, converts the index's virtual offsets to stream offsets and reads back one
// np2_bamrec_t + the CIGAR words per record for the admission pass (front_begin) — no payload byte is touched here, and the
// SEQ bytes never leave the device: a record's seq_off points into the inflated stream, which the columnariser reads in place.
//
// When: NP2_INFLATE=gpu, or — unset — when this rank's share of the host is under twelve CPUs (the two paths tie for an
// E. coli-sized contig with sixteen: 8.4 - 9 ms either way; eight ranks of a node on a 16-CPU quota have two each, where
// the pool's path takes 85 ms and this one 12), or when the reference's records are 128 MB of BAM and more.
// -S (SEQ of secondary records from their primaries) and reference-interval shards stay on the host path.
struct GpuFetch {
    np2h::DevBuf<uint8_t> d_comp, d_inf;
    np2h::DevBuf<np2::InfBlock> d_blk;
    np2h::DevBuf<uint32_t> d_status, d_cigar;
    np2h::DevBuf<uint64_t> d_starts, d_cig_src;
    np2h::DevBuf<uint2> d_info, d_off;
    np2h::DevBuf<np2_bamrec_t> d_recs;
    void *h_recs = nullptr, *h_cigar = nullptr; // pinned blocks the caller reads the records from (until the next fetch)
    size_t h_recs_cap = 0, h_cigar_cap = 0;
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    static constexpr size_t PIECE = (size_t)16 << 20;
    GpuFetch() { d_comp.cached = d_inf.cached = true; }
    ~GpuFetch() {
        (void)hipDeviceSynchronize();
        for (int i = 0; i < 2; ++i) {
            if (pin[i]) np2h::pinned_pool().put(pin[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
        if (h_recs) np2h::pinned_pool().put(h_recs);
        if (h_cigar) np2h::pinned_pool().put(h_cigar);
    }
    void *host_block(void *&p, size_t &cap, size_t bytes) {
        if (cap < bytes) {
            if (p) np2h::pinned_pool().put(p);
            p = nullptr, cap = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            p = np2h::pinned_pool().get(want);
            if (!p) throw np2h::Np2Error(NP2_E_NOMEM, "hipHostMalloc failed");
            cap = want;
        }
        return p;
    }
    // file bytes [src, src + n) -> dst (device), through two alternating pinned pieces filled by the host pool
    void upload(const uint8_t *src, size_t n, uint8_t *dst, hipStream_t s) {
        for (int i = 0; i < 2; ++i) {
            if (!pin[i]) {
                pin[i] = np2h::pinned_pool().get(PIECE);
                if (!pin[i]) throw np2h::Np2Error(NP2_E_NOMEM, "hipHostMalloc failed");
            }
            if (!ev[i]) HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        size_t piece = 0;
        for (size_t o = 0; o < n; o += PIECE, ++piece) {
            const int sl = (int)(piece & 1);
            if (piece >= 2) HIPCHK(hipEventSynchronize(ev[sl]));
            const size_t want = std::min(PIECE, n - o);
            uint8_t *stage = (uint8_t *)pin[sl];
            const size_t SUB = (size_t)1 << 20;
            IoPool::get().parallel_for((want + SUB - 1) / SUB, 16, [&](size_t k) {
                memcpy(stage + k * SUB, src + o + k * SUB, std::min(SUB, want - k * SUB));
            });
            HIPCHK(hipMemcpyAsync(dst + o, stage, want, hipMemcpyHostToDevice, s));
            HIPCHK(hipEventRecord(ev[sl], s));
        }
    }
};
np2_bam::~np2_bam() { delete gpu; }

namespace {
bool gpu_fetch_wanted(const np2_bam *bam, int tid) { // (read per contig, not per pass: a tool may switch between two reads of the same file)
    if (const char *e = getenv("NP2_INFLATE")) return !strcmp(e, "gpu");
    static const bool few_cpus = np2h::usable_cpus() / std::max(1u, np2h::local_ranks()) < 12u;
    if (few_cpus) return true;
    // with the host's CPUs to itself the pool's path ties for a bacterial contig (8.4 - 9 ms either way) and loses from a
    // few hundred MB of BAM on (a 248 Mb chromosome's 2 GB: front end 0.86 - 0.9 s here, 1.1 - 1.5 s there)
    const uint64_t lo = bam->ref_start[tid], hi = bam->ref_end[tid];
    return lo != ~0ull && hi && (hi >> 16) > (lo >> 16) && (hi >> 16) - (lo >> 16) >= ((uint64_t)128 << 20);
}
struct GpuRecs {
    const np2_bamrec_t *recs = nullptr;
    const uint32_t *cigar = nullptr;
    uint32_t n_recs = 0;
    const uint8_t *d_stream = nullptr;
    uint64_t stream_bytes = 0;
};
// File bytes [c_lo, read_end) -> pinned pieces (pread on the pool's threads: the page cache is copied from, no mapping of the
// file is faulted in — through the mmap the same 2 GB of a chromosome's BAM took 0.08 to 2.7 s) -> g.d_comp; the 18-byte block
// headers are parsed out of each piece while it is there.  blks: the blocks that lie wholly inside the range; eof: nothing
// (but the end-of-file marker) follows them in the file.
void read_blocks_to_device(GpuFetch &g, int fd, size_t file_len, size_t c_lo, size_t read_end, hipStream_t s, std::vector<GBlk> &blks, bool &eof) {
    const size_t c_bytes = read_end - c_lo;
    for (int i = 0; i < 2; ++i) {
        if (!g.pin[i]) {
            g.pin[i] = np2h::pinned_pool().get(GpuFetch::PIECE);
            if (!g.pin[i]) throw np2h::Np2Error(NP2_E_NOMEM, "hipHostMalloc failed");
        }
        if (!g.ev[i]) HIPCHK(hipEventCreateWithFlags(&g.ev[i], hipEventDisableTiming));
    }
    auto small_read = [&](uint64_t off, uint8_t *dst, size_t n) { // a few bytes that straddle a piece
        if (off + n > file_len || pread(fd, dst, n, (off_t)off) != (ssize_t)n) throw np2h::Np2Error(NP2_E_ARG, "truncated BGZF block");
    };
    uint64_t hdr_at = c_lo; // file offset of the next block header
    bool range_done = false;
    size_t piece = 0;
    for (size_t o = 0; o < c_bytes; o += GpuFetch::PIECE, ++piece) {
        const int sl = (int)(piece & 1);
        if (piece >= 2) HIPCHK(hipEventSynchronize(g.ev[sl]));
        const size_t want = std::min(GpuFetch::PIECE, c_bytes - o);
        uint8_t *stage = (uint8_t *)g.pin[sl];
        const size_t SUB = (size_t)1 << 20;
        std::atomic<int> bad{0};
        IoPool::get().parallel_for((want + SUB - 1) / SUB, 16, [&](size_t k) {
            size_t got = 0;
            const size_t n = std::min(SUB, want - k * SUB);
            while (got < n) {
                const ssize_t r = pread(fd, stage + k * SUB + got, n - got, (off_t)(c_lo + o + k * SUB + got));
                if (r <= 0) {
                    bad.store(1);
                    return;
                }
                got += (size_t)r;
            }
        });
        if (bad.load()) throw np2h::Np2Error(NP2_E_ARG, "truncated BAM");
        HIPCHK(hipMemcpyAsync(g.d_comp.p + o, stage, want, hipMemcpyHostToDevice, s));
        HIPCHK(hipEventRecord(g.ev[sl], s));
        // the headers that begin inside this piece
        const uint64_t p0 = c_lo + o, p1 = p0 + want;
        while (!range_done && hdr_at < p1) {
            if (hdr_at + 18 > file_len) throw np2h::Np2Error(NP2_E_ARG, "not a BGZF block");
            uint8_t hb[18 + 256];
            const uint8_t *hd = stage + (hdr_at - p0);
            if (hdr_at + 18 > p1) small_read(hdr_at, hb, 18), hd = hb;
            if (hd[0] != 31 || hd[1] != 139 || hd[2] != 8 || !(hd[3] & 4)) throw np2h::Np2Error(NP2_E_ARG, "not a BGZF block");
            const uint32_t xlen = hd[10] | (hd[11] << 8);
            if (xlen > 256) throw np2h::Np2Error(NP2_E_ARG, "BGZF block without BC field");
            if (hd == hb || hdr_at + 12 + xlen > p1) small_read(hdr_at, hb, 12 + (size_t)xlen), hd = hb;
            const uint8_t *ex = hd + 12;
            uint32_t bsize = 0;
            for (size_t q = 0; q + 4 <= xlen;) {
                const uint32_t slen = ex[q + 2] | (ex[q + 3] << 8);
                if (ex[q] == 'B' && ex[q + 1] == 'C' && slen == 2 && q + 6 <= xlen) bsize = (ex[q + 4] | (ex[q + 5] << 8)) + 1;
                q += 4 + slen;
            }
            if (!bsize) throw np2h::Np2Error(NP2_E_ARG, "BGZF block without BC field");
            if (bsize < 12 + xlen + 8 || hdr_at + bsize > file_len) throw np2h::Np2Error(NP2_E_ARG, "truncated BGZF block");
            if (hdr_at + bsize > read_end) { // (the block continues beyond what this round reads: not part of it)
                range_done = true;
                break;
            }
            uint8_t tb8[8];
            const uint64_t tail = hdr_at + bsize - 8;
            const uint8_t *tp = stage + (tail - p0);
            if (tail + 8 > p1) small_read(tail, tb8, 8), tp = tb8;
            GBlk b;
            b.file_off = hdr_at, b.hdr_len = 12 + xlen, b.clen = bsize - 12 - xlen - 8, b.bsize = bsize;
            b.isize = tp[4] | (tp[5] << 8) | (tp[6] << 16) | ((uint32_t)tp[7] << 24);
            blks.push_back(b);
            hdr_at += bsize;
        }
    }
    eof = false;
    if (hdr_at >= file_len) eof = true;
    else if (hdr_at + 28 == file_len) { // nothing but the end-of-file marker behind the range: the range IS the rest of the file
        uint8_t mk[28];
        small_read(hdr_at, mk, 28);
        if (mk[0] == 31 && mk[1] == 139 && (mk[24] | mk[25] | mk[26] | mk[27]) == 0) eof = true;
    }
}

// The resident stream of this handle's file (ResidentBam), built by the first caller; nullptr: not this file (one reference,
// too large, switched off, no room on the device, a damaged block — the per-reference path then says what is wrong with it).
//   NP2_BAM_RESIDENT_MB: largest file taken (default 1024; 0 = never)
ResidentBam *resident_for(np2_bam *bam, GpuFetch &g, hipStream_t s) {
    if (bam->resident) return bam->resident->built ? bam->resident.get() : nullptr;
    if (bam->resident_tried) return nullptr;
    bam->resident_tried = true;
    static const size_t max_mb = getenv("NP2_BAM_RESIDENT_MB") ? (size_t)std::max(0L, atol(getenv("NP2_BAM_RESIDENT_MB"))) : (size_t)1024;
    const size_t file_len = bam->map_len, c_lo = (size_t)(bam->first_rec >> 16);
    if (bam->ref_names.size() < 2 || !max_mb || file_len <= c_lo || file_len - c_lo > (max_mb << 20)) return nullptr;
    if (getenv("NP2_TEST_FETCH_NO_ROOM")) return nullptr; // (tests/test_gpu_inflate.py: a device without room)
    struct stat st;
    const int fd = fileno(bam->z.f);
    if (fstat(fd, &st) != 0) return nullptr;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static std::mutex reg_mu;
    static std::map<std::tuple<int, uint64_t, uint64_t, uint64_t, int64_t>, std::weak_ptr<ResidentBam>> reg;
    std::shared_ptr<ResidentBam> rb;
    {
        std::lock_guard<std::mutex> l(reg_mu);
        auto &w = reg[std::make_tuple(dev, (uint64_t)st.st_dev, (uint64_t)st.st_ino, (uint64_t)st.st_size, (int64_t)st.st_mtime)];
        rb = w.lock();
        if (!rb) {
            rb = std::make_shared<ResidentBam>();
            w = rb;
        }
    }
    bam->resident = rb;
    std::lock_guard<std::mutex> l(rb->mu);
    if (rb->built) return rb.get();
    if (rb->failed) return nullptr;
    const bool prof = getenv("NP2_IO_PROFILE") != nullptr;
    const double t0 = np2h::now_ms();
    rb->failed = true; // (until the stream stands)
    try {
        g.d_comp.ensure(file_len - c_lo + 64);
        bool eof = false;
        read_blocks_to_device(g, fd, file_len, c_lo, file_len, s, rb->blks, eof);
        const size_t n_blk = rb->blks.size();
        if (!n_blk || !eof) throw np2h::Np2Error(NP2_E_ARG, "BGZF blocks do not reach the end of the file");
        rb->out_off.assign(n_blk + 1, 0);
        for (size_t i = 0; i < n_blk; ++i) rb->out_off[i + 1] = rb->out_off[i] + rb->blks[i].isize;
        const uint64_t total = rb->out_off[n_blk];
        const double t1 = np2h::now_ms();
        rb->d_inf.ensure(total + 128);
        g.d_blk.ensure(n_blk + 1);
        g.d_status.ensure(n_blk + 8);
        std::vector<np2::InfBlock> tb(n_blk);
        for (size_t i = 0; i < n_blk; ++i) tb[i] = np2::InfBlock{rb->blks[i].file_off + rb->blks[i].hdr_len - c_lo, rb->out_off[i], rb->blks[i].clen, rb->blks[i].isize};
        np2::InfBlock *h_tb = (np2::InfBlock *)g.host_block(g.h_cigar, g.h_cigar_cap, n_blk * sizeof(np2::InfBlock));
        memcpy(h_tb, tb.data(), n_blk * sizeof(np2::InfBlock));
        HIPCHK(hipMemcpyAsync(g.d_blk.p, h_tb, n_blk * sizeof(np2::InfBlock), hipMemcpyHostToDevice, s));
        const size_t w_bad = (n_blk + 1) & ~(size_t)1;
        HIPCHK(hipMemsetAsync(g.d_status.p, 0, (w_bad + 4) * 4, s));
        HIPCHK(hipMemsetAsync(rb->d_inf.p + total, 0, 128, s)); // (the columnariser loads whole words behind the last SEQ)
        np2::launch_bgzf_inflate(s, g.d_blk.p, (uint32_t)n_blk, g.d_comp.p, rb->d_inf.p, g.d_status.p, g.d_status.p + w_bad);
        uint32_t n_bad = 0;
        HIPCHK(hipMemcpyAsync(h_tb, g.d_status.p + w_bad, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        memcpy(&n_bad, h_tb, 4);
        if (n_bad) throw np2h::Np2Error(NP2_E_ARG, "BGZF inflate failed");
        g.d_comp.release(); // (the file's bytes are not needed again)
        rb->built = true;
        rb->failed = false;
        if (prof)
            fprintf(stderr, "resident BAM: %zu blocks (%.1f MB -> %.1f MB) inflated on the device once: file -> device %.2f ms, inflate %.2f ms\n", n_blk,
                    (file_len - c_lo) / 1e6, total / 1e6, t1 - t0, np2h::now_ms() - t1);
    } catch (const np2h::Np2Error &) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(s);
        rb->blks.clear();
        rb->out_off.clear();
        rb->d_inf.release();
        return nullptr;
    }
    return rb.get();
}

// The records of reference `tid` (all of them: fetch(tid, 0, L)).  false: this BAM / index cannot take the device path
// (no linear index, an index entry that is not a record start) — the caller reads it the host way.
// zone [zone_lo, zone_hi): (0, L) for the whole contig, or a shard's interval (then `voffs` receives the records' BGZF
// virtual offsets: what the ranks exchange to number their reads contig-wide)
bool fetch_records_gpu(np2_bam *bam, int tid, uint32_t L, uint32_t zone_lo, uint32_t zone_hi, hipStream_t s, GpuRecs &out,
                       std::vector<uint64_t> *voffs = nullptr) {
    const bool prof = getenv("NP2_IO_PROFILE") != nullptr;
    const double t0 = np2h::now_ms();
    uint64_t start_off = bam->ref_start[tid];
    if (start_off == ~0ull) return true; // no record of this reference
    if (bam->lin[tid].empty()) return false;
    uint64_t end_hint = bam->ref_end[tid];
    {
        // where to start / stop: the smallest offset of a record overlapping the 16 kb window of zone_lo (an empty window
        // takes the next one's, like htslib); records that start at or beyond zone_hi lie behind the first record
        // overlapping the window AFTER zone_hi's
        const std::vector<uint64_t> &lin = bam->lin[tid];
        if (zone_lo > 0) {
            size_t w = std::min<size_t>(zone_lo >> 14, lin.size() - 1);
            while (w + 1 < lin.size() && lin[w] == 0) ++w;
            if (lin[w] != 0) start_off = lin[w];
        }
        if (zone_hi < L) {
            size_t w = (size_t)(zone_hi >> 14) + 1;
            while (w < lin.size() && lin[w] == 0) ++w;
            if (w < lin.size()) end_hint = lin[w];
        }
    }
    if (!bam->gpu) bam->gpu = new GpuFetch();
    GpuFetch &g = *bam->gpu;
    const int fd = fileno(bam->z.f);
    const size_t file_len = bam->map_len;
    const size_t c_lo = (size_t)(start_off >> 16);
    size_t c_hi = end_hint ? (size_t)(end_hint >> 16) : file_len; // file offset of the last block wanted
    if (getenv("NP2_TEST_FETCH_SHORT_HINT")) c_hi = c_lo; // test hook: an index that understates where the reference's records end
    std::vector<GBlk> blks;
    std::vector<uint64_t> out_off;
    size_t extra = 0; // bytes read beyond the index's end of the reference (an index that understates it costs a second round)
    ResidentBam *res = resident_for(bam, g, s); // the whole file on the device already (or now), or nullptr
    for (int round = 0;; ++round) {
        if (round > 40) throw np2h::Np2Error(NP2_E_ARG, "BAM/SAM parsing failed!");
        blks.clear();
        const size_t read_end = std::min(file_len, c_hi + 65536 + extra);
        const size_t c_bytes = read_end - c_lo;
        bool eof = false;
        const uint8_t *inf = nullptr; // the inflated stream of the blocks taken
        uint64_t total = 0;
        double t1 = t0, t_up = 0, t_inf = 0;
        size_t n_blk = 0;
        if (res) {
            // ---- the reference's stretch of the resident stream --------------------------------------------------------------
            const auto &all = res->blks;
            size_t first = (size_t)(std::lower_bound(all.begin(), all.end(), (uint64_t)c_lo, [](const GBlk &x, uint64_t v) { return x.file_off < v; }) - all.begin());
            if (first == all.size() || all[first].file_off != c_lo) return false; // the index does not point at a block of this file
            size_t last = first;
            while (last < all.size() && all[last].file_off + all[last].bsize <= read_end) ++last;
            blks.assign(all.begin() + (long)first, all.begin() + (long)last);
            eof = last == all.size() || (last + 1 == all.size() && all[last].isize == 0);
            n_blk = blks.size();
            if (!n_blk) return true;
            out_off.assign(n_blk + 1, 0);
            for (size_t i = 0; i <= n_blk; ++i) out_off[i] = res->out_off[first + i] - res->out_off[first];
            total = out_off[n_blk];
            inf = res->d_inf.p + res->out_off[first];
            g.d_status.ensure(n_blk + 8);
            t1 = t_up = t_inf = np2h::now_ms();
        } else {
            // ---- file bytes -> pinned pieces -> device; the block headers parsed on the way ---------------------------------------
            // (file bytes + inflated stream: 13 GB for a human chromosome.  No room on the device next to what else lives there ->
            // false: the host pool streams the same records through 128 MiB of host memory)
            static const bool test_no_room = getenv("NP2_TEST_FETCH_NO_ROOM") != nullptr; // (tests/test_gpu_inflate.py)
            auto room = [&](auto &buf, size_t n) {
                if (test_no_room && (void *)&buf == (void *)&g.d_inf) return false;
                try {
                    buf.ensure(n);
                } catch (const np2h::Np2Error &) {
                    (void)hipGetLastError();
                    return false;
                }
                return true;
            };
            if (!room(g.d_comp, c_bytes + 64)) return false;
            read_blocks_to_device(g, fd, file_len, c_lo, read_end, s, blks, eof);
            if (blks.empty()) {
                HIPCHK(hipStreamSynchronize(s));
                return true;
            }
            n_blk = blks.size();
            out_off.assign(n_blk + 1, 0);
            for (size_t i = 0; i < n_blk; ++i) out_off[i + 1] = out_off[i] + blks[i].isize;
            total = out_off[n_blk];
            t1 = np2h::now_ms();
            // ---- inflate --------------------------------------------------------------------------------------------------------
            if (!room(g.d_inf, total + 128)) {
                HIPCHK(hipStreamSynchronize(s)); // (the uploads into d_comp)
                return false;
            }
            inf = g.d_inf.p;
            g.d_blk.ensure(n_blk + 1);
            g.d_status.ensure(n_blk + 8);
            if (prof) {
                HIPCHK(hipStreamSynchronize(s));
                t_up = np2h::now_ms();
            }
        }
        np2::InfBlock *h_tb = (np2::InfBlock *)g.host_block(g.h_cigar, g.h_cigar_cap, std::max<size_t>(n_blk * sizeof(np2::InfBlock), 64)); // (free until the records come back)
        // status words: [0, n_blk) per block, then n_bad, walk flags, (pad), tail_at (64-bit, 8-byte aligned)
        const size_t w_bad = (n_blk + 1) & ~(size_t)1, w_flags = w_bad + 1, w_tail = w_bad + 2;
        HIPCHK(hipMemsetAsync(g.d_status.p, 0, (w_tail + 2) * 4, s));
        HIPCHK(hipMemsetAsync(g.d_status.p + w_tail, 0xFF, 8, s));
        if (!res) {
            for (size_t i = 0; i < n_blk; ++i) h_tb[i] = np2::InfBlock{blks[i].file_off + blks[i].hdr_len - c_lo, out_off[i], blks[i].clen, blks[i].isize};
            HIPCHK(hipMemcpyAsync(g.d_blk.p, h_tb, n_blk * sizeof(np2::InfBlock), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemsetAsync(g.d_inf.p + total, 0, 64, s)); // (the columnariser loads whole words behind the last SEQ)
            np2::launch_bgzf_inflate(s, g.d_blk.p, (uint32_t)n_blk, g.d_comp.p, g.d_inf.p, g.d_status.p, g.d_status.p + w_bad);
            if (prof) {
                HIPCHK(hipStreamSynchronize(s));
                t_inf = np2h::now_ms();
            }
        }
        // ---- chain starts: the linear index's record starts inside the range ----------------------------------------------------
        std::vector<uint64_t> starts;
        starts.push_back(out_off[0] + (start_off & 0xFFFF));
        {
            size_t bi = 0;
            uint64_t prev = start_off;
            for (uint64_t v : bam->lin[tid]) {
                if (v <= prev) continue; // (0 = empty window; entries repeat while one record spans several windows)
                const uint64_t fo = v >> 16;
                while (bi < n_blk && blks[bi].file_off < fo) ++bi;
                if (bi == n_blk) break;           // beyond the range read so far
                if (blks[bi].file_off != fo) return false; // not the start of a block: the index is not this file's
                const uint64_t so = out_off[bi] + (v & 0xFFFF);
                if ((v & 0xFFFF) >= blks[bi].isize) return false;
                starts.push_back(so);
                prev = v;
            }
        }
        const uint32_t n_chains = (uint32_t)starts.size();
        g.d_starts.ensure(n_chains + 1);
        g.d_info.ensure(n_chains + 1);
        g.d_off.ensure(n_chains + 1);
        uint64_t *h_st = (uint64_t *)g.host_block(g.h_recs, g.h_recs_cap, (size_t)n_chains * 8 + 64);
        memcpy(h_st, starts.data(), (size_t)n_chains * 8);
        HIPCHK(hipMemcpyAsync(g.d_starts.p, h_st, (size_t)n_chains * 8, hipMemcpyHostToDevice, s));
        np2::launch_bam_chain_count(s, inf, g.d_starts.p, n_chains, total, tid, L, zone_lo, zone_hi, g.d_info.p, g.d_status.p + w_flags,
                                    (unsigned long long *)(g.d_status.p + w_tail));
        // one wait: block statuses' summary, the walk's flags, the chains' counts
        std::vector<uint2> info(n_chains);
        uint32_t tailw[4];
        HIPCHK(hipMemcpyAsync(h_st, g.d_info.p, (size_t)n_chains * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h_tb, g.d_status.p + w_bad, 16, hipMemcpyDeviceToHost, s)); // (stream order: after the table's upload has read it)
        HIPCHK(hipStreamSynchronize(s));
        memcpy(info.data(), h_st, (size_t)n_chains * 8);
        memcpy(tailw, h_tb, 16);
        const double t2 = np2h::now_ms();
        if (tailw[0]) { // which block, and why
            std::vector<uint32_t> stv(n_blk);
            HIPCHK(hipMemcpy(stv.data(), g.d_status.p, n_blk * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n_blk; ++i)
                if (stv[i]) throw np2h::Np2Error(NP2_E_ARG, "BGZF inflate failed (block at file offset " + std::to_string(blks[i].file_off) + ", status " + std::to_string(stv[i]) + ")");
        }
        const uint32_t flags = tailw[1];
        if (flags & np2::WALK_BAD) throw np2h::Np2Error(NP2_E_ARG, "BAM/SAM parsing failed!");
        if (flags & np2::WALK_MISALIGNED) return false;
        if ((flags & (np2::WALK_TAIL | np2::WALK_AT_END)) && !eof) { // the reference's records go on beyond the index's end: further
            extra = std::max<size_t>((size_t)1 << 20, extra * 2 + c_bytes / 4);
            continue;
        }
        if (flags & np2::WALK_TAIL) throw np2h::Np2Error(NP2_E_ARG, "truncated BAM");
        // ---- offsets of the chains' records and CIGAR words, the records themselves ---------------------------------------------------
        std::vector<uint2> off(n_chains);
        uint64_t n_rec = 0, n_cig = 0;
        for (uint32_t c = 0; c < n_chains; ++c) {
            off[c] = make_uint2((uint32_t)n_rec, (uint32_t)n_cig);
            n_rec += info[c].x, n_cig += info[c].y;
        }
        if (n_rec > 0xFFFFFFF0ull || n_cig > 0xFFFFFFF0ull) throw np2h::Np2Error(NP2_E_NOMEM, "too many records for one contig");
        out.n_recs = (uint32_t)n_rec;
        out.d_stream = inf;
        out.stream_bytes = total + 16;
        if (n_rec) {
            g.d_recs.ensure(n_rec + 1);
            g.d_cig_src.ensure(n_rec + 1);
            g.d_cigar.ensure(n_cig + 1);
            memcpy(h_st, off.data(), (size_t)n_chains * 8);
            HIPCHK(hipMemcpyAsync(g.d_off.p, h_st, (size_t)n_chains * 8, hipMemcpyHostToDevice, s));
            np2::launch_bam_chain_write(s, inf, g.d_starts.p, n_chains, total, tid, L, zone_lo, zone_hi, g.d_off.p, g.d_recs.p, g.d_cig_src.p);
            np2::launch_bam_cigars(s, inf, g.d_recs.p, g.d_cig_src.p, (uint32_t)n_rec, g.d_cigar.p);
            HIPCHK(hipStreamSynchronize(s)); // (h_st is about to be given up for a larger block)
            if (prof) fprintf(stderr, "  fetch_records_gpu: offsets + write + cigars %.2f ms\n", np2h::now_ms() - t2);
            np2_bamrec_t *hr = (np2_bamrec_t *)g.host_block(g.h_recs, g.h_recs_cap, n_rec * sizeof(np2_bamrec_t) + 64);
            uint32_t *hc = (uint32_t *)g.host_block(g.h_cigar, g.h_cigar_cap, n_cig * 4 + 64);
            HIPCHK(hipMemcpyAsync(hr, g.d_recs.p, n_rec * sizeof(np2_bamrec_t), hipMemcpyDeviceToHost, s));
            if (n_cig) HIPCHK(hipMemcpyAsync(hc, g.d_cigar.p, n_cig * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            out.recs = hr, out.cigar = hc;
            if (voffs) { // stream offset of every record's start -> (file offset of its block << 16 | offset inside the block)
                std::vector<uint64_t> src(n_rec);
                HIPCHK(hipMemcpy(src.data(), g.d_cig_src.p, n_rec * 8, hipMemcpyDeviceToHost));
                voffs->resize(n_rec);
                IoPool::get().parallel_for((size_t)((n_rec + 4095) / 4096), 16, [&](size_t blk) {
                    for (uint64_t i = blk * 4096; i < std::min<uint64_t>(n_rec, (blk + 1) * 4096); ++i) {
                        const uint64_t at = src[i] - 36u - hr[i].pad; // (first byte of the record's block_size field)
                        const size_t bi = (size_t)(std::upper_bound(out_off.begin(), out_off.end(), at) - out_off.begin()) - 1;
                        (*voffs)[i] = (blks[bi].file_off << 16) | (at - out_off[bi]);
                    }
                });
            }
        }
        if (prof)
            fprintf(stderr, "fetch_records_gpu: %zu blocks (%.1f MB -> %.1f MB), %u chains, %llu records, %llu CIGAR words: file -> pinned pieces (+ block headers) %.2f ms, "
                            "rest of the upload %.2f ms, inflate %.2f ms, index + count + wait %.2f ms, records back %.2f ms%s\n", n_blk, c_bytes / 1e6, total / 1e6, n_chains,
                    (unsigned long long)n_rec, (unsigned long long)n_cig, t1 - t0, t_up - t1, t_inf - t_up, t2 - t_inf, np2h::now_ms() - t2, res ? (round ? " (stretch of the resident stream, after extending the range)" : " (stretch of the resident stream)") : (round ? " (after extending the range)" : ""));
        return true;
    }
}
} // namespace

int np2_bgzf_inflate_device(np2_ctx_t *cx, const uint8_t *bgzf, uint64_t n, uint8_t *out, uint64_t out_cap, uint64_t *out_len,
                            float *kernel_ms) {
    if (!cx || (!bgzf && n) || !out_len || (!out && out_cap)) return NP2_E_ARG;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    try {
        HIPCHK(hipSetDevice(cx->device));
        hipStream_t s = cx->stream;
        BgzfBatch hdr;
        hdr.map = bgzf, hdr.map_len = (size_t)n, hdr.fpos = 0;
        std::vector<np2::InfBlock> tb;
        uint64_t total = 0;
        for (;;) {
            BgzfBatch::Blk b;
            if (!hdr.read_raw(b)) break;
            tb.push_back(np2::InfBlock{(uint64_t)(b.c - bgzf), total, b.clen, b.isize});
            total += b.isize;
        }
        *out_len = total;
        if (total > out_cap) throw np2h::Np2Error(NP2_E_ARG, "output buffer too small");
        if (tb.empty()) return NP2_OK;
        GpuFetch g;
        g.d_comp.ensure(n + 64);
        g.d_inf.ensure(total + 64);
        g.d_blk.ensure(tb.size() + 1);
        g.d_status.ensure(tb.size() + 8);
        g.upload(bgzf, (size_t)n, g.d_comp.p, s);
        HIPCHK(hipMemcpyAsync(g.d_blk.p, tb.data(), tb.size() * sizeof(np2::InfBlock), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemsetAsync(g.d_status.p, 0, (tb.size() + 4) * 4, s));
        HIPCHK(hipStreamSynchronize(s)); // (tb is pageable: the copy must have read it before it goes)
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, s));
        np2h::DevBuf<uint64_t> d_prof;
        const bool kprof = getenv("NP2_INF_PROF") != nullptr; // (phase clocks of the inflate kernel: a tool's switch)
        if (kprof) {
            d_prof.ensure(tb.size() * 8 + 8);
            HIPCHK(hipMemsetAsync(d_prof.p, 0, tb.size() * 64, s));
        }
        np2::launch_bgzf_inflate(s, g.d_blk.p, (uint32_t)tb.size(), g.d_comp.p, g.d_inf.p, g.d_status.p, g.d_status.p + tb.size(),
                                 kprof ? (unsigned long long *)d_prof.p : nullptr, getenv("NP2_INF_PROBE") ? (uint32_t)atoi(getenv("NP2_INF_PROBE")) : 0u);
        HIPCHK(hipEventRecord(e1, s));
        if (kprof) {
            std::vector<uint64_t> pr(tb.size() * 8);
            HIPCHK(hipMemcpyAsync(pr.data(), d_prof.p, pr.size() * 8, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            double sum[8] = {0};
            for (size_t i = 0; i < tb.size(); ++i)
                for (int k = 0; k < 8; ++k) sum[k] += (double)pr[i * 8 + k];
            const double nb = (double)tb.size();
            fprintf(stderr, "[inf_prof] %zu blocks; mean clocks per block: total %.0f, wide decode %.0f, chain %.0f (of which match copies %.0f); tokens %.0f, matches %.0f; "
                            "table builds %.0f clocks over %.1f deflate blocks\n",
                    tb.size(), sum[0] / nb, sum[1] / nb, sum[2] / nb, sum[3] / nb, sum[4] / nb, sum[5] / nb, sum[6] / nb, sum[7] / nb);
        }
        std::vector<uint32_t> st(tb.size() + 1);
        HIPCHK(hipMemcpyAsync(st.data(), g.d_status.p, st.size() * 4, hipMemcpyDeviceToHost, s));
        if (total) HIPCHK(hipMemcpyAsync(out, g.d_inf.p, total, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (kernel_ms) HIPCHK(hipEventElapsedTime(kernel_ms, e0, e1));
        (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
        e0 = e1 = nullptr;
        for (size_t i = 0; i < tb.size(); ++i)
            if (st[i]) throw np2h::Np2Error(NP2_E_ARG, "BGZF inflate failed (block " + std::to_string(i) + ", status " + std::to_string(st[i]) + ")");
    } catch (const np2h::Np2Error &e) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipStreamSynchronize(cx->stream);
        return io_fail(e.code, e.what());
    } catch (const std::exception &ex) {
        (void)hipStreamSynchronize(cx->stream);
        return io_fail(NP2_E_NOMEM, ex.what());
    }
    return NP2_OK;
}

int np2_contig_from_bam(np2_ctx_t *cx, np2_bam_t *bam, const char *name, const uint8_t *ref, uint32_t L,
                        const np2_front_opts_t *opts, np2_contig_t **out) {
    if (!cx || !bam || !name || !ref || !opts || !out) return NP2_E_ARG;
    *out = nullptr;
    try {
        int tid = -1;
        for (size_t i = 0; i < bam->ref_names.size(); ++i)
            if (bam->ref_names[i] == name) tid = (int)i;
        if (tid < 0) throw np2h::Np2Error(NP2_E_ARG, std::string("Faield random access BAM/SAM! (contig not in the BAM header: ") + name + ")");
        std::vector<np2_bamrec_t> recs;
        std::vector<uint32_t> cigar;
        PinnedBytes &seq4 = bam->seq4;
        const bool prof = getenv("NP2_IO_PROFILE") != nullptr;
        const double t_p0 = np2h::now_ms();
        bam->batch.ms_read = bam->batch.ms_inflate = bam->batch.ms_drop = bam->batch.ms_walk = bam->batch.ms_size = bam->batch.ms_copy = 0;
        HIPCHK(hipSetDevice(cx->device));
        uint64_t seq_bytes = 0;
        if (gpu_fetch_wanted(bam, tid) && !opts->use_secondary) { // read extraction on the device (fetch_records_gpu)
            GpuRecs gr;
            if (fetch_records_gpu(bam, tid, L, 0, L, cx->stream, gr)) { // (false: not this BAM / index, or no room on the device)
                const double t_g1 = np2h::now_ms();
                FrontWork fw;
                front_begin(cx, ref, L, gr.recs, gr.n_recs, gr.cigar, nullptr, gr.n_recs ? gr.stream_bytes : 0, opts, nullptr, fw, gr.n_recs ? gr.d_stream : nullptr);
                front_finish(cx, fw, out);
                if (prof) fprintf(stderr, "np2_contig_from_bam %s: device read extraction %.2f ms (%u records), records->pileup %.2f ms\n", name, t_g1 - t_p0, gr.n_recs, np2h::now_ms() - t_g1);
                np2h::flush_timings(cx);
                return NP2_OK;
            }
        }
        fetch_records(bam, tid, L, 0, L, opts, recs, cigar, nullptr, cx->stream, &seq_bytes);
        const double t_p1 = np2h::now_ms();
        {
            const bool streamed = !opts->use_secondary;
            FrontWork fw;
            front_begin(cx, ref, L, recs.data(), (uint32_t)recs.size(), cigar.data(), seq4.data(), seq_bytes, opts, nullptr, fw,
                        streamed ? bam->seqs.dev.p : nullptr);
            front_finish(cx, fw, out);
        }
        if (prof)
            fprintf(stderr, "np2_contig_from_bam %s: inflate+parse %.2f ms (block headers %.2f, inflate [%s] %.2f, buffer moves %.2f, record "
                            "walk %.2f, array sizing %.2f, record copies %.2f; %zu records, %zu SEQ bytes), records->pileup %.2f ms\n",
                    name, t_p1 - t_p0, bam->batch.ms_read, Inflater::get().name(), bam->batch.ms_inflate, bam->batch.ms_drop,
                    bam->batch.ms_walk, bam->batch.ms_size, bam->batch.ms_copy, recs.size(), (size_t)seq_bytes, np2h::now_ms() - t_p1);
        np2h::flush_timings(cx);
    } catch (const np2h::Np2Error &e) {
        (void)hipStreamSynchronize(cx->stream);
        np2h::flush_timings(cx);
        return np2h::fail(cx, e);
    } catch (const std::exception &ex) {
        (void)hipStreamSynchronize(cx->stream);
        np2h::flush_timings(cx);
        return np2h::fail(cx, np2h::Np2Error(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what()));
    }
    return NP2_OK;
}

// ---- a reference-interval shard straight from the BAM -----------------------------------------------------------------
// Each rank reads only the records overlapping its zone (own interval +- halo; .bai linear index), admits and
// columnarises them on its GPU, and learns which of them are pushed (main.rs:1798-1813).  The contig-wide read numbers
// (the reference numbers pushed records in file order, main.rs:1813) come from one exchange: every rank publishes
// the BGZF virtual offsets of the pushed records that START in its own interval (it sees all of those); the lists,
// concatenated in rank order, are the contig's pushed records in file order, and a record's number is 1 + its place.
struct np2_shard_io {
    np2_ctx *cx = nullptr;
    FrontWork fw;
    np2_shard_plan_t plan{};
    uint32_t L = 0;
    std::vector<uint64_t> rec_voff;  // per fetched record
    std::vector<uint64_t> own_voff;  // pushed records starting in [own_lo, own_hi), file order
};

int np2_shard_bam_begin(np2_ctx_t *cx, np2_bam_t *bam, const char *name, const uint8_t *ref, uint32_t L, uint32_t own_lo,
                        uint32_t own_hi, uint32_t halo, const np2_front_opts_t *opts, np2_shard_io_t **out,
                        const uint64_t **own_voffsets, uint64_t *n_own) {
    if (!cx || !bam || !name || !ref || !opts || !out || !own_voffsets || !n_own || own_lo >= own_hi || own_hi > L)
        return NP2_E_ARG;
    *out = nullptr;
    np2_shard_io *io = new np2_shard_io();
    io->cx = cx;
    io->L = L;
    try {
        int tid = -1;
        for (size_t i = 0; i < bam->ref_names.size(); ++i)
            if (bam->ref_names[i] == name) tid = (int)i;
        if (tid < 0) throw np2h::Np2Error(NP2_E_ARG, std::string("Faield random access BAM/SAM! (contig not in the BAM header: ") + name + ")");
        np2_shard_plan_t &pl = io->plan;
        pl.own_lo = own_lo, pl.own_hi = own_hi;
        pl.zone_lo = own_lo > halo ? own_lo - halo : 0u;
        pl.zone_hi = (uint64_t)own_hi + halo < L ? own_hi + halo : L;
        std::vector<np2_bamrec_t> recs;
        std::vector<uint32_t> cigar;
        HIPCHK(hipSetDevice(cx->device));
        uint64_t seq_bytes = 0;
        // the interval's records: extracted on the device (fetch_records_gpu: a rank of eight has two of the node's CPUs) or
        // by the host pool
        GpuRecs gr;
        const bool on_device = gpu_fetch_wanted(bam, tid) && !opts->use_secondary &&
                               fetch_records_gpu(bam, tid, L, pl.zone_lo, pl.zone_hi, cx->stream, gr, &io->rec_voff);
        if (on_device) {
            recs.assign(gr.recs, gr.recs + gr.n_recs);
            seq_bytes = gr.n_recs ? gr.stream_bytes : 0;
        } else {
            io->rec_voff.clear();
            fetch_records(bam, tid, L, pl.zone_lo, pl.zone_hi, opts, recs, cigar, &io->rec_voff, cx->stream, &seq_bytes);
        }
        const uint32_t *cigar_p = on_device ? gr.cigar : cigar.data();
        // the sub-contig: from the first start to the last reference end among the fetched records
        uint32_t slo = pl.zone_lo, shi = pl.zone_hi;
        for (const np2_bamrec_t &r : recs) {
            uint64_t span = 0;
            for (uint32_t k = 0; k < r.n_cigar; ++k) {
                const uint32_t w = cigar_p[r.cigar_off + k], op = w & 15;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += w >> 4;
            }
            if (r.pos >= 0) slo = std::min<uint32_t>(slo, (uint32_t)r.pos);
            shi = (uint32_t)std::min<uint64_t>(L, std::max<uint64_t>(shi, (uint64_t)r.pos + span));
        }
        pl.sub_lo = own_lo == 0 ? 0u : (slo & ~63u);
        pl.sub_hi = own_hi == L ? L : shi;
        ShardSpec sp;
        sp.sub_lo = pl.sub_lo, sp.sub_hi = pl.sub_hi, sp.zone_lo = pl.zone_lo, sp.zone_hi = pl.zone_hi;
        front_begin(cx, ref, L, recs.data(), (uint32_t)recs.size(), cigar_p, on_device ? nullptr : bam->seq4.data(), seq_bytes, opts, &sp, io->fw,
                    on_device ? (gr.n_recs ? gr.d_stream : nullptr) : (opts->use_secondary ? nullptr : bam->seqs.dev.p));
        for (size_t i = 1; i < io->fw.reads.size(); ++i) {
            const np2_bamrec_t &r = recs[io->fw.rec_of[i]];
            if ((uint32_t)r.pos >= own_lo && (uint32_t)r.pos < own_hi) io->own_voff.push_back(io->rec_voff[io->fw.rec_of[i]]);
        }
        np2h::flush_timings(cx);
    } catch (const np2h::Np2Error &e) {
        (void)hipStreamSynchronize(cx->stream);
        delete io;
        return np2h::fail(cx, e);
    } catch (const std::exception &ex) {
        (void)hipStreamSynchronize(cx->stream);
        delete io;
        return np2h::fail(cx, np2h::Np2Error(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what()));
    }
    *own_voffsets = io->own_voff.data();
    *n_own = io->own_voff.size();
    *out = io;
    return NP2_OK;
}

void np2_shard_bam_abort(np2_shard_io_t *io) { delete io; }

int np2_shard_bam_finish(np2_shard_io_t *io, const uint64_t *all_voffsets, uint64_t n_all, np2_shard_plan_t *plan,
                         np2_contig_t **contig, uint32_t *n_reads_total) {
    if (!io || (n_all && !all_voffsets) || !plan || !contig || !n_reads_total) return NP2_E_ARG;
    np2_ctx *cx = io->cx;
    *contig = nullptr;
    int rc = NP2_OK;
    try {
        FrontWork &fw = io->fw;
        // contig-wide number of every pushed record this shard fetched
        const size_t n = fw.reads.size();
        std::vector<uint32_t> gid(n, 0);
        for (size_t i = 1; i < n; ++i) {
            const uint64_t v = io->rec_voff[fw.rec_of[i]];
            const uint64_t *it = std::lower_bound(all_voffsets, all_voffsets + n_all, v);
            if (it == all_voffsets + n_all || *it != v)
                throw np2h::Np2Error(NP2_E_ARG, "shard: a pushed record is missing from the exchanged offsets (do the ranks' own intervals tile the contig?)");
            gid[i] = 1 + (uint32_t)(it - all_voffsets);
            if (i > 1 && gid[i] <= gid[i - 1]) throw np2h::Np2Error(NP2_E_ARG, "shard: exchanged offsets are not in file order");
        }
        np2_shard_plan_t &pl = io->plan;
        pl.read_lo = n > 1 ? gid[1] : 1u;
        pl.read_hi = n > 1 ? gid[n - 1] + 1 : 1u;
        // holes: reads of the contig inside [read_lo, read_hi) this shard does not hold keep their number
        std::vector<np2_read_t> reads(1 + (size_t)(pl.read_hi - pl.read_lo));
        std::vector<uint8_t> lable(reads.size(), 0);
        np2_read_t hole;
        memset(&hole, 0, sizeof hole);
        hole.flags = NP2_READ_DROPPED;
        hole.nib_off = 0; // (never decoded; slot 0 is a valid aligned offset)
        std::fill(reads.begin(), reads.end(), hole);
        reads[0] = fw.reads[0];
        for (size_t i = 1; i < n; ++i) {
            reads[gid[i] - pl.read_lo + 1] = fw.reads[i];
            lable[gid[i] - pl.read_lo + 1] = fw.lable[i];
        }
        fw.reads.swap(reads);
        fw.lable.swap(lable);
        np2_contig *c = nullptr;
        front_finish(cx, fw, &c);
        *contig = c;
        *plan = pl;
        *n_reads_total = 1 + (uint32_t)n_all;
        np2h::flush_timings(cx);
    } catch (const np2h::Np2Error &e) {
        (void)hipStreamSynchronize(cx->stream);
        rc = np2h::fail(cx, e);
    } catch (const std::exception &ex) {
        (void)hipStreamSynchronize(cx->stream);
        rc = np2h::fail(cx, np2h::Np2Error(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what()));
    }
    delete io;
    return rc;
}

int np2_contig_export(np2_ctx_t *cx, np2_contig_t *c, np2_read_t **reads, uint32_t *n_reads, uint8_t **nibbles,
                      uint64_t *nib_bytes) {
    if (!cx || !c || !reads || !n_reads || !nibbles || !nib_bytes) return NP2_E_ARG;
    try {
        HIPCHK(hipSetDevice(cx->device));
        *reads = (np2_read_t *)malloc((size_t)c->R * sizeof(np2_read_t));
        *nibbles = (uint8_t *)malloc(c->nib_bytes);
        HIPCHK(hipMemcpy(*reads, c->reads.p, (size_t)c->R * sizeof(np2_read_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(*nibbles, c->nib.p, c->nib_bytes, hipMemcpyDeviceToHost));
        *n_reads = c->R;
        *nib_bytes = c->nib_bytes;
    } catch (const np2h::Np2Error &e) {
        return np2h::fail(cx, e);
    } catch (const std::exception &ex) {
        return np2h::fail(cx, np2h::Np2Error(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what()));
    }
    return NP2_OK;
}
}
