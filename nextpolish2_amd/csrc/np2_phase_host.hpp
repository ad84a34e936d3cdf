// Host side of the phasing vote: signed-weight Louvain over the read graph and the
// community ranking (src/utils/louvain.rs:59-356, src/main.rs:948-1015).
//
// Louvain local moving is Gauss-Seidel sequential per connected component, so it stays on
// the host; the +-1 edges it consumes are derived from the GPU-built
// candidate tables.  Tie order between conflicting communities in the reference follows the
// iteration order of Rust's FxHashMap/FxHashSet<u32> (hashbrown SwissTable, SSE2 group width
// 16, fxhash 0.2.1): SwissOrderMap below reproduces that *order* (insert / entry-insert /
// erase / retain / grow / in-place rehash), not the container's performance.
#pragma once
#include <emmintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <unordered_map>
#include <unordered_set>
#include <memory>
#include <mutex>
#include <exception>
#include <thread>
#include <utility>
#include "np2_hostcpu.hpp"
#include <vector>

namespace np2 {
namespace phase {

// Host threads for the order-free parts of a LARGE vote (the read graph of a chromosome-sized diploid contig: 3 x 10^5
// nodes, 2 x 10^7 pairs): building the adjacency rows and the per-community sums of an aggregation.  Everything whose
// ORDER the reference's hash containers make observable (key creation, community iteration, new_nid renaming, the
// Gauss-Seidel sweep itself) stays on one thread.  Small votes never start a thread.
inline unsigned host_threads() {
    static const unsigned n = [] {
        if (const char *e = getenv("NP2_VOTE_THREADS")) return (unsigned)std::max(1, atoi(e));
        return std::min(16u, std::max(1u, np2h::usable_cpus() / np2h::local_ranks())); // (this rank's share of the allowed CPUs)
    }();
    return n;
}
// f(t, lo, hi) over T = min(host_threads, n / min_per_thread) contiguous pieces of [0, n); the caller's thread takes piece 0
template <class F> void parallel_ranges(size_t n, size_t min_per_thread, F f) {
    const size_t T = std::max<size_t>(1, std::min<size_t>(host_threads(), min_per_thread ? n / min_per_thread : n));
    if (T <= 1) {
        f((unsigned)0, (size_t)0, n);
        return;
    }
    // (an exception in a piece — bad_alloc from a per-thread array — is carried to the caller's thread, not std::terminate)
    std::vector<std::exception_ptr> err(T);
    auto guarded = [&](size_t t, size_t lo, size_t hi) {
        try {
            f((unsigned)t, lo, hi);
        } catch (...) {
            err[t] = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    th.reserve(T);
    for (size_t t = 1; t < T; ++t) {
        try {
            th.emplace_back([&, t] { guarded(t, n * t / T, n * (t + 1) / T); });
        } catch (...) { // (no thread to be had: the piece runs here; the threads already started are joined below)
            guarded(t, n * t / T, n * (t + 1) / T);
        }
    }
    guarded(0, (size_t)0, n / T);
    for (auto &x : th) x.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}

template <class V> class SwissOrderMap {
  public:
    static constexpr uint8_t kEmpty = 0xFF, kDeleted = 0x80;
    static constexpr size_t kGroup = 16;
    static constexpr size_t npos = (size_t)-1;
    SwissOrderMap() : ctrl_(kGroup, kEmpty) {}

    size_t size() const { return items_; }
    bool empty() const { return items_ == 0; }

    size_t find(uint32_t key) const {
        if (!alloc_) return npos;
        const uint64_t h = hash(key);
        const uint8_t tag = (uint8_t)(h >> 57);
        size_t pos = (size_t)h & mask_, stride = 0;
        const __m128i vtag = _mm_set1_epi8((char)tag), vempty = _mm_set1_epi8((char)kEmpty);
        for (;;) { // (a group of 16 control bytes per step, like hashbrown's own SSE2 group)
            const __m128i g = _mm_loadu_si128(reinterpret_cast<const __m128i *>(ctrl_.data() + pos));
            unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(g, vtag));
            while (m) {
                const size_t i = (pos + (size_t)__builtin_ctz(m)) & mask_;
                m &= m - 1;
                if (keys_[i] == key && !(ctrl_[i] & 0x80)) return i;
            }
            if (_mm_movemask_epi8(_mm_cmpeq_epi8(g, vempty))) return npos;
            stride += kGroup;
            pos = (pos + stride) & mask_;
        }
    }
    bool has(uint32_t key) const { return find(key) != npos; }
    V *get(uint32_t key) {
        const size_t i = find(key);
        return i == npos ? nullptr : &vals_[i];
    }
    const V *get(uint32_t key) const {
        const size_t i = find(key);
        return i == npos ? nullptr : &vals_[i];
    }
    // HashMap::insert: grows only if the chosen slot is EMPTY and no growth is left
    void put(uint32_t key, V v) {
        const size_t f = find(key);
        if (f != npos) {
            vals_[f] = std::move(v);
            return;
        }
        const uint64_t h = hash(key);
        size_t s = probe_free(h);
        const uint8_t old = ctrl_[s];
        if (growth_ == 0 && (old & 1)) {
            grow_or_rehash(1);
            s = probe_free(h);
        }
        place(s, old, h, key, std::move(v));
    }
    // Entry::or_insert* on a vacant key: reserve(1) first (std's rustc_entry), then no-grow insert
    V &put_vacant(uint32_t key, V v) {
        if (growth_ < 1) grow_or_rehash(1);
        const uint64_t h = hash(key);
        const size_t s = probe_free(h);
        place(s, ctrl_[s], h, key, std::move(v));
        return vals_[s];
    }
    void reserve(size_t n) {
        if (n > growth_) grow_or_rehash(n);
    }
    bool take(uint32_t key, V *out) {
        const size_t i = find(key);
        if (i == npos) return false;
        if (out) *out = std::move(vals_[i]);
        erase_at(i);
        return true;
    }
    template <class P> void keep_if(P pred) {
        if (!alloc_) return;
        for (size_t i = 0; i <= mask_; ++i)
            if (!(ctrl_[i] & 0x80) && !pred(keys_[i], vals_[i])) erase_at(i);
    }
    template <class F> void each(F f) const {
        if (!alloc_) return;
        for (size_t i = 0; i <= mask_; ++i)
            if (!(ctrl_[i] & 0x80)) f(keys_[i], vals_[i]);
    }
    template <class F> void each_mut(F f) {
        if (!alloc_) return;
        for (size_t i = 0; i <= mask_; ++i)
            if (!(ctrl_[i] & 0x80)) f(keys_[i], vals_[i]);
    }
    std::vector<uint32_t> key_list() const {
        std::vector<uint32_t> k;
        k.reserve(items_);
        each([&](uint32_t key, const V &) { k.push_back(key); });
        return k;
    }
    static SwissOrderMap single(uint32_t key, V v) { // FromIterator<[T; 1]>
        SwissOrderMap m;
        m.reserve(1);
        m.put(key, std::move(v));
        return m;
    }

  private:
    size_t mask_ = 0, growth_ = 0, items_ = 0;
    bool alloc_ = false;
    std::vector<uint8_t> ctrl_;
    std::vector<uint32_t> keys_;
    std::vector<V> vals_;

    static uint64_t hash(uint32_t k) { return (uint64_t)k * 0x517cc1b727220a95ULL; } // FxHasher, one u32 word
    static size_t cap_of(size_t mask) { return mask < 8 ? mask : ((mask + 1) >> 3) * 7; }
    static size_t buckets_for(size_t cap) {
        if (cap < 4) return 4;
        if (cap < 8) return 8;
        size_t want = cap * 8 / 7, n = 1;
        while (n < want) n <<= 1;
        return n;
    }
    void mark(size_t i, uint8_t c) {
        ctrl_[i] = c;
        ctrl_[((i - kGroup) & mask_) + kGroup] = c;
    }
    size_t probe_free(uint64_t h) const {
        size_t pos = (size_t)h & mask_, stride = 0;
        for (;;) {
            // (empty and deleted control bytes have their top bit set)
            const unsigned m = (unsigned)_mm_movemask_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(ctrl_.data() + pos)));
            if (m) {
                size_t r = (pos + (size_t)__builtin_ctz(m)) & mask_;
                if (!(ctrl_[r] & 0x80)) // tiny table: matched a trailing byte, rescan from 0
                    for (r = 0; !(ctrl_[r] & 0x80); ++r) {
                    }
                return r;
            }
            stride += kGroup;
            pos = (pos + stride) & mask_;
        }
    }
    void place(size_t s, uint8_t old, uint64_t h, uint32_t key, V v) {
        growth_ -= (old & 1);
        mark(s, (uint8_t)(h >> 57));
        keys_[s] = key;
        vals_[s] = std::move(v);
        ++items_;
    }
    void erase_at(size_t i) {
        const size_t before = (i - kGroup) & mask_;
        size_t lead = 0, trail = 0;
        for (size_t b = kGroup; b-- > 0 && ctrl_[before + b] != kEmpty;) ++lead;
        for (size_t b = 0; b < kGroup && ctrl_[i + b] != kEmpty; ++b) ++trail;
        if (lead + trail >= kGroup) {
            mark(i, kDeleted);
        } else {
            mark(i, kEmpty);
            ++growth_;
        }
        --items_;
    }
    void grow_or_rehash(size_t extra) {
        const size_t want = items_ + extra;
        const size_t full = alloc_ ? cap_of(mask_) : 0;
        if (want <= full / 2)
            rehash_here();
        else
            rebuild(std::max(want, full + 1));
    }
    void rebuild(size_t cap) {
        const size_t nb = buckets_for(cap);
        SwissOrderMap n;
        n.alloc_ = true;
        n.mask_ = nb - 1;
        n.ctrl_.assign(nb + kGroup, kEmpty);
        n.keys_.assign(nb, 0);
        n.vals_.resize(nb);
        if (alloc_)
            for (size_t i = 0; i <= mask_; ++i) {
                if (ctrl_[i] & 0x80) continue;
                const uint64_t h = hash(keys_[i]);
                const size_t s = n.probe_free(h);
                n.mark(s, (uint8_t)(h >> 57));
                n.keys_[s] = keys_[i];
                n.vals_[s] = std::move(vals_[i]);
            }
        n.items_ = items_;
        n.growth_ = cap_of(n.mask_) - items_;
        *this = std::move(n);
    }
    void rehash_here() {
        const size_t nb = mask_ + 1;
        for (auto &c : ctrl_) c = (c & 0x80) ? kEmpty : kDeleted;
        if (nb < kGroup) {
            for (size_t i = 0; i < nb; ++i) ctrl_[kGroup + i] = ctrl_[i];
            for (size_t i = nb; i < kGroup; ++i) ctrl_[i] = kEmpty;
        } else {
            for (size_t i = 0; i < kGroup; ++i) ctrl_[nb + i] = ctrl_[i];
        }
        for (size_t i = 0; i < nb; ++i) {
            if (ctrl_[i] != kDeleted) continue;
            for (;;) {
                const uint64_t h = hash(keys_[i]);
                const size_t home = (size_t)h & mask_;
                const size_t j = probe_free(h);
                if ((((i - home) & mask_) / kGroup) == (((j - home) & mask_) / kGroup)) {
                    mark(i, (uint8_t)(h >> 57));
                    break;
                }
                const uint8_t prev = ctrl_[j];
                mark(j, (uint8_t)(h >> 57));
                if (prev == kEmpty) {
                    mark(i, kEmpty);
                    keys_[j] = keys_[i];
                    vals_[j] = std::move(vals_[i]);
                    break;
                }
                std::swap(keys_[i], keys_[j]);
                std::swap(vals_[i], vals_[j]);
            }
        }
        growth_ = cap_of(mask_) - items_;
    }
};

struct Nil {};
typedef SwissOrderMap<Nil> OrderSet;

// Signed weighted read graph.  `keys` reproduces the creation order of the outer keys of the reference's
// HashMap<u32, HashMap<u32, f32>> (only that order is observable); the rows are CSR adjacency lists
// (row iteration order only feeds exact small-integer f32 sums, so it is free).
// std::allocator whose value-less construct() leaves the element as it is: resize() of a 270 MB edge array must not
// zero what the fill loops overwrite a moment later (12 of the 19 ms the rows of a chromosome's graph took to build),
// and its pages are then first touched by the threads that fill them
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind {
        typedef NoInitAlloc<U> other;
    };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U> void construct(U *) noexcept {}
    template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a) {
        ::new ((void *)p) U(std::forward<A0>(a0), std::forward<A>(a)...);
    }
};
// f(t) on T threads (the caller's takes t = 0); an exception in any of them — bad_alloc from a per-thread array — is carried
// to the caller's thread instead of ending the process
template <class F> void parallel_each(size_t T, F f) {
    std::vector<std::exception_ptr> err(T);
    auto guarded = [&](size_t t) {
        try {
            f(t);
        } catch (...) {
            err[t] = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    th.reserve(T);
    for (size_t t = 1; t < T; ++t) {
        try {
            th.emplace_back([&, t] { guarded(t); });
        } catch (...) { // (no thread to be had: the piece runs here; the threads already started are joined below)
            guarded(t);
        }
    }
    guarded(0);
    for (auto &x : th) x.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}
struct Graph {
    typedef std::pair<uint32_t, float> Edge;
    typedef std::vector<Edge, NoInitAlloc<Edge>> EdgeVec;
    // The edge array of a chromosome's first-level graph is 150-300 MB.  Giving it back to the kernel page by page took
    // as long as a sweep over it, and the next vote's array faulted the same pages in again (most of its "fill"): ONE
    // retired array is kept for the next large graph of the process instead (rounds 4-5 freed it on a detached thread,
    // which could outlive the library's image: ADVICE round 5).
    struct Spare {
        std::mutex m;
        EdgeVec v;
    };
    static Spare &spare() {
        static Spare *s = new Spare(); // (leaked on purpose: no destructor order to get wrong at exit)
        return *s;
    }
    static void retire(EdgeVec &&v) {
        Spare &sp = spare();
        std::lock_guard<std::mutex> lk(sp.m);
        if (v.capacity() > sp.v.capacity()) sp.v.swap(v);
        // (the smaller of the two is freed here, on the caller's thread)
    }
    void edges_resize(size_t n) { // edges.resize(n), out of the retired array if that one is large enough and ours is not
        if (n > edges.capacity() && n >= ((size_t)1 << 20)) {
            Spare &sp = spare();
            std::lock_guard<std::mutex> lk(sp.m);
            if (sp.v.capacity() >= n) {
                edges.swap(sp.v);
                sp.v = EdgeVec();
            }
        }
        edges.resize(n);
    }
    bool rows_sorted = false; // every row: its neighbours below it ascending, then those above it ascending (a vote's rows)
    // row iteration helper
    struct Row {
        const Edge *b, *e;
        const Edge *begin() const { return b; }
        const Edge *end() const { return e; }
        bool empty() const { return b == e; }
    };
    OrderSet keys;
    std::vector<uint8_t> is_key; // dense mirror of `keys` (membership tests without hashing)
    std::vector<uint32_t> off;   // CSR row offsets, n_ids() + 1 entries
    EdgeVec edges; // directed copies of the undirected edges, grouped by source node

    uint32_t n_ids() const { return (uint32_t)is_key.size(); }
    void reserve_ids(uint32_t n) {
        if (n > is_key.size()) {
            is_key.resize(n, 0);
            off.resize((size_t)n + 1, (uint32_t)edges.size());
        }
    }
    void add_key(uint32_t k) { // Entry::or_insert_with on a vacant key
        reserve_ids(k + 1);
        if (!is_key[k]) {
            keys.put_vacant(k, Nil{});
            is_key[k] = 1;
        }
    }
    bool has_key(uint32_t k) const { return k < is_key.size() && is_key[k]; }
    Row adj(uint32_t v) const { return Row{edges.data() + off[v], edges.data() + off[v + 1]}; }
    // (Re)build the rows from a list of undirected weighted edges: count, prefix, fill.  Returns false if an
    // endpoint is not a key.
    // `skip` (optional, per node): edges with a flagged endpoint are left out — what drop_nodes would remove afterwards.
    template <class GetA, class GetB, class GetW>
    bool add_edges(uint64_t n, GetA ga, GetB gb, GetW gw, const uint8_t *skip = nullptr) {
        const uint32_t N = n_ids();
        rows_sorted = false;
        off.assign((size_t)N + 1, 0);
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t a = ga(i), b = gb(i);
            if (!has_key(a) || !has_key(b)) return false;
            if (skip && (skip[a] | skip[b])) continue;
            ++off[a + 1];
            ++off[b + 1];
        }
        for (uint32_t v = 0; v < N; ++v) off[v + 1] += off[v];
        edges.resize(off[N]);
        std::vector<uint32_t> cur(off.begin(), off.end() - 1);
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t a = ga(i), b = gb(i);
            if (skip && (skip[a] | skip[b])) continue;
            const float w = gw(i);
            edges[cur[a]++] = Edge(b, w);
            edges[cur[b]++] = Edge(a, w);
        }
        return true;
    }
    // The same rows from a pair list SORTED by (a << 32 | b) with a < b (what the vote delivers), on several threads and
    // still deterministic: every row ends up sorted by neighbour id — its partners below it, then those above it —,
    // which is also what the serial loop above produces for such a list.  A thread owns a range of ROWS: the partners
    // above a row are one contiguous run of the list, the partners below it are found by scanning the list from
    // `maxgap` rows before the range (b - a is bounded by the vote's band).  No atomics, two scans (count, fill).
    // Returns false if an endpoint is not a key; falls back to add_edges for a list that is not sorted like that.
    template <class GetW>
    bool add_edges_sorted(const uint64_t *key, uint64_t n, GetW gw, const uint8_t *skip = nullptr) {
        auto ga = [&](uint64_t i) { return (uint32_t)(key[i] >> 32); };
        auto gb = [&](uint64_t i) { return (uint32_t)key[i]; };
        if (n < (1u << 18) || host_threads() < 2) return add_edges(n, ga, gb, gw, skip);
        const uint32_t N = n_ids();
        const bool prof = getenv("NP2_PHASE_PROFILE") != nullptr;
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        // pass 0: endpoints are keys, the list is sorted with a < b, largest b - a
        struct Chk {
            bool ok = true, sorted = true;
            uint32_t maxgap = 0;
        };
        std::vector<Chk> chk(host_threads());
        parallel_ranges(n, 1u << 16, [&](unsigned t, size_t lo, size_t hi) {
            Chk c;
            for (size_t i = lo; i < hi; ++i) {
                const uint32_t a = ga(i), b = gb(i);
                if (!has_key(a) || !has_key(b)) c.ok = false;
                if (a >= b || (i && key[i - 1] >= key[i])) c.sorted = false;
                else c.maxgap = std::max(c.maxgap, b - a);
            }
            chk[t] = c;
        });
        uint32_t maxgap = 0;
        bool sorted = true;
        for (const Chk &c : chk) {
            if (!c.ok) return false;
            sorted = sorted && c.sorted;
            maxgap = std::max(maxgap, c.maxgap);
        }
        if (!sorted) return add_edges(n, ga, gb, gw, skip);
        const double t1 = now();
        // row ranges with about the same number of pairs each
        const size_t T = std::min<size_t>(host_threads(), n >> 16);
        std::vector<uint32_t> row_lo(T + 1, 0);
        for (size_t t = 1; t < T; ++t) row_lo[t] = std::max(row_lo[t - 1], ga(n * t / T));
        row_lo[T] = N;
        auto first_pair_of = [&](uint32_t row) { // first pair whose a >= row
            return (size_t)(std::lower_bound(key, key + n, (uint64_t)row << 32) - key);
        };
        off.assign((size_t)N + 1, 0);
        std::vector<uint32_t> n_lo((size_t)N, 0); // partners below each row
        auto scan = [&](size_t t, bool fill) {
            const uint32_t r0 = row_lo[t], r1 = row_lo[t + 1];
            if (r0 >= r1) return;
            const size_t i_own = first_pair_of(r0), i_end = first_pair_of(r1);
            const size_t i_beg = first_pair_of(r0 > maxgap ? r0 - maxgap : 0);
            std::vector<uint32_t> cur_lo, cur_up;
            if (fill) {
                cur_lo.assign(off.begin() + r0, off.begin() + r1);
                cur_up.resize(r1 - r0);
                for (uint32_t v = r0; v < r1; ++v) cur_up[v - r0] = off[v] + n_lo[v];
            }
            for (size_t i = i_beg; i < i_end; ++i) {
                const uint32_t a = ga(i), b = gb(i);
                if (skip && (skip[a] | skip[b])) continue;
                if (b >= r0 && b < r1) { // a partner below row b
                    if (fill) edges[cur_lo[b - r0]++] = Edge(a, gw(i));
                    else ++n_lo[b];
                }
                if (i >= i_own) { // (a in [r0, r1)) a partner above row a
                    if (fill) edges[cur_up[a - r0]++] = Edge(b, gw(i));
                    else ++off[a + 1];
                }
            }
        };
        parallel_each(T, [&](size_t t) { scan(t, false); });
        const double t2 = now();
        for (uint32_t v = 0; v < N; ++v) off[v + 1] += off[v] + n_lo[v];
        edges_resize(off[N]);
        rows_sorted = true;
        const double t3 = now();
        parallel_each(T, [&](size_t t) { scan(t, true); });
        if (prof) fprintf(stderr, "    rows: check %.2f ms, count %.2f ms, offsets + allocation %.2f ms, fill %.2f ms (%zu threads, largest b - a %u)\n", t1 - t0, t2 - t1, t3 - t2, now() - t3, T, maxgap);
        return true;
    }
    // The same rows from the COMPACT pair rows of the vote kernels (k_band_emit_compact): row a of the band = words
    // [row_off[a], row_off[a + 1]), word & 255 = b - a - 1, the rest decoded by gw(word).  The pairs come in the order of
    // the sorted (a, b) list, so the result is the one add_edges_sorted / add_edges give for that list: every row holds
    // its partners below it (ascending), then those above it (ascending).  Threads own row ranges; the partners below a
    // row sit in the rows of the 256 reads before it.
    template <class GetW>
    bool add_edges_rows(const uint32_t *row_off, const uint32_t *cp, uint32_t R, GetW gw, const uint8_t *skip = nullptr) {
        const uint32_t N = n_ids();
        if (R > N) return false;
        const uint64_t n = row_off[R];
        const uint32_t BAND = 256;
        const size_t T = (n < (1u << 18) || host_threads() < 2) ? 1 : std::min<size_t>(host_threads(), n >> 16);
        // endpoints are keys
        std::vector<uint8_t> bad_t(T, 0);
        std::vector<uint32_t> row_lo(T + 1, 0);
        for (size_t t = 1; t < T; ++t)
            row_lo[t] = std::max<uint32_t>(row_lo[t - 1], (uint32_t)(std::upper_bound(row_off, row_off + R + 1, (uint32_t)(n * t / T)) - row_off - 1));
        row_lo[T] = R;
        off.assign((size_t)N + 1, 0);
        std::vector<uint32_t> n_lo((size_t)N, 0); // partners below each row
        std::vector<uint32_t> cur_lo_all, cur_up_all;
        auto scan = [&](size_t t, bool fill) {
            const uint32_t r0 = row_lo[t], r1 = row_lo[t + 1];
            if (r0 >= r1) return;
            const uint32_t rb = r0 > BAND ? r0 - BAND : 0;
            std::vector<uint32_t> cur_lo, cur_up;
            if (fill) {
                cur_lo.assign(off.begin() + r0, off.begin() + r1);
                cur_up.resize(r1 - r0);
                for (uint32_t v = r0; v < r1; ++v) cur_up[v - r0] = off[v] + n_lo[v];
            }
            for (uint32_t a = rb; a < r1; ++a) {
                const uint32_t i0 = row_off[a], i1 = row_off[a + 1];
                if (i0 == i1) continue;
                if (!fill && a >= r0 && !has_key(a)) bad_t[t] = 1;
                const bool sa = skip && skip[a];
                for (uint32_t i = i0; i < i1; ++i) {
                    const uint32_t w = cp[i], b = a + 1 + (w & 255u);
                    if (!fill && a >= r0 && !has_key(b)) bad_t[t] = 1;
                    if (sa || b >= N || (skip && skip[b])) continue;
                    if (b >= r0 && b < r1) { // a partner below row b
                        if (fill) edges[cur_lo[b - r0]++] = Edge(a, gw(w));
                        else ++n_lo[b];
                    }
                    if (a >= r0) { // a partner above row a
                        if (fill) edges[cur_up[a - r0]++] = Edge(b, gw(w));
                        else ++off[a + 1];
                    }
                }
            }
        };
        auto run = [&](bool fill) { parallel_each(T, [&](size_t t) { scan(t, fill); }); };
        run(false);
        for (uint8_t b : bad_t)
            if (b) return false;
        for (uint32_t v = 0; v < N; ++v) off[v + 1] += off[v] + n_lo[v];
        edges_resize(off[N]);
        rows_sorted = true;
        run(true);
        return true;
    }
    // drop the rows of the flagged nodes and every edge pointing at one (row order is preserved)
    void drop_nodes(const uint8_t *bad) {
        const uint32_t N = n_ids();
        size_t w = 0;
        uint32_t row_begin = 0;
        for (uint32_t v = 0; v < N; ++v) {
            const uint32_t rb = off[v], re = off[v + 1];
            off[v] = row_begin;
            if (!bad[v])
                for (uint32_t i = rb; i < re; ++i)
                    if (!bad[edges[i].first]) edges[w++] = edges[i];
            row_begin = (uint32_t)w;
        }
        off[N] = (uint32_t)w;
        edges.resize(w);
    }
    // rows appended in increasing node order (used when the aggregated graph is built)
    void begin_rows(uint32_t n) {
        is_key.assign(n, 0);
        off.assign((size_t)n + 1, 0);
        edges.clear();
        rows_sorted = false;
        next_row_ = 0;
    }
    void append_row(uint32_t a, const std::vector<Edge> &row) { // a >= every earlier a
        for (; next_row_ <= a; ++next_row_) off[next_row_] = (uint32_t)edges.size();
        edges.insert(edges.end(), row.begin(), row.end());
    }
    void end_rows() {
        for (; next_row_ < off.size(); ++next_row_) off[next_row_] = (uint32_t)edges.size();
    }

  private:
    uint32_t next_row_ = 0;
};

struct Community {
    uint32_t id = 0;
    float weight = 0.f;
    std::vector<uint32_t> members; // union of original read ids
};

// Louvain with signed weights (louvain.rs:59-257).  Same decisions as the reference — sorted visit order,
// max-gain / smallest-id moves, negative-community declustering and community ordering follow the emulated
// hashbrown iteration order — but gains and inter-community weights are accumulated edge-wise (O(E)) instead of
// the reference's O(deg^2) / O(C^2) membership scans; all sums are exact small-integer f32, so results are equal.
class SignedLouvain {
  public:
    // communities: the community map of louvain.rs:65-68 built ahead by the caller (first_communities: it depends on the
    // graph's KEYS only, so a chromosome's vote builds it beside the edge rows), or nullptr
    explicit SignedLouvain(Graph g, OrderSet *communities = nullptr) : g_(std::move(g)) {
        const uint32_t n = g_.n_ids();
        node_id_.resize(n);
        node_w_.resize(n);
        cnt_.resize(n);
        if (communities) comm_keys_ = std::move(*communities);
        else // louvain.rs:65-68: a fresh map, the graph's keys inserted in the graph's iteration order
            for (uint32_t v : g_.keys.key_list()) comm_keys_.put(v, Nil{});
        // (members_: every node is {itself} until aggregated; the per-node state by a scan of the dense mirror — on all
        // threads for a chromosome's 6 x 10^5 ids, each touching its own pages first)
        parallel_ranges(n, (size_t)1 << 16, [&](unsigned, size_t lo, size_t hi) {
            for (size_t v = lo; v < hi; ++v) {
                node_id_[v] = (uint32_t)v; // (only read for keys)
                node_w_[v] = 0.f;
                cnt_[v] = g_.is_key[v] ? 1u : 0u;
            }
        });
    }
    // the first level's community map: a fresh map, the graph's keys inserted in the graph's iteration order
    static OrderSet first_communities(const OrderSet &graph_keys) {
        OrderSet c;
        for (uint32_t v : graph_keys.key_list()) c.put(v, Nil{});
        return c;
    }
    // returns false if the reference's weight<0 assertion (louvain.rs:234-237) would fire
    bool run(std::unordered_map<uint32_t, std::unordered_set<uint32_t>> &conflicts, std::vector<Community> &out) {
        const bool prof = getenv("NP2_PHASE_PROFILE") != nullptr;
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        for (;;) {
            double t0 = now();
            const bool moved = local_moving();
            double t1 = now();
            if (prof) fprintf(stderr, "  local_moving %.2f ms (nodes %zu)\n", t1 - t0, g_.keys.size());
            if (!moved) break;
            aggregate();
            if (prof) fprintf(stderr, "  aggregate %.2f ms\n", now() - t1);
        }
        double t2 = now();
        const bool ok = collect(conflicts, out);
        if (prof) fprintf(stderr, "  collect %.2f ms\n", now() - t2);
        return ok;
    }

  private:
    // The reference keeps communities as HashMap<u32, HashSet<u32>>.  Only two things about those containers are
    // observable: the iteration order of the outer map (fixed by the order its keys were inserted in: no key is
    // inserted or removed while nodes move) and, for a community that is declustered, the iteration order of its
    // member set (fixed by the sequence of inserts / removes it saw).  So the outer map is kept as a key-only order
    // set, membership as node_id_ + counters, and every community logs its member operations; a member set is only
    // rebuilt (by replaying the log into the emulated hash set) when a community has to be declustered.
    Graph g_;
    OrderSet comm_keys_;
    std::vector<uint32_t> node_id_; // community of each (possibly aggregated) node
    std::vector<float> node_w_;     // weight carried by an aggregated node
    bool level0_ = true;                        // nodes are still single reads: members of node v = {v}
    // original read ids of an aggregated node (from the first aggregation on); by node id, only the nodes that exist (a
    // vector over all ids was 600 k empty vectors per level for a chromosome's graph)
    std::unordered_map<uint32_t, std::vector<uint32_t>> members_;
    std::vector<uint32_t> cnt_;                 // members per community
    // member operations of this level in order: (community, +(v + 1) insert / -(v + 1) remove); one flat log instead
    // of a vector per community (thousands of small allocations per contig), filtered when a community is replayed
    std::vector<std::pair<uint32_t, int64_t>> oplog_;

    bool local_moving() { // first_stage, louvain.rs:72-117
        // The reference re-evaluates every node in every sweep until a sweep moves nothing.  A node's decision is a
        // pure function of its neighbours' community ids, so a node none of whose neighbours moved since its last
        // evaluation cannot move: only "dirty" nodes are evaluated (same visit order, identical outcome).
        //
        // EXACT DECOMPOSITION OVER COMPONENTS (round 6).  The move rule has no global term: the gain of a node towards a
        // community is the sum of its OWN edges into it (louvain.rs:84-96), a move touches the member sets of the two
        // communities only (:103-105), and a community never holds nodes of two connected components (a node only ever
        // joins the community of a neighbour).  So what a sweep does inside one component depends on nothing outside it:
        // the state of a component after the reference's global sweeps is the state after the same sweeps restricted to
        // its nodes in the same ascending order, a sweep over a component that has converged is a no-op (the reference's
        // outer loop (:78-114) just keeps visiting it), and each community's sequence of member inserts / removes — the one
        // thing whose ORDER is observable later, through the iteration order of a declustered community's hash set — comes
        // from one component.  Components are therefore swept to convergence independently, on all threads.  Reads are
        // numbered in start order and a read votes with its neighbours along the contig, so the cheap exact partition is by
        // CUT POINTS: an id c such that no edge joins an id below c to one at or above it (prefix maximum of the rows'
        // largest neighbours, O(nodes)); a piece between two cuts is a union of whole components.
        //
        // Sweeps after the first one of a large piece move next to nothing (a chromosome's read graph: 272 k nodes evaluated,
        // 60 moved), but every node is dirty, because all of its neighbours moved in the first sweep.  When the graph is
        // ONE piece such a sweep is evaluated AHEAD, on all threads, against the state the sweep starts from; the sweep
        // itself then walks the nodes in order and takes a node's pre-computed decision unless one of its neighbours has
        // moved earlier in this very sweep — then, and for nodes that became dirty during the sweep, the decision is
        // computed on the spot as before.  A pre-computed decision saw exactly the neighbour communities the serial sweep
        // would show it, so it is the same.
        std::vector<uint32_t> visit; // the keys in ascending order (louvain.rs:77: sorted): a scan of the dense mirror
        visit.reserve(g_.keys.size());
        for (uint32_t v = 0; v < g_.n_ids(); ++v)
            if (g_.has_key(v)) visit.push_back(v);
        std::vector<uint8_t> dirty(g_.n_ids(), 1);
        // gains per neighbouring community: a slot per community id, valid for the node whose stamp it carries (in the
        // first sweep every neighbour is a community of its own: a list searched per edge was quadratic in the degree).
        // Shared by the pieces: a piece only touches the slots of its own ids.
        std::vector<float> acc(g_.n_ids(), 0.f);
        std::vector<uint32_t> stamp(g_.n_ids(), 0u);
        static const bool no_ahead = getenv("NP2_VOTE_NO_AHEAD") != nullptr; // (A/B and tests)
        static const size_t ahead_min = getenv("NP2_VOTE_AHEAD_MIN") ? (size_t)atol(getenv("NP2_VOTE_AHEAD_MIN")) : (size_t)1 << 15; // (tests: small graphs too)
        static const bool no_pieces = getenv("NP2_VOTE_NO_PIECES") != nullptr; // (A/B and tests: one piece, as before round 6)
        static const size_t pieces_min = getenv("NP2_VOTE_PIECES_MIN") ? (size_t)atol(getenv("NP2_VOTE_PIECES_MIN")) : (size_t)1 << 13; // (tests: small graphs too)
        const bool prof = getenv("NP2_PHASE_PROFILE") != nullptr;
        std::vector<uint32_t> ahead; // pre-computed decisions of a sweep (NO_GUESS: none), one-piece graphs only
        std::vector<uint32_t> stale; // sweep in which a neighbour of the node last moved (0: never)
        static constexpr uint32_t NO_GUESS = 0xFFFFFFFFu;
        typedef std::vector<std::pair<uint32_t, int64_t>> OpLog;

        // all sweeps over visit[lo, hi) — a union of whole components — until one moves nothing
        auto sweep_piece = [&](size_t lo, size_t hi, OpLog &log, bool whole_graph) -> bool {
            bool moved_any = false;
            std::vector<uint32_t> touched;
            uint32_t tick = 0;
            const uint32_t id_lo = visit[lo], id_hi = visit[hi - 1] + 1; // (the piece's ids: its share of stamp[] / acc[])
            // decision of v against the current state: the community it moves to, or its own
            // (mark_all: the first sweep, in which nearly every node moves, flags the neighbours while it reads them — one
            // walk over the row instead of two; a node flagged without cause is evaluated once more and stays where it is)
            bool mark_all = false;
            auto decide = [&](uint32_t v) -> uint32_t {
                const uint32_t cur = node_id_[v];
                touched.clear();
                if (++tick == 0) { // (the stamps wrapped: start over)
                    std::fill(stamp.begin() + id_lo, stamp.begin() + id_hi, 0u);
                    tick = 1;
                }
                for (const auto &e : g_.adj(v)) {
                    if (mark_all) dirty[e.first] = 1;
                    const uint32_t c = node_id_[e.first];
                    if (stamp[c] != tick) {
                        stamp[c] = tick;
                        acc[c] = e.second;
                        touched.push_back(c);
                    } else {
                        acc[c] += e.second; // (edge order, like the list it replaces)
                    }
                }
                if (touched.empty()) return cur;
                uint32_t bc = touched[0]; // max weight, ties -> smaller community id (louvain.rs:99-101)
                float bw = acc[bc];
                for (size_t i = 1; i < touched.size(); ++i) {
                    const uint32_t c = touched[i];
                    if (acc[c] > bw || (acc[c] == bw && c < bc)) bc = c, bw = acc[c];
                }
                return bw > 0.f ? bc : cur;
            };
            uint32_t sweep = 0;
            size_t last_moved = 0;
            for (bool again = true; again;) {
                again = false;
                ++sweep;
                mark_all = sweep == 1 && level0_;
                size_t n_eval = 0, n_moved = 0, n_taken = 0;
                const auto t_sw = std::chrono::steady_clock::now();
                const bool use_ahead = whole_graph && !no_ahead && sweep > 1 && hi - lo >= ahead_min && last_moved >= ahead_min / 64 && host_threads() > 1; // (many moves in the last sweep: many dirty nodes now)
                if (use_ahead) {
                    ahead.assign(g_.n_ids(), NO_GUESS);
                    if (stale.empty()) stale.assign(g_.n_ids(), 0u);
                    parallel_ranges(visit.size(), 4096, [&](unsigned, size_t plo, size_t phi) {
                        std::vector<std::pair<uint32_t, float>> local; // (a node's neighbours lie in a handful of communities by now)
                        for (size_t i = plo; i < phi; ++i) {
                            const uint32_t v = visit[i];
                            if (!dirty[v]) continue;
                            local.clear();
                            bool give_up = false;
                            for (const auto &e : g_.adj(v)) {
                                const uint32_t c = node_id_[e.first];
                                bool hit = false;
                                for (auto &l : local)
                                    if (l.first == c) {
                                        l.second += e.second; // (exact small-integer sums: the order does not matter)
                                        hit = true;
                                        break;
                                    }
                                if (!hit) {
                                    if (local.size() >= 48) {
                                        give_up = true;
                                        break;
                                    }
                                    local.emplace_back(c, e.second);
                                }
                            }
                            if (give_up) continue;
                            uint32_t to = node_id_[v];
                            if (!local.empty()) {
                                uint32_t bc = local[0].first;
                                float bw = local[0].second;
                                for (size_t k = 1; k < local.size(); ++k)
                                    if (local[k].second > bw || (local[k].second == bw && local[k].first < bc)) bc = local[k].first, bw = local[k].second;
                                if (bw > 0.f) to = bc;
                            }
                            ahead[v] = to;
                        }
                    });
                }
                for (size_t vi = lo; vi < hi; ++vi) {
                    const uint32_t v = visit[vi];
                    if (!dirty[v]) continue;
                    dirty[v] = 0;
                    ++n_eval;
                    const uint32_t cur = node_id_[v];
                    uint32_t to;
                    const bool fresh = use_ahead && ahead[v] != NO_GUESS && stale[v] != sweep; // (stale: a neighbour moved earlier in this sweep)
                    if (fresh) to = ahead[v], ++n_taken;
                    else to = decide(v);
                    if (to != cur) {
                        node_id_[v] = to;
                        ++cnt_[to];
                        --cnt_[cur];
                        log.emplace_back(to, (int64_t)v + 1);
                        log.emplace_back(cur, -((int64_t)v + 1));
                        if (!mark_all)
                            for (const auto &e : g_.adj(v)) dirty[e.first] = 1; // their gains changed
                        if (use_ahead)
                            for (const auto &e : g_.adj(v)) stale[e.first] = sweep;
                        again = true;
                        moved_any = true;
                        ++n_moved;
                    }
                }
                last_moved = n_moved;
                if (prof && whole_graph) fprintf(stderr, "    sweep: %zu evaluated (%zu decided ahead), %zu moved, %.2f ms\n", n_eval, n_taken, n_moved, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sw).count());
            }
            return moved_any;
        };

        if (visit.empty()) return false;
        // cut points: visit index i starts a piece iff no node before it has a neighbour at or beyond visit[i]
        std::vector<size_t> cut; // piece starts (visit indices), ascending; cut.back() = visit.size()
        if (!no_pieces && visit.size() >= pieces_min && host_threads() > 1) {
            const auto t_c = std::chrono::steady_clock::now();
            // largest neighbour id over the nodes before visit[i] (prefix maximum): chunks on all threads, their maxima chained
            auto row_reach = [&](uint32_t v) -> uint32_t {
                const Graph::Row row = g_.adj(v);
                uint32_t m = 0;
                if (g_.rows_sorted) { // (a vote's rows: the largest neighbour is the last one)
                    if (!row.empty()) m = (row.e - 1)->first;
                } else {
                    for (const auto &e : row) m = std::max(m, e.first);
                }
                return m;
            };
            const size_t C = std::max<size_t>(1, std::min<size_t>(host_threads(), visit.size() >> 12));
            std::vector<uint32_t> cmax(C + 1, 0);
            std::vector<std::vector<size_t>> ccut(C);
            std::vector<uint32_t> rmax(visit.size());
            parallel_each(C, [&](size_t t) {
                uint32_t m = 0;
                for (size_t i = visit.size() * t / C; i < visit.size() * (t + 1) / C; ++i) rmax[i] = row_reach(visit[i]), m = std::max(m, rmax[i]);
                cmax[t + 1] = m;
            });
            for (size_t t = 0; t < C; ++t) cmax[t + 1] = std::max(cmax[t + 1], cmax[t]);
            parallel_each(C, [&](size_t t) {
                uint32_t reach = cmax[t];
                for (size_t i = visit.size() * t / C; i < visit.size() * (t + 1) / C; ++i) {
                    if (i == 0 || reach < visit[i]) ccut[t].push_back(i);
                    reach = std::max(reach, rmax[i]);
                }
            });
            for (const auto &cc : ccut) cut.insert(cut.end(), cc.begin(), cc.end());
            cut.push_back(visit.size());
            if (prof) fprintf(stderr, "    pieces: %zu between cut points (%.2f ms)\n", cut.size() - 1, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_c).count());
        }
        if (cut.size() < 3) { // one piece (or a small graph): as before
            if (level0_) oplog_.reserve(2 * visit.size()); // (the first sweep moves nearly every node: two log entries each)
            return sweep_piece(0, visit.size(), oplog_, true);
        }
        // Work units of about n / (8 T) nodes: consecutive pieces joined (a union of pieces is a union of components), taken
        // by the threads from a shared counter — largest first would need a sort; the pieces of a read graph are many and
        // small next to the few long ones, which come at arbitrary places.
        const size_t T = std::min<size_t>(host_threads(), cut.size() - 1);
        const size_t want = std::max<size_t>(256, visit.size() / (8 * T));
        std::vector<std::pair<size_t, size_t>> unit;
        for (size_t k = 0; k + 1 < cut.size();) {
            size_t k1 = k + 1;
            while (k1 + 1 < cut.size() && cut[k1] - cut[k] < want) ++k1;
            unit.emplace_back(cut[k], cut[k1]);
            k = k1;
        }
        std::vector<OpLog> logs(unit.size());
        std::vector<uint8_t> moved(unit.size(), 0);
        std::atomic<size_t> next{0};
        const auto t_p = std::chrono::steady_clock::now();
        parallel_each(std::min(T, unit.size()), [&](size_t) {
            for (size_t u; (u = next.fetch_add(1)) < unit.size();) {
                if (level0_) logs[u].reserve(2 * (unit[u].second - unit[u].first));
                moved[u] = sweep_piece(unit[u].first, unit[u].second, logs[u], false) ? 1 : 0;
            }
        });
        bool moved_any = false;
        size_t n_log = oplog_.size();
        for (const OpLog &l : logs) n_log += l.size();
        oplog_.reserve(n_log);
        for (size_t u = 0; u < unit.size(); ++u) { // (unit order = id order: deterministic whatever the thread count)
            oplog_.insert(oplog_.end(), logs[u].begin(), logs[u].end());
            moved_any = moved_any || moved[u];
        }
        if (prof) fprintf(stderr, "    sweeps: %zu work units on %zu threads, %zu log entries, %.2f ms\n", unit.size(), std::min(T, unit.size()), oplog_.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_p).count());
        return moved_any;
    }

    // member lists of all communities (CSR over the current nodes; order inside a community is not observable)
    void member_lists(std::vector<uint32_t> &off, std::vector<uint32_t> &list) const {
        const size_t n = node_id_.size();
        off.assign(n + 1, 0);
        // (the keys by a scan of the dense mirror, not in the table's order: a walk over a chromosome's table was 3 ms)
        const uint32_t nk = g_.n_ids();
        for (uint32_t v = 0; v < nk; ++v)
            if (g_.has_key(v)) ++off[node_id_[v] + 1];
        for (size_t i = 0; i < n; ++i) off[i + 1] += off[i];
        list.resize(g_.keys.size());
        std::vector<uint32_t> cur(off.begin(), off.end() - 1);
        for (uint32_t v = 0; v < nk; ++v)
            if (g_.has_key(v)) list[cur[node_id_[v]]++] = v;
    }
    // weight of a community = carried weights + half of every directed internal edge (louvain.rs:124-134)
    float internal_weight(uint32_t cid, const uint32_t *mb, const uint32_t *me, std::vector<uint32_t> &mem) const {
        collect_members(mb, me, mem);
        return weight_only(cid, mb, me);
    }
    void collect_members(const uint32_t *mb, const uint32_t *me, std::vector<uint32_t> &mem) const {
        for (const uint32_t *p = mb; p != me; ++p) {
            const uint32_t v = *p;
            if (level0_) mem.push_back(v);
            else {
                const std::vector<uint32_t> &m = members_.at(v);
                mem.insert(mem.end(), m.begin(), m.end());
            }
        }
    }
    float weight_only(uint32_t cid, const uint32_t *mb, const uint32_t *me) const {
        float w = 0.f;
        for (const uint32_t *p = mb; p != me; ++p) {
            const uint32_t v = *p;
            w += node_w_[v];
            for (const auto &e : g_.adj(v))
                if (node_id_[e.first] == cid) w += e.second / 2.0f;
        }
        return w;
    }
    // Weights of all live communities (the O(edges) part of an aggregation).  Large graph: threads take ranges of NODES
    // (after the first sweep a chromosome's graph is a few hundred communities of thousands of reads each) and add into
    // their own per-community partial sums, which are then added up in thread order.  Every term is a multiple of 0.5,
    // so the sums are exact — and equal to the member-order sums of weight_only — as long as they stay below 2^22; a
    // community beyond that is summed again the serial way.
    std::vector<float> all_weights(const std::vector<uint32_t> &moff, const std::vector<uint32_t> &mlist) const {
        const size_t n = node_id_.size();
        std::vector<float> w(n, 0.f);
        if (g_.edges.size() < (1u << 20) || host_threads() < 2) {
            for (uint32_t id = 0; id < n; ++id)
                if (cnt_[id]) w[id] = weight_only(id, mlist.data() + moff[id], mlist.data() + moff[id + 1]);
            return w;
        }
        // partial sums in double: every term is a multiple of 0.5 below 2^22 in magnitude, so a double holds any partial
        // and any total exactly whatever the order (a float partial beyond 2^23 would drop the halves even when the
        // community's total is small again); a total within the float-exact range equals the reference's member-order sum
        // (the live communities numbered densely: after the first sweep of a chromosome's graph they are a few hundred among
        // 6 x 10^5 ids — a partial array over all ids per thread was most of this step)
        std::vector<uint32_t> dense(n, 0xFFFFFFFFu), live;
        for (uint32_t id = 0; id < n; ++id)
            if (cnt_[id]) dense[id] = (uint32_t)live.size(), live.push_back(id);
        const size_t nc = live.size();
        std::vector<std::vector<double>> part(host_threads());
        parallel_ranges(n, 4096, [&](unsigned t, size_t lo, size_t hi) {
            std::vector<double> &p = part[t];
            p.assign(nc, 0.0);
            for (size_t v = lo; v < hi; ++v) {
                if (!g_.has_key((uint32_t)v)) continue;
                const uint32_t cid = node_id_[v];
                double acc = node_w_[v];
                for (const auto &e : g_.adj((uint32_t)v))
                    if (node_id_[e.first] == cid) acc += (double)e.second / 2.0;
                p[dense[cid]] += acc;
            }
        });
        std::vector<double> wd(nc, 0.0);
        for (const auto &p : part)
            if (!p.empty())
                for (size_t i = 0; i < nc; ++i) wd[i] += p[i];
        for (size_t i = 0; i < nc; ++i) {
            const uint32_t id = live[i];
            if (wd[i] > -4194304.0 && wd[i] < 4194304.0) w[id] = (float)wd[i];
            else w[id] = weight_only(id, mlist.data() + moff[id], mlist.data() + moff[id + 1]);
        }
        return w;
    }
    // iteration order of a community's member set: replay its history into the emulated hash set
    std::vector<uint32_t> member_order(uint32_t id) const {
        OrderSet set = OrderSet::single(id, Nil{}); // every community starts as {its own node}
        for (const auto &e : oplog_) {
            if (e.first != id) continue;
            const int64_t op = e.second;
            if (op > 0) set.put((uint32_t)(op - 1), Nil{});
            else set.take((uint32_t)(-op - 1), nullptr);
        }
        return set.key_list();
    }

    void aggregate() { // second_stage, louvain.rs:119-195
        const bool prof = getenv("NP2_PHASE_PROFILE") != nullptr && g_.edges.size() >= (1u << 20);
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t_m = now();
        auto mark = [&](const char *what) {
            if (prof) {
                const double t = now();
                fprintf(stderr, "    aggregate: %s %.2f ms\n", what, t - t_m);
                t_m = t;
            }
        };
        OrderSet ncomm;
        std::unordered_map<uint32_t, Community> nnode;
        std::vector<uint32_t> split;
        std::vector<uint32_t> moff, mlist;
        member_lists(moff, mlist);
        // new community key of every current node (members of split communities get their own key)
        std::vector<uint32_t> key_of(node_id_.size(), 0xFFFFFFFFu);
        mark("member lists");
        const std::vector<float> w_int = all_weights(moff, mlist);
        mark("community weights");
        // the live communities in the community map's iteration order (louvain.rs:123: self.communities.iter() filtered for
        // non-empty sets).  After the first sweep of a chromosome's graph they are a few hundred among 3 x 10^5 keys: found
        // by a scan of the member counts and ordered by their slot in the table, instead of a walk over the whole table
        std::vector<std::pair<size_t, uint32_t>> live;
        for (uint32_t id = 0; id < cnt_.size(); ++id)
            if (cnt_[id]) live.emplace_back(comm_keys_.find(id), id);
        std::sort(live.begin(), live.end());
        for (const auto &sl : live) {
            const uint32_t id = sl.second;
            if (sl.first == OrderSet::npos) continue; // (cannot happen: every community id is a key of the map)
            Community c;
            c.id = id;
            c.weight = w_int[id];
            collect_members(mlist.data() + moff[id], mlist.data() + moff[id + 1], c.members);
            for (uint32_t i = moff[id]; i < moff[id + 1]; ++i) key_of[mlist[i]] = id;
            if (c.weight < 0.f) {
                split.push_back(id);
            } else {
                ncomm.put(id, Nil{});
                nnode[id] = std::move(c);
            }
        }
        mark("community table");
        for (uint32_t id : split) { // decluster negative communities, louvain.rs:145-165
            for (uint32_t v : member_order(id)) {
                uint32_t nid = v;
                while (ncomm.has(nid) || nnode.count(nid)) ++nid;
                ncomm.put(nid, Nil{});
                Community c;
                c.id = nid;
                c.weight = node_w_[v];
                if (level0_) c.members = {v};
                else c.members = members_.at(v);
                nnode[nid] = std::move(c);
                key_of[v] = nid;
            }
        }
        mark("declustering");
        // inter-community weights, accumulated over the directed edges (louvain.rs:167-188 computes the same sums);
        // one pass per new community with a small local accumulator (a community touches few others)
        uint32_t max_id = 0;
        for (const auto &kv : nnode) max_id = std::max(max_id, kv.first);
        Graph ng;
        ng.begin_rows(max_id + 1);
        // nodes per new key.  Without a declustered community the new keys ARE the community ids and the lists the member
        // lists above (every id of moff is a valid row start: max_id < node_id_.size()); otherwise a second counting sort
        std::vector<uint32_t> koff, klist;
        if (split.empty()) {
            koff.swap(moff), klist.swap(mlist);
        } else {
            koff.assign((size_t)max_id + 2, 0);
            for (uint32_t v = 0; v < key_of.size(); ++v)
                if (key_of[v] != 0xFFFFFFFFu) ++koff[key_of[v] + 1];
            for (size_t i = 0; i + 1 < koff.size(); ++i) koff[i + 1] += koff[i];
            klist.resize(koff.back());
            std::vector<uint32_t> cur(koff.begin(), koff.end() - 1);
            for (uint32_t v = 0; v < key_of.size(); ++v)
                if (key_of[v] != 0xFFFFFFFFu) klist[cur[key_of[v]]++] = v;
        }
        // (the rows of a range of new ids are independent of every other range: several threads on a large graph, each
        // keeping its rows in id order; they are appended in id order afterwards)
        mark("key lists");
        struct Rows {
            std::vector<Graph::Edge> edges;
            std::vector<std::pair<uint32_t, uint32_t>> row; // (id, end offset in edges)
        };
        std::vector<Rows> part(host_threads());
        parallel_ranges((size_t)max_id + 1, g_.edges.size() >= (1u << 20) ? 4096 : (size_t)max_id + 2,
                        [&](unsigned t, size_t lo, size_t hi) {
            Rows &out = part[t];
            std::vector<std::pair<uint32_t, float>> local;
            for (uint32_t a = (uint32_t)lo; a < (uint32_t)hi; ++a) {
                if (koff[a] == koff[a + 1]) continue;
                local.clear();
                for (uint32_t i = koff[a]; i < koff[a + 1]; ++i)
                    for (const auto &e : g_.adj(klist[i])) {
                        const uint32_t b = key_of[e.first];
                        if (b == a) continue;
                        bool hit = false;
                        for (auto &l : local)
                            if (l.first == b) {
                                l.second += e.second;
                                hit = true;
                                break;
                            }
                        if (!hit) local.emplace_back(b, e.second);
                    }
                // a node enters the aggregated graph with its first non-zero weight (rows hold only those)
                local.erase(std::remove_if(local.begin(), local.end(), [](const Graph::Edge &l) { return l.second == 0.f; }),
                            local.end());
                if (!local.empty()) {
                    out.edges.insert(out.edges.end(), local.begin(), local.end());
                    out.row.emplace_back(a, (uint32_t)out.edges.size());
                }
            }
        });
        mark("inter-community rows");
        {
            std::vector<Graph::Edge> row;
            for (const Rows &r : part) { // (thread t took the t-th range of ids: already in id order)
                uint32_t b0 = 0;
                for (const auto &x : r.row) {
                    row.assign(r.edges.begin() + b0, r.edges.begin() + x.second);
                    ng.add_key(x.first);
                    ng.append_row(x.first, row);
                    b0 = x.second;
                }
            }
        }
        ng.end_rows();
        mark("new graph");
        if (g_.edges.size() >= (1u << 22)) Graph::retire(std::move(g_.edges)); // (kept for the next large vote: see Graph::Spare)
        g_ = std::move(ng);
        comm_keys_ = std::move(ncomm);
        node_id_.assign(max_id + 1, 0);
        node_w_.assign(max_id + 1, 0.f);
        members_.clear();
        members_.reserve(nnode.size());
        level0_ = false;
        cnt_.assign(max_id + 1, 0);
        oplog_.clear();
        for (auto &kv : nnode) {
            node_id_[kv.first] = kv.first;
            node_w_[kv.first] = kv.second.weight;
            members_[kv.first] = std::move(kv.second.members);
            cnt_[kv.first] = 1;
        }
        mark("state of the next level");
    }

    bool collect(std::unordered_map<uint32_t, std::unordered_set<uint32_t>> &conflicts,
                 std::vector<Community> &out) { // get_communities, louvain.rs:197-245
        out.clear();
        std::vector<uint32_t> moff, mlist;
        member_lists(moff, mlist);
        comm_keys_.each([&](uint32_t id, const Nil &) {
            if (cnt_[id] == 0) return;
            Community c;
            c.id = id;
            c.weight = internal_weight(id, mlist.data() + moff[id], mlist.data() + moff[id + 1], c.members);
            out.push_back(std::move(c));
        });
        // conflicts between final communities: sum of the weights between their members (must be < 0)
        std::vector<std::vector<uint32_t>> of_comm;
        for (uint32_t v = 0; v < g_.n_ids(); ++v) {
            if (g_.adj(v).empty()) continue;
            const uint32_t a = node_id_[v];
            if (a >= of_comm.size()) of_comm.resize(a + 1);
            of_comm[a].push_back(v);
        }
        bool ok = true;
        std::vector<std::pair<uint32_t, float>> local;
        for (uint32_t a = 0; a < of_comm.size(); ++a) {
            if (of_comm[a].empty()) continue;
            local.clear();
            for (uint32_t v : of_comm[a])
                for (const auto &e : g_.adj(v)) {
                    const uint32_t b = node_id_[e.first];
                    if (b <= a) continue;
                    bool hit = false;
                    for (auto &l : local)
                        if (l.first == b) {
                            l.second += e.second;
                            hit = true;
                            break;
                        }
                    if (!hit) local.emplace_back(b, e.second);
                }
            for (const auto &l : local) {
                if (l.second == 0.f) continue;
                if (!(l.second < 0.f)) ok = false;
                conflicts[a].insert(l.first);
                conflicts[l.first].insert(a);
            }
        }
        return ok;
    }
};

// phase_communities (louvain.rs:290-356): reads of the losing communities.
// ref_w / ref_seen: the reference haplotype's row (ref_data[0]), indexed by read id; have_ref = row exists
inline bool losing_reads(Graph graph, bool have_ref, const std::vector<float> &ref_w,
                         const std::vector<uint8_t> &ref_seen, std::vector<uint32_t> &losers, OrderSet *communities = nullptr) {
    const bool prof = getenv("NP2_PHASE_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_m = now();
    auto mark = [&](const char *what) {
        if (prof) {
            const double t = now();
            fprintf(stderr, "  losing_reads: %s %.2f ms\n", what, t - t_m);
            t_m = t;
        }
    };
    SignedLouvain lv(std::move(graph), communities);
    mark("first-level state");
    std::unordered_map<uint32_t, std::unordered_set<uint32_t>> conflicts;
    std::vector<Community> comms;
    if (!lv.run(conflicts, comms)) return false;
    mark("levels");
    if (have_ref) {
        std::vector<std::pair<int32_t, float>> key(comms.size());
        for (size_t i = 0; i < comms.size(); ++i) {
            int32_t cnt = 0;
            float w = 0.f;
            for (uint32_t m : comms[i].members) {
                if (m >= ref_seen.size() || !ref_seen[m]) continue;
                cnt += (ref_w[m] > 0.f) - (ref_w[m] < 0.f);
                w += ref_w[m];
            }
            key[i] = {cnt, w};
        }
        std::vector<size_t> ord(comms.size());
        for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
        std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return key[a] > key[b]; });
        std::vector<Community> tmp;
        tmp.reserve(comms.size());
        for (size_t i : ord) tmp.push_back(std::move(comms[i]));
        comms.swap(tmp);
    } else {
        std::stable_sort(comms.begin(), comms.end(),
                         [](const Community &a, const Community &b) { return a.weight > b.weight; });
    }
    std::unordered_set<uint32_t> lost;
    for (size_t p = 0; p < comms.size(); ++p) {
        if (lost.count(comms[p].id)) continue;
        auto it = conflicts.find(comms[p].id);
        if (it == conflicts.end()) continue;
        for (size_t q = p + 1; q < comms.size(); ++q)
            if (!lost.count(comms[q].id) && it->second.count(comms[q].id)) lost.insert(comms[q].id);
    }
    for (const Community &c : comms)
        if (lost.count(c.id)) losers.insert(losers.end(), c.members.begin(), c.members.end());
    mark("ranking");
    return true;
}

} // namespace phase
} // namespace np2
