// TEST INFRASTRUCTURE — CPU oracle for the NextPolish2 consensus hot path.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
// It is a literal, structure-preserving restatement (same iteration orders, tie-breaks,
// integer widths and wrapping) of Nextomics/NextPolish2 v0.2.2:
//     src/main.rs:30-1687, src/utils/kmer.rs:8-22,61-170,223-314, src/utils/louvain.rs:12-356
// taking the boundary format of include/np2.h (packed AlignSeq nibbles + yak words).
//
// PARITY UNPINNED: the reference ships no tests/golden vectors for this path
//  and cannot be built here (no Rust toolchain, no htslib); the only
// reference-derived known answer is the pass-through contig of test/hh.sh.  Everything
// else is pinned by hand-verified fixtures under tests/golden/.
//
// Third-party arithmetic restated from crate knowledge (sources absent from
// /root/reference, versions from Cargo.toml:8-24 semver ranges): itertools 0.10.5
// multi_cartesian_product order (last iterator fastest), fxhash 0.2.1 + hashbrown
// iteration order (hashbrown_emul.hpp), ordered-float 3.4 NotNan ordering.
#include "../include/np2.h"
#include "hashbrown_emul.hpp"

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace np2o {

struct Unsupported : std::runtime_error { // an input product and oracle both refuse (NP2_E_UNSUPPORTED)
    using std::runtime_error::runtime_error;
};
struct RefPanic : std::runtime_error {
    using std::runtime_error::runtime_error;
};
#define RPANIC_IF(cond, msg)                                                                       \
    do {                                                                                           \
        if (cond) throw RefPanic(msg);                                                             \
    } while (0)

// kmer.rs:11-22
static const uint8_t SEQ_NUM[128] = {
    65, 67, 71, 84, 45, 78, 77, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4,  4,  4,  4,  4,  4,  4,  4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4,  4,  4,  4,  4,  4,  4,  4, 4, 4, 4, 4, 4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 6,
    5,  4,  4,  4,  4,  4,  3,  3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 0, 4, 1, 4, 4, 4, 2,
    4,  4,  4,  4,  4,  6,  5,  4, 4, 4, 4, 4, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};

static const size_t LQSEQ_MAX_CAN_COUNT = 60; // main.rs:30
static const uint64_t INVALID_KMER = ~0ULL;    // main.rs:31

// main.rs:33-52
struct AlignBase {
    uint8_t q_base = 0;
    uint16_t delta = 0;
    uint32_t t_pos = 0;
    static AlignBase head(uint32_t t_pos, uint16_t delta) {
        AlignBase a;
        a.q_base = 0b1111;
        a.delta = delta;
        a.t_pos = t_pos;
        return a;
    }
    bool is_head() const { return q_base == 0b1111; }
    bool operator==(const AlignBase &o) const {
        return q_base == o.q_base && delta == o.delta && t_pos == o.t_pos;
    }
};

// main.rs:54-185
struct Kmer {
    uint16_t delta = 0;
    uint16_t bases = 0;
    uint32_t count = 0;
    uint32_t besti = 0;
    int64_t score = 0;

    static Kmer make(const AlignBase &b1, const AlignBase &b2, const AlignBase &b3) {
        uint16_t f = 0;
        if (b2.t_pos == b1.t_pos) f |= 0b0100;
        if (b2.t_pos == b3.t_pos) f |= 0b0001;
        Kmer k;
        k.delta = b1.delta;
        k.bases = (uint16_t)(((((unsigned)f << 4 | b1.q_base) << 4 | b2.q_base) << 4) | b3.q_base);
        k.count = 1;
        return k;
    }
    // p is the position of 3-base (main.rs:105-184); u32/u16 arithmetic wraps (release build)
    void get_bases(uint32_t p, AlignBase &a, AlignBase &b, AlignBase &c) const {
        a.q_base = (bases >> 8) & 0xF;
        b.q_base = (bases >> 4) & 0xF;
        c.q_base = bases & 0xF;
        if ((bases & 0x5000) == 0x5000) { // A--
            a.t_pos = p, a.delta = delta;
            b.t_pos = p, b.delta = (uint16_t)(delta + 1);
            c.t_pos = p, c.delta = (uint16_t)(delta + 2);
        } else if (bases & 0x1000) { // AA-
            a.t_pos = p - 1, a.delta = delta;
            b.t_pos = p, b.delta = 0;
            c.t_pos = p, c.delta = 1;
        } else if (bases & 0x4000) { // A-A
            a.t_pos = p - 1, a.delta = delta;
            b.t_pos = p - 1, b.delta = (uint16_t)(delta + 1);
            c.t_pos = p, c.delta = 0;
        } else { // AAA
            a.t_pos = p - 2, a.delta = delta;
            b.t_pos = p - 1, b.delta = 0;
            c.t_pos = p, c.delta = 0;
        }
    }
    uint16_t delta3() const {
        AlignBase a, b, c;
        get_bases(0, a, b, c);
        return c.delta;
    }
};

// main.rs:187-250
struct Msa {
    std::vector<Kmer> kmers;
    void push(const Kmer &v) {
        for (auto &k : kmers)
            if (k.bases == v.bases && k.delta == v.delta) {
                k.count += 1;
                RPANIC_IF(k.count == 0xFFFFFFFFu, "kmer count overflow!");
                return;
            }
        kmers.push_back(v);
    }
    void sort() { // sort_by_cached_key: stable
        std::stable_sort(kmers.begin(), kmers.end(),
                         [](const Kmer &x, const Kmer &y) { return x.delta3() < y.delta3(); });
    }
    int64_t coverage() const {
        int64_t c = 0;
        for (auto &k : kmers) {
            if (k.delta3() != 0) break;
            c += (int64_t)k.count;
        }
        return c;
    }
};

// main.rs:272-351 — a view over the boundary buffer
struct AlignSeq {
    uint32_t aln_t_s = 0, aln_t_e = 0;
    const uint8_t *align_bases = nullptr; // nullptr <=> Vec::new() (empty)
    bool empty() const { return align_bases == nullptr; }
    bool get_align_tag(size_t &p, AlignBase &ab) const { // main.rs:314-338
        uint8_t t = align_bases[p >> 1];
        if ((p & 1) == 0) t >>= 4;
        if ((t & 15) == 15) return false;
        ab.q_base = t & 7;
        if (p != 0) {
            if (t & 8) {
                ab.delta = (uint16_t)(ab.delta + 1);
            } else {
                ab.delta = 0;
                ab.t_pos += 1;
            }
        } else {
            ab.t_pos = aln_t_s;
            ab.delta = 0;
        }
        p += 1;
        return true;
    }
};

struct ConsensusBase {
    uint32_t pos;
    char base;
};

// main.rs:647-727
struct LqSeq {
    uint32_t order = 0;
    uint16_t kscore = 0;
    uint64_t kmer = 0;
    std::string seq;
};
static const uint8_t LABLE_TEMP = 0x01, LABLE_SUCC = 0x80, LABLE_HETE = 0x40, LABLE_RECH = 0x20;
struct LqSeqs {
    uint8_t lable = 0;
    uint32_t start = 0, end = 0;
    std::string sudoseed;
    std::vector<LqSeq> seqs;
    void set_lable(uint8_t l) { lable |= l; }
    void unset_lable(uint8_t l) { lable ^= l; }
    bool has_lable(uint8_t l) const { return (lable & l) != 0; }
};

// ---------------------------------------------------------------------------------------
// yak table (kmer.rs:61-221).  The reference keeps a candidate set and re-streams the file
// (kmer.rs:132-170); the observable result of insert(..,true)+retrieve_kmers+get is:
//   get(x) = count of the LAST file word in bucket (x & pmask) whose (w >> 10) == (x >> 10)
//            and whose count >= min_count; 0 otherwise.
// We hold the file words in memory, bucket by bucket, sorted by key for binary search.
// ---------------------------------------------------------------------------------------
static inline uint64_t yak_hash64(uint64_t key, uint64_t mask) { // kmer.rs:223-233
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

struct KmerInfo {
    uint32_t ksize = 0, pre = 0;
    uint64_t kmask = 0, pmask = 0;
    // per bucket: (key = w>>10, index in file order) sorted by key then index; counts[]
    typedef std::vector<std::vector<std::pair<uint64_t, uint16_t>>> Buckets;
    std::shared_ptr<Buckets> sets_p = std::make_shared<Buckets>(); // (w>>10, count) file order kept
    uint16_t built_min = 0xFFFF;
    // filtered by built_min, last wins.  Shared (read-only) by the clones of a context (np2o_ctx_clone): the CPU
    // baseline runs one oracle per host thread on ONE copy of the tables, like the reference's workers read one file
    std::shared_ptr<Buckets> sorted_p = std::make_shared<Buckets>();

    void load(const np2_yak_t &y) {
        ksize = y.k;
        pre = y.pre;
        kmask = (1ULL << (2 * (uint64_t)ksize)) - 1;
        pmask = (1ULL << pre) - 1;
        size_t nb = (size_t)1 << pre;
        Buckets &sets = *sets_p;
        sets.assign(nb, {});
        over_buckets(nb, y.n_words, [&](size_t b) {
            sets[b].reserve(y.bucket_off[b + 1] - y.bucket_off[b]);
            for (uint64_t i = y.bucket_off[b]; i < y.bucket_off[b + 1]; ++i)
                sets[b].emplace_back(y.words[i] >> 10, (uint16_t)(y.words[i] & 1023));
        });
    }
    // The file buckets are independent: a table of human size (10^8 words and more: the chromosome-scale parity tests) is
    // loaded and filtered on several host threads; the result does not depend on the thread count.
    template <class F> static void over_buckets(size_t nb, uint64_t n_words, F f) {
        unsigned nt = n_words < (1u << 24) ? 1 : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (nt == 1) {
            for (size_t b = 0; b < nb; ++b) f(b);
            return;
        }
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&]() {
                for (size_t b; (b = next.fetch_add(1)) < nb;) f(b);
            });
        for (auto &x : th) x.join();
    }
    // Variant (i) of the CPU baseline: the reference keeps no table in memory, every scoring phase
    // re-reads the whole .yak dump and probes its candidate set once per file word (retrieve_kmers, kmer.rs:132-170).
    // With a dump path set, stream_pass() does exactly that work on the real file: BufReader-sized reads, the 8-byte
    // word loop, the min_count test, one hash-set probe per surviving word, the histogram.  The counts themselves still
    // come from the in-memory table (they are the same numbers), so results do not depend on the mode.
    std::string path;
    uint64_t stream_pass(const std::vector<std::unordered_set<uint64_t>> &want, uint16_t min_count) const {
        if (path.empty()) return 0;
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) throw RefPanic("cannot open the yak dump for streaming");
        std::vector<uint8_t> buf(8192); // BufReader's default capacity
        setvbuf(f, nullptr, _IONBF, 0);
        size_t have = 0, at = 0;
        auto need = [&](size_t n) -> const uint8_t * { // read_exact over a refilled 8 KiB buffer
            if (have - at < n) {
                memmove(buf.data(), buf.data() + at, have - at);
                have -= at;
                at = 0;
                const size_t got = fread(buf.data() + have, 1, buf.size() - have, f);
                have += got;
                if (have < n) return nullptr;
            }
            const uint8_t *p = buf.data() + at;
            at += n;
            return p;
        };
        uint64_t hits = 0;
        std::vector<uint32_t> hist(1024, 0);
        if (need(16)) {
            for (size_t b = 0; b < ((size_t)1 << pre); ++b) {
                const uint8_t *h = need(8);
                if (!h) break;
                uint32_t size;
                memcpy(&size, h + 4, 4);
                const auto &set = want[b];
                for (uint32_t j = 0; j < size; ++j) {
                    const uint8_t *w = need(8);
                    if (!w) break;
                    uint64_t hash;
                    memcpy(&hash, w, 8);
                    const uint16_t count = (uint16_t)(hash & 1023);
                    hist[count] += 1;
                    if (count < min_count) continue;
                    if (set.count(hash >> 10)) ++hits;
                }
            }
        }
        fclose(f);
        return hits + hist[0];
    }
    uint64_t to_hash(uint64_t kmer) const { // kmer.rs:102-110
        return ksize < 32 ? yak_hash64(kmer, kmask) : kmer;
    }
    void prepare(uint16_t min_count) { // models retrieve_kmers(min_count) (kmer.rs:132-170)
        if (built_min == min_count) return;
        const Buckets &sets = *sets_p;
        sorted_p = std::make_shared<Buckets>(); // (a clone keeps the previous filter of its parent untouched)
        Buckets &sorted = *sorted_p;
        sorted.assign(sets.size(), {});
        uint64_t n_words = 0;
        for (auto &v : sets) n_words += v.size();
        over_buckets(sets.size(), n_words, [&](size_t b) {
            auto &s = sorted[b];
            for (auto &e : sets[b])
                if (e.second >= min_count) s.push_back(e);
            std::stable_sort(s.begin(), s.end(),
                             [](const std::pair<uint64_t, uint16_t> &x,
                                const std::pair<uint64_t, uint16_t> &y) { return x.first < y.first; });
            // last file word with the same key wins (HashSet::replace, kmer.rs:164-166)
            size_t w = 0;
            for (size_t i = 0; i < s.size(); ++i) {
                if (i + 1 < s.size() && s[i + 1].first == s[i].first) continue;
                s[w++] = s[i];
            }
            s.resize(w);
        });
        built_min = min_count;
    }
    uint16_t get_or0(uint64_t hash) const { // kmer.rs:123-125 + unwrap_or(0)
        const auto &s = (*sorted_p)[hash & pmask];
        uint64_t key = hash >> 10;
        auto it = std::lower_bound(
            s.begin(), s.end(), key,
            [](const std::pair<uint64_t, uint16_t> &e, uint64_t k) { return e.first < k; });
        if (it != s.end() && it->first == key) return it->second;
        return 0;
    }
};

// kmer.rs:255-287 (ksize < 32 path): canonical k-mers of a byte string, in order
template <class F> static void iter2kmer(const std::string &s, size_t ksize, F emit) {
    size_t l = 0;
    uint64_t shift = 2 * ((uint64_t)ksize - 1);
    uint64_t mask = (1ULL << (2 * (uint64_t)ksize)) - 1;
    uint64_t kmer[2] = {0, 0};
    for (unsigned char ch : s) {
        uint64_t c = SEQ_NUM[ch & 127];
        if (c < 4) {
            kmer[0] = (kmer[0] << 2 | c) & mask;
            kmer[1] = (kmer[1] >> 2) | (3 ^ c) << shift;
            l += 1;
        } else {
            l = 0;
        }
        if (l >= ksize) emit(kmer[0] < kmer[1] ? kmer[0] : kmer[1]);
    }
}

// min count over the k-mers of s; 0 if none  (main.rs:761-769, 1300-1315, 1342-1350)
// set while the candidate k-mers of a scoring phase are being collected for KmerInfo::stream_pass (variant (i) of the
// CPU baseline): min_kmer_count_of then records the hashes instead of looking them up
static thread_local std::vector<std::unordered_set<uint64_t>> *tl_collect = nullptr;

static uint16_t min_kmer_count_of(const std::string &s, const KmerInfo &ki) {
    if (tl_collect) {
        iter2kmer(s, ki.ksize, [&](uint64_t km) {
            const uint64_t h = ki.to_hash(km);
            (*tl_collect)[h & ki.pmask].insert(h >> 10);
        });
        return 0;
    }
    bool any = false;
    uint16_t mn = 0xFFFF;
    iter2kmer(s, ki.ksize, [&](uint64_t km) {
        uint16_t c = ki.get_or0(ki.to_hash(km));
        any = true;
        if (c < mn) mn = c;
    });
    return any ? mn : 0;
}

// ---------------------------------------------------------------------------------------
// Louvain (louvain.rs:12-356) on the hashbrown-order emulation
// ---------------------------------------------------------------------------------------
typedef std::unordered_map<uint32_t, float> Inner; // iteration only feeds exact f32 sums
typedef hb::FxMap<Inner> Data;

static void insert_data(Data &d, uint32_t k1, uint32_t k2, float v) { // louvain.rs:273-279
    Inner *in = d.get(k1);
    if (in) {
        auto it = in->find(k2);
        if (it != in->end())
            it->second += v;
        else
            (*in)[k2] = v;
    } else {
        Inner n;
        n[k2] = v;
        d.entry_insert_vacant(k1, std::move(n));
    }
}
static void assign_data(Data &d, uint32_t k1, uint32_t k2, float v) { // louvain.rs:282-288
    Inner *in = d.get(k1);
    if (in) {
        (*in)[k2] = v;
    } else {
        Inner n;
        n[k2] = v;
        d.entry_insert_vacant(k1, std::move(n));
    }
}

struct Node {
    uint32_t id = 0;
    float weight = 0.f;
    std::vector<uint32_t> nodes; // HashSet<u32>; only membership/union is observable
};
static void nodes_extend(std::vector<uint32_t> &dst, const std::vector<uint32_t> &src) {
    for (uint32_t x : src)
        if (std::find(dst.begin(), dst.end(), x) == dst.end()) dst.push_back(x);
}

struct Louvain {
    Data data;
    hb::FxMap<hb::FxSet> communities;
    std::unordered_map<uint32_t, Node> node;

    explicit Louvain(Data d) : data(std::move(d)) { // louvain.rs:60-70
        for (uint32_t vid : data.keys()) {
            communities.insert(vid, hb::FxSet::from_one(vid, hb::Unit{}));
            Node n;
            n.id = vid;
            n.weight = 0.f;
            n.nodes = {vid};
            node[vid] = n;
        }
    }

    bool first_stage() { // louvain.rs:72-117
        bool mod_inc = false;
        std::vector<uint32_t> visit_ids = data.keys();
        std::sort(visit_ids.begin(), visit_ids.end());
        std::vector<std::pair<uint32_t, float>> node_ids;
        for (;;) {
            bool can_stop = true;
            for (uint32_t v_id : visit_ids) {
                uint32_t v_nid = node[v_id].id;
                node_ids.clear();
                const Inner &row = *data.get(v_id);
                for (auto &kv : row) {
                    uint32_t w_nid = node[kv.first].id;
                    bool seen = false;
                    for (auto &e : node_ids)
                        if (e.first == w_nid) {
                            seen = true;
                            break;
                        }
                    if (seen) continue;
                    const hb::FxSet &comm = *communities.get(w_nid);
                    float s = 0.f;
                    for (auto &kv2 : row)
                        if (comm.contains(kv2.first)) s += kv2.second;
                    node_ids.emplace_back(w_nid, s);
                }
                // max_by(weight, then smaller id wins)
                bool have = false;
                uint32_t best_id = 0;
                float best_w = 0.f;
                for (auto &e : node_ids) {
                    if (!have || e.second > best_w || (e.second == best_w && e.first < best_id)) {
                        have = true;
                        best_id = e.first;
                        best_w = e.second;
                    }
                }
                if (have && best_w > 0.0f && best_id != v_nid) {
                    node[v_id].id = best_id;
                    communities.get(best_id)->insert(v_id, hb::Unit{});
                    communities.get(v_nid)->remove(v_id);
                    can_stop = false;
                    mod_inc = true;
                }
            }
            if (can_stop) break;
        }
        return mod_inc;
    }

    void second_stage() { // louvain.rs:119-195
        std::unordered_map<uint32_t, Node> nnode;
        hb::FxMap<hb::FxSet> ncomm;
        std::vector<uint32_t> decluster_ids;
        communities.for_each([&](uint32_t id, const hb::FxSet &nodes) {
            if (nodes.is_empty()) return;
            Node nn;
            nn.id = id;
            nn.weight = 0.f;
            nodes.for_each([&](uint32_t nid, const hb::Unit &) {
                const Node &vertex = node[nid];
                nodes_extend(nn.nodes, vertex.nodes);
                nn.weight += vertex.weight;
                const Inner *row = data.get(nid);
                if (row)
                    for (auto &kv : *row)
                        if (nodes.contains(kv.first)) nn.weight += kv.second / 2.0f;
            });
            if (nn.weight < 0.f) {
                decluster_ids.push_back(id);
            } else {
                ncomm.insert(id, hb::FxSet::from_one(id, hb::Unit{}));
                nnode[id] = nn;
            }
        });
        for (uint32_t id : decluster_ids) {
            hb::FxSet nodes;
            communities.remove(id, &nodes);
            for (uint32_t nid : nodes.keys()) {
                uint32_t new_nid = nid;
                while (ncomm.contains(new_nid) || nnode.count(new_nid)) new_nid += 1;
                ncomm.insert(new_nid, hb::FxSet::from_one(new_nid, hb::Unit{}));
                Node nn;
                nn.id = new_nid;
                nn.weight = node[nid].weight;
                nn.nodes = node[nid].nodes;
                nnode[new_nid] = nn;
                communities.insert(new_nid, hb::FxSet::from_one(nid, hb::Unit{}));
            }
        }
        Data ndata;
        communities.for_each([&](uint32_t nid1, const hb::FxSet &nodes1) {
            if (nodes1.is_empty()) return;
            communities.for_each([&](uint32_t nid2, const hb::FxSet &nodes2) {
                if (!(nid2 > nid1) || nodes2.is_empty()) return;
                float ew = 0.f;
                nodes1.for_each([&](uint32_t vid, const hb::Unit &) {
                    const Inner *row = data.get(vid);
                    if (row)
                        for (auto &kv : *row)
                            if (nodes2.contains(kv.first)) ew += kv.second;
                });
                if (ew != 0.f) {
                    insert_data(ndata, nid1, nid2, ew);
                    insert_data(ndata, nid2, nid1, ew);
                }
            });
        });
        data = std::move(ndata);
        communities = std::move(ncomm);
        node = std::move(nnode);
    }

    void get_communities(Data &odata, std::vector<Node> &out) { // louvain.rs:197-245
        out.clear();
        communities.for_each([&](uint32_t id, const hb::FxSet &nodes) {
            if (nodes.is_empty()) return;
            Node c;
            c.id = id;
            float weight = 0.f;
            nodes.for_each([&](uint32_t vid, const hb::Unit &) {
                const Node &v = node[vid];
                nodes_extend(c.nodes, v.nodes);
                weight += v.weight;
                const Inner *ks = data.get(vid);
                if (ks)
                    for (auto &kv : *ks)
                        if (nodes.contains(kv.first)) weight += kv.second / 2.0f;
            });
            c.weight = weight;
            out.push_back(c);
        });
        for (auto &c1 : out)
            for (auto &c2 : out) {
                if (!(c2.id > c1.id)) continue;
                float weight = 0.f;
                communities.get(c1.id)->for_each([&](uint32_t n1, const hb::Unit &) {
                    communities.get(c2.id)->for_each([&](uint32_t n2, const hb::Unit &) {
                        const Inner *row = data.get(n1);
                        if (row) {
                            auto it = row->find(n2);
                            if (it != row->end()) weight += it->second;
                        }
                    });
                });
                if (weight != 0.f) {
                    RPANIC_IF(!(weight < 0.f),
                              "the weight of two conflicting community is not less than 0");
                    insert_data(odata, c1.id, c2.id, weight);
                    insert_data(odata, c2.id, c1.id, weight);
                }
            }
    }

    void execute(Data &odata, std::vector<Node> &out) { // louvain.rs:247-256
        for (;;) {
            if (first_stage())
                second_stage();
            else {
                get_communities(odata, out);
                return;
            }
        }
    }
};

// louvain.rs:290-356
static std::vector<uint32_t> phase_communities(Data data, const Inner *ref_weight) {
    Louvain lv(std::move(data));
    Data cdata;
    std::vector<Node> communities;
    lv.execute(cdata, communities);

    if (ref_weight) {
        struct Key {
            int32_t count;
            float weight;
        };
        std::vector<Key> keys(communities.size());
        for (size_t i = 0; i < communities.size(); ++i) {
            int32_t count = 0;
            float weight = 0.f;
            for (uint32_t n : communities[i].nodes) {
                auto it = ref_weight->find(n);
                if (it != ref_weight->end()) {
                    if (it->second > 0.f)
                        count += 1;
                    else if (it->second < 0.f)
                        count -= 1;
                    weight += it->second;
                }
            }
            keys[i] = {count, weight};
        }
        std::vector<size_t> idx(communities.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
        // sort_by_cached_key(Reverse((count, weight))): stable, descending
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
            if (keys[a].count != keys[b].count) return keys[a].count > keys[b].count;
            return keys[a].weight > keys[b].weight;
        });
        std::vector<Node> sorted;
        sorted.reserve(idx.size());
        for (size_t i : idx) sorted.push_back(communities[i]);
        communities.swap(sorted);
    } else {
        std::stable_sort(communities.begin(), communities.end(),
                         [](const Node &a, const Node &b) { return a.weight > b.weight; });
    }

    std::unordered_set<uint32_t> invalid_ids;
    for (size_t p = 0; p < communities.size(); ++p) {
        if (invalid_ids.count(communities[p].id)) continue;
        const Inner *id_vs = cdata.get(communities[p].id);
        if (id_vs) {
            for (size_t q = p + 1; q < communities.size(); ++q) {
                if (invalid_ids.count(communities[q].id)) continue;
                if (id_vs->count(communities[q].id)) invalid_ids.insert(communities[q].id);
            }
        }
    }
    std::vector<uint32_t> invalid_nodes;
    for (auto &c : communities)
        if (invalid_ids.count(c.id))
            invalid_nodes.insert(invalid_nodes.end(), c.nodes.begin(), c.nodes.end());
    return invalid_nodes;
}

// ---------------------------------------------------------------------------------------
// Trace: named intermediate arrays per pass, for stage-level parity tests
// ---------------------------------------------------------------------------------------
struct Trace {
    bool on = false;
    std::map<std::string, std::vector<uint8_t>> items;
    template <class T> void put(int pass, const char *name, const std::vector<T> &v) {
        if (!on) return;
        std::string key = std::to_string(pass) + ":" + name;
        auto &dst = items[key];
        dst.resize(v.size() * sizeof(T));
        if (!v.empty()) memcpy(dst.data(), v.data(), dst.size());
    }
};

struct Opt {
    std::vector<KmerInfo> yak;
    uint16_t min_kmer_count = 5;
    long max_indel_len = 20;
    size_t iter_count = 2;
    bool model_ref = true;
    bool use_all_reads = false;
};

struct Stats {
    uint64_t kmer_probes = 0, n_regions = 0, n_candidates = 0, n_nodes = 0, n_invalid = 0;
};

struct Ctx {
    Opt opt;
    Trace trace;
    Stats stats;
    std::string err;
    int pass = 0;
};

// main.rs:740-778
static void retrieve_kmer_count(std::vector<LqSeqs> &lqseqs, KmerInfo &ki, uint16_t min_kmer_count,
                                Stats &st) {
    ki.prepare(min_kmer_count);
    size_t ksize = ki.ksize;
    if (!ki.path.empty()) { // variant (i): insert the candidates (main.rs:745-755), then stream the file (756)
        std::vector<std::unordered_set<uint64_t>> want((size_t)1 << ki.pre);
        for (auto &lq : lqseqs)
            for (auto &seq : lq.seqs) {
                if (seq.seq.size() > ksize)
                    iter2kmer(seq.seq, ksize, [&](uint64_t km) {
                        const uint64_t h = ki.to_hash(km);
                        want[h & ki.pmask].insert(h >> 10);
                    });
                else if (seq.kmer != INVALID_KMER)
                    want[seq.kmer & ki.pmask].insert(seq.kmer >> 10);
            }
        st.n_invalid += 0 * ki.stream_pass(want, min_kmer_count);
    }
    for (auto &lq : lqseqs)
        for (auto &seq : lq.seqs) {
            if (seq.seq.size() > ksize) {
                seq.kscore = min_kmer_count_of(seq.seq, ki);
                st.kmer_probes += seq.seq.size() - ksize + 1;
            } else if (seq.kmer != INVALID_KMER) {
                seq.kscore = ki.get_or0(seq.kmer);
                st.kmer_probes += 1;
            }
        }
}

// main.rs:780-801
static bool is_valid_snp(const std::string &s1, const std::string &s2) {
    size_t i = 0, j = 0;
    while (i < s1.size() && j < s2.size()) {
        if (s1[i] != s2[j]) return true;
        while (i + 1 < s1.size() && s1[i] == s1[i + 1]) i += 1;
        while (j + 1 < s2.size() && s2[j] == s2[j + 1]) j += 1;
        i += 1;
        j += 1;
    }
    return false;
}

static size_t get_min_count(size_t c) { return c >= 9 ? 3 : (c >= 6 ? 2 : 1); } // main.rs:803-811

// order_stat: HashMap<u32,usize> — only get/insert/entry are used (order-free)
typedef std::unordered_map<uint32_t, size_t> OrderStat;

// main.rs:813-849
static void fill_order_stat(const LqSeqs &lq, size_t *stats, OrderStat &order_stat, size_t &max1_c,
                            size_t &max1_p, size_t &max2_c, size_t &max2_p) {
    max1_c = max1_p = max2_c = max2_p = 0;
    for (size_t i = 0; i < LQSEQ_MAX_CAN_COUNT; ++i) stats[i] = 0;
    order_stat.clear();
    for (size_t p1 = 0; p1 < lq.seqs.size(); ++p1) {
        const LqSeq &seq = lq.seqs[p1];
        if (!(seq.kscore > 0)) continue;
        if (stats[p1] > 0) continue;
        size_t c = 0;
        for (size_t q = p1; q < lq.seqs.size(); ++q)
            if (lq.seqs[q].seq == seq.seq) c += 1;
        order_stat[lq.seqs[p1].order] = c;
        for (size_t q = p1; q < lq.seqs.size(); ++q)
            if (lq.seqs[q].seq == seq.seq) stats[q] = c;
        if (c > max1_c || (c == max1_c && seq.order == 0)) {
            max2_c = max1_c;
            max2_p = max1_p;
            max1_c = c;
            max1_p = p1;
        } else if (max1_p == max2_p || c > max2_c) {
            max2_c = c;
            max2_p = p1;
        }
    }
}

// main.rs:851-860
static bool no_dupseq_lqseq(const LqSeqs &lq) {
    for (size_t p1 = 1; p1 < lq.seqs.size(); ++p1)
        for (size_t p2 = p1 + 1; p2 < lq.seqs.size(); ++p2)
            if (lq.seqs[p1].seq == lq.seqs[p2].seq) return false;
    return true;
}

// main.rs:714-726
static void retain_sort_seqs(LqSeqs &lq, const OrderStat &stat, size_t min_c) {
    auto get = [&](uint32_t order) -> size_t {
        auto it = stat.find(order);
        return it == stat.end() ? 0 : it->second;
    };
    std::stable_sort(lq.seqs.begin(), lq.seqs.end(),
                     [&](const LqSeq &a, const LqSeq &b) { return get(a.order) > get(b.order); });
    size_t c = 0;
    for (auto &s : lq.seqs) {
        if (get(s.order) < min_c) break;
        c += 1;
    }
    lq.seqs.resize(c);
}

// main.rs:862-914
static void fill_seed_lqseqs(std::vector<LqSeqs> &lqseqs, long max_indel_len) {
    size_t stats[LQSEQ_MAX_CAN_COUNT];
    OrderStat order_stat;
    for (auto &lq : lqseqs) {
        size_t max1_c, max1_p, max2_c, max2_p;
        fill_order_stat(lq, stats, order_stat, max1_c, max1_p, max2_c, max2_p);
        RPANIC_IF(lq.seqs.empty(), "index out of bounds: lqseq.seqs[max1_p]");
        lq.sudoseed = lq.seqs[max1_p].seq;
        lq.set_lable(LABLE_SUCC);
        lq.set_lable(LABLE_RECH);
        size_t min_c = get_min_count(lq.seqs.size());
        RPANIC_IF(lq.seqs[0].order != 0, "the first lqseq is not ref.");

        auto it0 = order_stat.find(0);
        if (it0 != order_stat.end()) {
            if (it0->second > 1 && it0->second < min_c) it0->second = min_c;
        } else {
            size_t c = 0;
            for (auto &x : lq.seqs)
                if (x.seq == lq.seqs[0].seq) c += 1;
            if (c > 1) order_stat[0] = min_c;
        }

        if (max1_p != 0 && max1_c < min_c && (max1_c > 1 || no_dupseq_lqseq(lq))) {
            auto itp = order_stat.find(lq.seqs[max1_p].order);
            RPANIC_IF(itp == order_stat.end(), "unwrap on None: order_stat.get_mut(max1_p.order)");
            itp->second = min_c;
            order_stat[0] = min_c;
        } else if (max1_c < min_c) {
            order_stat[0] = min_c;
        }

        retain_sort_seqs(lq, order_stat, min_c);

        // NB: if retain_sort_seqs emptied seqs, the reference indexes seqs[0] and panics
        RPANIC_IF(lq.seqs.empty(), "index out of bounds: lqseq.seqs[0] after retain_sort_seqs");
        long d = (long)lq.sudoseed.size() - (long)lq.seqs[0].seq.size();
        bool skip_long_lqseq = (d < 0 ? -d : d) > max_indel_len;
        if (lq.seqs.size() <= 1 || skip_long_lqseq) {
            if (!lq.seqs.empty() || skip_long_lqseq) {
                lq.sudoseed = lq.seqs[0].seq;
            }
            lq.unset_lable(LABLE_RECH);
            lq.seqs.clear();
        }
    }
}

// main.rs:916-946
static void mark_hete_lqseqs(std::vector<LqSeqs> &lqseqs) {
    size_t stats[LQSEQ_MAX_CAN_COUNT];
    OrderStat order_stat;
    for (auto &lq : lqseqs) {
        size_t max1_c, max1_p, max2_c, max2_p;
        fill_order_stat(lq, stats, order_stat, max1_c, max1_p, max2_c, max2_p);
        size_t min_c = get_min_count(lq.seqs.size());
        if (max2_c >= min_c &&
            (lq.seqs[max1_p].seq.size() == lq.seqs[max2_p].seq.size() ||
             (lq.seqs.size() >= 6 && max2_c >= max1_c / 2)) &&
            is_valid_snp(lq.seqs[max1_p].seq, lq.seqs[max2_p].seq)) {
            lq.set_lable(LABLE_HETE);
            for (size_t p = 0; p < lq.seqs.size(); ++p)
                if (lq.seqs[p].kscore > 0 && stats[p] < min_c) lq.seqs[p].kscore = 0;
        }
    }
}

// main.rs:948-1015
static std::vector<uint32_t> phase_reads_by_lqseqs(const std::vector<LqSeqs> &lqseqs, bool asref,
                                                   bool use_all_reads) {
    Data data, dif, ref_data;
    std::unordered_set<uint32_t> invalid_ids;
    for (auto &lq : lqseqs) {
        if (!lq.has_lable(LABLE_HETE)) continue;
        for (size_t i = 0; i < lq.seqs.size(); ++i) {
            const LqSeq &s1 = lq.seqs[i];
            if (s1.kscore == 0) continue;
            for (size_t j = i + 1; j < lq.seqs.size(); ++j) {
                const LqSeq &s2 = lq.seqs[j];
                if (s2.kscore == 0) continue;
                float w = (s1.seq == s2.seq) ? 1.f : -1.f;
                if (s1.order == 0) {
                    if (asref) insert_data(ref_data, s1.order, s2.order, w);
                    if (w < 0.f && !use_all_reads) invalid_ids.insert(s2.order);
                    continue;
                }
                RPANIC_IF(s2.order == 0, "seq2 order is equal to 0");
                if (w == -1.f) {
                    insert_data(dif, s1.order, s2.order, -1.f);
                    insert_data(dif, s2.order, s1.order, -1.f);
                }
                insert_data(data, s1.order, s2.order, w);
                insert_data(data, s2.order, s1.order, w);
            }
        }
    }
    dif.for_each([&](uint32_t n1, const Inner &row) {
        for (auto &kv : row)
            if (kv.second <= -3.f) assign_data(data, n1, kv.first, kv.second);
    });
    if (!use_all_reads) {
        data.retain([&](uint32_t k, Inner &) { return invalid_ids.count(k) == 0; });
        data.for_each_mut([&](uint32_t, Inner &row) {
            for (auto it = row.begin(); it != row.end();)
                if (invalid_ids.count(it->first))
                    it = row.erase(it);
                else
                    ++it;
        });
    }
    const Inner *rw = nullptr;
    Inner rw_copy;
    {
        std::vector<uint32_t> ks = ref_data.keys();
        if (!ks.empty()) {
            rw_copy = *ref_data.get(ks[0]);
            rw = &rw_copy;
        }
    }
    std::vector<uint32_t> out = phase_communities(std::move(data), rw);
    for (uint32_t id : invalid_ids) out.push_back(id);
    return out;
}

// main.rs:1017-1025 (usize wrapping)
static size_t get_lqseqs_next_idx_by_lable(const std::vector<LqSeqs> &lqseqs, size_t i, uint8_t lable) {
    i -= 1;
    while (i < lqseqs.size() && !lqseqs[i].has_lable(lable)) i -= 1;
    return i;
}

// main.rs:1027-1058
static std::vector<ConsensusBase> update_consensus_with_lqseqs(const std::vector<LqSeqs> &lqseqs,
                                                               const std::vector<ConsensusBase> &consensus,
                                                               uint8_t lable) {
    std::vector<ConsensusBase> out;
    out.reserve(consensus.size());
    size_t i = 0;
    size_t li = get_lqseqs_next_idx_by_lable(lqseqs, lqseqs.size(), lable);
    while (i < consensus.size()) {
        uint32_t p = consensus[i].pos;
        if (li < lqseqs.size() && p == lqseqs[li].start) {
            for (char b : lqseqs[li].sudoseed) out.push_back({p, b});
            while (i < consensus.size() && consensus[i].pos <= lqseqs[li].end) i += 1;
            li = get_lqseqs_next_idx_by_lable(lqseqs, li, lable);
        } else {
            out.push_back(consensus[i]);
            i += 1;
        }
    }
    return out;
}

// main.rs:1060-1420
struct Reupdate {
    const std::vector<ConsensusBase> &consensus;
    explicit Reupdate(const std::vector<ConsensusBase> &c) : consensus(c) {}
    const ConsensusBase &at(size_t i) const {
        RPANIC_IF(i >= consensus.size(), "index out of bounds: consensus[i] in reupdate");
        return consensus[i];
    }
    // main.rs:1068-1097 (not include s & e)
    void iter_consensus_region(size_t &idx, uint32_t s, uint32_t e, size_t &si, size_t &ei) const {
        size_t i = idx;
        while (at(i).pos <= s) i += 1;
        while (at(i).pos > s) i -= 1;
        i += 1;
        RPANIC_IF(!(at(i).pos > s && at(i - 1).pos <= s), "assert iter_consensus_region 1");
        si = i;
        while (at(i).pos >= e) i -= 1;
        while (at(i).pos < e) i += 1;
        i -= 1;
        RPANIC_IF(!(at(i).pos < e && at(i + 1).pos >= e), "assert iter_consensus_region 2");
        idx = i;
        ei = i + 1;
    }
    // main.rs:1100-1139 (not include p)
    void iter_consensus_extend(size_t &idx, uint32_t p, size_t l, bool toleft, size_t &si,
                               size_t &ei) const {
        size_t i = idx;
        if (toleft) {
            while (at(i).pos >= p) i -= 1;
            while (at(i).pos < p) i += 1;
            RPANIC_IF(!(at(i).pos >= p && at(i - 1).pos < p), "assert iter_consensus_extend l");
            idx = i;
            ei = i;
            si = i > l ? i - l : 0;
        } else {
            while (at(i).pos <= p) i += 1;
            while (at(i).pos > p) i -= 1;
            RPANIC_IF(!(at(i).pos <= p && at(i + 1).pos > p), "assert iter_consensus_extend r");
            idx = i;
            si = i + 1;
            ei = (i + l < consensus.size()) ? i + l + 1 : consensus.size();
        }
    }
    void append(std::string &s, size_t si, size_t ei) const {
        RPANIC_IF(si > ei || ei > consensus.size(), "slice index out of range in reupdate");
        for (size_t i = si; i < ei; ++i) s.push_back(consensus[i].base);
    }
};

static std::vector<ConsensusBase> reupdate_consensus_with_lqseqs(std::vector<LqSeqs> &lqseqs,
                                                                 const std::vector<ConsensusBase> &consensus,
                                                                 KmerInfo &ki, uint16_t min_kmer_count,
                                                                 size_t iter_count, Stats &st) {
    ki.prepare(min_kmer_count);
    size_t ksize = ki.ksize;
    std::vector<size_t> rech_idxs;
    for (size_t i = lqseqs.size(); i-- > 0;)
        if (lqseqs[i].has_lable(LABLE_RECH)) rech_idxs.push_back(i);

    Reupdate ru(consensus);
    // The reference walks the groups twice (insert pass main.rs:1193-1265, score pass
    // 1269-1369) with the same cursor logic; the first pass only fills the candidate set
    // whose observable effect is subsumed by KmerInfo::prepare.  We must still replay the
    // cursor walk of pass one, because its asserts/panics are observable; it starts from
    // idx = 0 in both passes and is deterministic, so one walk suffices.
    size_t idx = 0, sj = 0, ej;
    struct Ks {
        size_t i, p;
        uint16_t k;
    };
    std::vector<Ks> kscore_buf;
    std::string buf;
    auto walk = [&]() {
    idx = 0, sj = 0;
    while (sj < rech_idxs.size()) {
        ej = sj + 1;
        while (ej < rech_idxs.size() &&
               lqseqs[rech_idxs[ej]].start < lqseqs[rech_idxs[ej - 1]].end + ki.ksize) {
            ej += 1;
            if (ej > sj + 5) break;
        }
        size_t si_l, ei_l, si_r, ei_r;
        ru.iter_consensus_extend(idx, lqseqs[rech_idxs[sj]].start, ksize - 1, true, si_l, ei_l);
        ru.iter_consensus_extend(idx, lqseqs[rech_idxs[ej - 1]].end, ksize - 1, false, si_r, ei_r);
        if (ej == sj + 1) {
            for (auto &seq : lqseqs[rech_idxs[sj]].seqs) {
                buf.clear();
                ru.append(buf, si_l, ei_l);
                buf += seq.seq;
                ru.append(buf, si_r, ei_r);
                seq.kscore = min_kmer_count_of(buf, ki);
                if (buf.size() >= ksize) st.kmer_probes += buf.size() - ksize + 1;
            }
        } else {
            kscore_buf.clear();
            size_t n = ej - sj;
            std::vector<size_t> cur(n, 0), lens(n);
            bool empty = false;
            for (size_t x = 0; x < n; ++x) {
                lens[x] = lqseqs[rech_idxs[sj + x]].seqs.size();
                if (lens[x] == 0) empty = true;
            }
            // multi_cartesian_product: lexicographic, last iterator fastest
            while (!empty) {
                buf.clear();
                ru.append(buf, si_l, ei_l);
                for (size_t i = 0; i < n; ++i) {
                    const std::string &seq = lqseqs[rech_idxs[sj + i]].seqs[cur[i]].seq;
                    if (i < n - 1) {
                        uint32_t s = lqseqs[rech_idxs[sj + i]].end;
                        uint32_t e = lqseqs[rech_idxs[sj + i + 1]].start;
                        buf += seq;
                        if (s + 1 != e) {
                            size_t si, ei;
                            ru.iter_consensus_region(idx, s, e, si, ei);
                            ru.append(buf, si, ei);
                        }
                    } else {
                        buf += seq;
                        ru.append(buf, si_r, ei_r);
                    }
                }
                uint16_t kscore = min_kmer_count_of(buf, ki);
                if (buf.size() >= ksize) st.kmer_probes += buf.size() - ksize + 1;
                if (kscore > 0)
                    for (size_t i = 0; i < n; ++i) kscore_buf.push_back({rech_idxs[sj + i], cur[i], kscore});
                // advance
                size_t d = n;
                while (d-- > 0) {
                    if (++cur[d] < lens[d]) break;
                    cur[d] = 0;
                    if (d == 0) {
                        empty = true;
                    }
                }
            }
            for (size_t x = sj; x < ej; ++x)
                for (auto &seq : lqseqs[rech_idxs[x]].seqs) seq.kscore = 0;
            for (auto &k : kscore_buf) lqseqs[k.i].seqs[k.p].kscore = k.k;
        }
        sj = ej;
    }
    };
    if (!ki.path.empty()) { // variant (i): insert pass (main.rs:1193-1265), file stream (1267), then the score pass
        std::vector<std::unordered_set<uint64_t>> want((size_t)1 << ki.pre);
        Stats scratch = st;
        tl_collect = &want;
        try {
            walk();
        } catch (...) {
            tl_collect = nullptr;
            throw;
        }
        tl_collect = nullptr;
        st = scratch; // (the insert pass does not count as probes)
        (void)ki.stream_pass(want, min_kmer_count);
    }
    walk();

    for (auto &lq : lqseqs) {
        if (!lq.has_lable(LABLE_RECH)) continue;
        size_t c = 0, valid_count = 0;
        for (size_t p = 0; p < lq.seqs.size(); ++p) {
            if (lq.seqs[p].kscore != 0) {
                if (c == 0 || lq.seqs[p].order == 0) c = p + 1;
                valid_count += 1;
            }
        }
        if (valid_count > 1) lq.set_lable(LABLE_TEMP);
        if (c != 0) {
            lq.sudoseed = lq.seqs[c - 1].seq;
        } else if (iter_count == 1) {
            size_t i = 0;
            for (size_t p = 0; p < lq.seqs.size(); ++p)
                if (lq.seqs[p].order == 0) {
                    i = p;
                    break;
                }
            RPANIC_IF(lq.seqs.empty(), "index out of bounds: lqseq.seqs[i] in reupdate");
            lq.sudoseed = lq.seqs[i].seq;
        }
    }
    std::vector<ConsensusBase> out = update_consensus_with_lqseqs(lqseqs, consensus, LABLE_RECH);
    for (auto &lq : lqseqs) {
        if (!lq.has_lable(LABLE_RECH)) continue;
        if (lq.has_lable(LABLE_TEMP))
            lq.unset_lable(LABLE_TEMP);
        else
            lq.unset_lable(LABLE_RECH);
    }
    return out;
}

static void trace_lqseqs(Ctx &cx, const char *tag, const std::vector<LqSeqs> &lqseqs) {
    if (!cx.trace.on) return;
    std::vector<uint32_t> start, end, cand_off, order, seq_off, sudo_off;
    std::vector<uint16_t> kscore;
    std::vector<uint64_t> kmer;
    std::vector<uint8_t> lable, seqs, sudo;
    cand_off.push_back(0);
    seq_off.push_back(0);
    sudo_off.push_back(0);
    for (auto &lq : lqseqs) {
        start.push_back(lq.start);
        end.push_back(lq.end);
        lable.push_back(lq.lable);
        sudo.insert(sudo.end(), lq.sudoseed.begin(), lq.sudoseed.end());
        sudo_off.push_back((uint32_t)sudo.size());
        for (auto &s : lq.seqs) {
            order.push_back(s.order);
            kscore.push_back(s.kscore);
            kmer.push_back(s.kmer);
            seqs.insert(seqs.end(), s.seq.begin(), s.seq.end());
            seq_off.push_back((uint32_t)seqs.size());
        }
        cand_off.push_back((uint32_t)order.size());
    }
    std::string t(tag);
    cx.trace.put(cx.pass, (t + ".start").c_str(), start);
    cx.trace.put(cx.pass, (t + ".end").c_str(), end);
    cx.trace.put(cx.pass, (t + ".lable").c_str(), lable);
    cx.trace.put(cx.pass, (t + ".sudo_off").c_str(), sudo_off);
    cx.trace.put(cx.pass, (t + ".sudo").c_str(), sudo);
    cx.trace.put(cx.pass, (t + ".cand_off").c_str(), cand_off);
    cx.trace.put(cx.pass, (t + ".order").c_str(), order);
    cx.trace.put(cx.pass, (t + ".kscore").c_str(), kscore);
    cx.trace.put(cx.pass, (t + ".kmer").c_str(), kmer);
    cx.trace.put(cx.pass, (t + ".seq_off").c_str(), seq_off);
    cx.trace.put(cx.pass, (t + ".seq").c_str(), seqs);
}
static void trace_cns(Ctx &cx, const char *tag, const std::vector<ConsensusBase> &cns) {
    if (!cx.trace.on) return;
    std::vector<uint32_t> pos(cns.size());
    std::vector<uint8_t> base(cns.size());
    for (size_t i = 0; i < cns.size(); ++i) {
        pos[i] = cns[i].pos;
        base[i] = (uint8_t)cns[i].base;
    }
    std::string t(tag);
    cx.trace.put(cx.pass, (t + ".pos").c_str(), pos);
    cx.trace.put(cx.pass, (t + ".base").c_str(), base);
}

// main.rs:1422-1553
static bool generate_lqseqs_from_tags_kmer(Ctx &cx, std::vector<AlignSeq> &alignseqs,
                                           std::vector<LqSeqs> lqseqs,
                                           std::vector<ConsensusBase> consensus, bool out_cns,
                                           std::vector<ConsensusBase> &out) {
    Opt &opt = cx.opt;
    std::vector<AlignBase> align_bases;
    RPANIC_IF(opt.yak.empty(), "index out of bounds: opt.yak[0]");
    KmerInfo &ki0 = opt.yak[0];
    uint64_t ksize = ki0.ksize;
    uint64_t shift = 2 * (ksize - 1);
    uint64_t mask = (1ULL << (2 * ksize)) - 1;
    uint64_t kmers[2];
    uint64_t l;
    size_t j;
    size_t s = lqseqs.size() - 1;
    for (size_t idx = 0; idx < alignseqs.size(); ++idx) {
        const AlignSeq &ab = alignseqs[idx];
        if (ab.empty()) continue;
        while (s > 0 && lqseqs[s].start < ab.aln_t_s) s -= 1;
        if (lqseqs[s].start < ab.aln_t_s || lqseqs[s].end > ab.aln_t_e) continue;
        j = s;
        while (j > 0 && lqseqs[j].end <= ab.aln_t_e) j -= 1;
        if (lqseqs[j].end > ab.aln_t_e) j += 1;

        align_bases.clear();
        size_t p = 0;
        AlignBase a;
        while (ab.get_align_tag(p, a)) {
            align_bases.push_back(a);
            if (a.t_pos > lqseqs[j].end + (uint32_t)ksize) break;
        }

        for (size_t r = j; r <= s; ++r) {
            LqSeqs &lq = lqseqs[r];
            if (lq.seqs.size() >= LQSEQ_MAX_CAN_COUNT) continue;
            l = 0;
            kmers[0] = kmers[1] = 0;
            std::string seq;
            size_t from = (size_t)lq.start - (size_t)ab.aln_t_s;
            RPANIC_IF(from > align_bases.size(), "slice start out of range: align_bases[start-aln_t_s..]");
            for (size_t q = from; q < align_bases.size(); ++q) {
                const AlignBase &b = align_bases[q];
                if (b.t_pos >= lq.start && b.q_base != 4) {
                    if (b.t_pos <= lq.end) seq.push_back((char)SEQ_NUM[b.q_base]);
                    if (l < ksize) {
                        kmers[0] = (kmers[0] << 2 | (uint64_t)b.q_base) & mask;
                        kmers[1] = (kmers[1] >> 2) | (3 ^ (uint64_t)b.q_base) << shift;
                        l += 1;
                    }
                    if (b.t_pos > lq.end && l >= ksize) break;
                }
            }
            uint64_t kmer = l >= ksize ? (kmers[0] < kmers[1] ? kmers[0] : kmers[1]) : INVALID_KMER;
            if (!seq.empty()) {
                LqSeq ls;
                ls.order = (uint32_t)idx;
                ls.kscore = 0;
                ls.kmer = kmer != INVALID_KMER ? ki0.to_hash(kmer) : INVALID_KMER;
                ls.seq = std::move(seq);
                lq.seqs.push_back(std::move(ls));
            }
        }
    }

    retrieve_kmer_count(lqseqs, ki0, opt.min_kmer_count, cx.stats);
    cx.stats.n_regions += lqseqs.size();
    for (auto &lq : lqseqs) cx.stats.n_candidates += lq.seqs.size();
    trace_lqseqs(cx, "cand", lqseqs);

    if (out_cns) {
        fill_seed_lqseqs(lqseqs, opt.max_indel_len);
        trace_lqseqs(cx, "seed", lqseqs);
        consensus = update_consensus_with_lqseqs(lqseqs, consensus, LABLE_SUCC);
        trace_cns(cx, "cns_succ", consensus);
        for (size_t p = 0; p < opt.yak.size(); ++p) {
            consensus = reupdate_consensus_with_lqseqs(lqseqs, consensus, opt.yak[p],
                                                       opt.min_kmer_count, p + 1, cx.stats);
            std::string t = "rech" + std::to_string(p);
            trace_lqseqs(cx, t.c_str(), lqseqs);
            trace_cns(cx, ("cns_" + t).c_str(), consensus);
        }
        out = std::move(consensus);
        return true;
    } else {
        mark_hete_lqseqs(lqseqs);
        trace_lqseqs(cx, "hete", lqseqs);
        std::vector<uint32_t> invalid = phase_reads_by_lqseqs(lqseqs, opt.model_ref, opt.use_all_reads);
        std::vector<uint32_t> inv_sorted(invalid);
        std::sort(inv_sorted.begin(), inv_sorted.end());
        inv_sorted.erase(std::unique(inv_sorted.begin(), inv_sorted.end()), inv_sorted.end());
        cx.trace.put(cx.pass, "invalid_ids", inv_sorted);
        cx.stats.n_invalid += inv_sorted.size();
        for (uint32_t id : invalid) {
            RPANIC_IF(id >= alignseqs.size(), "index out of bounds: alignseqs[id]");
            alignseqs[id].align_bases = nullptr;
        }
        return false;
    }
}

// main.rs:1555-1643
static bool generate_cns_from_best_score_lq(Ctx &cx, const std::vector<Msa> &msas,
                                            std::vector<AlignSeq> &alignseqs, const Kmer *best,
                                            bool out_cns, std::vector<ConsensusBase> &out) {
    std::vector<LqSeqs> lqseqs;
    std::vector<ConsensusBase> cns;
    cns.reserve(msas.size());
    const int64_t hq_min_qv = 95;
    const size_t lq_min_length = 2;
    bool has_lq = false;
    size_t lq_s = (size_t)-1, lq_e = 0, p = 0;

    AlignBase b1, base2, base3;
    best->get_bases((uint32_t)msas.size() - 1, b1, base2, base3);
    for (;;) {
        if (base3.q_base != 4) {
            RPANIC_IF(base3.t_pos >= msas.size(), "index out of bounds: msas[base3.t_pos]");
            int64_t coverage = msas[base3.t_pos].coverage();
            RPANIC_IF(coverage == 0, "attempt to divide by zero (coverage)");
            int64_t qv = (int64_t)best->count * 100 / coverage;
            cns.push_back({base3.t_pos, (char)SEQ_NUM[base3.q_base]});
            if (coverage < 2) {
                has_lq = false;
                lq_s = (size_t)-1;
            } else if (qv < hq_min_qv) {
                if (lq_s == (size_t)-1) lq_s = p;
                lq_e = p;
                has_lq = true;
            } else if (has_lq && p - lq_e > 2 * lq_min_length && cns[p - 1].pos != cns[p - 2].pos &&
                       cns[p - 1].base != cns[p - 2].base) {
                lq_e = p - 2;
                lq_s = lq_s > lq_min_length ? lq_s - lq_min_length : 1;
                while (lq_s > 1 &&
                       (cns[lq_s - 1].pos == cns[lq_s].pos || cns[lq_s - 1].base == cns[lq_s].base))
                    lq_s -= 1;
                size_t n = lqseqs.size();
                if (n >= 1 && cns[lq_s].pos >= lqseqs[n - 1].start) {
                    lqseqs[n - 1].start = cns[lq_e].pos;
                } else {
                    LqSeqs lq;
                    lq.end = cns[lq_s].pos;
                    lq.start = cns[lq_e].pos;
                    lqseqs.push_back(std::move(lq));
                }
                has_lq = false;
                lq_s = (size_t)-1;
            }
            p += 1;
        }
        if (base2.is_head()) break;
        RPANIC_IF(base2.t_pos >= msas.size(), "index out of bounds: msas[base2.t_pos]");
        const Msa &m = msas[base2.t_pos];
        RPANIC_IF(best->besti >= m.kmers.size(), "index out of bounds: kmers[besti]");
        best = &m.kmers[best->besti];
        uint32_t pp = base2.t_pos;
        best->get_bases(pp, b1, base2, base3);
    }
    std::reverse(cns.begin(), cns.end());
    trace_cns(cx, "cns_raw", cns);
    {
        std::vector<uint32_t> st, en;
        for (auto &lq : lqseqs) {
            st.push_back(lq.start);
            en.push_back(lq.end);
        }
        cx.trace.put(cx.pass, "lq.start", st);
        cx.trace.put(cx.pass, "lq.end", en);
    }
    if (lqseqs.empty()) {
        out = std::move(cns);
        return true;
    }
    return generate_lqseqs_from_tags_kmer(cx, alignseqs, std::move(lqseqs), std::move(cns), out_cns, out);
}

// main.rs:1645-1687
static bool get_cns_from_align_tags(Ctx &cx, std::vector<Msa> &msas, std::vector<AlignSeq> &alignseqs,
                                    bool out_cns, std::vector<ConsensusBase> &out) {
    Kmer dflt;
    const Kmer *global_best = &dflt;
    AlignBase base1, base2, base3, pb1, pb2, pb3;
    const size_t L = msas.size();
    for (size_t p = 0; p < L; ++p) {
        Msa &msa = msas[p];
        for (Kmer &kmer : msa.kmers) {
            kmer.get_bases((uint32_t)p, base1, base2, base3);
            int64_t coverage = msa.coverage();
            uint32_t besti = 0;
            int64_t kmer_score;
            if (base2.is_head()) {
                kmer_score = 10 * (int64_t)kmer.count - 4 * coverage;
            } else {
                kmer_score = INT64_MIN >> 1;
                RPANIC_IF(base2.t_pos >= L, "index out of bounds: msas[base2.t_pos] (dp)");
                const Msa &pm = msas[base2.t_pos];
                // Msa::get(base1, base2) (main.rs:209-225)
                uint8_t base23 = (uint8_t)(base1.q_base << 4 | base2.q_base);
                uint16_t delta23 = base1.t_pos == base2.t_pos ? 1 : 0;
                for (size_t pi = 0; pi < pm.kmers.size(); ++pi) {
                    const Kmer &v = pm.kmers[pi];
                    if (!((uint8_t)v.bases == base23 && ((v.bases >> 12) & 1) == delta23)) continue;
                    v.get_bases(base2.t_pos, pb1, pb2, pb3);
                    if (!(pb2 == base1 && pb3 == base2)) continue;
                    if (base2.t_pos >= 3 && pb1.is_head()) continue;
                    int64_t score = v.score + 10 * (int64_t)kmer.count - 4 * coverage;
                    if (score > kmer_score || (score == kmer_score && pb1.q_base != 4)) {
                        kmer_score = score;
                        besti = (uint32_t)pi;
                    }
                }
            }
            kmer.score = kmer_score;
            kmer.besti = besti;
            if (p == L - 1 && kmer_score >= global_best->score) global_best = &kmer;
        }
    }
    // No node at the last position with a score >= 0: the reference backtracks from its *default* Kmer here
    // (main.rs:1651,1680: a spurious 'A' at L - 1 with count 0, then node 0 of position L - 2) — the artefact of a contig
    // whose best path has under 40 % support end to end.  Restated literally (global_best stays &dflt); the product does
    // the same since round 6 (k_dp_finish / k_default_tail).
    return generate_cns_from_best_score_lq(cx, msas, alignseqs, global_best, out_cns, out);
}

// main.rs:576-589
static void update_msas(std::vector<Msa> &msas, const std::vector<AlignSeq> &alignseqs) {
    for (const AlignSeq &as : alignseqs) {
        if (as.empty()) continue;
        size_t p = 0;
        AlignBase b1 = AlignBase::head(as.aln_t_s - 1, 0);
        AlignBase b2 = AlignBase::head(as.aln_t_s - 1, 1);
        AlignBase b3;
        while (as.get_align_tag(p, b3)) {
            RPANIC_IF(b3.t_pos >= msas.size(), "index out of bounds: msas[b3.t_pos]");
            msas[b3.t_pos].push(Kmer::make(b1, b2, b3));
            b1 = b2;
            b2 = b3;
        }
    }
}

static void trace_graph(Ctx &cx, const std::vector<Msa> &msas) {
    if (!cx.trace.on) return;
    std::vector<uint32_t> off(msas.size() + 1, 0), count;
    std::vector<uint16_t> bases, delta;
    for (size_t p = 0; p < msas.size(); ++p) {
        for (auto &k : msas[p].kmers) {
            bases.push_back(k.bases);
            delta.push_back(k.delta);
            count.push_back(k.count);
        }
        off[p + 1] = (uint32_t)count.size();
    }
    cx.trace.put(cx.pass, "graph.off", off);
    cx.trace.put(cx.pass, "graph.bases", bases);
    cx.trace.put(cx.pass, "graph.delta", delta);
    cx.trace.put(cx.pass, "graph.count", count);
}

// the loop main.rs:1819-1836
static void polish(Ctx &cx, uint32_t L, const np2_read_t *reads, uint32_t n_reads,
                   const uint8_t *nibbles, std::vector<ConsensusBase> &out) {
    std::vector<AlignSeq> alignseqs(n_reads);
    for (uint32_t i = 0; i < n_reads; ++i) {
        alignseqs[i].aln_t_s = reads[i].aln_t_s;
        alignseqs[i].aln_t_e = reads[i].aln_t_e;
        alignseqs[i].align_bases = (reads[i].flags & NP2_READ_DROPPED) ? nullptr : nibbles + reads[i].nib_off;
    }
    std::vector<Msa> msas(L);
    size_t i = 0;
    cx.stats = Stats();
    for (;;) {
        cx.pass = (int)i;
        update_msas(msas, alignseqs);
        for (auto &m : msas) m.sort();
        for (auto &m : msas) cx.stats.n_nodes += m.kmers.size();
        trace_graph(cx, msas);
        if (i + 1 == cx.opt.iter_count) {
            bool some = get_cns_from_align_tags(cx, msas, alignseqs, true, out);
            RPANIC_IF(!some, "unwrap on None consensus");
            return;
        } else {
            std::vector<ConsensusBase> dummy;
            get_cns_from_align_tags(cx, msas, alignseqs, false, dummy);
            for (auto &m : msas) m.kmers.clear();
        }
        i += 1;
    }
}

} // namespace np2o

// ---------------------------------------------------------------------------------------
// C ABI (np2o_*) — mirrors include/np2.h for the tests
// ---------------------------------------------------------------------------------------
using namespace np2o;
extern "C" {

void *np2o_ctx_create(const np2_yak_t *yaks, int n_yak) {
    Ctx *cx = new Ctx();
    for (int i = 0; i < n_yak; ++i) {
        if (yaks[i].k >= 32 || yaks[i].k < 2) {
            delete cx;
            return nullptr;
        }
        cx->opt.yak.emplace_back();
        cx->opt.yak.back().load(yaks[i]);
    }
    return cx;
}
// variant (i) of the CPU baseline: yak dump paths (one per table, same order); NULL / "" = in-memory only
int np2o_set_yak_files(void *c, const char *const *paths, int n) {
    Ctx *cx = (Ctx *)c;
    if ((size_t)n != cx->opt.yak.size()) return NP2_E_ARG;
    for (int i = 0; i < n; ++i) cx->opt.yak[i].path = paths && paths[i] ? paths[i] : "";
    return 0;
}
// a further context over the same (read-only) k-mer tables, filtered for `min_kmer_count` once in the parent
void *np2o_ctx_clone(void *parent, uint16_t min_kmer_count) {
    Ctx *p = (Ctx *)parent;
    for (auto &ki : p->opt.yak) ki.prepare(min_kmer_count);
    Ctx *cx = new Ctx();
    cx->opt.yak = p->opt.yak; // shared_ptr copies
    return cx;
}
void np2o_ctx_destroy(void *c) { delete (Ctx *)c; }
const char *np2o_last_error(void *c) { return ((Ctx *)c)->err.c_str(); }
void np2o_set_trace(void *c, int on) { ((Ctx *)c)->trace.on = on != 0; }

int np2o_polish_contig(void *c, const uint8_t *ref, uint32_t L, const np2_read_t *reads,
                       uint32_t n_reads, const uint8_t *nibbles, const np2_opts_t *o,
                       uint8_t **out_bases, uint32_t **out_pos, uint64_t *out_len) {
    (void)ref; // reads[0] carries the contig (main.rs:1732-1739)
    Ctx &cx = *(Ctx *)c;
    cx.opt.min_kmer_count = o->min_kmer_count;
    cx.opt.max_indel_len = o->max_indel_len;
    cx.opt.iter_count = o->iter_count;
    cx.opt.model_ref = o->model_ref != 0;
    cx.opt.use_all_reads = o->use_all_reads != 0;
    cx.trace.items.clear();
    if (o->iter_count < 1 || L < 3 || n_reads < 1) {
        cx.err = "bad argument";
        return NP2_E_ARG;
    }
    std::vector<ConsensusBase> out;
    try {
        polish(cx, L, reads, n_reads, nibbles, out);
    } catch (const RefPanic &e) {
        cx.err = std::string("reference would panic: ") + e.what();
        return NP2_E_REFPANIC;
    } catch (const Unsupported &e) {
        cx.err = e.what();
        return NP2_E_UNSUPPORTED;
    }
    *out_len = out.size();
    *out_bases = (uint8_t *)malloc(out.size() + 1);
    *out_pos = (uint32_t *)malloc(sizeof(uint32_t) * (out.size() + 1));
    for (size_t i = 0; i < out.size(); ++i) {
        (*out_bases)[i] = (uint8_t)out[i].base;
        (*out_pos)[i] = out[i].pos;
    }
    return NP2_OK;
}
void np2o_free(void *p) { free(p); }

int np2o_trace_get(void *c, int pass, const char *name, const void **data, uint64_t *nbytes) {
    Ctx &cx = *(Ctx *)c;
    auto it = cx.trace.items.find(std::to_string(pass) + ":" + name);
    if (it == cx.trace.items.end()) return NP2_E_ARG;
    *data = it->second.data();
    *nbytes = it->second.size();
    return NP2_OK;
}

// stats of the last polish: [kmer_probes, n_regions, n_candidates, n_nodes, n_invalid]
void np2o_last_stats(void *c, uint64_t *out5) {
    Ctx &cx = *(Ctx *)c;
    out5[0] = cx.stats.kmer_probes;
    out5[1] = cx.stats.n_regions;
    out5[2] = cx.stats.n_candidates;
    out5[3] = cx.stats.n_nodes;
    out5[4] = cx.stats.n_invalid;
}

int np2o_score_strings(void *c, int yak_idx, const uint8_t *strs, const uint64_t *off, uint64_t n,
                       uint16_t min_kmer_count, uint16_t *scores) {
    Ctx &cx = *(Ctx *)c;
    if (yak_idx < 0 || (size_t)yak_idx >= cx.opt.yak.size()) return NP2_E_ARG;
    KmerInfo &ki = cx.opt.yak[yak_idx];
    ki.prepare(min_kmer_count);
    for (uint64_t i = 0; i < n; ++i) {
        std::string s((const char *)strs + off[i], (const char *)strs + off[i + 1]);
        scores[i] = min_kmer_count_of(s, ki);
    }
    return NP2_OK;
}
int np2o_lookup_hashes(void *c, int yak_idx, const uint64_t *hashes, uint64_t n,
                       uint16_t min_kmer_count, uint16_t *counts) {
    Ctx &cx = *(Ctx *)c;
    if (yak_idx < 0 || (size_t)yak_idx >= cx.opt.yak.size()) return NP2_E_ARG;
    KmerInfo &ki = cx.opt.yak[yak_idx];
    ki.prepare(min_kmer_count);
    for (uint64_t i = 0; i < n; ++i) counts[i] = ki.get_or0(hashes[i]);
    return NP2_OK;
}
uint64_t np2o_yak_hash64(uint64_t key, uint32_t k) { return yak_hash64(key, (1ULL << (2 * (uint64_t)k)) - 1); }

// SwissTable script for the hand-traced order vectors (tests/test_swiss_vectors.py): op 0 = HashMap::insert(key),
// 1 = remove(key), 2 = entry(key).or_insert (reserve(1) first when vacant); returns the keys in iteration order
int np2o_swiss_order(const uint32_t *ops, const uint32_t *keys, uint32_t n, uint32_t *out, uint32_t *n_out) {
    hb::FxMap<int> m;
    for (uint32_t i = 0; i < n; ++i) {
        if (ops[i] == 0)
            m.insert(keys[i], 0);
        else if (ops[i] == 1)
            m.remove(keys[i]);
        else if (!m.contains(keys[i]))
            m.entry_insert_vacant(keys[i], 0);
    }
    uint32_t c = 0;
    m.for_each([&](uint32_t k, const int &) { out[c++] = k; });
    *n_out = c;
    return 0;
}

// Louvain entry for unit tests: edges (a,b,w) applied with insert_data in order, optional ref row
int np2o_phase_communities(const uint32_t *ea, const uint32_t *eb, const float *ew, uint64_t n_edges,
                           const uint32_t *ref_ids, const float *ref_w, uint64_t n_ref, int has_ref,
                           uint32_t *out_ids, uint64_t *n_out) {
    Data d;
    for (uint64_t i = 0; i < n_edges; ++i) insert_data(d, ea[i], eb[i], ew[i]);
    Inner rw;
    for (uint64_t i = 0; i < n_ref; ++i) rw[ref_ids[i]] = ref_w[i];
    try {
        std::vector<uint32_t> r = phase_communities(std::move(d), has_ref ? &rw : nullptr);
        std::sort(r.begin(), r.end());
        *n_out = r.size();
        for (size_t i = 0; i < r.size(); ++i) out_ids[i] = r[i];
    } catch (const RefPanic &) {
        return NP2_E_REFPANIC;
    }
    return NP2_OK;
}
}
