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
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

namespace np2 {
namespace phase {

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
        for (;;) {
            bool saw_empty = false;
            for (size_t b = 0; b < kGroup; ++b) {
                const uint8_t c = ctrl_[pos + b];
                if (c == tag) {
                    const size_t i = (pos + b) & mask_;
                    if (keys_[i] == key && !(ctrl_[i] & 0x80)) return i;
                }
                saw_empty |= (c == kEmpty);
            }
            if (saw_empty) return npos;
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
            for (size_t b = 0; b < kGroup; ++b)
                if (ctrl_[pos + b] & 0x80) {
                    size_t r = (pos + b) & mask_;
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
typedef std::unordered_map<uint32_t, float> Row; // inner maps only feed exact small-integer f32 sums
typedef SwissOrderMap<Row> Weights;

inline void add_weight(Weights &w, uint32_t a, uint32_t b, float v) { // insert_data, louvain.rs:273-279
    if (Row *r = w.get(a)) {
        (*r)[b] += v;
    } else {
        Row n;
        n[b] = v;
        w.put_vacant(a, std::move(n));
    }
}
inline void set_weight(Weights &w, uint32_t a, uint32_t b, float v) { // assign_data, louvain.rs:282-288
    if (Row *r = w.get(a)) {
        (*r)[b] = v;
    } else {
        Row n;
        n[b] = v;
        w.put_vacant(a, std::move(n));
    }
}

struct Community {
    uint32_t id = 0;
    float weight = 0.f;
    std::vector<uint32_t> members; // union of original read ids
};

class SignedLouvain {
  public:
    explicit SignedLouvain(Weights w) : w_(std::move(w)) {
        for (uint32_t v : w_.key_list()) { // louvain.rs:65-68
            comm_.put(v, OrderSet::single(v, Nil{}));
            Community c;
            c.id = v;
            c.members.push_back(v);
            node_[v] = c;
        }
    }
    // returns false if the reference's weight<0 assertion (louvain.rs:234-237) would fire
    bool run(Weights &conflicts, std::vector<Community> &out) {
        while (local_moving()) aggregate();
        return collect(conflicts, out);
    }

  private:
    Weights w_;
    SwissOrderMap<OrderSet> comm_;
    std::unordered_map<uint32_t, Community> node_;

    bool local_moving() { // first_stage, louvain.rs:72-117
        bool moved_any = false;
        std::vector<uint32_t> visit = w_.key_list();
        std::sort(visit.begin(), visit.end());
        std::vector<std::pair<uint32_t, float>> gains;
        for (bool again = true; again;) {
            again = false;
            for (uint32_t v : visit) {
                const uint32_t cur = node_[v].id;
                const Row &row = *w_.get(v);
                gains.clear();
                for (const auto &e : row) {
                    const uint32_t c = node_[e.first].id;
                    bool seen = false;
                    for (const auto &g : gains) seen |= (g.first == c);
                    if (seen) continue;
                    const OrderSet &set = *comm_.get(c);
                    float s = 0.f;
                    for (const auto &e2 : row)
                        if (set.has(e2.first)) s += e2.second;
                    gains.emplace_back(c, s);
                }
                if (gains.empty()) continue;
                size_t best = 0; // max weight, ties -> smaller community id (louvain.rs:99-101)
                for (size_t i = 1; i < gains.size(); ++i)
                    if (gains[i].second > gains[best].second ||
                        (gains[i].second == gains[best].second && gains[i].first < gains[best].first))
                        best = i;
                if (gains[best].second > 0.f && gains[best].first != cur) {
                    node_[v].id = gains[best].first;
                    comm_.get(gains[best].first)->put(v, Nil{});
                    comm_.get(cur)->take(v, nullptr);
                    again = true;
                    moved_any = true;
                }
            }
        }
        return moved_any;
    }

    float internal_weight(const OrderSet &set, std::vector<uint32_t> &members) const {
        float wsum = 0.f;
        set.each([&](uint32_t v, const Nil &) {
            const Community &c = node_.at(v);
            for (uint32_t m : c.members)
                if (std::find(members.begin(), members.end(), m) == members.end()) members.push_back(m);
            wsum += c.weight;
            if (const Row *row = w_.get(v))
                for (const auto &e : *row)
                    if (set.has(e.first)) wsum += e.second / 2.0f;
        });
        return wsum;
    }

    void aggregate() { // second_stage, louvain.rs:119-195
        std::unordered_map<uint32_t, Community> nnode;
        SwissOrderMap<OrderSet> ncomm;
        std::vector<uint32_t> split;
        comm_.each([&](uint32_t id, const OrderSet &set) {
            if (set.empty()) return;
            Community c;
            c.id = id;
            c.weight = internal_weight(set, c.members);
            if (c.weight < 0.f) {
                split.push_back(id);
            } else {
                ncomm.put(id, OrderSet::single(id, Nil{}));
                nnode[id] = std::move(c);
            }
        });
        for (uint32_t id : split) { // decluster negative communities, louvain.rs:145-165
            OrderSet set;
            comm_.take(id, &set);
            for (uint32_t v : set.key_list()) {
                uint32_t nid = v;
                while (ncomm.has(nid) || nnode.count(nid)) ++nid;
                ncomm.put(nid, OrderSet::single(nid, Nil{}));
                Community c = node_[v];
                c.id = nid;
                nnode[nid] = std::move(c);
                comm_.put(nid, OrderSet::single(v, Nil{}));
            }
        }
        Weights nw;
        comm_.each([&](uint32_t a, const OrderSet &sa) {
            if (sa.empty()) return;
            comm_.each([&](uint32_t b, const OrderSet &sb) {
                if (b <= a || sb.empty()) return;
                float e = 0.f;
                sa.each([&](uint32_t v, const Nil &) {
                    if (const Row *row = w_.get(v))
                        for (const auto &x : *row)
                            if (sb.has(x.first)) e += x.second;
                });
                if (e != 0.f) {
                    add_weight(nw, a, b, e);
                    add_weight(nw, b, a, e);
                }
            });
        });
        w_ = std::move(nw);
        comm_ = std::move(ncomm);
        node_ = std::move(nnode);
    }

    bool collect(Weights &conflicts, std::vector<Community> &out) { // get_communities, louvain.rs:197-245
        out.clear();
        comm_.each([&](uint32_t id, const OrderSet &set) {
            if (set.empty()) return;
            Community c;
            c.id = id;
            c.weight = internal_weight(set, c.members);
            out.push_back(std::move(c));
        });
        bool ok = true;
        for (const Community &a : out)
            for (const Community &b : out) {
                if (b.id <= a.id) continue;
                float e = 0.f;
                comm_.get(a.id)->each([&](uint32_t x, const Nil &) {
                    const Row *row = w_.get(x);
                    if (!row) return;
                    comm_.get(b.id)->each([&](uint32_t y, const Nil &) {
                        auto it = row->find(y);
                        if (it != row->end()) e += it->second;
                    });
                });
                if (e != 0.f) {
                    if (!(e < 0.f)) ok = false;
                    add_weight(conflicts, a.id, b.id, e);
                    add_weight(conflicts, b.id, a.id, e);
                }
            }
        return ok;
    }
};

// phase_communities (louvain.rs:290-356): reads of the losing communities
inline bool losing_reads(Weights graph, const Row *ref_row, std::vector<uint32_t> &losers) {
    SignedLouvain lv(std::move(graph));
    Weights conflicts;
    std::vector<Community> comms;
    if (!lv.run(conflicts, comms)) return false;
    if (ref_row) {
        std::vector<std::pair<int32_t, float>> key(comms.size());
        for (size_t i = 0; i < comms.size(); ++i) {
            int32_t cnt = 0;
            float w = 0.f;
            for (uint32_t m : comms[i].members) {
                auto it = ref_row->find(m);
                if (it == ref_row->end()) continue;
                cnt += (it->second > 0.f) - (it->second < 0.f);
                w += it->second;
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
        const Row *adj = conflicts.get(comms[p].id);
        if (!adj) continue;
        for (size_t q = p + 1; q < comms.size(); ++q)
            if (!lost.count(comms[q].id) && adj->count(comms[q].id)) lost.insert(comms[q].id);
    }
    for (const Community &c : comms)
        if (lost.count(c.id)) losers.insert(losers.end(), c.members.begin(), c.members.end());
    return true;
}

} // namespace phase
} // namespace np2
