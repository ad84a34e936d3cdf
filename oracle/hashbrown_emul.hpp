// TEST INFRASTRUCTURE (oracle) — not part of the product path.
//
// Emulation of the *iteration order* of Rust's std HashMap/HashSet<u32, _, FxBuildHasher>
// as used by the reference's phasing code (src/utils/louvain.rs:1, src/main.rs:3).
// The reference's Louvain tie order leaks hashbrown bucket order (SURVEY.md §7.3 H1,
// Appendix B): louvain.rs:65, 123-142, 145-165, 199-217.
//
// Third-party code absent from /root/reference (no Cargo.lock is checked in):
//   * fxhash 0.2.1  — FxHasher64::write_u32: h = (rotl(h,5) ^ k) * 0x517cc1b727220a95, h0 = 0
//   * hashbrown 0.12.x (std of Rust 1.64-1.68, contemporary with NextPolish2 v0.2.x):
//     SwissTable, SSE2 group width 16, h2 = top 7 hash bits, triangular probing,
//     capacity_to_buckets / bucket_mask_to_capacity, RawTable::insert (grow only when the
//     found slot is EMPTY and growth_left == 0), rustc_entry (reserve(1) for vacant
//     entries), erase (EMPTY vs DELETED rule), reserve_rehash (rehash_in_place vs resize),
//     iteration in ascending bucket index.
// PARITY UNPINNED: no Rust toolchain in this environment, so this restates the published
// algorithm from crate knowledge; it cannot be run against the real crate here.
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace hb {

static inline uint64_t fxhash_u32(uint32_t k) { return (uint64_t)k * 0x517cc1b727220a95ULL; }

template <class V> class FxMap {
  public:
    static constexpr uint8_t EMPTY = 0xFF, DELETED = 0x80;
    static constexpr size_t W = 16; // Group::WIDTH (SSE2)
    static constexpr size_t NPOS = (size_t)-1;

    FxMap() { ctrl_.assign(W, EMPTY); }

    size_t len() const { return items_; }
    bool is_empty() const { return items_ == 0; }
    size_t buckets() const { return bucket_mask_ + 1; }

    // ---- lookup -------------------------------------------------------------------
    size_t find(uint32_t key) const {
        if (singleton_) return NPOS;
        uint64_t h = fxhash_u32(key);
        uint8_t tag = h2(h);
        size_t pos = (size_t)h & bucket_mask_, stride = 0;
        for (;;) {
            for (size_t b = 0; b < W; ++b)
                if (ctrl_[pos + b] == tag) {
                    size_t idx = (pos + b) & bucket_mask_;
                    if (keys_[idx] == key && is_full(ctrl_[idx])) return idx;
                }
            for (size_t b = 0; b < W; ++b)
                if (ctrl_[pos + b] == EMPTY) return NPOS;
            stride += W;
            pos = (pos + stride) & bucket_mask_;
        }
    }
    bool contains(uint32_t key) const { return find(key) != NPOS; }
    V *get(uint32_t key) {
        size_t i = find(key);
        return i == NPOS ? nullptr : &vals_[i];
    }
    const V *get(uint32_t key) const {
        size_t i = find(key);
        return i == NPOS ? nullptr : &vals_[i];
    }

    // ---- HashMap::insert (hashbrown 0.12 map.rs insert -> RawTable::insert) ----------
    // returns true if the key was newly inserted
    bool insert(uint32_t key, V val) {
        size_t i = find(key);
        if (i != NPOS) {
            vals_[i] = std::move(val);
            return false;
        }
        uint64_t h = fxhash_u32(key);
        size_t slot = find_insert_slot(h);
        uint8_t old = ctrl_[slot];
        if (growth_left_ == 0 && (old & 1)) {
            reserve_rehash(1);
            slot = find_insert_slot(h);
        }
        record_insert(slot, old, h, key, std::move(val));
        return true;
    }

    // ---- Entry API for a vacant key (rustc_entry: reserve(1) then insert_no_grow) -----
    // caller has checked !contains(key)
    V &entry_insert_vacant(uint32_t key, V val) {
        reserve(1);
        uint64_t h = fxhash_u32(key);
        size_t slot = find_insert_slot(h);
        uint8_t old = ctrl_[slot];
        record_insert(slot, old, h, key, std::move(val));
        return vals_[slot];
    }

    void reserve(size_t additional) {
        if (additional > growth_left_) reserve_rehash(additional);
    }

    // ---- remove / retain ---------------------------------------------------------------
    bool remove(uint32_t key, V *out = nullptr) {
        size_t i = find(key);
        if (i == NPOS) return false;
        if (out) *out = std::move(vals_[i]);
        erase(i);
        return true;
    }
    template <class F> void retain(F keep) {
        if (singleton_) return;
        for (size_t i = 0; i < buckets(); ++i)
            if (is_full(ctrl_[i]) && !keep(keys_[i], vals_[i])) erase(i);
    }
    void clear() {
        if (singleton_) return;
        // clear_no_drop: all ctrl EMPTY, items 0, growth_left = capacity
        ctrl_.assign(ctrl_.size(), EMPTY);
        items_ = 0;
        growth_left_ = mask_to_cap(bucket_mask_);
    }

    // ---- iteration: ascending bucket index ------------------------------------------
    template <class F> void for_each(F f) const {
        if (singleton_) return;
        for (size_t i = 0; i < buckets(); ++i)
            if (is_full(ctrl_[i])) f(keys_[i], vals_[i]);
    }
    template <class F> void for_each_mut(F f) {
        if (singleton_) return;
        for (size_t i = 0; i < buckets(); ++i)
            if (is_full(ctrl_[i])) f(keys_[i], vals_[i]);
    }
    std::vector<uint32_t> keys() const {
        std::vector<uint32_t> out;
        out.reserve(items_);
        for_each([&](uint32_t k, const V &) { out.push_back(k); });
        return out;
    }

    // FromIterator of one element: with_hasher(default) + extend -> reserve(1) + insert
    static FxMap from_one(uint32_t key, V val) {
        FxMap m;
        m.reserve(1);
        m.insert(key, std::move(val));
        return m;
    }
    // HashMap::extend reservation rule (map.rs Extend impl)
    void extend_reserve(size_t hint) { reserve(is_empty() ? hint : (hint + 1) / 2); }

  private:
    size_t bucket_mask_ = 0, growth_left_ = 0, items_ = 0;
    bool singleton_ = true;
    std::vector<uint8_t> ctrl_;
    std::vector<uint32_t> keys_;
    std::vector<V> vals_;

    static bool is_full(uint8_t c) { return (c & 0x80) == 0; }
    static uint8_t h2(uint64_t h) { return (uint8_t)(h >> 57); }
    static size_t mask_to_cap(size_t mask) { return mask < 8 ? mask : ((mask + 1) / 8) * 7; }
    static size_t cap_to_buckets(size_t cap) {
        if (cap < 8) return cap < 4 ? 4 : 8;
        size_t adj = cap * 8 / 7, n = 1;
        while (n < adj) n <<= 1;
        return n;
    }
    void set_ctrl(size_t i, uint8_t c) {
        size_t i2 = ((i - W) & bucket_mask_) + W;
        ctrl_[i] = c;
        ctrl_[i2] = c;
    }
    size_t find_insert_slot(uint64_t h) const {
        size_t pos = (size_t)h & bucket_mask_, stride = 0;
        for (;;) {
            for (size_t b = 0; b < W; ++b)
                if (ctrl_[pos + b] & 0x80) {
                    size_t r = (pos + b) & bucket_mask_;
                    if (is_full(ctrl_[r])) {
                        // table smaller than a group: the hit was a trailing/mirror byte
                        for (size_t c = 0; c < W; ++c)
                            if (ctrl_[c] & 0x80) return c;
                    }
                    return r;
                }
            stride += W;
            pos = (pos + stride) & bucket_mask_;
        }
    }
    void record_insert(size_t slot, uint8_t old, uint64_t h, uint32_t key, V val) {
        growth_left_ -= (old & 1);
        set_ctrl(slot, h2(h));
        keys_[slot] = key;
        vals_[slot] = std::move(val);
        items_ += 1;
    }
    void erase(size_t index) {
        size_t before = (index - W) & bucket_mask_;
        // empty_before.leading_zeros(): non-EMPTY bytes at the top of the group before
        size_t lz = 0;
        for (size_t b = W; b-- > 0;) {
            if (ctrl_[before + b] == EMPTY) break;
            ++lz;
        }
        size_t tz = 0;
        for (size_t b = 0; b < W; ++b) {
            if (ctrl_[index + b] == EMPTY) break;
            ++tz;
        }
        uint8_t c;
        if (lz + tz >= W) {
            c = DELETED;
        } else {
            growth_left_ += 1;
            c = EMPTY;
        }
        set_ctrl(index, c);
        items_ -= 1;
    }
    void reserve_rehash(size_t additional) {
        size_t new_items = items_ + additional;
        size_t full_cap = singleton_ ? 0 : mask_to_cap(bucket_mask_);
        if (new_items <= full_cap / 2)
            rehash_in_place();
        else
            resize(new_items > full_cap + 1 ? new_items : full_cap + 1);
    }
    void resize(size_t capacity) {
        size_t nb = cap_to_buckets(capacity);
        FxMap n;
        n.singleton_ = false;
        n.bucket_mask_ = nb - 1;
        n.ctrl_.assign(nb + W, EMPTY);
        n.keys_.assign(nb, 0);
        n.vals_.assign(nb, V());
        if (!singleton_) {
            for (size_t i = 0; i < buckets(); ++i) {
                if (!is_full(ctrl_[i])) continue;
                uint64_t h = fxhash_u32(keys_[i]);
                size_t slot = n.find_insert_slot(h);
                n.set_ctrl(slot, h2(h));
                n.keys_[slot] = keys_[i];
                n.vals_[slot] = std::move(vals_[i]);
            }
        }
        n.items_ = items_;
        n.growth_left_ = mask_to_cap(n.bucket_mask_) - items_;
        *this = std::move(n);
    }
    void rehash_in_place() {
        size_t nb = buckets();
        // prepare_rehash_in_place: FULL -> DELETED, DELETED -> EMPTY (whole groups)
        for (size_t i = 0; i < ctrl_.size(); ++i) ctrl_[i] = is_full(ctrl_[i]) ? DELETED : EMPTY;
        if (nb < W) {
            for (size_t i = 0; i < nb; ++i) ctrl_[W + i] = ctrl_[i];
            for (size_t i = nb; i < W; ++i) ctrl_[i] = EMPTY;
        } else {
            for (size_t i = 0; i < W; ++i) ctrl_[nb + i] = ctrl_[i];
        }
        for (size_t i = 0; i < nb; ++i) {
            if (ctrl_[i] != DELETED) continue;
            for (;;) {
                uint64_t h = fxhash_u32(keys_[i]);
                size_t new_i = find_insert_slot(h);
                size_t start = (size_t)h & bucket_mask_;
                size_t pi = ((i - start) & bucket_mask_) / W;
                size_t pn = ((new_i - start) & bucket_mask_) / W;
                if (pi == pn) {
                    set_ctrl(i, h2(h));
                    break;
                }
                uint8_t prev = ctrl_[new_i];
                set_ctrl(new_i, h2(h));
                if (prev == EMPTY) {
                    set_ctrl(i, EMPTY);
                    keys_[new_i] = keys_[i];
                    vals_[new_i] = std::move(vals_[i]);
                    break;
                }
                // prev == DELETED: swap and keep processing the element now at i
                std::swap(keys_[i], keys_[new_i]);
                std::swap(vals_[i], vals_[new_i]);
            }
        }
        growth_left_ = mask_to_cap(bucket_mask_) - items_;
    }
};

struct Unit {};
using FxSet = FxMap<Unit>;

} // namespace hb
