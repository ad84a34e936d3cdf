// Host driver of the MI355X NextPolish2 hot path: C ABI (include/np2.h), HBM residency,
// per-pass kernel orchestration and the host-resident region logic (vote / seed / splice /
// recheck control and the Louvain phasing vote).  Replaces the per-contig loop
// src/main.rs:1819-1836 of the reference.  No CPU fallback: every entry point needs a HIP device.
#include "../../include/np2.h"
#include "np2_common.hpp"
#include "np2_kernels.hpp"
#include "np2_phase_host.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

using namespace np2;

namespace {

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

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    T *ensure(size_t n) {
        if (n > cap) {
            release();
            size_t want = n + n / 8 + 64;
            HIPCHK(hipMalloc((void **)&p, want * sizeof(T)));
            cap = want;
        }
        return p;
    }
};

struct YakTable {
    uint32_t k = 0, cap_log2 = 0;
    DevBuf<uint64_t> table;
    YakDev dev() const { return YakDev{table.p, cap_log2, k}; }
};

struct Timing {
    std::vector<std::string> names;
    std::vector<float> ms;
    std::string joined;
};

} // namespace

struct np2_contig {
    uint32_t L = 0, R = 0;
    uint64_t nib_bytes = 0, n_cols = 0, n_ckpt = 0;
    DevBuf<np2_read_t> reads;
    DevBuf<uint8_t> nib;
    DevBuf<uint8_t> refnib; // nibble-packed contig codes (+ padding), also viewed as uint64_t words
    DevBuf<uint64_t> ck_off;
    DevBuf<uint32_t> ckpt;
};

struct np2_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<YakTable> yaks;
    std::string err;
    bool trace = false;
    std::map<std::string, std::vector<uint8_t>> trace_items;
    Timing timing;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending_events;

    // scratch (reused across contigs)
    DevBuf<uint8_t> tmp;
    DevBuf<uint64_t> keys_raw, keys;
    DevBuf<uint32_t> vals_raw, vals, shard_cnt, gcount, gmin, flag, idx;
    DevBuf<uint64_t> shard_off;
    DevBuf<uint32_t> npos, ncount, nminr, nbesti, node_cnt, node_off, run_start, run_end, n0_besti, emit, eoff;
    DevBuf<uint16_t> nbases, ndelta;
    DevBuf<int64_t> nscore;
    DevBuf<int32_t> covd, cov, mval, smin;
    DevBuf<uint8_t> alive, cns_base, cns_cls, lq_kind, lq_nothead;
    DevBuf<uint32_t> cns_pos, lq_next, rflag, rstart, rend, ridx, raw_start, raw_end, headflag, hidx, lq_start,
        lq_end;
    DevBuf<uint32_t> pj, pcount, poff, pair_region, pair_read, pair_region_s, pair_read_s, reg_npairs, reg_poff,
        pair_len, pair_keep, keepflag, cand_idx, seq_off, reg_ncand, cand_off, cand_order, cand_seq_off, kill_ids;
    DevBuf<uint64_t> cand_kmer;
    DevBuf<uint8_t> cand_seq;
    DevBuf<uint16_t> kscore;
    DevBuf<uint32_t> scal; // device scalars: see enum below
    DevBuf<uint8_t> sstr;
    DevBuf<uint64_t> soff;
    DevBuf<uint16_t> sscore;
};

namespace {

enum Scal { S_ERR = 0, S_NNODES, S_NRUNS, S_BEST, S_PATHBEGIN, S_NRAW, S_NREG, S_DUP, S_LAST0, S_LAST1, S_GAIN0,
            S_GAIN1, S_COUNT = 16 };

struct EventTimer {
    np2_ctx *cx;
    hipEvent_t a, b;
    EventTimer(np2_ctx *c, const char *name) : cx(c) {
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, cx->stream);
        cx->pending_events.push_back({name, {a, b}});
    }
    ~EventTimer() { (void)hipEventRecord(b, cx->stream); }
};

void flush_timings(np2_ctx *cx) {
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

template <class T> std::vector<T> d2h(np2_ctx *cx, const T *d, size_t n) {
    std::vector<T> v(n);
    if (n) {
        HIPCHK(hipMemcpyAsync(v.data(), d, n * sizeof(T), hipMemcpyDeviceToHost, cx->stream));
        HIPCHK(hipStreamSynchronize(cx->stream));
    }
    return v;
}
template <class T> void trace_put(np2_ctx *cx, int pass, const std::string &name, const std::vector<T> &v) {
    if (!cx->trace) return;
    auto &dst = cx->trace_items[std::to_string(pass) + ":" + name];
    dst.resize(v.size() * sizeof(T));
    if (!v.empty()) memcpy(dst.data(), v.data(), dst.size());
}

uint32_t exclusive_total(np2_ctx *cx, const uint32_t *in, uint32_t *out, size_t n_plus1) {
    // scans n_plus1 elements (caller guarantees in[n_plus1-1] == 0); returns out[n_plus1-1] lazily on device
    int rc = prim_exclusive_sum_u32(cx->stream, cx->tmp.p, cx->tmp.cap, in, out, n_plus1);
    if (rc) throw Np2Error(NP2_E_DEVICE, "rocprim exclusive_scan failed");
    return 0;
}
void zero32(np2_ctx *cx, void *p, size_t n_elems, size_t elem = 4) {
    if (n_elems) HIPCHK(hipMemsetAsync(p, 0, n_elems * elem, cx->stream));
}

// ------------------------------------------------------------------------------------------
// host-side region logic (operates on the GPU-built candidate tables)
// ------------------------------------------------------------------------------------------
static const uint8_t LB_TEMP = 0x01, LB_SUCC = 0x80, LB_HETE = 0x40, LB_RECH = 0x20; // main.rs:655-658

struct Cand {
    uint32_t order;
    uint16_t kscore;
    uint32_t so, len; // sequence = pool[so, so+len)
};
struct Region {
    uint32_t start, end;
    uint8_t lable = 0;
    std::string sudoseed;
    std::vector<Cand> seqs;
};
struct Cns {
    std::vector<uint32_t> pos;
    std::vector<uint8_t> base;
    size_t size() const { return pos.size(); }
};
struct RegionSet {
    std::vector<Region> regs;
    std::vector<uint8_t> pool;
    bool same(const Cand &a, const Cand &b) const {
        return a.len == b.len && memcmp(pool.data() + a.so, pool.data() + b.so, a.len) == 0;
    }
    std::string str(const Cand &c) const { return std::string((const char *)pool.data() + c.so, c.len); }
};

struct GroupStat { // fill_order_stat, main.rs:813-849
    size_t stats[LQSEQ_MAX_CAN_COUNT];
    std::vector<std::pair<uint32_t, size_t>> by_order; // HashMap<u32, usize>: order -> count
    size_t max1_c = 0, max1_p = 0, max2_c = 0, max2_p = 0;
    size_t *find(uint32_t order) {
        for (auto &e : by_order)
            if (e.first == order) return &e.second;
        return nullptr;
    }
    size_t get_or0(uint32_t order) {
        size_t *p = find(order);
        return p ? *p : 0;
    }
    void set(uint32_t order, size_t v) {
        size_t *p = find(order);
        if (p)
            *p = v;
        else
            by_order.emplace_back(order, v);
    }
};

void group_stats(const RegionSet &rs, const Region &rg, GroupStat &g) {
    g.max1_c = g.max1_p = g.max2_c = g.max2_p = 0;
    std::fill(g.stats, g.stats + LQSEQ_MAX_CAN_COUNT, 0);
    g.by_order.clear();
    const size_t n = rg.seqs.size();
    for (size_t a = 0; a < n; ++a) {
        if (rg.seqs[a].kscore == 0 || g.stats[a] > 0) continue;
        size_t c = 0;
        for (size_t b = a; b < n; ++b) c += rs.same(rg.seqs[b], rg.seqs[a]);
        g.set(rg.seqs[a].order, c);
        for (size_t b = a; b < n; ++b)
            if (rs.same(rg.seqs[b], rg.seqs[a])) g.stats[b] = c;
        if (c > g.max1_c || (c == g.max1_c && rg.seqs[a].order == 0)) {
            g.max2_c = g.max1_c, g.max2_p = g.max1_p;
            g.max1_c = c, g.max1_p = a;
        } else if (g.max1_p == g.max2_p || c > g.max2_c) {
            g.max2_c = c, g.max2_p = a;
        }
    }
}
inline size_t min_support(size_t n) { return n >= 9 ? 3 : (n >= 6 ? 2 : 1); } // get_min_count, main.rs:803-811

bool differs_after_hp_compression(const uint8_t *a, size_t na, const uint8_t *b, size_t nb) { // is_valid_snp
    size_t i = 0, j = 0;
    while (i < na && j < nb) {
        if (a[i] != b[j]) return true;
        while (i + 1 < na && a[i] == a[i + 1]) ++i;
        while (j + 1 < nb && b[j] == b[j + 1]) ++j;
        ++i, ++j;
    }
    return false;
}

// phasing pass: mark_hete_lqseqs (main.rs:916-946) + phase_reads_by_lqseqs (948-1015)
std::vector<uint32_t> phasing_vote(RegionSet &rs, bool asref, bool use_all_reads) {
    GroupStat g;
    for (Region &rg : rs.regs) {
        group_stats(rs, rg, g);
        const size_t min_c = min_support(rg.seqs.size());
        if (g.max2_c < min_c) continue;
        const Cand &c1 = rg.seqs[g.max1_p], &c2 = rg.seqs[g.max2_p];
        if (!(c1.len == c2.len || (rg.seqs.size() >= 6 && g.max2_c >= g.max1_c / 2))) continue;
        if (!differs_after_hp_compression(rs.pool.data() + c1.so, c1.len, rs.pool.data() + c2.so, c2.len)) continue;
        rg.lable |= LB_HETE;
        for (size_t p = 0; p < rg.seqs.size(); ++p)
            if (rg.seqs[p].kscore > 0 && g.stats[p] < min_c) rg.seqs[p].kscore = 0;
    }
    phase::Weights data, dif, ref_data;
    std::unordered_set<uint32_t> bad;
    for (const Region &rg : rs.regs) {
        if (!(rg.lable & LB_HETE)) continue;
        for (size_t i = 0; i < rg.seqs.size(); ++i) {
            const Cand &a = rg.seqs[i];
            if (a.kscore == 0) continue;
            for (size_t j = i + 1; j < rg.seqs.size(); ++j) {
                const Cand &b = rg.seqs[j];
                if (b.kscore == 0) continue;
                const float w = rs.same(a, b) ? 1.f : -1.f;
                if (a.order == 0) {
                    if (asref) phase::add_weight(ref_data, a.order, b.order, w);
                    if (w < 0.f && !use_all_reads) bad.insert(b.order);
                    continue;
                }
                REFPANIC_IF(b.order == 0, "seq2 order is equal to 0");
                if (w == -1.f) {
                    phase::add_weight(dif, a.order, b.order, -1.f);
                    phase::add_weight(dif, b.order, a.order, -1.f);
                }
                phase::add_weight(data, a.order, b.order, w);
                phase::add_weight(data, b.order, a.order, w);
            }
        }
    }
    dif.each([&](uint32_t n1, const phase::Row &row) {
        for (const auto &e : row)
            if (e.second <= -3.f) phase::set_weight(data, n1, e.first, e.second);
    });
    if (!use_all_reads) {
        data.keep_if([&](uint32_t k, phase::Row &) { return bad.count(k) == 0; });
        data.each_mut([&](uint32_t, phase::Row &row) {
            for (auto it = row.begin(); it != row.end();) it = bad.count(it->first) ? row.erase(it) : std::next(it);
        });
    }
    phase::Row ref_row;
    bool have_ref = false;
    ref_data.each([&](uint32_t, const phase::Row &row) {
        if (!have_ref) ref_row = row, have_ref = true;
    });
    std::vector<uint32_t> losers;
    if (!phase::losing_reads(std::move(data), have_ref ? &ref_row : nullptr, losers))
        throw Np2Error(NP2_E_REFPANIC,
                       "reference would panic: the weight of two conflicting community is not less than 0");
    for (uint32_t b : bad) losers.push_back(b);
    std::sort(losers.begin(), losers.end());
    losers.erase(std::unique(losers.begin(), losers.end()), losers.end());
    return losers;
}

// final pass part 1: fill_seed_lqseqs (main.rs:862-914) with retain_sort_seqs (714-726)
void choose_seeds(RegionSet &rs, long max_indel_len) {
    GroupStat g;
    for (Region &rg : rs.regs) {
        group_stats(rs, rg, g);
        REFPANIC_IF(rg.seqs.empty(), "index out of bounds: lqseq.seqs[max1_p]");
        rg.sudoseed = rs.str(rg.seqs[g.max1_p]);
        rg.lable |= LB_SUCC | LB_RECH;
        const size_t min_c = min_support(rg.seqs.size());
        REFPANIC_IF(rg.seqs[0].order != 0, "the first lqseq is not ref.");
        if (size_t *v = g.find(0)) {
            if (*v > 1 && *v < min_c) *v = min_c;
        } else {
            size_t c = 0;
            for (const Cand &x : rg.seqs) c += rs.same(x, rg.seqs[0]);
            if (c > 1) g.set(0, min_c);
        }
        bool nodup = true; // no_dupseq_lqseq (main.rs:851-860): evaluated lazily below
        auto no_dup = [&]() {
            for (size_t a = 1; a < rg.seqs.size(); ++a)
                for (size_t b = a + 1; b < rg.seqs.size(); ++b)
                    if (rs.same(rg.seqs[a], rg.seqs[b])) return false;
            return true;
        };
        (void)nodup;
        if (g.max1_p != 0 && g.max1_c < min_c && (g.max1_c > 1 || no_dup())) {
            size_t *v = g.find(rg.seqs[g.max1_p].order);
            REFPANIC_IF(!v, "unwrap on None: order_stat.get_mut");
            *v = min_c;
            g.set(0, min_c);
        } else if (g.max1_c < min_c) {
            g.set(0, min_c);
        }
        // retain_sort_seqs: stable sort by group count descending, cut below min_c
        std::stable_sort(rg.seqs.begin(), rg.seqs.end(),
                         [&](const Cand &a, const Cand &b) { return g.get_or0(a.order) > g.get_or0(b.order); });
        size_t keep = 0;
        while (keep < rg.seqs.size() && g.get_or0(rg.seqs[keep].order) >= min_c) ++keep;
        rg.seqs.resize(keep);
        REFPANIC_IF(rg.seqs.empty(), "index out of bounds: lqseq.seqs[0] after retain_sort_seqs");
        long d = (long)rg.sudoseed.size() - (long)rg.seqs[0].len;
        const bool too_long = (d < 0 ? -d : d) > max_indel_len;
        if (rg.seqs.size() <= 1 || too_long) {
            rg.sudoseed = rs.str(rg.seqs[0]);
            rg.lable ^= LB_RECH;
            rg.seqs.clear();
        }
    }
}

// update_consensus_with_lqseqs (main.rs:1027-1058) incl. the wrapping cursor of 1017-1025
Cns splice(const std::vector<Region> &regs, const Cns &in, uint8_t lable) {
    Cns out;
    out.pos.reserve(in.size());
    out.base.reserve(in.size());
    auto next_with = [&](size_t i) {
        i -= 1;
        while (i < regs.size() && !(regs[i].lable & lable)) i -= 1;
        return i;
    };
    size_t i = 0, li = next_with(regs.size());
    while (i < in.size()) {
        const uint32_t p = in.pos[i];
        if (li < regs.size() && p == regs[li].start) {
            for (char b : regs[li].sudoseed) {
                out.pos.push_back(p);
                out.base.push_back((uint8_t)b);
            }
            while (i < in.size() && in.pos[i] <= regs[li].end) ++i;
            li = next_with(li);
        } else {
            out.pos.push_back(p);
            out.base.push_back(in.base[i]);
            ++i;
        }
    }
    return out;
}

// cursor helpers of reupdate_consensus_with_lqseqs (main.rs:1068-1139); out-of-range indexing
// is a panic in the reference
struct CnsCursor {
    const Cns &c;
    size_t idx = 0;
    explicit CnsCursor(const Cns &cns) : c(cns) {}
    uint32_t pos(size_t i) const {
        REFPANIC_IF(i >= c.size(), "index out of bounds: consensus[i] in reupdate");
        return c.pos[i];
    }
    void left_flank(uint32_t p, size_t l, size_t &si, size_t &ei) {
        size_t i = idx;
        while (pos(i) >= p) i -= 1;
        while (pos(i) < p) i += 1;
        REFPANIC_IF(!(pos(i) >= p && pos(i - 1) < p), "assert iter_consensus_extend (left)");
        idx = i, ei = i, si = i > l ? i - l : 0;
    }
    void right_flank(uint32_t p, size_t l, size_t &si, size_t &ei) {
        size_t i = idx;
        while (pos(i) <= p) i += 1;
        while (pos(i) > p) i -= 1;
        REFPANIC_IF(!(pos(i) <= p && pos(i + 1) > p), "assert iter_consensus_extend (right)");
        idx = i, si = i + 1, ei = (i + l < c.size()) ? i + l + 1 : c.size();
    }
    void between(uint32_t s, uint32_t e, size_t &si, size_t &ei) {
        size_t i = idx;
        while (pos(i) <= s) i += 1;
        while (pos(i) > s) i -= 1;
        i += 1;
        REFPANIC_IF(!(pos(i) > s && pos(i - 1) <= s), "assert iter_consensus_region (1)");
        si = i;
        while (pos(i) >= e) i -= 1;
        while (pos(i) < e) i += 1;
        i -= 1;
        REFPANIC_IF(!(pos(i) < e && pos(i + 1) >= e), "assert iter_consensus_region (2)");
        idx = i, ei = i + 1;
    }
};

void gpu_score_strings(np2_ctx *cx, int yak_idx, const std::vector<uint8_t> &blob, const std::vector<uint64_t> &off,
                       uint16_t min_kmer_count, std::vector<uint16_t> &scores) {
    const size_t n = off.size() - 1;
    scores.assign(n, 0);
    if (!n) return;
    cx->sstr.ensure(blob.size() + 16);
    cx->soff.ensure(off.size());
    cx->sscore.ensure(n);
    HIPCHK(hipMemcpyAsync(cx->sstr.p, blob.data(), blob.size(), hipMemcpyHostToDevice, cx->stream));
    HIPCHK(hipMemcpyAsync(cx->soff.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, cx->stream));
    {
        EventTimer t(cx, "score_strings");
        launch_score_strings(cx->stream, cx->yaks[yak_idx].dev(), cx->sstr.p, cx->soff.p, n, min_kmer_count,
                             cx->sscore.p);
    }
    HIPCHK(hipMemcpyAsync(scores.data(), cx->sscore.p, n * 2, hipMemcpyDeviceToHost, cx->stream));
    HIPCHK(hipStreamSynchronize(cx->stream));
}

// reupdate_consensus_with_lqseqs (main.rs:1060-1420): strings are assembled on the host, scored
// in one batch by the k-mer kernel, then the selection rules are applied.
Cns recheck(np2_ctx *cx, RegionSet &rs, const Cns &cns, int yak_idx, uint16_t min_kmer_count, size_t iter_count) {
    const uint32_t ksize = cx->yaks[yak_idx].k;
    std::vector<size_t> rech;
    for (size_t i = rs.regs.size(); i-- > 0;)
        if (rs.regs[i].lable & LB_RECH) rech.push_back(i);

    struct Group {
        size_t sj, ej;
        size_t first_job;
        std::vector<uint32_t> lens; // product radix (chains only)
    };
    std::vector<Group> groups;
    std::vector<uint8_t> blob;
    std::vector<uint64_t> off(1, 0);
    CnsCursor cur(cns);
    auto add_cns = [&](size_t si, size_t ei) {
        REFPANIC_IF(si > ei || ei > cns.size(), "slice index out of range in reupdate");
        blob.insert(blob.end(), cns.base.begin() + (long)si, cns.base.begin() + (long)ei);
    };
    auto add_seq = [&](const Cand &c) { blob.insert(blob.end(), rs.pool.begin() + c.so, rs.pool.begin() + c.so + c.len); };
    size_t sj = 0;
    while (sj < rech.size()) {
        size_t ej = sj + 1;
        while (ej < rech.size() && rs.regs[rech[ej]].start < rs.regs[rech[ej - 1]].end + ksize) {
            ej += 1;
            if (ej > sj + 5) break; // at most 6 chained regions (main.rs:1202-1205)
        }
        size_t sl, el, sr, er;
        cur.left_flank(rs.regs[rech[sj]].start, ksize - 1, sl, el);
        cur.right_flank(rs.regs[rech[ej - 1]].end, ksize - 1, sr, er);
        Group g{sj, ej, off.size() - 1, {}};
        if (ej == sj + 1) {
            for (const Cand &c : rs.regs[rech[sj]].seqs) {
                add_cns(sl, el);
                add_seq(c);
                add_cns(sr, er);
                off.push_back(blob.size());
            }
        } else {
            const size_t n = ej - sj;
            std::vector<size_t> pick(n, 0);
            bool done = false;
            for (size_t x = 0; x < n; ++x) {
                g.lens.push_back((uint32_t)rs.regs[rech[sj + x]].seqs.size());
                done |= g.lens.back() == 0;
            }
            while (!done) { // multi_cartesian_product: last iterator fastest
                add_cns(sl, el);
                for (size_t x = 0; x < n; ++x) {
                    add_seq(rs.regs[rech[sj + x]].seqs[pick[x]]);
                    if (x + 1 < n) {
                        const uint32_t s = rs.regs[rech[sj + x]].end, e = rs.regs[rech[sj + x + 1]].start;
                        if (s + 1 != e) {
                            size_t si, ei;
                            cur.between(s, e, si, ei);
                            add_cns(si, ei);
                        }
                    } else {
                        add_cns(sr, er);
                    }
                }
                off.push_back(blob.size());
                size_t d = n;
                while (d-- > 0) {
                    if (++pick[d] < g.lens[d]) break;
                    pick[d] = 0;
                    if (d == 0) done = true;
                }
            }
        }
        groups.push_back(std::move(g));
        sj = ej;
    }

    std::vector<uint16_t> scores;
    blob.resize(blob.size() + 8, 0);
    gpu_score_strings(cx, yak_idx, blob, off, min_kmer_count, scores);

    for (const Group &g : groups) {
        size_t job = g.first_job;
        if (g.ej == g.sj + 1) {
            for (Cand &c : rs.regs[rech[g.sj]].seqs) c.kscore = scores[job++];
        } else {
            const size_t n = g.ej - g.sj;
            for (size_t x = 0; x < n; ++x)
                for (Cand &c : rs.regs[rech[g.sj + x]].seqs) c.kscore = 0;
            size_t total = 1;
            for (uint32_t l : g.lens) total *= l;
            std::vector<size_t> pick(n, 0);
            for (size_t t = 0; t < total; ++t) { // later products overwrite earlier ones (main.rs:1364-1366)
                const uint16_t ks = scores[job++];
                if (ks > 0)
                    for (size_t x = 0; x < n; ++x) rs.regs[rech[g.sj + x]].seqs[pick[x]].kscore = ks;
                size_t d = n;
                while (d-- > 0) {
                    if (++pick[d] < g.lens[d]) break;
                    pick[d] = 0;
                }
            }
        }
    }

    for (Region &rg : rs.regs) {
        if (!(rg.lable & LB_RECH)) continue;
        size_t c = 0, valid = 0;
        for (size_t p = 0; p < rg.seqs.size(); ++p)
            if (rg.seqs[p].kscore != 0) {
                if (c == 0 || rg.seqs[p].order == 0) c = p + 1;
                ++valid;
            }
        if (valid > 1) rg.lable |= LB_TEMP;
        if (c != 0) {
            rg.sudoseed = rs.str(rg.seqs[c - 1]);
        } else if (iter_count == 1) {
            size_t i = 0;
            for (size_t p = 0; p < rg.seqs.size(); ++p)
                if (rg.seqs[p].order == 0) {
                    i = p;
                    break;
                }
            REFPANIC_IF(rg.seqs.empty(), "index out of bounds: lqseq.seqs[i] in reupdate");
            rg.sudoseed = rs.str(rg.seqs[i]);
        }
    }
    Cns out = splice(rs.regs, cns, LB_RECH);
    for (Region &rg : rs.regs) {
        if (!(rg.lable & LB_RECH)) continue;
        rg.lable ^= (rg.lable & LB_TEMP) ? LB_TEMP : LB_RECH;
    }
    return out;
}

void trace_regions(np2_ctx *cx, int pass, const std::string &tag, const RegionSet &rs) {
    if (!cx->trace) return;
    std::vector<uint32_t> start, end, cand_off(1, 0), order, seq_off(1, 0), sudo_off(1, 0);
    std::vector<uint16_t> kscore;
    std::vector<uint8_t> lable, seqs, sudo;
    for (const Region &rg : rs.regs) {
        start.push_back(rg.start);
        end.push_back(rg.end);
        lable.push_back(rg.lable);
        sudo.insert(sudo.end(), rg.sudoseed.begin(), rg.sudoseed.end());
        sudo_off.push_back((uint32_t)sudo.size());
        for (const Cand &c : rg.seqs) {
            order.push_back(c.order);
            kscore.push_back(c.kscore);
            seqs.insert(seqs.end(), rs.pool.begin() + c.so, rs.pool.begin() + c.so + c.len);
            seq_off.push_back((uint32_t)seqs.size());
        }
        cand_off.push_back((uint32_t)order.size());
    }
    trace_put(cx, pass, tag + ".start", start);
    trace_put(cx, pass, tag + ".end", end);
    trace_put(cx, pass, tag + ".lable", lable);
    trace_put(cx, pass, tag + ".sudo_off", sudo_off);
    trace_put(cx, pass, tag + ".sudo", sudo);
    trace_put(cx, pass, tag + ".cand_off", cand_off);
    trace_put(cx, pass, tag + ".order", order);
    trace_put(cx, pass, tag + ".kscore", kscore);
    trace_put(cx, pass, tag + ".seq_off", seq_off);
    trace_put(cx, pass, tag + ".seq", seqs);
}
void trace_cns(np2_ctx *cx, int pass, const std::string &tag, const Cns &c) {
    trace_put(cx, pass, tag + ".pos", c.pos);
    trace_put(cx, pass, tag + ".base", c.base);
}

// ------------------------------------------------------------------------------------------
// the per-contig pipeline
// ------------------------------------------------------------------------------------------
struct PassOut {
    bool has_regions = false;
    Cns cns; // raw consensus of this pass (host copy, only when needed)
};

void run_diff(np2_ctx *cx, np2_contig *c, uint32_t &T) {
    hipStream_t s = cx->stream;
    const uint32_t L = c->L, R = c->R;
    uint64_t cap_total = std::max<uint64_t>(c->n_cols / 24 + (uint64_t)R * 8 + 65536, 1u << 20);
    for (int attempt = 0; attempt < 3; ++attempt) {
        uint32_t shard_cap = (uint32_t)((cap_total + NSHARD - 1) / NSHARD);
        uint64_t cap = (uint64_t)shard_cap * NSHARD;
        cx->keys_raw.ensure(cap);
        cx->vals_raw.ensure(cap);
        cx->shard_cnt.ensure(NSHARD * SHARD_STRIDE);
        zero32(cx, cx->shard_cnt.p, NSHARD * SHARD_STRIDE);
        zero32(cx, cx->scal.p, S_COUNT);
        {
            EventTimer t(cx, "diff_reads");
            launch_diff_reads(s, c->reads.p, R, c->nib.p, (const uint64_t *)c->refnib.p, c->refnib.p, L,
                              cx->keys_raw.p, cx->vals_raw.p, cx->shard_cnt.p, shard_cap, c->ck_off.p, c->ckpt.p,
                              cx->scal.p + S_ERR);
        }
        std::vector<uint32_t> cnt = d2h(cx, cx->shard_cnt.p, (size_t)NSHARD * SHARD_STRIDE);
        std::vector<uint32_t> sc = d2h(cx, cx->scal.p, S_COUNT);
        if (sc[S_ERR] & 2u)
            throw Np2Error(NP2_E_ARG, "packed read inconsistent with its descriptor (n_cols / aln_t_e / terminator)");
        uint32_t mx = 0;
        uint64_t total = 0;
        std::vector<uint64_t> off(NSHARD + 1, 0);
        for (int i = 0; i < NSHARD; ++i) {
            mx = std::max(mx, cnt[(size_t)i * SHARD_STRIDE]);
            off[i + 1] = off[i] + cnt[(size_t)i * SHARD_STRIDE];
        }
        total = off[NSHARD];
        if (mx > shard_cap) { // a shard overflowed: grow and redo the dense pass
            cap_total = (uint64_t)mx * NSHARD * 5 / 4 + 65536;
            continue;
        }
        if (total >= 0xFFFFFFF0ull) throw Np2Error(NP2_E_NOMEM, "too many exception nodes");
        T = (uint32_t)total;
        cx->keys.ensure(T + 1);
        cx->vals.ensure(T + 1);
        cx->keys_raw.ensure(std::max<uint64_t>(cap, T + 1));
        cx->shard_off.ensure(NSHARD + 1);
        HIPCHK(hipMemcpyAsync(cx->shard_off.p, off.data(), (NSHARD + 1) * 8, hipMemcpyHostToDevice, s));
        cx->tmp.ensure(prim_temp_bytes(std::max<size_t>({(size_t)T + 1, (size_t)L + 2, (size_t)R + 1})));
        {
            EventTimer t(cx, "sort_exceptions");
            // compact into keys/vals, sort back into keys_raw/vals_raw, then swap roles
            launch_compact_shards(s, cx->keys_raw.p, cx->vals_raw.p, shard_cap, cx->shard_cnt.p, cx->shard_off.p,
                                  cx->keys.p, cx->vals.p);
            unsigned pos_bits = 1;
            while ((1ull << pos_bits) < (uint64_t)L + 1) ++pos_bits;
            int rc = prim_sort_pairs_u64_u32(s, cx->tmp.p, cx->tmp.cap, cx->keys.p, cx->keys_raw.p, cx->vals.p,
                                             cx->vals_raw.p, T, 32 + pos_bits);
            if (rc) throw Np2Error(NP2_E_DEVICE, "rocprim radix_sort_pairs failed");
        }
        HIPCHK(hipStreamSynchronize(s)); // shard_off host vector goes out of scope
        return;
    }
    throw Np2Error(NP2_E_NOMEM, "exception buffer kept overflowing");
}

// sorted exception tuples live in keys_raw / vals_raw after run_diff
void build_graph(np2_ctx *cx, np2_contig *c, uint32_t T, uint32_t &n_nodes, uint32_t &n_runs) {
    hipStream_t s = cx->stream;
    const uint32_t L = c->L, R = c->R;
    EventTimer t(cx, "build_graph");
    cx->gcount.ensure(T + 1);
    cx->gmin.ensure(T + 1);
    cx->flag.ensure(std::max<size_t>((size_t)T + 1, (size_t)L + 2));
    cx->idx.ensure(std::max<size_t>((size_t)T + 1, (size_t)L + 2));
    cx->npos.ensure(T + 1);
    cx->nbases.ensure(T + 1);
    cx->ndelta.ensure(T + 1);
    cx->ncount.ensure(T + 1);
    cx->nminr.ensure(T + 1);
    cx->nscore.ensure(T + 1);
    cx->nbesti.ensure(T + 1);
    cx->node_cnt.ensure(L + 2);
    cx->node_off.ensure(L + 2);
    cx->covd.ensure(L + 2);
    cx->cov.ensure(L + 2);
    cx->run_start.ensure(L + 2);
    cx->run_end.ensure(L + 2);
    zero32(cx, cx->node_cnt.p, L + 2);
    zero32(cx, cx->scal.p, S_COUNT);
    NodeArrays nd{cx->npos.p, cx->nbases.p, cx->ndelta.p, cx->ncount.p, cx->nminr.p};
    if (T) {
        launch_group_nodes(s, cx->keys_raw.p, cx->vals_raw.p, T, cx->alive.p, cx->gcount.p, cx->gmin.p, cx->flag.p);
        exclusive_total(cx, cx->flag.p, cx->idx.p, T);
        launch_scatter_nodes(s, cx->keys_raw.p, cx->gcount.p, cx->gmin.p, cx->idx.p, T, nd, cx->node_cnt.p,
                             cx->scal.p + S_NNODES);
    }
    exclusive_total(cx, cx->node_cnt.p, cx->node_off.p, (size_t)L + 1);
    launch_order_nodes(s, cx->node_off.p, L, nd);
    zero32(cx, cx->covd.p, L + 2);
    launch_cov_delta(s, c->reads.p, R, cx->alive.p, cx->covd.p);
    if (prim_inclusive_sum_i32(s, cx->tmp.p, cx->tmp.cap, cx->covd.p, cx->cov.p, (size_t)L + 1))
        throw Np2Error(NP2_E_DEVICE, "rocprim inclusive_scan failed");
    launch_mark_runs(s, cx->node_off.p, L, cx->flag.p);
    exclusive_total(cx, cx->flag.p, cx->idx.p, L);
    launch_scatter_idx(s, cx->flag.p, cx->idx.p, L, cx->run_start.p, cx->scal.p + S_NRUNS);
    std::vector<uint32_t> sc = d2h(cx, cx->scal.p, S_COUNT);
    n_nodes = sc[S_NNODES];
    n_runs = sc[S_NRUNS];
}

GraphPtrs graph_ptrs(np2_ctx *cx, np2_contig *c) {
    NodeArrays nd{cx->npos.p, cx->nbases.p, cx->ndelta.p, cx->ncount.p, cx->nminr.p};
    return GraphPtrs{c->refnib.p, cx->node_off.p, nd, cx->cov.p, c->L};
}

void trace_graph(np2_ctx *cx, np2_contig *c, int pass, uint32_t n_nodes) {
    if (!cx->trace) return;
    const uint32_t L = c->L;
    auto off = d2h(cx, cx->node_off.p, (size_t)L + 1);
    auto nb = d2h(cx, cx->nbases.p, n_nodes);
    auto ndl = d2h(cx, cx->ndelta.p, n_nodes);
    auto nc = d2h(cx, cx->ncount.p, n_nodes);
    auto cov = d2h(cx, cx->cov.p, (size_t)L);
    auto rn = d2h(cx, c->refnib.p, (size_t)(L + 1) / 2);
    auto code = [&](uint32_t p) -> unsigned { return (rn[p >> 1] >> (4 * (p & 1))) & 7; };
    std::vector<uint32_t> goff(L + 1, 0), gcount;
    std::vector<uint16_t> gbases, gdelta;
    for (uint32_t p = 0; p < L; ++p) {
        uint16_t b, d;
        if (p >= 2)
            b = (uint16_t)((code(p - 2) << 8) | (code(p - 1) << 4) | code(p)), d = 0;
        else if (p == 1)
            b = (uint16_t)(0x0F00 | (code(0) << 4) | code(1)), d = 1;
        else
            b = (uint16_t)(0x4FF0 | code(0)), d = 0;
        uint32_t e0 = 0;
        for (uint32_t i = off[p]; i < off[p + 1]; ++i)
            if (node_delta3(nb[i], ndl[i]) == 0) e0 += nc[i];
        gbases.push_back(b);
        gdelta.push_back(d);
        gcount.push_back((uint32_t)cov[p] - e0);
        for (uint32_t i = off[p]; i < off[p + 1]; ++i) {
            gbases.push_back(nb[i]);
            gdelta.push_back(ndl[i]);
            gcount.push_back(nc[i]);
        }
        goff[p + 1] = (uint32_t)gcount.size();
    }
    trace_put(cx, pass, "graph.off", goff);
    trace_put(cx, pass, "graph.bases", gbases);
    trace_put(cx, pass, "graph.delta", gdelta);
    trace_put(cx, pass, "graph.count", gcount);
}

// DP + backtrack + LQ regions; returns consensus length M and region count
void consensus_and_regions(np2_ctx *cx, np2_contig *c, uint32_t n_runs, uint32_t &M, uint32_t &n_reg) {
    hipStream_t s = cx->stream;
    const uint32_t L = c->L;
    GraphPtrs gp = graph_ptrs(cx, c);
    cx->n0_besti.ensure(L + 2);
    cx->emit.ensure(L + 2);
    cx->eoff.ensure(L + 2);
    {
        EventTimer t(cx, "dp_backtrack");
        zero32(cx, cx->n0_besti.p, L + 2);
        zero32(cx, cx->scal.p + S_BEST, S_COUNT - S_BEST); // best, path_begin, n_raw, n_reg, dup, last, gain
        launch_dp(s, gp, cx->run_start.p, cx->scal.p + S_NRUNS, n_runs, cx->nscore.p, cx->nbesti.p, cx->n0_besti.p,
                  cx->run_end.p, (int64_t *)(cx->scal.p + S_LAST0), (unsigned long long *)(cx->scal.p + S_GAIN0),
                  cx->scal.p + S_BEST);
        launch_bt_count(s, gp, cx->run_start.p, cx->run_end.p, cx->scal.p + S_NRUNS, n_runs, cx->nbesti.p,
                        cx->n0_besti.p, cx->scal.p + S_BEST, cx->emit.p, cx->scal.p + S_PATHBEGIN);
        zero32(cx, cx->emit.p + L, 1);
        exclusive_total(cx, cx->emit.p, cx->eoff.p, (size_t)L + 1);
    }
    std::vector<uint32_t> sc = d2h(cx, cx->scal.p, S_COUNT);
    if (sc[S_BEST] == 0xFFFFFFFFu)
        throw Np2Error(NP2_E_UNSUPPORTED,
                       "best path score is negative at the contig end (reference would emit its default node)");
    M = d2h(cx, cx->eoff.p + L, 1)[0];
    if (M == 0) throw Np2Error(NP2_E_REFPANIC, "reference would panic: empty consensus");
    cx->cns_pos.ensure(M + 2);
    cx->cns_base.ensure(M + 2);
    cx->cns_cls.ensure(M + 2);
    cx->lq_kind.ensure(M + 2);
    cx->lq_next.ensure(M + 2);
    cx->lq_nothead.ensure(M + 2);
    cx->rflag.ensure(M + 2);
    cx->rstart.ensure(M + 2);
    cx->rend.ensure(M + 2);
    cx->ridx.ensure(M + 2);
    cx->tmp.ensure(prim_temp_bytes((size_t)M + 2));
    {
        EventTimer t(cx, "dp_backtrack");
        launch_bt_write(s, gp, cx->run_start.p, cx->run_end.p, cx->scal.p + S_NRUNS, n_runs, cx->nbesti.p,
                        cx->n0_besti.p, cx->scal.p + S_BEST, cx->emit.p, cx->eoff.p, cx->cns_pos.p, cx->cns_base.p,
                        cx->cns_cls.p);
    }
    uint32_t n_raw = 0;
    {
        EventTimer t(cx, "lq_regions");
        zero32(cx, cx->lq_nothead.p, M + 2, 1);
        launch_lq_scan(s, cx->cns_pos.p, cx->cns_base.p, cx->cns_cls.p, M, cx->lq_kind.p, cx->lq_next.p,
                       cx->lq_nothead.p, cx->rflag.p, cx->rstart.p, cx->rend.p);
        exclusive_total(cx, cx->rflag.p, cx->ridx.p, M);
        cx->raw_start.ensure(M + 2);
        cx->raw_end.ensure(M + 2);
        launch_scatter_regions(s, cx->rflag.p, cx->ridx.p, cx->rstart.p, cx->rend.p, M, cx->raw_start.p,
                               cx->raw_end.p, cx->scal.p + S_NRAW);
        n_raw = d2h(cx, cx->scal.p + S_NRAW, 1)[0];
        n_reg = 0;
        if (n_raw) {
            cx->headflag.ensure(n_raw + 2);
            cx->hidx.ensure(n_raw + 2);
            cx->lq_start.ensure(n_raw + 2);
            cx->lq_end.ensure(n_raw + 2);
            launch_lq_merge_flag(s, cx->raw_start.p, cx->raw_end.p, cx->scal.p + S_NRAW, n_raw, cx->headflag.p);
            exclusive_total(cx, cx->headflag.p, cx->hidx.p, n_raw);
            launch_lq_merge_write(s, cx->raw_start.p, cx->raw_end.p, cx->scal.p + S_NRAW, n_raw, cx->headflag.p,
                                  cx->hidx.p, cx->lq_start.p, cx->lq_end.p, cx->scal.p + S_NREG);
            n_reg = d2h(cx, cx->scal.p + S_NREG, 1)[0];
        }
    }
}

// candidate extraction + first-yak scoring; fills the host RegionSet
void extract_candidates(np2_ctx *cx, np2_contig *c, uint32_t n_reg, uint16_t min_kmer_count, int pass,
                        RegionSet &rs) {
    hipStream_t s = cx->stream;
    const uint32_t R = c->R;
    REFPANIC_IF(cx->yaks.empty(), "index out of bounds: opt.yak[0]");
    uint32_t NP = 0, NC = 0, SB = 0;
    {
        EventTimer t(cx, "candidates");
        cx->mval.ensure(R + 2);
        cx->smin.ensure(R + 2);
        cx->pj.ensure(R + 2);
        cx->pcount.ensure(R + 2);
        cx->poff.ensure(R + 2);
        launch_read_m(s, c->reads.p, R, cx->alive.p, cx->lq_start.p, n_reg, cx->mval.p);
        if (prim_inclusive_min_i32(s, cx->tmp.p, cx->tmp.cap, cx->mval.p, cx->smin.p, R))
            throw Np2Error(NP2_E_DEVICE, "rocprim min-scan failed");
        launch_pair_count(s, c->reads.p, R, cx->alive.p, cx->lq_start.p, cx->lq_end.p, n_reg, cx->smin.p, cx->pj.p,
                          cx->pcount.p);
        zero32(cx, cx->pcount.p + R, 1);
        exclusive_total(cx, cx->pcount.p, cx->poff.p, (size_t)R + 1);
        NP = d2h(cx, cx->poff.p + R, 1)[0];
    }
    cx->reg_npairs.ensure(n_reg + 2);
    cx->reg_poff.ensure(n_reg + 2);
    cx->reg_ncand.ensure(n_reg + 2);
    cx->cand_off.ensure(n_reg + 2);
    cx->pair_region.ensure(NP + 2);
    cx->pair_read.ensure(NP + 2);
    cx->pair_region_s.ensure(NP + 2);
    cx->pair_read_s.ensure(NP + 2);
    cx->pair_len.ensure(NP + 2);
    cx->pair_keep.ensure(NP + 2);
    cx->keepflag.ensure(NP + 2);
    cx->cand_idx.ensure(NP + 2);
    cx->seq_off.ensure(NP + 2);
    cx->tmp.ensure(prim_temp_bytes(std::max<size_t>((size_t)NP + 2, (size_t)n_reg + 2)));
    CandPtrs cp{c->reads.p, c->nib.p, c->ck_off.p, c->ckpt.p, cx->lq_start.p, cx->lq_end.p, cx->pj.p, cx->yaks[0].k};
    {
        EventTimer t(cx, "candidates");
        zero32(cx, cx->reg_npairs.p, n_reg + 2);
        launch_pair_fill(s, R, cx->pj.p, cx->pcount.p, cx->poff.p, cx->pair_region.p, cx->pair_read.p,
                         cx->reg_npairs.p);
        unsigned bits = 1;
        while ((1ull << bits) < (uint64_t)n_reg + 1) ++bits;
        if (prim_sort_pairs_u32_u32(s, cx->tmp.p, cx->tmp.cap, cx->pair_region.p, cx->pair_region_s.p,
                                    cx->pair_read.p, cx->pair_read_s.p, NP, bits))
            throw Np2Error(NP2_E_DEVICE, "rocprim pair sort failed");
        exclusive_total(cx, cx->reg_npairs.p, cx->reg_poff.p, (size_t)n_reg + 1);
        launch_cand_measure(s, cp, cx->pair_region_s.p, cx->pair_read_s.p, NP, cx->pair_len.p);
        launch_region_rank(s, cx->reg_poff.p, n_reg, cx->pair_len.p, cx->pair_keep.p, cx->reg_ncand.p);
        zero32(cx, cx->reg_ncand.p + n_reg, 1);
        exclusive_total(cx, cx->reg_ncand.p, cx->cand_off.p, (size_t)n_reg + 1);
        if (NP) {
            // candidate slot = exclusive count of kept pairs; sequence offset = exclusive sum of kept lengths
            // (pair_keep holds the kept length, 0 for dropped pairs)
            zero32(cx, cx->pair_keep.p + NP, 1);
            exclusive_total(cx, cx->pair_keep.p, cx->seq_off.p, (size_t)NP + 1);
        }
    }
    // keep flags -> candidate index
    {
        EventTimer t(cx, "candidates");
        if (NP) {
            launch_flag_nonzero(s, cx->pair_keep.p, NP, cx->keepflag.p);
            zero32(cx, cx->keepflag.p + NP, 1);
            exclusive_total(cx, cx->keepflag.p, cx->cand_idx.p, (size_t)NP + 1);
            NC = d2h(cx, cx->cand_idx.p + NP, 1)[0];
            SB = d2h(cx, cx->seq_off.p + NP, 1)[0];
        }
    }
    cx->cand_order.ensure(NC + 2);
    cx->cand_kmer.ensure(NC + 2);
    cx->cand_seq_off.ensure(NC + 2);
    cx->cand_seq.ensure((size_t)SB + 64);
    cx->kscore.ensure(NC + 2);
    {
        EventTimer t(cx, "candidates");
        launch_cand_write(s, cp, cx->pair_region_s.p, cx->pair_read_s.p, cx->pair_keep.p, cx->cand_idx.p,
                          cx->seq_off.p, NP, cx->cand_order.p, cx->cand_kmer.p, cx->cand_seq_off.p, cx->cand_seq.p);
        HIPCHK(hipMemcpyAsync(cx->cand_seq_off.p + NC, &SB, 4, hipMemcpyHostToDevice, s));
    }
    {
        EventTimer t(cx, "kmer_score");
        launch_cand_score(s, cx->yaks[0].dev(), cx->cand_seq_off.p, cx->cand_seq.p, cx->cand_kmer.p, NC,
                          min_kmer_count, cx->kscore.p);
    }
    HIPCHK(hipStreamSynchronize(s));
    // D2H of the candidate tables
    auto start = d2h(cx, cx->lq_start.p, n_reg), end = d2h(cx, cx->lq_end.p, n_reg);
    auto coff = d2h(cx, cx->cand_off.p, (size_t)n_reg + 1);
    auto order = d2h(cx, cx->cand_order.p, NC);
    auto ks = d2h(cx, cx->kscore.p, NC);
    auto soff = d2h(cx, cx->cand_seq_off.p, (size_t)NC + 1);
    rs.pool = d2h(cx, cx->cand_seq.p, SB);
    rs.regs.assign(n_reg, Region());
    for (uint32_t g = 0; g < n_reg; ++g) {
        Region &rg = rs.regs[g];
        rg.start = start[g];
        rg.end = end[g];
        rg.seqs.reserve(coff[g + 1] - coff[g]);
        for (uint32_t i = coff[g]; i < coff[g + 1]; ++i) rg.seqs.push_back(Cand{order[i], ks[i], soff[i], soff[i + 1] - soff[i]});
    }
    if (cx->trace) {
        trace_regions(cx, pass, "cand", rs);
        trace_put(cx, pass, "cand.kmer", d2h(cx, cx->cand_kmer.p, NC));
    }
}

Cns fetch_cns(np2_ctx *cx, uint32_t M) {
    Cns c;
    c.pos = d2h(cx, cx->cns_pos.p, M);
    c.base = d2h(cx, cx->cns_base.p, M);
    return c;
}

void polish_impl(np2_ctx *cx, np2_contig *c, const np2_opts_t *o, Cns &result) {
    if (o->iter_count < 1) throw Np2Error(NP2_E_ARG, "iter_count must be >= 1");
    hipStream_t s = cx->stream;
    HIPCHK(hipSetDevice(cx->device));
    cx->trace_items.clear();
    cx->scal.ensure(S_COUNT);
    cx->alive.ensure(c->R + 2);
    uint32_t T = 0;
    run_diff(cx, c, T);
    launch_init_alive(s, c->reads.p, c->R, cx->alive.p);
    for (uint32_t pass = 0; pass < o->iter_count; ++pass) {
        const bool out_cns = pass + 1 == o->iter_count;
        uint32_t n_nodes = 0, n_runs = 0, M = 0, n_reg = 0;
        build_graph(cx, c, T, n_nodes, n_runs);
        trace_graph(cx, c, (int)pass, n_nodes);
        consensus_and_regions(cx, c, n_runs, M, n_reg);
        if (cx->trace) {
            trace_cns(cx, (int)pass, "cns_raw", fetch_cns(cx, M));
            trace_put(cx, (int)pass, "lq.start", d2h(cx, cx->lq_start.p, n_reg));
            trace_put(cx, (int)pass, "lq.end", d2h(cx, cx->lq_end.p, n_reg));
        }
        if (n_reg == 0) {
            if (out_cns) {
                result = fetch_cns(cx, M);
                return;
            }
            continue;
        }
        RegionSet rs;
        extract_candidates(cx, c, n_reg, o->min_kmer_count, (int)pass, rs);
        if (!out_cns) {
            std::vector<uint32_t> losers = phasing_vote(rs, o->model_ref != 0, o->use_all_reads != 0);
            trace_regions(cx, (int)pass, "hete", rs);
            trace_put(cx, (int)pass, "invalid_ids", losers);
            for (uint32_t id : losers) REFPANIC_IF(id >= c->R, "index out of bounds: alignseqs[id]");
            if (!losers.empty()) {
                cx->kill_ids.ensure(losers.size());
                HIPCHK(hipMemcpyAsync(cx->kill_ids.p, losers.data(), losers.size() * 4, hipMemcpyHostToDevice, s));
                launch_kill_reads(s, cx->kill_ids.p, (uint32_t)losers.size(), cx->alive.p);
                HIPCHK(hipStreamSynchronize(s));
            }
        } else {
            Cns cns = fetch_cns(cx, M);
            choose_seeds(rs, o->max_indel_len);
            trace_regions(cx, (int)pass, "seed", rs);
            cns = splice(rs.regs, cns, LB_SUCC);
            trace_cns(cx, (int)pass, "cns_succ", cns);
            for (size_t y = 0; y < cx->yaks.size(); ++y) {
                cns = recheck(cx, rs, cns, (int)y, o->min_kmer_count, y + 1);
                trace_regions(cx, (int)pass, "rech" + std::to_string(y), rs);
                trace_cns(cx, (int)pass, "cns_rech" + std::to_string(y), cns);
            }
            result = std::move(cns);
            return;
        }
    }
    throw Np2Error(NP2_E_ARG, "unreachable: no final pass");
}

int fail(np2_ctx *cx, const Np2Error &e) {
    if (cx) cx->err = e.what();
    return e.code;
}

} // namespace


// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int np2_ctx_create(np2_ctx_t **out, int device, const np2_yak_t *yaks, int n_yak) {
    if (!out) return NP2_E_ARG;
    *out = nullptr;
    np2_ctx *cx = new np2_ctx();
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw Np2Error(NP2_E_DEVICE, "no HIP device available (the np2 hot path has no CPU fallback)");
        if (device < 0 || device >= ndev) throw Np2Error(NP2_E_ARG, "bad device index");
        cx->device = device;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreateWithFlags(&cx->stream, hipStreamNonBlocking));
        cx->scal.ensure(S_COUNT);
        cx->yaks.resize(n_yak);
        for (int i = 0; i < n_yak; ++i) {
            const np2_yak_t &y = yaks[i];
            if (y.k >= 32 || y.k < 2) throw Np2Error(NP2_E_UNSUPPORTED, "yak k must be in [2, 32) (main.rs:1433-1434)");
            if (y.pre != 10) throw Np2Error(NP2_E_UNSUPPORTED, "yak pre must be 10 (kmer.rs:52-54,123-125)");
            if (i && yaks[i - 1].k > y.k) throw Np2Error(NP2_E_ARG, "yak tables must be sorted by k (option.rs:238)");
            uint64_t mx = 0;
            for (uint32_t b = 0; b < 1024; ++b) mx = std::max(mx, y.bucket_off[b + 1] - y.bucket_off[b]);
            uint32_t cl = 4;
            while ((1ull << cl) < mx * 2 + 2) ++cl;
            YakTable &t = cx->yaks[i];
            t.k = y.k;
            t.cap_log2 = cl;
            const size_t slots = (size_t)1024 << cl;
            t.table.ensure(slots);
            HIPCHK(hipMemsetAsync(t.table.p, 0xFF, slots * 8, cx->stream));
            DevBuf<uint64_t> dw, doff;
            dw.ensure(y.n_words + 1);
            doff.ensure(1025);
            HIPCHK(hipMemcpyAsync(dw.p, y.words, y.n_words * 8, hipMemcpyHostToDevice, cx->stream));
            HIPCHK(hipMemcpyAsync(doff.p, y.bucket_off, 1025 * 8, hipMemcpyHostToDevice, cx->stream));
            zero32(cx, cx->scal.p, S_COUNT);
            launch_yak_insert(cx->stream, dw.p, doff.p, 1024, mx, t.table.p, cl, cx->scal.p + S_DUP);
            auto sc = d2h(cx, cx->scal.p, S_COUNT);
            if (sc[S_DUP]) throw Np2Error(NP2_E_UNSUPPORTED, "duplicate k-mer key inside one yak bucket");
        }
    } catch (const Np2Error &e) {
        fprintf(stderr, "np2_ctx_create: %s\n", e.what());
        int code = e.code;
        if (cx->stream) (void)hipStreamDestroy(cx->stream);
        delete cx;
        return code;
    }
    *out = cx;
    return NP2_OK;
}

void np2_ctx_destroy(np2_ctx_t *cx) {
    if (!cx) return;
    (void)hipSetDevice(cx->device);
    if (cx->stream) {
        (void)hipStreamSynchronize(cx->stream);
        (void)hipStreamDestroy(cx->stream);
    }
    delete cx;
}
const char *np2_last_error(np2_ctx_t *cx) { return cx ? cx->err.c_str() : "null context"; }
void *np2_ctx_stream(np2_ctx_t *cx) { return cx ? (void *)cx->stream : nullptr; }
void np2_ctx_set_trace(np2_ctx_t *cx, int enable) {
    if (cx) cx->trace = enable != 0;
}

int np2_contig_upload(np2_ctx_t *cx, const uint8_t *ref, uint32_t L, const np2_read_t *reads, uint32_t n_reads,
                      const uint8_t *nibbles, uint64_t nib_bytes, np2_contig_t **out) {
    if (!cx || !out) return NP2_E_ARG;
    *out = nullptr;
    np2_contig *c = new np2_contig();
    try {
        (void)ref;
        HIPCHK(hipSetDevice(cx->device));
        if (L < 3 || n_reads < 1 || !reads || !nibbles) throw Np2Error(NP2_E_ARG, "bad contig arguments");
        if (reads[0].aln_t_s != 0 || reads[0].n_cols != L || reads[0].aln_t_e != L - 1 ||
            (reads[0].flags & NP2_READ_DROPPED))
            throw Np2Error(NP2_E_ARG, "reads[0] must be the contig aligned to itself (main.rs:1732-1739)");
        std::vector<uint64_t> ck(n_reads + 1, 0);
        uint64_t cols = 0;
        for (uint32_t r = 0; r < n_reads; ++r) {
            const np2_read_t &rd = reads[r];
            if (rd.nib_off & 15) throw Np2Error(NP2_E_ARG, "nib_off must be a multiple of 16");
            if (rd.aln_t_e >= L || rd.aln_t_s > rd.aln_t_e) throw Np2Error(NP2_E_ARG, "read span outside the contig");
            if (rd.nib_off + ((uint64_t)(rd.n_cols + 1) >> 1) + 1 + 16 > nib_bytes)
                throw Np2Error(NP2_E_ARG, "nibble stream (plus 16 B tail padding) exceeds the buffer");
            const uint32_t first = (rd.aln_t_s + CKPT - 1) >> CKPT_SHIFT, last = rd.aln_t_e >> CKPT_SHIFT;
            ck[r + 1] = ck[r] + (last >= first ? last - first + 1 : 0);
            cols += rd.n_cols;
        }
        c->L = L;
        c->R = n_reads;
        c->nib_bytes = nib_bytes;
        c->n_cols = cols;
        c->n_ckpt = ck[n_reads];
        c->reads.ensure(n_reads);
        c->nib.ensure(nib_bytes + 64);
        const uint32_t refbytes = ((L + 1) >> 1) + 96; // padding so that 128-bit probes near the end stay in bounds
        c->refnib.ensure(((size_t)refbytes + 7) & ~(size_t)7);
        c->ck_off.ensure(n_reads + 1);
        c->ckpt.ensure(c->n_ckpt + 1);
        hipStream_t s = cx->stream;
        HIPCHK(hipMemcpyAsync(c->reads.p, reads, (size_t)n_reads * sizeof(np2_read_t), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(c->nib.p, nibbles, nib_bytes, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(c->ck_off.p, ck.data(), (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, s));
        cx->scal.ensure(S_COUNT);
        zero32(cx, cx->scal.p, S_COUNT);
        launch_encode_ref(s, c->nib.p + reads[0].nib_off, L, c->refnib.p, refbytes, cx->scal.p + S_ERR);
        auto sc = d2h(cx, cx->scal.p, S_COUNT);
        if (sc[S_ERR]) throw Np2Error(NP2_E_ARG, "reads[0] is not a plain self-alignment of the contig");
    } catch (const Np2Error &e) {
        delete c;
        return fail(cx, e);
    }
    *out = c;
    return NP2_OK;
}
void np2_contig_free(np2_ctx_t *cx, np2_contig_t *c) {
    if (cx) (void)hipSetDevice(cx->device);
    delete c;
}

int np2_polish_resident(np2_ctx_t *cx, np2_contig_t *c, const np2_opts_t *opts, uint8_t **out_bases,
                        uint32_t **out_pos, uint64_t *out_len) {
    if (!cx || !c || !opts || !out_bases || !out_pos || !out_len) return NP2_E_ARG;
    Cns r;
    try {
        polish_impl(cx, c, opts, r);
        flush_timings(cx);
    } catch (const Np2Error &e) {
        (void)hipStreamSynchronize(cx->stream);
        flush_timings(cx);
        return fail(cx, e);
    }
    *out_len = r.size();
    *out_bases = (uint8_t *)malloc(r.size() + 1);
    *out_pos = (uint32_t *)malloc((r.size() + 1) * sizeof(uint32_t));
    if (!*out_bases || !*out_pos) return NP2_E_NOMEM;
    memcpy(*out_bases, r.base.data(), r.size());
    memcpy(*out_pos, r.pos.data(), r.size() * sizeof(uint32_t));
    return NP2_OK;
}

int np2_polish_contig(np2_ctx_t *cx, const uint8_t *ref, uint32_t L, const np2_read_t *reads, uint32_t n_reads,
                      const uint8_t *nibbles, uint64_t nib_bytes, const np2_opts_t *opts, uint8_t **out_bases,
                      uint32_t **out_pos, uint64_t *out_len) {
    np2_contig_t *c = nullptr;
    int rc = np2_contig_upload(cx, ref, L, reads, n_reads, nibbles, nib_bytes, &c);
    if (rc) return rc;
    rc = np2_polish_resident(cx, c, opts, out_bases, out_pos, out_len);
    np2_contig_free(cx, c);
    return rc;
}
void np2_free(void *p) { free(p); }

int np2_score_strings(np2_ctx_t *cx, int yak_idx, const uint8_t *strs, const uint64_t *off, uint64_t n,
                      uint16_t min_kmer_count, uint16_t *scores) {
    if (!cx || yak_idx < 0 || (size_t)yak_idx >= cx->yaks.size()) return NP2_E_ARG;
    try {
        HIPCHK(hipSetDevice(cx->device));
        std::vector<uint8_t> blob(strs, strs + off[n]);
        blob.resize(blob.size() + 8, 0);
        std::vector<uint64_t> o(off, off + n + 1);
        std::vector<uint16_t> sc;
        gpu_score_strings(cx, yak_idx, blob, o, min_kmer_count, sc);
        memcpy(scores, sc.data(), n * 2);
        flush_timings(cx);
    } catch (const Np2Error &e) {
        return fail(cx, e);
    }
    return NP2_OK;
}

int np2_lookup_hashes(np2_ctx_t *cx, int yak_idx, const uint64_t *hashes, uint64_t n, uint16_t min_kmer_count,
                      uint16_t *counts) {
    if (!cx || yak_idx < 0 || (size_t)yak_idx >= cx->yaks.size()) return NP2_E_ARG;
    try {
        HIPCHK(hipSetDevice(cx->device));
        cx->soff.ensure(n + 1);
        cx->sscore.ensure(n + 1);
        HIPCHK(hipMemcpyAsync(cx->soff.p, hashes, n * 8, hipMemcpyHostToDevice, cx->stream));
        launch_lookup(cx->stream, cx->yaks[yak_idx].dev(), cx->soff.p, n, min_kmer_count, cx->sscore.p);
        HIPCHK(hipMemcpyAsync(counts, cx->sscore.p, n * 2, hipMemcpyDeviceToHost, cx->stream));
        HIPCHK(hipStreamSynchronize(cx->stream));
    } catch (const Np2Error &e) {
        return fail(cx, e);
    }
    return NP2_OK;
}

int np2_trace_get(np2_ctx_t *cx, int pass, const char *name, const void **data, uint64_t *nbytes) {
    if (!cx) return NP2_E_ARG;
    auto it = cx->trace_items.find(std::to_string(pass) + ":" + name);
    if (it == cx->trace_items.end()) return NP2_E_ARG;
    *data = it->second.data();
    *nbytes = it->second.size();
    return NP2_OK;
}

int np2_last_timings(np2_ctx_t *cx, const char **names, const float **ms, int *n) {
    if (!cx) return NP2_E_ARG;
    *names = cx->timing.joined.c_str();
    *ms = cx->timing.ms.data();
    *n = (int)cx->timing.ms.size();
    return NP2_OK;
}
}
