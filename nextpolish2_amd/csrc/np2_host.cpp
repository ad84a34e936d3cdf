// Host driver of the MI355X NextPolish2 hot path: C ABI (include/np2.h), HBM residency,
// per-pass kernel orchestration and the host-resident region logic (vote / seed / splice /
// recheck control and the Louvain phasing vote).  Replaces the per-contig loop
// src/main.rs:1819-1836 of the reference.  No CPU fallback: every entry point needs a HIP device.
#include "../../include/np2.h"
#include "np2_common.hpp"
#include "np2_kernels.hpp"
#include "np2_ctx.hpp"
#include "np2_phase_host.hpp"

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <future>
#include <functional>
#include <vector>

using namespace np2;

namespace np2 {
Recorder *&tl_recorder() {
    static thread_local Recorder *r = nullptr;
    return r;
}
} // namespace np2


namespace {

// ------------------------------------------------------------------------------------------
// region state (device-resident; host keeps only counters)
// ------------------------------------------------------------------------------------------
static const uint8_t LB_SUCC = 0x80, LB_RECH = 0x20; // main.rs:655-658

struct Cns {
    std::vector<uint32_t> pos;
    std::vector<uint8_t> base;
    size_t size() const { return pos.size(); }
};
void trace_cns(np2_ctx *cx, int pass, const std::string &tag, const Cns &c) {
    trace_put(cx, pass, tag + ".pos", c.pos);
    trace_put(cx, pass, tag + ".base", c.base);
}

struct PassCounts {
    uint32_t n_reg = 0, NC = 0, SB = 0, M = 0;
    uint32_t grow = 0; // upper bound of the consensus growth of one splice round (sum of the longest kept candidates)
    // NC / SB / grow are produced on the device (scal S_NC / S_SB / S_GROW) and picked up by the next read-back that
    // happens anyway; until then buffers and launches use the bounds below
    bool known = false;
    uint32_t NC_cap = 0; // 60 candidates per region
    uint32_t SB_cap = 0; // candidate strings are disjoint pieces of the reads: at most the pileup's columns
    void resolve(const std::vector<uint32_t> &sc) {
        NC = sc[S_NC], SB = sc[S_SB], grow = sc[S_GROW];
        known = true;
    }
};

RegionTables region_tables(np2_ctx *cx, uint32_t n_reg) {
    return RegionTables{cx->cand_off.p, cx->cand_order.p, cx->kscore.p, cx->cand_seq_off.p, cx->cand_seq.p, n_reg};
}

// candidate tables as the oracle traces them ("cand" / "hete": every candidate; "seed" / "rechN": kept lists)
void trace_region_tables(np2_ctx *cx, int pass, const std::string &tag, const PassCounts &pc, bool kept_view) {
    if (!cx->trace) return;
    const uint32_t n_reg = pc.n_reg;
    auto start = d2h(cx, cx->lq_start.p, n_reg), end = d2h(cx, cx->lq_end.p, n_reg);
    auto coff = d2h(cx, cx->cand_off.p, (size_t)n_reg + 1);
    auto order = d2h(cx, cx->cand_order.p, pc.NC);
    auto ks = d2h(cx, cx->kscore.p, pc.NC);
    auto soff = d2h(cx, cx->cand_seq_off.p, (size_t)pc.NC + 1);
    auto pool = d2h(cx, cx->cand_seq.p, pc.SB);
    std::vector<uint8_t> lable(n_reg, 0);
    std::vector<uint32_t> seedc, keepn, keepl;
    std::vector<uint16_t> keepks;
    if (tag != "cand") lable = d2h(cx, cx->reg_lable.p, n_reg);
    if (kept_view) {
        seedc = d2h(cx, cx->seed_cand.p, n_reg);
        keepn = d2h(cx, cx->keep_n.p, n_reg);
        keepl = d2h(cx, cx->keep_list.p, pc.NC);
        keepks = d2h(cx, cx->keep_ks.p, pc.NC);
    }
    std::vector<uint32_t> o_coff(1, 0), o_order, o_soff(1, 0), sudo_off(1, 0);
    std::vector<uint16_t> o_ks;
    std::vector<uint8_t> o_seq, sudo;
    for (uint32_t g = 0; g < n_reg; ++g) {
        if (kept_view) {
            const uint32_t c = seedc[g];
            if (c != 0xFFFFFFFFu) sudo.insert(sudo.end(), pool.begin() + soff[c], pool.begin() + soff[c + 1]);
            for (uint32_t t = 0; t < keepn[g]; ++t) {
                const uint32_t ci = keepl[coff[g] + t];
                o_order.push_back(order[ci]);
                o_ks.push_back(keepks[coff[g] + t]);
                o_seq.insert(o_seq.end(), pool.begin() + soff[ci], pool.begin() + soff[ci + 1]);
                o_soff.push_back((uint32_t)o_seq.size());
            }
        } else {
            for (uint32_t ci = coff[g]; ci < coff[g + 1]; ++ci) {
                o_order.push_back(order[ci]);
                o_ks.push_back(ks[ci]);
                o_seq.insert(o_seq.end(), pool.begin() + soff[ci], pool.begin() + soff[ci + 1]);
                o_soff.push_back((uint32_t)o_seq.size());
            }
        }
        sudo_off.push_back((uint32_t)sudo.size());
        o_coff.push_back((uint32_t)o_order.size());
    }
    trace_put(cx, pass, tag + ".start", start);
    trace_put(cx, pass, tag + ".end", end);
    trace_put(cx, pass, tag + ".lable", lable);
    trace_put(cx, pass, tag + ".sudo_off", sudo_off);
    trace_put(cx, pass, tag + ".sudo", sudo);
    trace_put(cx, pass, tag + ".cand_off", o_coff);
    trace_put(cx, pass, tag + ".order", o_order);
    trace_put(cx, pass, tag + ".kscore", o_ks);
    trace_put(cx, pass, tag + ".seq_off", o_soff);
    trace_put(cx, pass, tag + ".seq", o_seq);
    if (tag == "cand") trace_put(cx, pass, "cand.kmer", d2h(cx, cx->cand_kmer.p, pc.NC));
}

// the final pass ran its splice rounds with a guessed growth allowance that turned out too small (nothing was written
// out of bounds: k_splice_plan): run_final_pass repeats the pass with the exact one
struct GrowRetry : Np2Error {
    GrowRetry() : Np2Error(NP2_E_DEVICE, "internal: guessed growth allowance too small") {}
};
void check_region_err(np2_ctx *cx, uint32_t e) {
    (void)cx;
    if (e & 4u) throw Np2Error(NP2_E_REFPANIC, "reference would panic: seq2 order is equal to 0");
    if (e & 8u) throw Np2Error(NP2_E_REFPANIC, "reference would panic: index out of bounds: lqseq.seqs[max1_p]");
    if (e & 16u) throw Np2Error(NP2_E_REFPANIC, "reference would panic: the first lqseq is not ref.");
    if (e & 32u) throw Np2Error(NP2_E_REFPANIC, "reference would panic: lqseq.seqs[0] after retain_sort_seqs");
    if (e & 64u) throw Np2Error(NP2_E_REFPANIC, "reference would panic: consensus index out of bounds in reupdate");
    if (e & 128u) throw Np2Error(NP2_E_NOMEM, "cartesian product of chained LQ regions is too large");
    if (e & LB_ERR) throw Np2Error(NP2_E_DEVICE, "device-wide scan timed out waiting for a predecessor block");
    if (e & GROW_ERR) throw GrowRetry();
    if (e & LQ_LIST_ERR) throw Np2Error(NP2_E_DEVICE, "internal: more low-quality bases than their list was sized for");
}

// What one phasing pass hands to the host side of the vote (phase_reads_by_lqseqs, main.rs:948-1015).  Everything in
// it is additive over the HETE regions it was collected from, so the shards of one contig (np2_shard_*) each collect
// theirs over the regions they own and the contig's owner merges them before the Louvain.
struct VoteData {
    uint32_t R = 0;                   // reads of the (sub-)contig, local numbering
    bool any = false;                 // false: no read votes anywhere, nobody can lose
    std::vector<uint64_t> pair_key;   // a << 32 | b (a < b), ascending
    std::vector<uint32_t> pair_cnt;   // HETE regions in which the pair agrees | disagrees << 16
    // ... or, instead of the two vectors, a view of a caller's arrays (np2_vote_decide with one sorted vote: a
    // chromosome's pair list is 200 MB, not to be copied for nothing)
    const uint64_t *view_key = nullptr;
    const uint32_t *view_cnt = nullptr;
    uint64_t view_n = 0;
    // ... or the compact rows of the plain pipeline (k_band_emit_compact: 4 bytes per pair; row a of the reads' band =
    // words [c_off[a], c_off[a + 1]), word = (b - a - 1) | agree << 8 | disagree << 20), again a view of the staging
    const uint32_t *c_off = nullptr, *c_pairs = nullptr;
    std::vector<uint32_t> c_off_own, c_pairs_own;
    bool compact() const { return c_off != nullptr; }
    const uint64_t *keys() const { return view_key ? view_key : pair_key.data(); }
    const uint32_t *cnts() const { return view_key ? view_cnt : pair_cnt.data(); }
    uint64_t n_pairs() const { return compact() ? (uint64_t)c_off[R] : view_key ? view_n : (uint64_t)pair_key.size(); }
    void own() { // copy a view into the vectors (the viewed memory is about to be reused)
        if (compact() && c_off_own.empty()) {
            c_off_own.assign(c_off, c_off + R + 1);
            c_pairs_own.assign(c_pairs, c_pairs + c_off[R]);
            c_off = c_off_own.data(), c_pairs = c_pairs_own.data();
        }
        if (!view_key) return;
        pair_key.assign(view_key, view_key + view_n);
        pair_cnt.assign(view_cnt, view_cnt + view_n);
        view_key = nullptr, view_cnt = nullptr, view_n = 0;
    }
    std::vector<uint32_t> first_key;  // per read: creation rank of its key in the reference's weight map = index of the
                                      // first HETE region it votes in (0xFFFFFFFF: none); shards: see shard_vote_export
    std::vector<int32_t> ref_w;       // per read: summed weight against the contig's own candidate (ref_data[0])
    std::vector<uint8_t> ref_seen, bad;
    const uint8_t *d_bad = nullptr;   // the `bad` flags where the vote kernel left them on the device (until the next vote)
    bool key_added = false;           // (wide form) the pair keys already carry the shard's read-number shift
};

// GPU part of the phasing pass: mark_hete (main.rs:916-946), pair votes (948-1002) over the regions whose start lies in
// [own_lo, own_hi)
// `wide`: pairs as (a << 32 | b, counts) — what the shards of a contig export and merge; otherwise the compact rows, read
// back in the SAME wait as the counters that size them (one device round trip and 8 bytes per pair less)
void vote_collect(np2_ctx *cx, np2_contig *c, PassCounts &pc, bool asref, bool use_all, int pass, uint32_t own_lo,
                  uint32_t own_hi, VoteData &vd, bool wide = false, uint64_t key_add = 0) {
    hipStream_t s = cx->stream;
    const uint32_t R = c->R, n_reg = pc.n_reg;
    RegionTables rt = region_tables(cx, n_reg);
    uint32_t NE = 0, NU = 0;
    vd = VoteData();
    vd.R = R;
    {
        EventTimer t(cx, "vote_phase");
        cx->reg_lable.ensure(n_reg + 2);
        cx->grp.ensure((size_t)pc.NC_cap + 2);
        cx->ecount.ensure(n_reg + 2);
        cx->eoff.ensure(std::max<size_t>(n_reg + 2, (size_t)c->L + 2));
        // per-read vote outputs live in one buffer: [first_reg u32 x RP][ref_w i32 x RP][ref_seen u8 x RP][bad u8 x RP]
        // (compact: followed by the row offsets and the pair words — one piece to read back)
        const size_t RP = ((size_t)R + 63) & ~(size_t)63;
        const size_t band_words = (size_t)R * EDGE_BAND;
        const size_t b_v = RP * 10, b_off = (((size_t)R + 2) * 4 + 15) & ~(size_t)15;
        uint8_t *vbuf;
        uint32_t *row_off;
        if (wide) {
            cx->votebuf.ensure(b_v);
            cx->band_off.ensure((size_t)R + 2);
            vbuf = cx->votebuf.p, row_off = cx->band_off.p;
        } else {
            cx->votepack.ensure(b_v + b_off + band_words * 4 + 64);
            vbuf = cx->votepack.p, row_off = (uint32_t *)(cx->votepack.p + b_v);
        }
        uint32_t *v_first = (uint32_t *)vbuf;
        int32_t *v_refw = (int32_t *)(vbuf + RP * 4);
        uint8_t *v_seen = vbuf + RP * 8, *v_bad = vbuf + RP * 9;
        op_fill(cx, v_first, 0xFF, RP * 4);
        op_fill(cx, v_refw, 0, RP * 6);
        launch_vote_phase(s, rt, asref, use_all, cx->lq_start.p, own_lo, own_hi, cx->reg_lable.p, cx->grp.p, cx->ecount.p,
                          v_refw, v_seen, v_bad, v_first, cx->scal.p + S_ERR);
        exclusive_total_n(cx, cx->ecount.p, cx->eoff.p, n_reg);
        launch_vote_counts(s, v_first, v_bad, R, cx->scal.p + S_M1); // S_M1 = graph keys, S_M2 = invalid reads
        // distinct read pairs + their vote counts, queued behind the vote without waiting for its counts (one read-back
        // less per phasing pass; with no HETE region the three kernels find nothing to do).  Normal case: the raw pair
        // votes are accumulated in the banded matrix (reads are numbered in start order, partners are close); rows read
        // in order = the sorted unique list, at most EDGE_BAND pairs per read.
        cx->band.ensure(band_words + 4);
        cx->band_n.ensure((size_t)R + 2);
        op_fill(cx, cx->scal.p + S_M3, 0, 4);
        launch_edges_row(s, rt, cx->grp.p, cx->ecount.p, cx->pj.p, cx->pcount.p, cx->alive.p, R, cx->band.p, cx->band_n.p,
                         cx->scal.p + S_M3);
        exclusive_total_n(cx, cx->band_n.p, row_off, R);
        if (wide) {
            cx->ekey.ensure(band_words + 2);
            cx->eval.ensure(band_words + 2);
            launch_band_emit(s, cx->band.p, R, row_off, cx->ekey.p, cx->eval.p, cx->scal.p + S_NRAW, key_add);
        } else {
            launch_band_emit_compact(s, cx->band.p, R, row_off, (uint32_t *)(cx->votepack.p + b_v + b_off), cx->scal.p + S_NRAW,
                                     cx->scal.p + S_M3);
        }
    }
    const size_t RP = ((size_t)R + 63) & ~(size_t)63;
    const size_t b_v = RP * 10, b_off = (((size_t)R + 2) * 4 + 15) & ~(size_t)15;
    auto per_read = [&](const uint8_t *vb) {
        vd.first_key.assign((const uint32_t *)vb, (const uint32_t *)vb + R);
        vd.ref_w.assign((const int32_t *)(vb + RP * 4), (const int32_t *)(vb + RP * 4) + R);
        vd.ref_seen.assign(vb + RP * 8, vb + RP * 8 + R);
        vd.bad.assign(vb + RP * 9, vb + RP * 9 + R);
    };
    bool far = false, have_per_read = false;
    {
        std::vector<uint32_t> sc;
        uint8_t *pin = nullptr;
        // pairs the first copy has room for: 64 per read (a 30x diploid pileup has ~36); more follow in a second copy
        const uint32_t likely = (uint32_t)std::min<size_t>((size_t)R * EDGE_BAND, std::max<size_t>((size_t)R * 64, 1u << 14));
        if (wide) {
            sc = fetch_scal(cx, cx->scal.p + S_M0, cx->eoff.p + n_reg);
        } else {
            // counters, per-read arrays, row offsets and the pair words in ONE wait: the mailbox post, two copies (the second
            // one sized on the device by the number of pairs), one synchronisation
            pin = (uint8_t *)cx->pin_d2h.ensure(b_v + b_off + (size_t)likely * 4 + 64);
            const uint32_t seq = ++cx->mbox_seq;
            launch_post(cx->stream, cx->scal.p, S_COUNT, cx->mbox_dev, seq, cx->scal.p + S_M0, cx->eoff.p + n_reg, nullptr, nullptr,
                        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
            op_d2h(cx, pin, cx->votepack.p, b_v + b_off);
            launch_copy_counted(cx->stream, (uint32_t *)(pin + b_v + b_off), (const uint32_t *)(cx->votepack.p + b_v + b_off),
                                (const uint32_t *)(cx->votepack.p + b_v) + R, likely);
            op_sync(cx);
            if (__atomic_load_n(&cx->mbox_host[0], __ATOMIC_ACQUIRE) != seq)
                throw Np2Error(NP2_E_DEVICE, "mailbox not posted after a synchronisation");
            sc.assign(cx->mbox_host + 1, cx->mbox_host + 1 + S_COUNT);
        }
        check_region_err(cx, sc[S_ERR]);
        pc.resolve(sc);
        NE = sc[S_M0];
        if (!cx->trace && sc[S_M1] == 0 && sc[S_M2] == 0) { // no read votes anywhere: nobody can lose
            if (NE) throw Np2Error(NP2_E_DEVICE, "internal: pair edges without voting reads");
            return;
        }
        NU = sc[S_NRAW];
        far = sc[S_M3] != 0 || cx->hooks.edge_sort; // (test hook: force the sort-based path)
        if (!wide) {
            per_read(pin);
            have_per_read = true;
            vd.d_bad = cx->votepack.p + RP * 9;
            if (!far) {
                if (NU > likely) { // more pairs than the first copy had room for: fetch the whole piece again
                    pin = (uint8_t *)cx->pin_d2h.ensure(b_v + b_off + (size_t)NU * 4 + 64);
                    op_d2h(cx, pin, cx->votepack.p, b_v + b_off + (size_t)NU * 4);
                    op_sync(cx);
                }
                vd.c_off = (const uint32_t *)(pin + b_v), vd.c_pairs = (const uint32_t *)(pin + b_v + b_off);
                if (vd.c_off[R] != NU) throw Np2Error(NP2_E_DEVICE, "internal: vote rows and pair count disagree");
            }
        }
    }
    vd.any = true;
    if (NE && far) { // some pair lies outside the band (deep pileup) or a count outgrew 12 bits: sort the raw votes instead
        {
            EventTimer t(cx, "vote_phase");
            cx->ekey.ensure(NE + 2);
            cx->ekey_s.ensure(NE + 2);
            cx->eval.ensure(NE + 2);
            cx->eval_s.ensure(NE + 2);
            cx->eflag.ensure(NE + 2);
            cx->eidx.ensure(NE + 2);
            cx->ew.ensure(NE + 2);
            cx->tmp.ensure(prim_temp_bytes((size_t)NE + 2));
            launch_edges_write(s, rt, cx->reg_lable.p, cx->grp.p, cx->ecount.p, cx->eoff.p, cx->ekey.p, cx->eval.p);
            unsigned rbits = 1;
            while ((1ull << rbits) < (uint64_t)R + 1) ++rbits;
            prim_op(cx, [=](hipStream_t st) {
                if (prim_sort_pairs_u64_u32(st, cx->tmp.p, cx->tmp.cap, cx->ekey.p, cx->ekey_s.p, cx->eval.p, cx->eval_s.p, NE,
                                            32 + rbits))
                    throw Np2Error(NP2_E_DEVICE, "rocprim edge sort failed");
            });
            launch_edge_reduce(s, cx->ekey_s.p, cx->eval_s.p, NE, cx->eflag.p, (uint32_t *)cx->ew.p);
            exclusive_total(cx, cx->eflag.p, cx->eidx.p, NE);
            // compact into ekey / eval (packed agree | disagree << 16 counts)
            launch_edge_compact(s, cx->ekey_s.p, cx->eflag.p, cx->eidx.p, (const uint32_t *)cx->ew.p, NE, cx->ekey.p,
                                cx->eval.p, cx->scal.p + S_NRAW);
            NU = fetch_scal(cx)[S_NRAW];
        }
    } else if (!NE) {
        NU = 0;
    }
    if (wide || far) {
        // one wait for everything the host side of the vote still needs: unique pairs, their counts, (wide:) the per-read
        // vote arrays
        const size_t b_key = ((size_t)NU * 8 + 15) & ~(size_t)15, b_w = ((size_t)NU * 4 + 15) & ~(size_t)15;
        const size_t b_pr = have_per_read ? 0 : b_v;
        uint8_t *pin = (uint8_t *)cx->pin_d2h.ensure(b_key + b_w + b_pr + 64);
        if (tl_recorder()) {
            // batch driver: the pieces are gathered into one device buffer by copy kernels (one launch each for all
            // contigs of the batch) and cross the bus as ONE transfer per contig
            cx->votepack2.ensure(b_key + b_w + b_pr + 64);
            op_copy_d2d(cx, cx->votepack2.p, cx->ekey.p, (size_t)NU * 8);
            op_copy_d2d(cx, cx->votepack2.p + b_key, cx->eval.p, (size_t)NU * 4);
            if (b_pr) op_copy_d2d(cx, cx->votepack2.p + b_key + b_w, cx->votebuf.p, b_pr);
            op_d2h(cx, pin, cx->votepack2.p, b_key + b_w + b_pr);
        } else {
            op_d2h(cx, pin, cx->ekey.p, (size_t)NU * 8);
            op_d2h(cx, pin + b_key, cx->eval.p, (size_t)NU * 4);
            if (b_pr) op_d2h(cx, pin + b_key + b_w, cx->votebuf.p, b_pr);
        }
        op_sync(cx);
        // the pairs stay where they landed (the context's pinned read-back staging: valid until its next read-back; a
        // chromosome's list is 200 MB): the plain pipeline decides the vote right away, a shard copies them first
        vd.view_key = (const uint64_t *)pin, vd.view_cnt = (const uint32_t *)(pin + b_key), vd.view_n = NU;
        vd.key_added = wide && !far; // (the sort path's keys are local)
        if (b_pr) {
            per_read(pin + b_key + b_w);
            if (wide) vd.d_bad = cx->votebuf.p + RP * 9; // (where the vote kernel left the flags: np2_shard_vote's early start)
        }
    }
    if (cx->trace) {
        vd.own(); // (the traces below read back through the same staging)
        trace_put(cx, pass, "hete.lable", d2h(cx, cx->reg_lable.p, n_reg));
        trace_put(cx, pass, "hete.kscore", d2h(cx, cx->kscore.p, pc.NC));
    }
}

// Host part of the vote: rebuild the weight maps in the reference's key-creation order, then Louvain
// (louvain.rs:290-356).  Outer keys of `data` are created region by region (index order), valid non-ref candidates in
// position (= read index) order, first occurrence wins.
std::vector<uint32_t> vote_decide(np2_ctx *cx, const VoteData &vd, bool use_all) {
    if (!vd.any) return {};
    const uint32_t R = vd.R;
    const uint64_t NU = vd.n_pairs();
    const uint32_t *const pair_cnt = vd.compact() ? nullptr : vd.cnts();
    const double t_host0 = now_ms();
    // reads with a key, by (creation rank, read): the reads come in read order, so a STABLE sort by rank alone does it —
    // two 16-bit counting passes (a comparison sort of a chromosome's 3 x 10^5 keys was 12 ms)
    std::vector<std::pair<uint32_t, uint32_t>> keys;
    for (uint32_t r = 0; r < R; ++r)
        if (vd.first_key[r] != 0xFFFFFFFFu) keys.emplace_back(vd.first_key[r], r);
    if (keys.size() < 4096) {
        std::sort(keys.begin(), keys.end());
    } else {
        std::vector<std::pair<uint32_t, uint32_t>> tmp(keys.size());
        std::vector<uint32_t> cnt(65537);
        for (int pass = 0; pass < 2; ++pass) {
            const int sh = 16 * pass;
            std::fill(cnt.begin(), cnt.end(), 0u);
            for (const auto &k : keys) ++cnt[((k.first >> sh) & 0xFFFFu) + 1];
            for (size_t i = 0; i < 65536; ++i) cnt[i + 1] += cnt[i];
            for (const auto &k : keys) tmp[cnt[(k.first >> sh) & 0xFFFFu]++] = k;
            keys.swap(tmp);
        }
    }
    const bool prof = getenv("NP2_PHASE_PROFILE") != nullptr;
    double t_mark = t_host0;
    auto mark = [&](const char *what) {
        if (prof) {
            const double t = now_ms();
            fprintf(stderr, "  vote host: %s %.2f ms\n", what, t - t_mark);
            t_mark = t;
        }
    };
    phase::Graph data;
    data.reserve_ids(R);
    mark("key order");
    for (auto &k : keys) data.add_key(k.second);
    mark("key table");
    // data weight of a pair = sum(w), unless the pair disagrees in 3 or more regions: then -(#disagreements)
    // (dif <= -3 overwrites, main.rs:996-1002)
    auto weight = [&](uint64_t i) {
        const int32_t same = (int32_t)(pair_cnt[i] & 0xFFFFu), neg = (int32_t)(pair_cnt[i] >> 16);
        return (float)(neg >= 3 ? -neg : same - neg);
    };
    // data.retain(..) + per-row retain (main.rs:1004-1010) drop the reads flagged bad and every edge pointing at one:
    // those edges are left out while the rows are built (the rows' relative order is all that is kept of them)
    auto weight_c = [](uint32_t w) {
        const int32_t same = (int32_t)((w >> 8) & 0xFFFu), neg = (int32_t)(w >> 20);
        return (float)(neg >= 3 ? -neg : same - neg);
    };
    // data.retain(..) on the keys themselves (erase order = bucket order) BEFORE the rows: the rows leave the flagged reads'
    // edges out by themselves, and with the key table final the first level's community map — the keys inserted into a
    // fresh map in the table's iteration order, 10 ms of hash-order emulation for a chromosome — is built on a helper
    // thread beside the rows
    std::vector<uint32_t> bad;
    for (uint32_t r = 0; r < R; ++r)
        if (vd.bad[r]) bad.push_back(r);
    if (!use_all) data.keys.keep_if([&](uint32_t k, phase::Nil &) { return !vd.bad[k]; }); // (is_key follows after the rows: they check their endpoints against it)
    mark("retain");
    std::future<phase::OrderSet> communities;
    if (data.keys.size() >= (1u << 15) && phase::host_threads() > 1)
        communities = std::async(std::launch::async, [&data] { return phase::SignedLouvain::first_communities(data.keys); });
    bool ok = false;
    try {
        ok = vd.compact() ? data.add_edges_rows(vd.c_off, vd.c_pairs, R, weight_c, use_all ? nullptr : vd.bad.data())
                          : data.add_edges_sorted(vd.keys(), NU, weight, use_all ? nullptr : vd.bad.data());
    } catch (...) {
        if (communities.valid()) communities.wait(); // (it reads data.keys)
        throw;
    }
    phase::OrderSet comm_pre;
    if (communities.valid()) comm_pre = communities.get();
    const bool have_pre = !comm_pre.empty();
    if (!ok) throw Np2Error(NP2_E_DEVICE, "internal: edge endpoint without a key");
    if (!use_all)
        for (uint32_t b : bad) data.is_key[b] = 0;
    mark("keys + edges");
    std::vector<float> ref_row(R, 0.f);
    std::vector<uint8_t> ref_have(R, 0);
    bool have_ref = false;
    for (uint32_t r = 0; r < R; ++r)
        if (vd.ref_seen[r]) {
            ref_row[r] = (float)vd.ref_w[r];
            ref_have[r] = 1;
            have_ref = true;
        }
    std::vector<uint32_t> losers;
    if (!phase::losing_reads(std::move(data), have_ref, ref_row, ref_have, losers, have_pre ? &comm_pre : nullptr))
        throw Np2Error(NP2_E_REFPANIC,
                       "reference would panic: the weight of two conflicting community is not less than 0");
    mark("louvain + ranking");
    if (cx) cx->timing.host.push_back({"wall_louvain", (float)(now_ms() - t_host0)});
    // ascending, each once (flags over the read ids: a comparison sort of a chromosome's 3 x 10^5 losers was 12 ms)
    std::vector<uint8_t> lost(R, 0);
    for (uint32_t b : bad) lost[b] = 1;
    for (uint32_t l : losers)
        if (l < R) lost[l] = 1;
        else throw Np2Error(NP2_E_DEVICE, "internal: loser outside the contig's reads");
    losers.clear();
    for (uint32_t r = 0; r < R; ++r)
        if (lost[r]) losers.push_back(r);
    return losers;
}

// consensus double buffer: cns_pos/cns_base (A) <-> cns_pos2/cns_base2 (B).  The length lives on the device
// (M_p); the host only carries an upper bound (M_cap) to size launches and buffers.
struct CnsDev {
    uint32_t *pos;
    uint8_t *base;
    const uint32_t *M_p;
    uint32_t M_cap;
};

// update_consensus_with_lqseqs on the device; returns the new consensus (in the other buffer).  No read-back: the
// new length is chained on the device into cx->mlen[version].
CnsDev splice_gpu(np2_ctx *cx, const CnsDev &in, uint32_t n_reg, uint8_t lable, uint32_t grow_bound, bool to_b,
                  uint32_t version) {
    hipStream_t s = cx->stream;
    EventTimer t(cx, "splice");
    cx->sp_idx_s.ensure(n_reg + 2);
    cx->sp_idx_e.ensure(n_reg + 2);
    cx->ap_g.ensure(n_reg + 2);
    cx->ap_s.ensure(n_reg + 2);
    cx->ap_e.ensure(n_reg + 2);
    cx->ap_delta.ensure(n_reg + 2);
    cx->ap_shift.ensure(n_reg + 2);
    cx->mlen.ensure(16);
    const uint32_t out_cap = in.M_cap + grow_bound;
    const size_t cap = (size_t)out_cap + 64;
    uint32_t *opos = to_b ? cx->cns_pos2.ensure(cap) : cx->cns_pos.ensure(cap);
    uint8_t *obase = to_b ? cx->cns_base2.ensure(cap) : cx->cns_base.ensure(cap);
    // per-round counters (stuck, n_ap) live behind the posted scalar block; run_diff zeroes them once per contig
    uint32_t *const stuck_p = cx->scal.p + S_COUNT + 2 * version, *const nap_p = stuck_p + 1;
    launch_splice_find(s, in.pos, in.M_p, cx->lq_start.p, cx->lq_end.p, cx->reg_lable.p, lable, n_reg, cx->sp_idx_s.p,
                       cx->sp_idx_e.p, stuck_p);
    launch_splice_plan(s, next_lookback(cx, region_lb_blocks(n_reg)), cx->reg_lable.p, lable, n_reg, stuck_p,
                       cx->sp_idx_s.p, cx->sp_idx_e.p, cx->seed_cand.p, cx->cand_seq_off.p, cx->ap_g.p, cx->ap_s.p,
                       cx->ap_e.p, cx->ap_delta.p, cx->ap_shift.p, nap_p, in.M_p, cx->mlen.p + version, out_cap,
                       cx->scal.p + S_ERR);
    launch_splice_write(s, in.pos, in.base, in.M_p, in.M_cap, cx->ap_g.p, cx->ap_s.p, cx->ap_e.p, cx->ap_delta.p,
                        cx->ap_shift.p, nap_p, n_reg, cx->lq_start.p, cx->seed_cand.p, cx->cand_seq_off.p,
                        cx->cand_seq.p, opos, obase);
    return CnsDev{opos, obase, cx->mlen.p + version, out_cap};
}

Cns fetch_cns_dev(np2_ctx *cx, const CnsDev &c) { // trace only
    Cns r;
    const uint32_t M = d2h(cx, c.M_p, 1)[0];
    r.pos = d2h(cx, c.pos, M);
    r.base = d2h(cx, c.base, M);
    return r;
}

// reupdate_consensus_with_lqseqs for one yak table, on the device
CnsDev recheck_gpu(np2_ctx *cx, const CnsDev &in, const PassCounts &pc, int yak_idx, uint16_t min_kmer_count,
                   bool first_yak, bool to_b, uint32_t version) {
    hipStream_t s = cx->stream;
    const uint32_t n_reg = pc.n_reg, ksize = cx->yaks[yak_idx].k;
    uint32_t n_rech = 0, n_groups = 0, n_jobs = 0;
    uint64_t blob_bound = 0; // upper bound of the recheck strings' bytes (sizes their buffer without a read-back)
    {
        // RECH region list, chain groups and job offsets: two look-back passes, then one read-back
        EventTimer t(cx, "recheck");
        cx->rech.ensure(n_reg + 2);
        cx->rech_groups.ensure((size_t)(n_reg + 2) * rech_group_bytes());
        cx->rech_joboff.ensure(n_reg + 2);
        cx->rech_groups_tmp.ensure((size_t)(n_reg + 2) * rech_group_bytes());
        cx->rech_headjobs.ensure(n_reg + 2);
        launch_rech_list(s, next_lookback(cx, region_lb_blocks(n_reg)), cx->reg_lable.p, n_reg, cx->rech.p, cx->scal.p + S_NRECH,
                         (unsigned long long *)(cx->scal.p + S_M1), cx->scal.p + S_ERR);
        launch_rech_groups(s, next_lookback(cx, region_lb_blocks(n_reg)), cx->rech.p, cx->scal.p + S_NRECH, n_reg, in.pos, in.M_p,
                           cx->lq_start.p, cx->lq_end.p, cx->keep_n.p, ksize, cx->reg_maxlen.p, cx->rech_groups_tmp.p,
                           cx->rech_headjobs.p, cx->rech_groups.p,
                           cx->rech_joboff.p, cx->scal.p + S_NGROUPS,
                           cx->scal.p + S_M0, (unsigned long long *)(cx->scal.p + S_M1), cx->scal.p + S_ERR);
        std::vector<uint32_t> sc = fetch_scal(cx);
        blob_bound = (uint64_t)sc[S_M1] | ((uint64_t)sc[S_M2] << 32);
        check_region_err(cx, sc[S_ERR]);
        n_rech = sc[S_NRECH];
        n_groups = sc[S_NGROUPS];
        n_jobs = sc[S_M0];
    }
    if (n_rech) {
        RechPtrs rp{cx->rech_groups.p, cx->rech_joboff.p, cx->rech.p, cx->cand_off.p, cx->keep_list.p,
                    cx->cand_seq_off.p, cx->cand_seq.p, in.base, n_groups};
        if (n_jobs) {
            EventTimer t(cx, "recheck");
            cx->job_len.ensure(n_jobs + 2);
            cx->job_off32.ensure(n_jobs + 2);
            cx->soff.ensure(n_jobs + 2);
            cx->sscore.ensure(n_jobs + 2);
            cx->tmp.ensure(prim_temp_bytes((size_t)n_jobs + 2));
            launch_rech_job_len(s, rp, n_jobs, cx->job_len.p);
            exclusive_total_n(cx, cx->job_len.p, cx->job_off32.p, n_jobs);
            if (blob_bound >= 0xFFFFFFF0ull) // 32-bit string offsets: let the exact total decide
                blob_bound = fetch_scal(cx, cx->scal.p + S_M0, cx->job_off32.p + n_jobs)[S_M0];
            cx->sstr.ensure((size_t)blob_bound + 64);
            launch_rech_job_build(s, rp, n_jobs, cx->job_off32.p, cx->soff.p, cx->sstr.p);
            launch_score_strings(s, cx->yaks[yak_idx].dev(), cx->sstr.p, cx->soff.p, n_jobs, min_kmer_count,
                                 cx->sscore.p, true);
            launch_rech_apply(s, rp, cx->sscore.p, cx->keep_ks.p);
        }
        EventTimer t(cx, "recheck");
        launch_rech_select(s, cx->rech.p, cx->scal.p + S_NRECH, n_rech, cx->cand_off.p, cx->keep_n.p, cx->keep_list.p,
                           cx->keep_ks.p, cx->cand_order.p, first_yak, cx->reg_lable.p, cx->seed_cand.p);
    }
    CnsDev out = splice_gpu(cx, in, n_reg, LB_RECH, pc.grow, to_b, version);
    launch_rech_relabel(s, cx->reg_lable.p, n_reg);
    return out;
}

void gpu_score_strings(np2_ctx *cx, int yak_idx, const std::vector<uint8_t> &blob, const std::vector<uint64_t> &off,
                       uint16_t min_kmer_count, std::vector<uint16_t> &scores) {
    const size_t n = off.size() - 1;
    scores.assign(n, 0);
    if (!n) return;
    cx->sstr.ensure(blob.size() + 16);
    cx->soff.ensure(off.size());
    cx->sscore.ensure(n);
    {
        op_sync(cx);
        uint8_t *pin = (uint8_t *)cx->pin_h2d.ensure(blob.size() + off.size() * 8 + 16);
        const size_t o2 = (blob.size() + 7) & ~(size_t)7;
        memcpy(pin, blob.data(), blob.size());
        memcpy(pin + o2, off.data(), off.size() * 8);
        op_h2d(cx, cx->sstr.p, pin, blob.size());
        op_h2d(cx, cx->soff.p, pin + o2, off.size() * 8);
        cx->h2d_inflight = true;
    }
    {
        EventTimer t(cx, "score_strings");
        launch_score_strings(cx->stream, cx->yaks[yak_idx].dev(), cx->sstr.p, cx->soff.p, n, min_kmer_count,
                             cx->sscore.p, false);
    }
    scores = d2h(cx, cx->sscore.p, n);
}

// ------------------------------------------------------------------------------------------
// the per-contig pipeline
// ------------------------------------------------------------------------------------------
struct PassOut {
    bool has_regions = false;
    Cns cns; // raw consensus of this pass (host copy, only when needed)
};

// Dense pass + exception sort.  Afterwards the sorted (node key, read) records of contig tile t are
// keys_raw / vals_raw [a, a + tile_n[t]) with a = t * cx->bucket_cap (bucketed layout, the normal case) or
// a = tile_scan[t] when cx->bucket_cap == 0 (compact layout after the device-wide sort).
void run_diff(np2_ctx *cx, np2_contig *c, uint32_t &T) {
    hipStream_t s = cx->stream;
    const uint32_t L = c->L, R = c->R, NCH = c->n_chunks;
    const uint32_t n_tiles = (L + TILE - 1) >> TILE_SHIFT;
    const uint32_t bcap = std::min(cx->tile_cap, c->tile_cap); // (cx: the NP2_TILE_CAP test hook)
    const uint64_t buckets = (uint64_t)n_tiles * bcap;
    uint64_t ovf_cap = std::max<uint64_t>(c->n_cols / 256 + 65536, 1u << 18); // spill area of full buckets
    cx->tmp.ensure(prim_temp_bytes(std::max<size_t>((size_t)NCH + 2, (size_t)L + 2)));
    cx->tile_n.ensure(n_tiles + 2);
    cx->tile_scan.ensure(n_tiles + 2);
    cx->tile_scanb.ensure(n_tiles + 2);
    cx->tile_nn.ensure(n_tiles + 2);
    cx->tile_nr.ensure(n_tiles + 2);
    cx->tile_noff.ensure(n_tiles + 2);
    cx->tile_roff.ensure(n_tiles + 2);
    cx->tile_gain.ensure(n_tiles + 2);
    if (cx->tile_cur.cap < (size_t)n_tiles + 2) { // the cursors are left zeroed by k_tile_layout; clear new storage
        cx->tile_cur.ensure(n_tiles + 2);
        zero32(cx, cx->tile_cur.p, cx->tile_cur.cap);
    }
    if (cx->chunk_st.cap < (size_t)NCH + 2) { // status words carry a launch epoch: cleared once, never again
        cx->chunk_st.ensure(NCH + 2);
        zero32(cx, cx->chunk_st.p, cx->chunk_st.cap, 8);
    }
    // chunk counts ahead of the dense pass (launch_chunk_counts): when this process shares the device with another one
    // (NP2_DENSE_PRECOUNT: set by the host side for ranks rehearsing on one GPU), or once a chunk's wait gave up
    static const bool precount_env = getenv("NP2_DENSE_PRECOUNT") != nullptr;
    bool precount = precount_env;
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (ovf_cap >= 0xFFFFFFF0ull) throw Np2Error(NP2_E_NOMEM, "too many exception nodes");
        cx->keys_raw.ensure(buckets + ovf_cap + 1);
        cx->vals_raw.ensure(buckets + ovf_cap + 1);
        zero32(cx, cx->scal.p, SCAL_TOTAL);
        if (++cx->chunk_epoch == 0) ++cx->chunk_epoch; // 0 = never written
        if (precount) launch_chunk_counts(s, c->descs.p, NCH, c->nib.p, cx->chunk_st.p, cx->chunk_epoch);
        {
            EventTimer t(cx, "diff_reads", true);
            launch_diff_reads(s, c->descs.p, NCH, c->nib.p, (const uint64_t *)c->refnib.p, c->refnib.p, c->refnib.p + c->ref_stride, c->ref_stride, L,
                              cx->keys_raw.p, cx->vals_raw.p, cx->tile_cur.p, n_tiles, bcap, buckets, (uint32_t)ovf_cap,
                              cx->scal.p + S_M3, c->ckpt.p, cx->chunk_st.p, cx->chunk_epoch, cx->scal.p + S_ERR);
        }
        static const uint32_t probe = getenv("NP2_DENSE_PROBE") ? (uint32_t)atoi(getenv("NP2_DENSE_PROBE")) : 0u;
        if (probe) { // (timing experiment: a part of the dense pass once more, tools/dense_probe.sh)
            EventTimer t(cx, "diff_probe", true);
            launch_diff_reads(s, c->descs.p, NCH, c->nib.p, (const uint64_t *)c->refnib.p, c->refnib.p, c->refnib.p + c->ref_stride, c->ref_stride, L,
                              cx->keys_raw.p, cx->vals_raw.p, cx->tile_cur.p, n_tiles, bcap, buckets, (uint32_t)ovf_cap,
                              cx->scal.p + S_M3, c->ckpt.p, cx->chunk_st.p, cx->chunk_epoch, cx->scal.p + S_ERR, probe);
        }
        {
            EventTimer t(cx, "sort_exceptions");
            // per-tile counts / layouts + the counters the host needs, all in the mailbox: S_M0 = T, S_M1 = largest
            // tile, S_M2 = records spilled from full buckets
            // (contigs of a few thousand tiles and more: a handful of blocks chained by the look-back)
            const bool wide = n_tiles >= 2048;
            Lookback lb{};
            if (wide) lb = next_lookback(cx, tile_scan_blocks(n_tiles));
            launch_tile_layout(s, cx->tile_cur.p, n_tiles, bcap, cx->tile_n.p, cx->tile_scan.p, cx->tile_scanb.p,
                               cx->scal.p + S_M3, cx->scal.p + S_M0, wide ? &lb : nullptr, cx->scal.p + S_ERR);
        }
        std::vector<uint32_t> sc = fetch_scal(cx);
        // (a wait that gave up — in the dense pass or in k_tile_layout — first: what follows it in the kernel is undefined,
        // the descriptor check included)
        static const bool test_fail = getenv("NP2_TEST_DENSE_LB_FAIL") != nullptr; // (test hook: the first attempt "gave up")
        if (test_fail && attempt == 0 && !precount) sc[S_ERR] |= LB_ERR;
        if ((sc[S_ERR] & LB_ERR) && !precount) { // (cursors, scalars and epoch start over with the next attempt)
            precount = true;
            continue;
        }
        if (sc[S_ERR] & ~2u) check_region_err(cx, sc[S_ERR] & ~2u);
        if (sc[S_ERR] & 2u)
            throw Np2Error(NP2_E_ARG, "packed read inconsistent with its descriptor (n_cols / aln_t_e / terminator)");
        if (sc[S_M2] > ovf_cap) { // the spill area itself overflowed: grow and redo the dense pass
            ovf_cap = (uint64_t)sc[S_M2] * 5 / 4 + 65536;
            continue;
        }
        T = sc[S_M0];
        if (T >= 0xFFFFFFF0u) throw Np2Error(NP2_E_NOMEM, "too many exception nodes");
        cx->tmp.ensure(prim_temp_bytes(std::max<size_t>({(size_t)T + 1, (size_t)L + 2, (size_t)R + 1})));
        EventTimer t(cx, "sort_exceptions");
        if (sc[S_M2] == 0) {
            // raw records -> node keys, sorted by (key, read) inside LDS, tile by tile, in place
            cx->tile_pidx.ensure((size_t)n_tiles * (TILE / 16) + 64);
            launch_tile_sort(s, cx->tile_pidx.p, c->reads.p, c->nib.p, cx->tile_n.p, n_tiles, bcap, sc[S_M1], cx->keys_raw.p,
                             cx->vals_raw.p, cx->scal.p + S_ERR);
            cx->bucket_cap = bcap;
            cx->pidx_valid = true;
        } else {
            // some tile holds more records than a bucket (e.g. a long insertion carried by every read): gather
            // buckets + spill area into one array and sort it device-wide
            cx->keys.ensure((size_t)T + 1);
            cx->vals.ensure((size_t)T + 1);
            const uint32_t nb = T - sc[S_M2];
            launch_gather_buckets(s, c->reads.p, c->nib.p, cx->tile_n.p, cx->tile_scanb.p, n_tiles, bcap, cx->keys_raw.p,
                                  cx->vals_raw.p, cx->keys.p, cx->vals.p);
            launch_gather_spill(s, c->reads.p, c->nib.p, cx->keys_raw.p + buckets, cx->vals_raw.p + buckets, sc[S_M2],
                                cx->keys.p + nb, cx->vals.p + nb);
            unsigned pos_bits = 1;
            while ((1ull << pos_bits) < (uint64_t)L + 1) ++pos_bits;
            prim_op(cx, [=](hipStream_t st) {
                if (prim_sort_pairs_u64_u32(st, cx->tmp.p, cx->tmp.cap, cx->keys.p, cx->keys_raw.p, cx->vals.p, cx->vals_raw.p,
                                            T, 32 + pos_bits))
                    throw Np2Error(NP2_E_DEVICE, "rocprim radix_sort_pairs failed");
            });
            cx->bucket_cap = 0;
            cx->pidx_valid = false;
        }
        return;
    }
    throw Np2Error(NP2_E_NOMEM, "exception buffer kept overflowing");
}

void build_graph(np2_ctx *cx, np2_contig *c, uint32_t T, uint32_t &n_nodes, uint32_t &n_runs) {
    hipStream_t s = cx->stream;
    const uint32_t L = c->L;
    const uint32_t n_tiles = (L + TILE - 1) >> TILE_SHIFT;
    EventTimer t(cx, "build_graph");
    cx->npos.ensure(T + 1);
    cx->nbases.ensure(T + 1);
    cx->ndelta.ensure(T + 1);
    cx->ncount.ensure(T + 1);
    cx->nminr.ensure(T + 1);
    cx->nscore.ensure(T + 1);
    cx->nbesti.ensure(T + 1);
    cx->nrec.ensure((size_t)T + 32); // the DP kernels prefetch a fixed number of records per position
    cx->node_off.ensure(L + 16); // k_dp_bt_short reads fixed windows past a run's start
    cx->cov.ensure(L + 16);
    cx->pflag.ensure((size_t)L + TILE + 16); // (k_tile_write stores four flags at a time, up to the end of the last tile)
    cx->run_start.ensure(L + 2);
    cx->run_end.ensure(L + 2);
    cx->emit.ensure(L + 2);
    NodeArrays nd{cx->npos.p, cx->nbases.p, cx->ndelta.p, cx->ncount.p, cx->nminr.p};
    launch_tile_count(s, cx->keys_raw.p, cx->vals_raw.p, cx->tile_n.p, cx->tile_scan.p, cx->bucket_cap, n_tiles,
                      cx->alive.p, cx->tile_nn.p, cx->tile_nr.p);
    // (also resets the per-pass scalars S_BEST .. S_NLQ)
    {
        const bool wide = n_tiles >= 2048;
        Lookback lb{};
        if (wide) lb = next_lookback(cx, tile_scan_blocks(n_tiles));
        launch_tile_offsets(s, cx->tile_nn.p, cx->tile_nr.p, n_tiles, cx->tile_noff.p, cx->tile_roff.p,
                            cx->scal.p + S_NNODES, cx->scal.p + S_NRUNS, cx->scal.p + S_BEST, S_NLQ + 1 - S_BEST,
                            wide ? &lb : nullptr, cx->scal.p + S_ERR);
    }
    launch_tile_write(s, cx->keys_raw.p, cx->vals_raw.p, cx->tile_n.p, cx->tile_scan.p, cx->bucket_cap, cx->tile_noff.p,
                      cx->tile_roff.p, n_tiles, cx->alive.p, L, nd, cx->nrec.p, cx->node_off.p, cx->run_start.p,
                      c->reads.p, c->tile_rd_off.p, c->tile_rd.p, cx->cov.p, c->refnib.p, cx->emit.p,
                      (long long *)cx->tile_gain.p, cx->scal.p + S_DEEP /* reset by launch_tile_offsets above */, cx->deep_min,
                      cx->pflag.p);
    // no read-back: downstream kernels are launched with the bound below and check the device-side counters
    n_runs = std::min<uint32_t>(T, L);
    n_nodes = T;
    if (cx->trace) n_nodes = d2h(cx, cx->scal.p + S_NNODES, 1)[0];
}

GraphPtrs graph_ptrs(np2_ctx *cx, np2_contig *c) {
    NodeArrays nd{cx->npos.p, cx->nbases.p, cx->ndelta.p, cx->ncount.p, cx->nminr.p};
    return GraphPtrs{c->refnib.p, cx->node_off.p, nd, cx->cov.p, c->L, cx->nrec.p, cx->scal.p + S_DEEP, cx->pflag.p};
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
// One read-back at the end: consensus length, region count, error word.  The consensus length stays on the device
// (eoff[L]) while the consensus and the LQ regions are built; launches and buffers are sized by M <= L + T.
struct CnsBounds {
    uint32_t M_cap, lq_cap, n_words;
};
CnsBounds cns_buffers(np2_ctx *cx, np2_contig *c, uint32_t n_nodes, uint32_t n_runs, uint32_t T) {
    const uint32_t L = c->L;
    if ((uint64_t)L + T + 2 >= 0xFFFFFFF0ull) throw Np2Error(NP2_E_NOMEM, "consensus bound exceeds 32 bits");
    const uint32_t M_cap = L + T + 2; // a path node emits at most one base; exception nodes <= T
    cx->eoff.ensure(L + 2);
    cx->cns_pos.ensure(M_cap + 2);
    cx->cns_base.ensure(M_cap + 2);
    cx->cns_cls.ensure(M_cap + 2);
    cx->lq_kind.ensure(M_cap + 2);
    cx->lq_next.ensure(M_cap + 2);
    cx->lq_nothead.ensure(M_cap + 2);
    cx->rflag.ensure(M_cap + 2);
    cx->rstart.ensure(M_cap + 2);
    cx->rend.ensure(M_cap + 2);
    cx->ridx.ensure(M_cap + 2);
    cx->raw_start.ensure(M_cap + 2);
    cx->raw_end.ensure(M_cap + 2);
    cx->headflag.ensure(M_cap + 2);
    cx->hidx.ensure(M_cap + 2);
    cx->lq_start.ensure(M_cap + 2);
    cx->lq_end.ensure(M_cap + 2);
    cx->tmp.ensure(prim_temp_bytes((size_t)M_cap + 2));
    // the low-quality bases (they only come out of dirty runs) as a list of consensus indices; a path through a run has
    // at most one entry per dirty position and per exception node
    const uint32_t lq_cap = (uint32_t)std::min<uint64_t>(2ull * n_nodes + n_runs + 2, M_cap);
    const uint32_t n_words = (M_cap + 31) / 32;
    cx->lq_list.ensure((size_t)lq_cap + 2);
    cx->hbits.ensure((size_t)n_words + 2);
    return CnsBounds{M_cap, lq_cap, n_words};
}
// raw LQ regions from the list of low-quality bases (n_lq of them, a device-side count): chain heads marked in a bitmap
// over the emission indices, heads per word scanned, regions written out in bit order (right -> left, the reference's
// numbering), then merged (main.rs:1613-1615)
void lq_regions_issue(np2_ctx *cx, const CnsBounds &b, const uint32_t *M_p, const uint32_t *n_lq) {
    hipStream_t s = cx->stream;
    EventTimer t(cx, "lq_regions");
    // (the head bitmap is cleared by k_lq_scan, its words counted inside the scan, the merge flags formed inside theirs:
    // three launches less per pass than one kernel per step)
    launch_lq_scan(s, cx->cns_pos.p, cx->cns_base.p, cx->cns_cls.p, M_p, cx->lq_list.p, n_lq, b.lq_cap, cx->lq_kind.p,
                   cx->lq_next.p, cx->lq_nothead.p, cx->hbits.p, b.n_words + 1, cx->rstart.p, cx->rend.p);
    launch_scan_lb_popc(s, next_lookback(cx, scan_lb_blocks((size_t)b.n_words + 1)), cx->hbits.p, cx->ridx.p, b.n_words,
                        cx->scal.p + S_ERR);
    launch_scatter_regions(s, cx->hbits.p, b.n_words, cx->ridx.p, cx->rstart.p, cx->rend.p, cx->raw_start.p,
                           cx->raw_end.p, cx->scal.p + S_NRAW);
    launch_lq_merge_scan_lb(s, next_lookback(cx, lq_merge_lb_blocks()), cx->raw_start.p, cx->raw_end.p, cx->scal.p + S_NRAW, b.M_cap,
                            cx->headflag.p, cx->hidx.p, cx->scal.p + S_ERR);
    // (the merged regions written by the single-block scan kernel itself measured slower: 52 + 15 -> 95 us per step)
    launch_lq_merge_write(s, cx->raw_start.p, cx->raw_end.p, cx->scal.p + S_NRAW, cx->headflag.p, cx->hidx.p,
                          cx->lq_start.p, cx->lq_end.p, cx->scal.p + S_NREG);
}

void consensus_and_regions_issue(np2_ctx *cx, np2_contig *c, uint32_t n_nodes, uint32_t n_runs, uint32_t T) {
    hipStream_t s = cx->stream;
    const uint32_t L = c->L;
    const CnsBounds cb = cns_buffers(cx, c, n_nodes, n_runs, T);
    const uint32_t M_cap = cb.M_cap, lq_cap = cb.lq_cap;
    GraphPtrs gp = graph_ptrs(cx, c);
    cx->n0_besti.ensure(L + 2);
    cx->run_gain.ensure((size_t)n_runs + 2);
    cx->run_flag.ensure((size_t)n_runs + 2);
    cx->emit.ensure(L + 2);
    cx->bt_path.ensure((size_t)M_cap + 2);
    cx->lqc.ensure((size_t)n_runs + 4);
    cx->lqoff.ensure((size_t)n_runs + 4);
    const uint32_t *const n_lq = cx->lqoff.p + n_runs; // total of the scanned per-run counts (n_runs: host-side bound)
    const uint32_t *M_p = cx->eoff.p + L;
    {
        EventTimer t(cx, "dp_backtrack");
        // (scalars were zeroed by build_graph; S_GAIN already holds the clean-position gains)
        // long runs on the second stream, short runs on the main one: the kernels touch disjoint runs and the long
        // kernel is a latency chain (a few lanes walking runs of dozens of positions) that would otherwise sit alone
        // on the device for as long as the short kernel takes
        // The short kernel goes first and lists the runs it leaves alone, so that the long-run kernels walk that list
        // instead of classifying every run again.
        // (NP2_DP_FORK: the earlier scheme — short and long-run kernels side by side on two streams, each classifying
        // the runs itself; measured ~1 % slower on the E. coli-sized contig once the short kernel had become the
        // shorter of the two; kept as a tested alternative)
        const bool forked = tl_recorder() == nullptr && cx->hooks.dp_fork;
        uint32_t *dp_list = nullptr, *n_dp_list = nullptr;
        if (forked) {
            HIPCHK(hipEventRecord(cx->ev_fork, s));
            HIPCHK(hipStreamWaitEvent(cx->stream2, cx->ev_fork, 0));
        } else {
            cx->dp_list.ensure((size_t)n_runs + 2);
            dp_list = cx->dp_list.p, n_dp_list = cx->scal.p + S_NDPLIST; // (zeroed with the per-pass scalars)
            launch_dp_short(s, gp, c->refnib.p, cx->run_start.p, cx->scal.p + S_NRUNS, n_runs, cx->run_end.p,
                            cx->run_gain.p, cx->emit.p, cx->scal.p + S_PATHBEGIN, cx->bt_path.p, dp_list, n_dp_list);
        }
        launch_dp_long(forked ? cx->stream2 : s, gp, cx->run_start.p, cx->scal.p + S_NRUNS, n_runs, cx->nrec.p, cx->nscore.p,
                       cx->nbesti.p, cx->n0_besti.p, cx->run_end.p, (int64_t *)(cx->scal.p + S_LAST0), cx->run_gain.p,
                       cx->emit.p, cx->scal.p + S_PATHBEGIN, cx->bt_path.p, cx->run_flag.p, dp_list, n_dp_list);
        if (forked) {
            HIPCHK(hipEventRecord(cx->ev_join, cx->stream2));
            launch_dp_short(s, gp, c->refnib.p, cx->run_start.p, cx->scal.p + S_NRUNS, n_runs, cx->run_end.p,
                            cx->run_gain.p, cx->emit.p, cx->scal.p + S_PATHBEGIN, cx->bt_path.p, nullptr, nullptr);
            HIPCHK(hipStreamWaitEvent(s, cx->ev_join, 0));
        }
        launch_dp_finish(s, gp, cx->run_start.p, cx->scal.p + S_NRUNS, cx->nscore.p, cx->nbesti.p, cx->n0_besti.p,
                         (const int64_t *)(cx->scal.p + S_LAST0), (unsigned long long *)(cx->scal.p + S_GAIN0),
                         cx->scal.p + S_DUP /* block counter: reset with the per-pass scalars */, cx->scal.p + S_BEST,
                         cx->run_gain.p, (const long long *)cx->tile_gain.p,
                         (c->L + TILE - 1) >> TILE_SHIFT, cx->emit.p, cx->scal.p + S_PATHBEGIN, cx->bt_path.p);
        exclusive_total(cx, cx->emit.p, cx->eoff.p, (size_t)L + 1);
        launch_bt_write(s, gp, cx->run_start.p, cx->scal.p + S_NRUNS, n_runs, cx->emit.p, cx->eoff.p, cx->bt_path.p,
                        cx->cns_pos.p, cx->cns_base.p, cx->cns_cls.p, cx->lq_nothead.p, cx->lqc.p);
        exclusive_total(cx, cx->lqc.p, cx->lqoff.p, (size_t)n_runs + 1); // (lqc[n_runs] = 0)
        launch_lq_list(s, gp, cx->run_start.p, cx->scal.p + S_NRUNS, n_runs, cx->emit.p, cx->eoff.p, cx->bt_path.p,
                       cx->lqoff.p, lq_cap, cx->lq_list.p, cx->scal.p + S_ERR);
        launch_default_tail(s, gp, cx->scal.p + S_BEST, M_p, cx->cns_base.p, cx->cns_cls.p, cx->lq_list.p,
                            cx->lqoff.p + n_runs, lq_cap, cx->scal.p + S_ERR);
    }
    lq_regions_issue(cx, cb, M_p, n_lq);
}

// The same stage through the fused pass front (np2_passfront.hip): records of a tile -> the tile's piece of the consensus
// in one kernel, one scan over the per-tile counts, one compaction; no graph arrays in memory.  Needs the bucketed record
// layout with k_tile_sort's position index.  A pass the fused kernels cannot hold (PF_REDO) or whose best path score is
// negative is redone by the kernels above (pass_front_finish).
bool front_can_fuse(np2_ctx *cx) {
    return !cx->hooks.front_unfused && cx->bucket_cap != 0 && cx->pidx_valid;
}
void pass_front_fused_issue(np2_ctx *cx, np2_contig *c, uint32_t T) {
    hipStream_t s = cx->stream;
    const uint32_t L = c->L;
    const uint32_t n_tiles = (L + TILE - 1) >> TILE_SHIFT;
    const CnsBounds cb = cns_buffers(cx, c, T, std::min<uint32_t>(T, L), T);
    cx->pf_slots.ensure(pf_slot_entries(n_tiles, T));
    cx->pf_bad.ensure((size_t)n_tiles + 2);
    cx->pf_bad2.ensure((size_t)n_tiles + 2);
    // (test hooks: lower the LDS variants' limits so that small inputs take the big variant / the unfused redo)
    const np2_ctx::Hooks &hk = cx->hooks;
    const uint32_t cap_lim = hk.has_pf_cap ? hk.pf_cap : PF_CAP, cap_big = hk.has_pf_cap_big ? hk.pf_cap_big : PF_CAP_BIG,
                   halo_lim = (hk.has_pf_halo ? hk.pf_halo : PF_HALO) & ~15u, cov_max = hk.has_pf_cov_max ? hk.pf_cov_max : PF_COV_MAX;
    PfTile a{cx->keys_raw.p, cx->vals_raw.p, cx->tile_n.p, cx->tile_scan.p, cx->tile_pidx.p, cx->alive.p, c->reads.p,
             c->tile_rd_off.p, c->tile_rd.p, c->refnib.p, cx->pf_slots.p, cx->tile_nn.p, cx->tile_nr.p,
             (long long *)cx->tile_gain.p, cx->scal.p + S_PF, cx->scal.p + S_NBAD, cx->pf_bad.p, cx->scal.p + S_NBAD2, cx->pf_bad2.p,
             (long long *)(cx->scal.p + S_PFEND0), (unsigned long long *)(cx->scal.p + S_PFGAIN0), nullptr, L, n_tiles, cx->bucket_cap,
             cap_lim, cap_big, halo_lim, std::min(cov_max, cx->deep_min), cx->pf_big ? 1u : 0u};
    uint32_t *const M_p = cx->eoff.p + L;
    uint32_t *const n_lq = cx->scal.p + S_NRUNS;
    const bool prof = cx->hooks.pf_prof && tl_recorder() == nullptr; // (phase timers of the tile kernel: a tool's switch)
    if (prof) {
        cx->pf_prof.ensure((size_t)n_tiles * 8 + 8);
        zero32(cx, cx->pf_prof.p, (size_t)n_tiles * 8, 8);
        a.prof = (unsigned long long *)cx->pf_prof.p;
    }
    {
        EventTimer t(cx, "pass_front");
        launch_pf_tile(s, a);
        if (prof) {
            auto st = d2h(cx, cx->pf_prof.p, (size_t)n_tiles * 8);
            double sum[8] = {0}, mx[8] = {0};
            size_t cnt = 0;
            for (uint32_t t = 0; t < n_tiles; ++t) {
                const uint64_t *q = st.data() + (size_t)t * 8;
                if (!q[7] || !q[0]) continue; // (a tile the small variant did not finish)
                ++cnt;
                for (int i = 1; i < 8; ++i) {
                    const double d = (double)(q[i] - q[i - 1]);
                    sum[i] += d, mx[i] = std::max(mx[i], d);
                }
            }
            {
                auto tn = d2h(cx, cx->tile_n.p, n_tiles);
                uint32_t h[6] = {0};
                for (uint32_t v : tn) ++h[v <= 480 ? 0 : v <= 960 ? 1 : v <= 1440 ? 2 : v <= 1920 ? 3 : v <= 3584 ? 4 : 5];
                fprintf(stderr, "[pf_prof] records per tile <=480 / 960 / 1440 / 1920 / 3584 / more: %u %u %u %u %u %u; ", h[0], h[1], h[2], h[3], h[4], h[5]);
            }
            fprintf(stderr, "listed for the middle / the big variant: %u / %u; ", d2h(cx, cx->scal.p + S_NBAD, 1)[0], d2h(cx, cx->scal.p + S_NBAD2, 1)[0]);
            fprintf(stderr, "tiles %zu of %u; mean / max clocks per phase (loads, cover+nodes, offsets+sort, DP, reduce, count+scan, write):", cnt, n_tiles);
            for (int i = 1; i < 8; ++i) fprintf(stderr, " %.0f/%.0f", cnt ? sum[i] / cnt : 0.0, mx[i]);
            fprintf(stderr, "\n");
        }
        // consensus offset and low-quality offset of every tile, their totals (consensus length -> eoff[L]), the total of the
        // gains; also resets the per-pass scalars S_BEST .. S_NLQ like the unfused build
        const bool wide = n_tiles >= 2048;
        Lookback lb{};
        if (wide) lb = next_lookback(cx, tile_scan_blocks(n_tiles));
        launch_tile_offsets(s, cx->tile_nn.p, cx->tile_nr.p, n_tiles, cx->tile_noff.p, cx->tile_roff.p, M_p, n_lq,
                            cx->scal.p + S_BEST, S_NLQ + 1 - S_BEST, wide ? &lb : nullptr, cx->scal.p + S_ERR,
                            (const long long *)cx->tile_gain.p, (unsigned long long *)(cx->scal.p + S_PFGAIN0));
        launch_pf_compact(s, n_tiles, cx->pf_slots.p, cx->tile_scan.p, cx->tile_nn.p, cx->tile_noff.p, cx->tile_roff.p,
                          cx->cns_pos.p, cx->cns_base.p, cx->cns_cls.p, cx->lq_nothead.p, cx->lq_list.p, cb.lq_cap,
                          cx->scal.p + S_ERR, cx->scal.p + S_PF, cx->scal.p + S_PFOUT, cx->scal.p + S_NBAD, cx->scal.p + S_NBAD2);
    }
    lq_regions_issue(cx, cb, M_p, n_lq);
}

// graph + consensus + LQ regions of a pass, issued (no wait): fused where possible
void pass_front_issue(np2_ctx *cx, np2_contig *c, uint32_t T, int pass, bool force_unfused = false) {
    cx->front_fused = !force_unfused && front_can_fuse(cx);
    if (!cx->front_fused || cx->trace) { // (stage traces of the graph come from the unfused build)
        uint32_t n_nodes = 0, n_runs = 0;
        {
            WallTimer w(cx, "wall_graph");
            build_graph(cx, c, T, n_nodes, n_runs);
        }
        trace_graph(cx, c, pass, n_nodes);
        if (!cx->front_fused) {
            consensus_and_regions_issue(cx, c, n_nodes, n_runs, T);
            return;
        }
    }
    pass_front_fused_issue(cx, c, T);
}
// ... and the read-back that ends the stage: consensus length, region count, the error word
void consensus_and_regions_finish(np2_ctx *cx, np2_contig *c, uint32_t T, int pass, uint32_t &M, uint32_t &n_reg, bool whole_contig = true) {
    const uint32_t *M_p = cx->eoff.p + c->L;
    std::vector<uint32_t> sc = fetch_scal(cx, cx->scal.p + S_M0, M_p);
    if (cx->front_fused) {
        const int64_t total = (int64_t)(((uint64_t)sc[S_PFGAIN1] << 32) | sc[S_PFGAIN0]);
        const int64_t end_rel = (int64_t)(((uint64_t)sc[S_PFEND1] << 32) | sc[S_PFEND0]);
        const bool negative = end_rel <= SCORE_NEG / 2 || total + end_rel < 0; // main.rs:1651,1680: no end node reaches 0
        if (sc[S_PFOUT] & PF_NEED_BIG) cx->pf_big = true; // (sticky: the next fused pass of this context launches it)
        if ((sc[S_PFOUT] & PF_REDO) || negative) {
            ++cx->front_redos;
            cx->timing.host.push_back({"front_redo", 1.0f}); // (a count, read through np2_last_timings by the tests)
            pass_front_issue(cx, c, T, pass, true);
            sc = fetch_scal(cx, cx->scal.p + S_M0, M_p);
        }
    }
    check_region_err(cx, sc[S_ERR]);
    // No end node at score >= 0: k_dp_finish / k_default_tail have walked back from the reference's default node
    // (main.rs:1651,1680).  Whether the score is negative is a property of the WHOLE contig: a shard cannot tell.
    if (!cx->front_fused && sc[S_BEST] == 0xFFFFFFFFu && !whole_contig)
        throw Np2Error(NP2_E_UNSUPPORTED,
                       "best path score is negative at the end of a shard (the reference's default node is chosen by the whole "
                       "contig's score): polish this contig unsharded");
    M = sc[S_M0];
    if (M == 0) throw Np2Error(NP2_E_REFPANIC, "reference would panic: empty consensus");
    n_reg = sc[S_NRAW] ? sc[S_NREG] : 0;
}

// candidate extraction + first-yak scoring; fills the host RegionSet
void extract_candidates(np2_ctx *cx, np2_contig *c, uint32_t n_reg, uint16_t min_kmer_count, int pass,
                        PassCounts &pc) {
    hipStream_t s = cx->stream;
    const uint32_t R = c->R;
    REFPANIC_IF(cx->yaks.empty(), "index out of bounds: opt.yak[0]");
    if ((uint64_t)n_reg * LQSEQ_MAX_CAN_COUNT >= 0xFFFFFFF0ull) throw Np2Error(NP2_E_NOMEM, "too many LQ regions");
    const uint32_t NC_cap = n_reg * LQSEQ_MAX_CAN_COUNT;
    // candidate strings are disjoint pieces of the reads: at most the pileup's columns.  A chromosome-sized contig does not
    // allocate by that bound (7.7 GB for a 248 Mb contig, 1 % of it used: half a second of the context's first polish) but
    // reads the exact byte count back once the offsets are known — a read-back of its own, ~60 us, which a contig that
    // size does not notice and a yeast-sized batch would
    static const uint64_t exact_from = getenv("NP2_CAND_EXACT_FROM") ? strtoull(getenv("NP2_CAND_EXACT_FROM"), nullptr, 10) : (1ull << 28); // (tests lower it)
    const bool exact_bytes = c->n_cols >= exact_from && !cx->trace;
    uint32_t SB_cap = (uint32_t)std::min<uint64_t>(c->n_cols + 64, 0xFFFFFFF0ull);
    cx->mval.ensure(R + 2);
    cx->smin.ensure(R + 2);
    cx->pj.ensure(R + 2);
    cx->pcount.ensure(R + 2);
    cx->rinfo.ensure(R + 2);
    cx->reg_ncand.ensure(n_reg + 2);
    cx->reg_bytes.ensure(n_reg + 2);
    cx->reg_soff.ensure(n_reg + 2);
    cx->reg_maxlen.ensure(n_reg + 2);
    cx->blk_sum.ensure(3 * ((size_t)n_reg / 4 + 2));
    cx->blk_coff.ensure((size_t)n_reg / 4 + 2);
    cx->blk_soff.ensure((size_t)n_reg / 4 + 2);
    cx->cand_off.ensure(n_reg + 2);
    cx->kept_read.ensure((size_t)NC_cap + 2);
    cx->kept_len.ensure((size_t)NC_cap + 2);
    cx->kept_col.ensure((size_t)NC_cap + 2);
    cx->cand_order.ensure((size_t)NC_cap + 2);
    cx->cand_kmer.ensure((size_t)NC_cap + 2);
    cx->cand_seq_off.ensure((size_t)NC_cap + 2);
    if (!exact_bytes) cx->cand_seq.ensure((size_t)SB_cap + 64);
    cx->kscore.ensure((size_t)NC_cap + 2);
    cx->long_list.ensure((size_t)NC_cap + 2);
    CandPtrs cp{c->reads.p,   c->nib.p,   c->ck_off.p,      c->ckpt.p,    cx->lq_start.p, cx->lq_end.p, cx->pj.p,
                cx->pcount.p, cx->alive.p, cx->rinfo.p, c->tile_rd_off.p, c->tile_rd.p, c->n_tiles, cx->yaks[0].k,
                cx->keys_raw.p, cx->vals_raw.p, cx->tile_n.p,
                (cx->pidx_valid && cx->bucket_cap && !cx->hooks.cand_decode_all) ? cx->tile_pidx.p : nullptr, cx->bucket_cap,
                (const uint32_t *)c->refnib.p, c->L};
    {
        EventTimer t(cx, "candidates");
        // every live read covers a contiguous interval [pj, pj + pcount) of the region list
        // (both in one single-block kernel measured 58 us against 7.5 + 6: a block's worth of binary searches is a latency chain)
        launch_read_m(s, c->reads.p, R, cx->alive.p, cx->lq_start.p, n_reg, cx->mval.p);
        scan_incl_min(cx, cx->mval.p, cx->smin.p, R);
        launch_pair_count(s, c->reads.p, R, cx->alive.p, cx->lq_start.p, cx->lq_end.p, n_reg, cx->smin.p, cx->pj.p,
                          cx->pcount.p, c->ck_off.p, cx->rinfo.p);
        // one wavefront per region: find its reads, measure the candidates, keep the first 60 non-empty ones
        launch_region_measure(s, cp, n_reg, cx->kept_read.p, cx->kept_len.p, cx->kept_col.p, cx->reg_ncand.p,
                              cx->reg_bytes.p, cx->reg_maxlen.p, cx->blk_sum.p);
        {
            // (the chained variant whenever there is a region at all: a threshold had the contigs of a batch pick different
            // kernels here, and everything up to the vote went out twice, once per half of the batch)
            const bool wide = n_reg > 0;
            Lookback lb{};
            if (wide) lb = next_lookback(cx, cand_offsets_blocks(n_reg));
            launch_cand_offsets(s, cx->blk_sum.p, n_reg, cx->blk_coff.p, cx->blk_soff.p, cx->cand_off.p, cx->reg_soff.p,
                                cx->scal.p + S_NC, cx->scal.p + S_SB, cx->scal.p + S_GROW, wide ? &lb : nullptr,
                                cx->scal.p + S_ERR);
        }
        if (exact_bytes) {
            const std::vector<uint32_t> sc = fetch_scal(cx);
            check_region_err(cx, sc[S_ERR]);
            pc.resolve(sc);
            SB_cap = (uint32_t)std::min<uint64_t>((uint64_t)pc.SB + 64, SB_cap);
            cx->cand_seq.ensure((size_t)SB_cap + 64);
        }
        launch_region_write(s, cp, n_reg, cx->kept_read.p, cx->kept_len.p, cx->kept_col.p, cx->reg_ncand.p,
                            cx->reg_bytes.p, cx->blk_coff.p, cx->blk_soff.p, cx->cand_off.p, cx->reg_soff.p, NC_cap + 1,
                            SB_cap, cx->cand_order.p, cx->cand_kmer.p, cx->cand_seq_off.p, cx->cand_seq.p);
    }
    {
        EventTimer t(cx, "kmer_score");
        launch_cand_score(s, cx->yaks[0].dev(), cx->cand_seq_off.p, cx->cand_seq.p, cx->cand_kmer.p, cx->scal.p + S_NC,
                          NC_cap, min_kmer_count, cx->kscore.p, cx->long_list.p, cx->scal.p + S_NLONG);
    }
    pc.n_reg = n_reg;
    pc.NC_cap = NC_cap;
    pc.SB_cap = SB_cap;
    pc.known = exact_bytes; // (NC / SB / grow read back above)
    if (cx->trace) pc.resolve(fetch_scal(cx));
    trace_region_tables(cx, pass, "cand", pc, false);
}

Cns fetch_cns(np2_ctx *cx, uint32_t M) {
    Cns c;
    c.pos = d2h(cx, cx->cns_pos.p, M);
    c.base = d2h(cx, cx->cns_base.p, M);
    return c;
}

struct ResultOut {
    uint8_t *bases = nullptr;
    uint32_t *pos = nullptr;
    uint64_t len = 0;
    bool want_pos = true;
    bool want_bases = true; // false: the sequence stays on the device (np2_last_result_device / np2_result_fetch_begin)
};
// M_cap: what the host knows the length is at most (the device buffers hold that many elements).  The sequence is copied
// out up to that bound in the SAME wait that brings the length back (one device round trip per contig less than
// "length first, then the copy"; the bound exceeds the length by the splice rounds' growth allowance, a few per cent).
void fetch_result(np2_ctx *cx, const uint32_t *dpos, const uint8_t *dbase, const uint32_t *M_p, uint32_t M_cap, ResultOut &r,
                  uint32_t M_likely = 0xFFFFFFFFu) {
    const double t0 = now_ms();
    // final length + the error word of everything that ran without a read-back since the last one
    // (the same post carries the first and the last consensus position: the FASTA header span)
    std::vector<uint32_t> sc;
    // (a second copy of the bases, device to device, at its device-side length: into the caller's sink — np2_batch_set_sink —,
    // done when this call's last wait returns)
    if (cx->sink_dst && cx->sink_cap) launch_copy_len(cx->stream, cx->sink_dst, dbase, M_p, 1, cx->sink_cap);
    if (r.want_bases || r.want_pos) {
        const uint32_t seq = ++cx->mbox_seq;
        launch_post(cx->stream, cx->scal.p, S_COUNT, cx->mbox_dev, seq, cx->scal.p + S_M0, M_p, nullptr, nullptr, nullptr, nullptr,
                    nullptr, nullptr, dpos, cx->scal.p + S_M1);
        if (r.want_bases) r.bases = (uint8_t *)pinned_pool().get((size_t)M_cap + 1);
        if (r.want_pos) r.pos = (uint32_t *)pinned_pool().get(((size_t)M_cap + 1) * 4);
        if ((r.want_bases && !r.bases) || (r.want_pos && !r.pos)) throw Np2Error(NP2_E_NOMEM, "pinned result allocation failed");
        // (M_likely: a tighter guess of the length — every splice round reserves the full growth allowance, a region
        // is spliced in one of them —; should the sequence be longer after all, its rest follows in a second copy)
        // (under the batch driver the copy kernel reads the length on the device and moves exactly that)
        uint32_t first = std::min(M_cap, M_likely);
        const bool exact_b = r.want_bases && op_d2h_len(cx, r.bases, dbase, M_p, 1, M_cap);
        const bool exact_p = r.want_pos && op_d2h_len(cx, r.pos, dpos, M_p, 4, (size_t)M_cap * 4);
        if (r.want_bases && !exact_b) op_d2h(cx, r.bases, dbase, first);
        if (r.want_pos && !exact_p) op_d2h(cx, r.pos, dpos, (size_t)first * 4);
        if ((exact_b || !r.want_bases) && (exact_p || !r.want_pos)) first = M_cap; // (nothing left for a second copy)
        op_sync(cx);
        if (__atomic_load_n(&cx->mbox_host[0], __ATOMIC_ACQUIRE) != seq)
            throw Np2Error(NP2_E_DEVICE, "mailbox not posted after a synchronisation");
        sc.assign(cx->mbox_host + 1, cx->mbox_host + 1 + S_COUNT);
        if (sc[S_M0] > first && sc[S_M0] <= M_cap) {
            const uint32_t rest = sc[S_M0] - first;
            if (r.want_bases) op_d2h(cx, r.bases + first, dbase + first, rest);
            if (r.want_pos) op_d2h(cx, r.pos + first, dpos + first, (size_t)rest * 4);
            op_sync(cx);
        }
    } else {
        sc = fetch_scal(cx, cx->scal.p + S_M0, M_p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, dpos, cx->scal.p + S_M1);
    }
    check_region_err(cx, sc[S_ERR]);
    const uint32_t M = sc[S_M0];
    if (M == 0) throw Np2Error(NP2_E_REFPANIC, "reference would panic: empty consensus");
    if (M > M_cap) throw Np2Error(NP2_E_DEVICE, "internal: consensus longer than its bound");
    r.len = M;
    cx->last_first_pos = sc[S_M1];
    cx->last_last_pos = sc[S_M2];
    cx->last_dbase = dbase;
    cx->last_dpos = dpos;
    cx->last_len = M;
    if (cx->stage_timing) cx->timing.host.push_back({"wall_fetch_result", (float)(now_ms() - t0)});
}

// The per-contig loop (main.rs:1819-1836) as a stepper: begin (dense pass), then for every pass either
// vote_pass + apply_losers (a phasing pass) or final_pass.  np2_polish_resident drives it straight through; the shards
// of one contig (np2_shard_*) stop after vote_pass, exchange their votes, and continue with the contig-wide losers.
struct PolishRun {
    np2_ctx *cx = nullptr;
    np2_contig *c = nullptr;
    np2_opts_t o{};
    uint32_t own_lo = 0, own_hi = 0xFFFFFFFFu; // regions this run votes over (sub-contig coordinates)
    uint32_t T = 0, pass = 0, M = 0, n_reg = 0;
    bool reuse = false;
    bool front_issued = false; // graph, DP, consensus and LQ regions of pass `pass` are already on their way (polish_impl)
    uint32_t grow_prev = 0xFFFFFFFFu; // growth bound of the previous (phasing) pass, read back with its vote for free
    bool wide_votes = false; // the shards of a contig export (a << 32 | b, counts) pairs; the plain pipeline reads compact rows
    uint64_t key_add = 0;    // ... with the shard's read-number shift added on the device (s << 32 | s)
    PassCounts pc;
    bool final_pass() const { return pass + 1 == o.iter_count; }
};

void run_begin(PolishRun &r) {
    np2_ctx *cx = r.cx;
    if (r.o.iter_count < 1) throw Np2Error(NP2_E_ARG, "iter_count must be >= 1");
    HIPCHK(hipSetDevice(cx->device));
    cx->trace_items.clear();
    cx->last_dbase = nullptr;
    cx->last_dpos = nullptr;
    cx->scal.ensure(SCAL_TOTAL);
    cx->alive.ensure(r.c->R + 2);
    {
        WallTimer w(cx, "wall_diff");
        run_diff(cx, r.c, r.T);
    }
    launch_init_alive(cx->stream, r.c->reads.p, r.c->R, cx->alive.p);
    r.pass = 0;
    r.reuse = false;
}

// graph, consensus, LQ regions and candidates of pass r.pass.  If a phasing pass voted out no read, this pass would
// rebuild a byte-identical graph, consensus, region list and candidate table (they are pure functions of the pileup and
// the live-read set): reuse them.  The reference recomputes (main.rs:1819-1836); the result is identical by construction.
void run_pass_front(PolishRun &r) {
    np2_ctx *cx = r.cx;
    np2_contig *c = r.c;
    if (!r.reuse) {
        {
            WallTimer w(cx, "wall_cns_lq");
            if (!r.front_issued) pass_front_issue(cx, c, r.T, (int)r.pass);
            r.front_issued = false;
            consensus_and_regions_finish(cx, c, r.T, (int)r.pass, r.M, r.n_reg, !r.wide_votes);
        }
        if (cx->trace) {
            trace_cns(cx, (int)r.pass, "cns_raw", fetch_cns(cx, r.M));
            trace_put(cx, (int)r.pass, "lq.start", d2h(cx, cx->lq_start.p, r.n_reg));
            trace_put(cx, (int)r.pass, "lq.end", d2h(cx, cx->lq_end.p, r.n_reg));
        }
    }
    if (r.n_reg == 0) return;
    if (!r.reuse) {
        r.pc = PassCounts();
        r.pc.M = r.M;
        WallTimer w(cx, "wall_extract");
        extract_candidates(cx, c, r.n_reg, r.o.min_kmer_count, (int)r.pass, r.pc);
    } else if (r.pc.NC_cap) { // undo mark_hete's kscore edits of the previous (identical) pass
        op_copy_d2d(cx, cx->kscore.p, cx->kscore_saved.p, (size_t)(r.pc.known ? r.pc.NC : r.pc.NC_cap) * 2);
    }
}

// a phasing pass up to the votes (vd.any == false: nobody votes)
void run_vote_pass(PolishRun &r, VoteData &vd, const std::function<void()> *after_front = nullptr) {
    np2_ctx *cx = r.cx;
    vd = VoteData();
    vd.R = r.c->R;
    run_pass_front(r);
    if (after_front) (*after_front)(); // (polish_impl: the previous pass's vote is settled before this pass votes)
    if (r.n_reg == 0) return;
    WallTimer w(cx, "wall_vote");
    const bool may_reuse = cx->reuse_identical_pass && !cx->trace;
    if (may_reuse && r.pc.NC_cap) {
        cx->kscore_saved.ensure((size_t)r.pc.NC_cap + 2);
        op_copy_d2d(cx, cx->kscore_saved.p, cx->kscore.p, (size_t)(r.pc.known ? r.pc.NC : r.pc.NC_cap) * 2);
    }
    vote_collect(cx, r.c, r.pc, r.o.model_ref != 0, r.o.use_all_reads != 0, (int)r.pass, r.own_lo, r.own_hi, vd, r.wide_votes, r.key_add);
    if (r.pc.known) r.grow_prev = r.pc.grow;
}

// the reads the vote removed (align_bases = empty, main.rs:1548-1550), then on to the next pass
void run_apply_losers(PolishRun &r, const std::vector<uint32_t> &losers) {
    np2_ctx *cx = r.cx;
    if (r.n_reg) trace_put(cx, (int)r.pass, "invalid_ids", losers); // (a pass without regions never reaches the vote)
    for (uint32_t id : losers) REFPANIC_IF(id >= r.c->R, "index out of bounds: alignseqs[id]");
    const bool may_reuse = cx->reuse_identical_pass && !cx->trace;
    r.reuse = false;
    if (!losers.empty()) {
        cx->kill_ids.ensure(losers.size());
        h2d_staged(cx, cx->kill_ids.p, losers.data(), losers.size() * 4);
        launch_kill_reads(cx->stream, cx->kill_ids.p, (uint32_t)losers.size(), cx->alive.p);
        // (no wait: the ids sit in the pinned staging buffer, guarded by h2d_inflight)
    } else if (may_reuse) {
        r.reuse = true; // (also when the pass had no region at all: no read was voted out)
    }
    ++r.pass;
}

void run_final_pass(PolishRun &r, ResultOut &result, bool exact_grow = false) {
    np2_ctx *cx = r.cx;
    np2_contig *c = r.c;
    hipStream_t s = cx->stream;
    const uint32_t n_reg_prev = r.n_reg;
    (void)n_reg_prev;
    run_pass_front(r);
    const uint32_t n_reg = r.n_reg;
    if (n_reg == 0) {
        fetch_result(cx, cx->cns_pos.p, cx->cns_base.p, cx->eoff.p + c->L, r.M, result);
        return;
    }
    PassCounts &pc = r.pc;
    WallTimer w(cx, "wall_final");
    RegionTables rt = region_tables(cx, n_reg);
    cx->reg_lable.ensure(n_reg + 2);
    cx->seed_cand.ensure(n_reg + 2);
    cx->keep_n.ensure(n_reg + 2);
    // The splice rounds need an allowance for the consensus' growth (the sum of the regions' longest kept candidates, on
    // the device: S_GROW).  Reading it back is a device round trip of its own; after a phasing pass its value there is
    // known — the final pass has fewer reads and mostly fewer regions — so twice that, guarded on the device, is used
    // instead, and the pass is repeated with the exact figure should a round outgrow it (never observed).
    const bool no_guess = cx->hooks.exact_grow;
    bool guessed = false;
    if (!pc.known) {
        if (r.grow_prev != 0xFFFFFFFFu && !exact_grow && !no_guess && !cx->trace && (uint64_t)r.grow_prev * 2 + 4096 < 0x7FFFFFFFull) {
            pc.grow = r.grow_prev * 2 + 4096;
            if (cx->hooks.has_grow_guess) pc.grow = cx->hooks.grow_guess; // test hook: force the retry
            guessed = true;
        } else {
            pc.resolve(fetch_scal(cx));
        }
    }
    try {
        cx->keep_list.ensure((size_t)pc.NC_cap + 2);
        cx->keep_ks.ensure((size_t)pc.NC_cap + 2);
        {
            EventTimer t(cx, "seed");
            launch_seed(s, rt, r.o.max_indel_len, cx->reg_lable.p, cx->seed_cand.p, cx->keep_n.p, cx->keep_list.p,
                        cx->keep_ks.p, cx->scal.p + S_ERR);
        }
        if (cx->trace) check_region_err(cx, d2h(cx, cx->scal.p + S_ERR, 1)[0]);
        trace_region_tables(cx, (int)r.pass, "seed", pc, true);
        CnsDev cur{cx->cns_pos.p, cx->cns_base.p, cx->eoff.p + c->L, r.M}; // eoff[L] = length of the raw consensus
        bool to_b = true;
        uint32_t version = 0;
        cur = splice_gpu(cx, cur, n_reg, LB_SUCC, pc.grow, to_b, version++);
        to_b = !to_b;
        if (cx->trace) trace_cns(cx, (int)r.pass, "cns_succ", fetch_cns_dev(cx, cur));
        for (size_t y = 0; y < cx->yaks.size(); ++y) {
            cur = recheck_gpu(cx, cur, pc, (int)y, r.o.min_kmer_count, y == 0, to_b, version++);
            to_b = !to_b;
            trace_region_tables(cx, (int)r.pass, "rech" + std::to_string(y), pc, true);
            if (cx->trace) trace_cns(cx, (int)r.pass, "cns_rech" + std::to_string(y), fetch_cns_dev(cx, cur));
        }
        fetch_result(cx, cur.pos, cur.base, cur.M_p, cur.M_cap, result, guessed ? r.M + r.grow_prev : r.M + pc.grow);
    } catch (const GrowRetry &) {
        if (!guessed) throw;
        if (result.bases) pinned_pool().put(result.bases), result.bases = nullptr;
        if (result.pos) pinned_pool().put(result.pos), result.pos = nullptr;
        op_fill(cx, cx->scal.p + S_ERR, 0, 4); // (every other bit of the word would have been reported first)
        r.reuse = false;
        run_final_pass(r, result, true);
    }
}

void polish_impl(np2_ctx *cx, np2_contig *c, const np2_opts_t *o, ResultOut &result) {
    PolishRun r;
    r.cx = cx, r.c = c, r.o = *o;
    run_begin(r);
    const bool no_spec = cx->hooks.no_speculate;
    const bool use_all = o->use_all_reads != 0;
    // The host side of the vote (key order, rows, Louvain: ~1 ms for a 1.5 Mb diploid contig, 100 ms for a chromosome) is
    // the longest host phase of a contig, and the device has nothing to do for it meanwhile.  Without -r the vote kernel
    // has already flagged the reads that disagree with the contig at a marker (main.rs:977); the Louvain only adds the
    // reads of conflicting communities AMONG the others, which is rare (never on the synthetic diploid workloads: the
    // removed reads are exactly the flagged ones).  So the vote is decided on a helper thread (a contig on its own
    // context) or next to the first part of the pass (batch driver) while this pipeline goes on
    // with the next pass on the flagged reads alone: its graph, DP, consensus and LQ regions go to the device at once
    // (a flush nobody waits for in the batch driver), the rest of the pass follows; the decision is awaited before the
    // next vote is collected, or before the result is handed out, and if it removes other reads as well the pass is
    // simply done again with them (its results are a pure function of the live reads).
    struct Decision {
        std::vector<uint32_t> losers;
        double ms = 0;
    };
    struct Pending {
        std::future<Decision> fut;
        std::shared_ptr<VoteData> vd;
        size_t n_bad = 0;
        bool active = false;
    } pend;
    // -> true if the pass under way is the right one; false: the reads the decision removes beyond the flagged ones are
    // taken out now and the pass has to be done again
    auto settle = [&]() -> bool {
        if (!pend.active) return true;
        pend.active = false;
        Decision d = pend.fut.get(); // (rethrows what the decision threw: the reference's panics)
        cx->timing.host.push_back({"wall_louvain", (float)d.ms});
        std::vector<uint32_t> extra;
        for (uint32_t id : d.losers) {
            REFPANIC_IF(id >= c->R, "index out of bounds: alignseqs[id]");
            if (!pend.vd->bad[id]) extra.push_back(id);
        }
        if (extra.empty() && d.losers.size() == pend.n_bad && !cx->hooks.test_misspeculate) return true; // (test hook)
        if (!extra.empty()) {
            cx->kill_ids.ensure(extra.size() + 1);
            h2d_staged(cx, cx->kill_ids.p, extra.data(), extra.size() * 4);
            launch_kill_reads(cx->stream, cx->kill_ids.p, (uint32_t)extra.size(), cx->alive.p);
        }
        r.reuse = false;
        r.front_issued = false;
        return false;
    };
    const std::function<void()> settle_then_redo = [&]() {
        if (!settle()) run_pass_front(r); // (again, on the right reads)
    };
    while (!r.final_pass()) {
        auto vd = std::make_shared<VoteData>();
        run_vote_pass(r, *vd, &settle_then_redo);
        size_t n_bad = 0;
        if (vd->any && vd->d_bad && !use_all && !cx->trace && !no_spec)
            for (uint8_t b : vd->bad) n_bad += b;
        if (n_bad == 0) {
            run_apply_losers(r, vote_decide(cx, *vd, use_all));
            continue;
        }
        launch_kill_flagged(cx->stream, vd->d_bad, c->R, cx->alive.p);
        ++r.pass; // (what run_apply_losers does; the pass cannot be a reuse of the last one: reads are going)
        r.reuse = false;
        // (the tile kernel's phase clocks are read back inside pass_front_issue — through the very staging the pairs still sit
        // in, which may move: tools/pf_prof.py died there)
        if (cx->hooks.pf_prof) vd->own();
        pass_front_issue(cx, c, r.T, (int)r.pass);
        r.front_issued = true;
        op_submit(cx);
        vd->own(); // (the pairs sit in the context's read-back staging, which the pipeline goes on using)
        pend.vd = vd;
        pend.n_bad = n_bad;
        pend.active = true;
        auto decide = [vd, use_all]() {
            Decision d;
            const double t0 = now_ms();
            d.losers = vote_decide(nullptr, *vd, use_all);
            d.ms = now_ms() - t0;
            return d;
        };
        if (tl_recorder() == nullptr) {
            // one contig on its own context (a chromosome: ~100 ms of host vote next to ~50 ms of device work for the pass)
            pend.fut = std::async(std::launch::async, decide);
        } else {
            // Under the batch driver the decision is taken here and now, next to the part of the pass that is already on
            // its way: with four batch groups the device is the busy part of a step, and seventeen more host threads
            // deciding votes beside the pipelines' own cost more than the rest of the overlap gave (3.72 ms per
            // yeast-sized assembly this way, 3.9 - 4.1 with helper threads; 4.06 without any early start).
            std::promise<Decision> p;
            pend.fut = p.get_future();
            try {
                p.set_value(decide());
            } catch (...) {
                p.set_exception(std::current_exception());
            }
            (void)settle(); // (a decision that removes more reads: the pass is started again by the next run_pass_front)
        }
    }
    for (;;) {
        run_final_pass(r, result);
        if (settle()) break;
        if (result.bases) pinned_pool().put(result.bases), result.bases = nullptr;
        if (result.pos) pinned_pool().put(result.pos), result.pos = nullptr;
    }
}


// ---- shards of one contig ----------------------------------------------------------------------------------------------
// A shard polishes the sub-contig [sub_lo, sub_hi) that holds every read overlapping its owned interval widened by a halo,
// as if it were a contig of its own; inside [own_lo - halo, own_hi + halo) every read of the whole contig is present, so
// graph, DP (exact between clean positions), LQ regions, candidates and recheck chains there are those of
// the whole contig.  What is not local: the phasing vote (collected per shard over the regions it owns, decided once per
// contig on the merged votes, main.rs:948-1015) and the reads it removes (applied everywhere).
struct ShardRun {
    PolishRun run;
    np2_shard_plan_t plan{};
    uint32_t verify = 0; // positions beyond the owned interval that are emitted as well (checked by the stitcher)
    // last exported vote (borrowed by the caller until the next call)
    std::vector<uint64_t> v_key;
    std::vector<uint32_t> v_cnt, v_read, v_first;
    std::vector<int32_t> v_refw;
    std::vector<uint8_t> v_flags;
    // np2_shard_final_device: the owned slice of the device-resident result
    uint64_t own_off = 0, own_len = 0;
    bool have_piece = false;
    // the pass started on the reads the vote kernel flagged while the ranks' votes are merged and decided (np2_shard_vote /
    // np2_shard_apply): the flags it went by, local read numbers
    bool spec = false;
    std::vector<uint8_t> spec_bad;
    size_t spec_n_bad = 0;
};
inline uint32_t shard_global_read(const np2_shard_plan_t &pl, uint32_t local) { return local == 0 ? 0u : pl.read_lo + local - 1; }

// A splice cursor that got stuck (update_consensus_with_lqseqs, main.rs:1036-1056) leaves every region to its right
// untouched — contig-wide: a shard cannot reproduce that on its own, whichever shard sees it (the first one included:
// the shards to its right would go on splicing).  Only a region stuck beyond the right end of the zone is harmless:
// coverage is partial there (an artefact of the cut) and everything it freezes lies outside of what this shard emits.
void shard_check_stuck(ShardRun *sr) {
    np2_ctx *cx = sr->run.cx;
    const np2_shard_plan_t &pl = sr->plan;
    if (!sr->run.n_reg) return;
    std::vector<uint32_t> rounds = d2h(cx, cx->scal.p + S_COUNT, 2 * (cx->yaks.size() + 1));
    uint32_t worst = 0; // highest region index + 1 = the leftmost stuck region over all rounds
    for (size_t v = 0; v < rounds.size(); v += 2) worst = std::max(worst, rounds[v]);
    if (!worst) return;
    if (worst > sr->run.n_reg) throw Np2Error(NP2_E_DEVICE, "internal: stuck region index out of range");
    const uint32_t at = d2h(cx, cx->lq_start.p + (worst - 1), 1)[0];
    if ((uint64_t)at + pl.sub_lo < pl.zone_hi)
        throw Np2Error(NP2_E_UNSUPPORTED, "splice cursor stuck inside a shard: polish this contig unsharded");
}

} // namespace



namespace np2h {
int fail(np2_ctx *cx, const Np2Error &e) {
    if (cx) cx->err = e.what();
    return e.code;
}

void finish_contig(np2_ctx *cx, np2_contig *c, const np2_read_t *reads, uint32_t n_reads, uint32_t L,
                   uint64_t nib_bytes) {
    if (reads[0].aln_t_s != 0 || reads[0].n_cols != L || reads[0].aln_t_e != L - 1 ||
        (reads[0].flags & NP2_READ_DROPPED))
        throw Np2Error(NP2_E_ARG, "reads[0] must be the contig aligned to itself (main.rs:1732-1739)");
    // what goes to the device is built in pinned blocks of the process-wide pool (pageable sources are staged by the
    // runtime at a fraction of the bus rate, synchronously: 4 MB of descriptors per E. coli-sized contig)
    struct PinnedTmp {
        void *p = nullptr;
        explicit PinnedTmp(size_t bytes) : p(pinned_pool().get(std::max<size_t>(bytes, 64))) {
            if (!p) throw Np2Error(NP2_E_NOMEM, "hipHostMalloc failed");
        }
        ~PinnedTmp() {
            if (std::uncaught_exceptions() > 0) (void)hipDeviceSynchronize(); // (an upload out of it may still be in flight)
            pinned_pool().put(p);
        }
    };
    uint64_t total_chunks = 0;
    for (uint32_t r = 1; r < n_reads; ++r)
        if (!(reads[r].flags & NP2_READ_DROPPED)) total_chunks += ((uint64_t)reads[r].n_cols + DENSE_COLS - 1) / DENSE_COLS;
    if (total_chunks >= 0xFFFFFFF0ull) throw Np2Error(NP2_E_NOMEM, "too many pileup columns for one contig");
    PinnedTmp ck_pin((size_t)(n_reads + 1) * 8), descs_pin((size_t)(total_chunks + 1) * sizeof(ChunkDesc));
    uint64_t *ck = (uint64_t *)ck_pin.p;
    ChunkDesc *descs = (ChunkDesc *)descs_pin.p;
    uint32_t n_descs = 0;
    ck[0] = 0;
    uint64_t cols = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        const np2_read_t &rd = reads[r];
        const bool dropped = rd.flags & NP2_READ_DROPPED;
        if (rd.nib_off & 15) throw Np2Error(NP2_E_ARG, "nib_off must be a multiple of 16");
        if (rd.nib_off >> 36) throw Np2Error(NP2_E_NOMEM, "a contig's nibble streams must stay below 64 GiB");
        if (!dropped && (rd.aln_t_e >= L || rd.aln_t_s > rd.aln_t_e))
            throw Np2Error(NP2_E_ARG, "read span outside the contig");
        if (rd.nib_off + ((uint64_t)(rd.n_cols + 1) >> 1) + 1 + 16 > nib_bytes)
            throw Np2Error(NP2_E_ARG, "nibble stream (plus 16 B tail padding) exceeds the buffer");
        uint32_t nck = 0, nch = 0;
        if (!dropped) {
            const uint32_t first = (rd.aln_t_s + CKPT - 1) >> CKPT_SHIFT, last = rd.aln_t_e >> CKPT_SHIFT;
            nck = last >= first ? last - first + 1 : 0;
            if (r != 0) nch = (uint32_t)(((uint64_t)rd.n_cols + DENSE_COLS - 1) / DENSE_COLS);
            cols += rd.n_cols;
        }
        ck[r + 1] = ck[r] + nck;
        const uint32_t first_chunk = n_descs;
        for (uint32_t k = 0; k < nch; ++k) {
            ChunkDesc &d = descs[n_descs++];
            memset(&d, 0, sizeof d);
            d.nib_off = rd.nib_off;
            d.ckbase = ck[r];
            d.read = r;
            d.ts = rd.aln_t_s;
            d.c0 = k * DENSE_COLS;
            d.ncols = rd.n_cols;
            d.first_chunk = first_chunk;
            d.aln_t_e = rd.aln_t_e;
            d.nck = nck;
        }
    }
    // reads overlapping each contig tile, ascending read index (candidate extraction looks reads up by position)
    const uint32_t n_tiles = (L + TILE - 1) >> TILE_SHIFT;
    PinnedTmp trd_off_pin(((size_t)n_tiles + 1) * 4);
    uint32_t *trd_off = (uint32_t *)trd_off_pin.p;
    memset(trd_off, 0, ((size_t)n_tiles + 1) * 4);
    for (uint32_t r = 0; r < n_reads; ++r)
        if (!(reads[r].flags & NP2_READ_DROPPED))
            for (uint32_t t = reads[r].aln_t_s >> TILE_SHIFT; t <= reads[r].aln_t_e >> TILE_SHIFT; ++t) ++trd_off[t + 1];
    for (uint32_t t = 0; t < n_tiles; ++t) trd_off[t + 1] += trd_off[t];
    const size_t n_trd = trd_off[n_tiles];
    PinnedTmp trd_pin((n_trd + 1) * 4);
    uint32_t *trd = (uint32_t *)trd_pin.p;
    {
        std::vector<uint32_t> cur(trd_off, trd_off + n_tiles);
        for (uint32_t r = 0; r < n_reads; ++r)
            if (!(reads[r].flags & NP2_READ_DROPPED))
                for (uint32_t t = reads[r].aln_t_s >> TILE_SHIFT; t <= reads[r].aln_t_e >> TILE_SHIFT; ++t) trd[cur[t]++] = r;
    }
    {
        uint32_t deepest = 0;
        for (uint32_t t = 0; t < n_tiles; ++t) deepest = std::max(deepest, trd_off[t + 1] - trd_off[t]);
        const uint64_t want = ((uint64_t)deepest * TILE / 24 + 255) & ~255ull; // ~4 % of the tile's columns
        c->tile_cap = (uint32_t)std::min<uint64_t>(TILE_CAP, std::max<uint64_t>(1024, want));
    }
    hipStream_t s = cx->stream;
    c->L = L;
    c->R = n_reads;
    c->n_tiles = n_tiles;
    c->tile_rd_off.ensure((size_t)n_tiles + 1);
    c->tile_rd.ensure(n_trd + 1);
    HIPCHK(hipMemcpyAsync(c->tile_rd_off.p, trd_off, ((size_t)n_tiles + 1) * 4, hipMemcpyHostToDevice, s));
    if (n_trd) HIPCHK(hipMemcpyAsync(c->tile_rd.p, trd, n_trd * 4, hipMemcpyHostToDevice, s));
    c->nib_bytes = nib_bytes;
    c->n_cols = cols;
    c->n_ckpt = ck[n_reads];
    if (c->n_ckpt >> 32) throw Np2Error(NP2_E_NOMEM, "too many pileup columns for one contig (checkpoint offsets are 32-bit)");
    c->n_chunks = n_descs;
    c->reads.ensure(n_reads);
    const uint32_t refbytes = ((L + 1) >> 1) + 96; // padding so that 128-bit probes near the end stay in bounds
    c->ref_stride = (refbytes + 15) & ~15u; // (the dense pass's two copies follow the codes: launch_encode_ref)
    c->refnib.ensure(3 * (size_t)c->ref_stride);
    c->ck_off.ensure(n_reads + 1);
    c->ckpt.ensure(c->n_ckpt + 1);
    c->descs.ensure(c->n_chunks + 1);
    HIPCHK(hipMemcpyAsync(c->reads.p, reads, (size_t)n_reads * sizeof(np2_read_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->ck_off.p, ck, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, s));
    if (c->n_chunks)
        HIPCHK(hipMemcpyAsync(c->descs.p, descs, (size_t)c->n_chunks * sizeof(ChunkDesc), hipMemcpyHostToDevice, s));
    cx->scal.ensure(SCAL_TOTAL);
    zero32(cx, cx->scal.p, 24);
    launch_encode_ref(s, c->nib.p + reads[0].nib_off, L, c->refnib.p, refbytes, c->ref_stride, cx->scal.p);
    auto sc = d2h(cx, cx->scal.p, 1); // also syncs: the pinned staging blocks above go back to the pool
    if (sc[0]) throw Np2Error(NP2_E_ARG, "reads[0] is not a plain self-alignment of the contig");
}
} // namespace np2h

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

// the context's streams, events and mailbox go back to the pool (synchronised), its result staging to the pinned pool
static void destroy_streams(np2_ctx *cx) {
    for (int i = 0; i < 2; ++i) {
        if (cx->out_host[i]) pinned_pool().put(cx->out_host[i]);
        cx->out_host[i] = nullptr;
        cx->out_host_cap[i] = 0;
    }
    if (cx->borrowed_state) { // (streams and mailbox are the batch driver's)
        if (cx->ev_out) (void)hipEventDestroy(cx->ev_out);
        if (cx->ev_fork) (void)hipEventDestroy(cx->ev_fork);
        if (cx->ev_join) (void)hipEventDestroy(cx->ev_join);
        cx->stream = cx->stream2 = cx->stream_out = nullptr;
        cx->ev_out = cx->ev_fork = cx->ev_join = nullptr;
        cx->mbox_host = cx->mbox_dev = nullptr;
        return;
    }
    CtxDeviceState st;
    st.stream = cx->stream, st.stream2 = cx->stream2, st.stream_out = cx->stream_out;
    st.ev_out = cx->ev_out, st.ev_fork = cx->ev_fork, st.ev_join = cx->ev_join;
    st.mbox_host = cx->mbox_host, st.mbox_dev = cx->mbox_dev;
    cx->stream = cx->stream2 = cx->stream_out = nullptr;
    cx->ev_out = cx->ev_fork = cx->ev_join = nullptr;
    cx->mbox_host = cx->mbox_dev = nullptr;
    const bool complete = st.stream && st.stream2 && st.stream_out && st.ev_out && st.ev_fork && st.ev_join && st.mbox_host;
    bool idle = complete;
    if (complete) // (nothing of this context may still be running on a set the next context takes over)
        idle = hipStreamSynchronize(st.stream) == hipSuccess && hipStreamSynchronize(st.stream2) == hipSuccess &&
               hipStreamSynchronize(st.stream_out) == hipSuccess;
    if (idle)
        ctx_state_pool().put(cx->device, st);
    else
        st.destroy();
}

// streams, events, mailbox: everything of a context but its k-mer tables
static void init_ctx_device(np2_ctx *cx, int device, hipStream_t borrow = nullptr, uint32_t *mbox_host = nullptr, uint32_t *mbox_dev = nullptr) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw Np2Error(NP2_E_DEVICE, "no HIP device available (the np2 hot path has no CPU fallback)");
    if (device < 0 || device >= ndev) throw Np2Error(NP2_E_ARG, "bad device index");
    cx->device = device;
    cx->hooks.read();
    HIPCHK(hipSetDevice(device));
    CtxDeviceState st;
    if (borrow) {
        cx->borrowed_state = true;
        cx->stream = cx->stream2 = cx->stream_out = borrow;
        cx->mbox_host = mbox_host, cx->mbox_dev = mbox_dev;
        HIPCHK(hipEventCreateWithFlags(&cx->ev_out, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&cx->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&cx->ev_join, hipEventDisableTiming));
    } else if (ctx_state_pool().get(device, st)) {
        cx->stream = st.stream, cx->stream2 = st.stream2, cx->stream_out = st.stream_out;
        cx->ev_out = st.ev_out, cx->ev_fork = st.ev_fork, cx->ev_join = st.ev_join;
        cx->mbox_host = st.mbox_host, cx->mbox_dev = st.mbox_dev;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&cx->stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&cx->stream2, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&cx->stream_out, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&cx->ev_out, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&cx->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&cx->ev_join, hipEventDisableTiming));
        HIPCHK(hipHostMalloc((void **)&cx->mbox_host, 64 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipHostGetDevicePointer((void **)&cx->mbox_dev, cx->mbox_host, 0));
    }
    cx->scal.ensure(SCAL_TOTAL);
    memset(cx->mbox_host, 0, 64 * sizeof(uint32_t));
    if (const char *e = getenv("NP2_TILE_CAP")) // test hook: smaller buckets force the spill / device-wide sort path
        cx->tile_cap = (uint32_t)std::min<long>(TILE_CAP, std::max<long>(1, atol(e)));
    if (const char *e = getenv("NP2_TEST_DEEP_COV")) // test hook: treat shallower pileups as too deep for the on-chip DP
        cx->deep_min = (uint32_t)std::min<long>(65536, std::max<long>(1, atol(e)));
}

int np2_ctx_create(np2_ctx_t **out, int device, const np2_yak_t *yaks, int n_yak) {
    if (!out) return NP2_E_ARG;
    *out = nullptr;
    np2_ctx *cx = new np2_ctx();
    try {
        const double t_c0 = now_ms();
        init_ctx_device(cx, device);
        if (getenv("NP2_CTX_PROFILE")) fprintf(stderr, "np2_ctx_create: streams, events, mailbox %.2f ms\n", now_ms() - t_c0);
        // the final pass runs one splice round + one per yak table; their per-round device counters live in fixed slots
        if (n_yak < 0 || n_yak > NP2_MAX_YAK || (n_yak && !yaks))
            throw Np2Error(NP2_E_ARG, "n_yak must be in [0, 15]");
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
            t.table = std::make_shared<DevBuf<uint64_t>>();
            t.table->ensure(slots);
            HIPCHK(hipMemsetAsync(t.table->p, 0xFF, slots * 8, cx->stream));
            HIPCHK(hipStreamSynchronize(cx->stream)); // large fill + pageable H2D below: do not rely on their ordering
            DevBuf<uint64_t> dw, doff;
            dw.ensure(y.n_words + 1);
            doff.ensure(1025);
            HIPCHK(hipMemcpyAsync(dw.p, y.words, y.n_words * 8, hipMemcpyHostToDevice, cx->stream));
            HIPCHK(hipMemcpyAsync(doff.p, y.bucket_off, 1025 * 8, hipMemcpyHostToDevice, cx->stream));
            zero32(cx, cx->scal.p, S_COUNT);
            launch_yak_insert(cx->stream, dw.p, doff.p, 1024, mx, t.table->p, cl, cx->scal.p + S_DUP);
            auto sc = d2h(cx, cx->scal.p, S_COUNT);
            if (sc[S_DUP]) { // a repeated key (yak writes none): a slot per word, the winner chosen at lookup (kmer.rs:148-167)
                t.ord = std::make_shared<DevBuf<uint32_t>>();
                t.ord->ensure(slots);
                HIPCHK(hipMemsetAsync(t.table->p, 0xFF, slots * 8, cx->stream));
                launch_yak_insert_dup(cx->stream, dw.p, doff.p, 1024, mx, t.table->p, cl, t.ord->p);
                HIPCHK(hipStreamSynchronize(cx->stream)); // (dw / doff are released at the end of this scope)
            }
        }
    } catch (const Np2Error &e) {
        fprintf(stderr, "np2_ctx_create: %s\n", e.what());
        int code = e.code;
        destroy_streams(cx);
        delete cx;
        return code;
    } catch (const std::exception &ex) {
        fprintf(stderr, "np2_ctx_create: %s\n", ex.what());
        int code = NP2_E_NOMEM;
        destroy_streams(cx);
        delete cx;
        return code;
    }
    *out = cx;
    return NP2_OK;
}

int np2_ctx_create_shared(np2_ctx_t **out, np2_ctx_t *parent) {
    if (!out || !parent) return NP2_E_ARG;
    *out = nullptr;
    np2_ctx *cx = new np2_ctx();
    try {
        init_ctx_device(cx, parent->device);
        cx->yaks = parent->yaks; // the HBM tables are reference-counted: freed with the last context using them
        cx->tile_cap = parent->tile_cap;
    } catch (const Np2Error &e) {
        fprintf(stderr, "np2_ctx_create_shared: %s\n", e.what());
        int code = e.code;
        destroy_streams(cx);
        delete cx;
        return code;
    }
    *out = cx;
    return NP2_OK;
}

} // extern "C"
// a batch driver's slot: the tables shared with `parent`, the stream and the mailbox the driver's (np2_batch.cpp)
np2_ctx *np2h::ctx_create_slot(np2_ctx *parent, hipStream_t s, uint32_t *mbox_host, uint32_t *mbox_dev) {
    np2_ctx *cx = new np2_ctx();
    try {
        init_ctx_device(cx, parent->device, s, mbox_host, mbox_dev);
        cx->yaks = parent->yaks;
        cx->tile_cap = parent->tile_cap;
    } catch (...) {
        destroy_streams(cx);
        delete cx;
        throw;
    }
    return cx;
}
void np2h::ctx_slot_set_stream(np2_ctx *cx, hipStream_t s) {
    if (cx->borrowed_state) cx->stream = cx->stream2 = cx->stream_out = s;
}
extern "C" {

void np2_ctx_destroy(np2_ctx_t *cx) {
    if (!cx) return;
    const bool prof = getenv("NP2_CTX_PROFILE") != nullptr;
    const double t0 = now_ms();
    (void)hipSetDevice(cx->device);
    if (cx->stream) (void)hipStreamSynchronize(cx->stream);
    if (cx->stream2) (void)hipStreamSynchronize(cx->stream2);
    if (cx->stream_out) (void)hipStreamSynchronize(cx->stream_out);
    const double t1 = now_ms();
    destroy_streams(cx);
    const double t2 = now_ms();
    {
        DevSyncScope idle; // the context's ~130 buffers go back to the slabs / the cache behind ONE device synchronisation
        delete cx;
    }
    if (prof)
        fprintf(stderr, "np2_ctx_destroy: stream syncs %.2f ms, streams/events/result staging %.2f ms, buffers %.2f ms\n", t1 - t0,
                t2 - t1, now_ms() - t2);
}
const char *np2_last_error(np2_ctx_t *cx) { return cx ? cx->err.c_str() : "null context"; }
void *np2_ctx_stream(np2_ctx_t *cx) { return cx ? (void *)cx->stream : nullptr; }
void np2_ctx_set_trace(np2_ctx_t *cx, int enable) {
    if (cx) cx->trace = enable != 0;
}
void np2_ctx_set_timing(np2_ctx_t *cx, int enable) {
    if (cx) cx->stage_timing = enable != 0;
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
        c->nib.ensure(nib_bytes + 64);
        // (the caller's pageable buffer in one call: the runtime locks its pages and copies at the link's rate — 56 GB/s for
        // 32 MiB and more, tools/ubench_h2d.hip; a ring of pinned blocks filled by helper threads measured 38-51)
        HIPCHK(hipMemcpyAsync(c->nib.p, nibbles, nib_bytes, hipMemcpyHostToDevice, cx->stream));
        finish_contig(cx, c, reads, n_reads, L, nib_bytes);
    } catch (const Np2Error &e) {
        (void)hipStreamSynchronize(cx->stream); // (the nibble upload may still be in flight: the blocks go back to the cache)
        delete c;
        return fail(cx, e);
    } catch (const std::exception &ex) {
        (void)hipStreamSynchronize(cx->stream);
        delete c;
        return fail(cx, Np2Error(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what()));
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
    if (!cx || !c || !opts || !out_len) return NP2_E_ARG;
    ResultOut r;
    r.want_pos = out_pos != nullptr;
    r.want_bases = out_bases != nullptr;
    const double t_wall0 = now_ms(), t_cpu0 = thread_cpu_ms();
    try {
        polish_impl(cx, c, opts, r);
        cx->timing.host.push_back({"wall_polish", (float)(now_ms() - t_wall0)});
        cx->timing.host.push_back({"cpu_polish", (float)(thread_cpu_ms() - t_cpu0)});
        flush_timings(cx);
    } catch (const Np2Error &e) {
        (void)hipStreamSynchronize(cx->stream);
        if (cx->stream2) (void)hipStreamSynchronize(cx->stream2); // (a failure between fork and join leaves work there)
        flush_timings(cx);
        if (r.bases) pinned_pool().put(r.bases);
        if (r.pos) pinned_pool().put(r.pos);
        return fail(cx, e);
    } catch (const std::exception &ex) {
        (void)hipStreamSynchronize(cx->stream);
        if (cx->stream2) (void)hipStreamSynchronize(cx->stream2); // (a failure between fork and join leaves work there)
        flush_timings(cx);
        if (r.bases) pinned_pool().put(r.bases);
        if (r.pos) pinned_pool().put(r.pos);
        return fail(cx, Np2Error(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what()));
    }
    *out_len = r.len;
    if (out_bases) *out_bases = r.bases;
    if (out_pos) *out_pos = r.pos;
    return NP2_OK;
}

int np2_result_fetch_begin(np2_ctx_t *cx) {
    if (!cx || !cx->last_dbase || cx->out_pending) return NP2_E_ARG;
    try {
        HIPCHK(hipSetDevice(cx->device));
        const size_t n = cx->last_len;
        cx->out_snap.ensure(n + 1);
        uint8_t *&host = cx->out_host[cx->out_slot];
        if (cx->out_host_cap[cx->out_slot] < n + 1) {
            if (host) pinned_pool().put(host);
            host = nullptr;
            cx->out_host_cap[cx->out_slot] = 0;
            const size_t cap = n + n / 8 + 4096;
            host = (uint8_t *)pinned_pool().get(cap);
            if (!host) throw Np2Error(NP2_E_NOMEM, "pinned result allocation failed");
            cx->out_host_cap[cx->out_slot] = cap;
        }
        // snapshot on the main stream (in order before the next contig overwrites the consensus buffers), host copy on
        // the output stream behind it
        HIPCHK(hipMemcpyAsync(cx->out_snap.p, cx->last_dbase, n, hipMemcpyDeviceToDevice, cx->stream));
        HIPCHK(hipEventRecord(cx->ev_out, cx->stream));
        HIPCHK(hipStreamWaitEvent(cx->stream_out, cx->ev_out, 0));
        HIPCHK(hipMemcpyAsync(host, cx->out_snap.p, n, hipMemcpyDeviceToHost, cx->stream_out));
        cx->out_len = n;
        cx->out_pending = true;
    } catch (const Np2Error &e) {
        return fail(cx, e);
    }
    return NP2_OK;
}

int np2_result_fetch_end(np2_ctx_t *cx, const uint8_t **bases, uint64_t *len) {
    if (!cx || !bases || !len || !cx->out_pending) return NP2_E_ARG;
    try {
        HIPCHK(hipStreamSynchronize(cx->stream_out));
    } catch (const Np2Error &e) {
        return fail(cx, e);
    }
    *bases = cx->out_host[cx->out_slot];
    *len = cx->out_len;
    cx->out_slot ^= 1;
    cx->out_pending = false;
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
void np2_free(void *p) {
    if (!p) return;
    if (!pinned_pool().put(p)) free(p);
}

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
    } catch (const std::exception &ex) {
        return fail(cx, Np2Error(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what()));
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
    } catch (const std::exception &ex) {
        return fail(cx, Np2Error(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what()));
    }
    return NP2_OK;
}

int np2_last_span(np2_ctx_t *cx, uint32_t *first_pos, uint32_t *last_pos) {
    if (!cx || !first_pos || !last_pos) return NP2_E_ARG;
    *first_pos = cx->last_first_pos;
    *last_pos = cx->last_last_pos;
    return NP2_OK;
}

int np2_last_result_device(np2_ctx_t *cx, const uint8_t **dev_bases, uint64_t *len) {
    if (!cx || !dev_bases || !len) return NP2_E_ARG;
    if (!cx->last_dbase) return NP2_E_ARG;
    *dev_bases = cx->last_dbase;
    *len = cx->last_len;
    return NP2_OK;
}

int np2_phase_vote(const uint32_t *keys, uint32_t n_keys, const uint32_t *pa, const uint32_t *pb, const float *pw,
                   uint64_t n_pairs, const uint32_t *ref_ids, const float *ref_w, uint32_t n_ref, int has_ref,
                   uint32_t *out_ids, uint32_t *n_out) {
    if (!keys || !out_ids || !n_out) return NP2_E_ARG;
    try {
    uint32_t mx = 0;
    for (uint32_t i = 0; i < n_keys; ++i) mx = std::max(mx, keys[i]);
    for (uint32_t i = 0; i < n_ref; ++i) mx = std::max(mx, ref_ids[i]);
    phase::Graph g;
    g.reserve_ids(mx + 1);
    for (uint32_t i = 0; i < n_keys; ++i) g.add_key(keys[i]);
    if (!g.add_edges(n_pairs, [&](uint64_t i) { return pa[i]; }, [&](uint64_t i) { return pb[i]; },
                     [&](uint64_t i) { return pw[i]; }))
        return NP2_E_ARG;
    std::vector<float> rw(mx + 1, 0.f);
    std::vector<uint8_t> rs(mx + 1, 0);
    for (uint32_t i = 0; i < n_ref; ++i) {
        rw[ref_ids[i]] = ref_w[i];
        rs[ref_ids[i]] = 1;
    }
    std::vector<uint32_t> losers;
    if (!phase::losing_reads(std::move(g), has_ref != 0, rw, rs, losers)) return NP2_E_REFPANIC;
    std::sort(losers.begin(), losers.end());
    losers.erase(std::unique(losers.begin(), losers.end()), losers.end());
    *n_out = (uint32_t)losers.size();
    for (size_t i = 0; i < losers.size(); ++i) out_ids[i] = losers[i];
    } catch (const std::exception &) {
        return NP2_E_NOMEM;
    }
    return NP2_OK;
}

// host-only test hook: iteration order of the product's SwissTable order model (np2_phase_host.hpp) after a script of
// op 0 = insert(key), 1 = remove(key), 2 = entry(key).or_insert — checked against hand-traced vectors
int np2_swiss_order(const uint32_t *ops, const uint32_t *keys, uint32_t n, uint32_t *out, uint32_t *n_out) {
    if (!ops || !keys || !out || !n_out) return NP2_E_ARG;
    phase::SwissOrderMap<int> m;
    for (uint32_t i = 0; i < n; ++i) {
        if (ops[i] == 0)
            m.put(keys[i], 0);
        else if (ops[i] == 1)
            m.take(keys[i], nullptr);
        else if (!m.has(keys[i]))
            m.put_vacant(keys[i], 0);
    }
    uint32_t c = 0;
    m.each([&](uint32_t k, const int &) { out[c++] = k; });
    *n_out = c;
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

// ---- shards of one contig (multi-GPU: reference intervals of a long contig) -----------------------------------------
int np2_shard_plan(const np2_read_t *reads, uint32_t n_reads, uint32_t L, uint32_t n_shards, uint32_t halo,
                   np2_shard_plan_t *out) {
    if (!reads || !out || n_reads < 1 || n_shards < 1 || L < 3) return NP2_E_ARG;
    for (uint32_t k = 0; k < n_shards; ++k) {
        np2_shard_plan_t &p = out[k];
        p.own_lo = (uint32_t)((uint64_t)L * k / n_shards);
        p.own_hi = (uint32_t)((uint64_t)L * (k + 1) / n_shards);
        if (k) p.own_lo &= ~1023u; // (cuts on tile boundaries; the last shard ends at L)
        if (k + 1 < n_shards) p.own_hi &= ~1023u;
        if (p.own_hi <= p.own_lo) return NP2_E_ARG; // contig too short for that many shards
        const uint32_t zlo = p.own_lo > halo ? p.own_lo - halo : 0u;
        const uint32_t zhi = (uint64_t)p.own_hi + halo < L ? p.own_hi + halo : L;
        // every read overlapping the zone, whole; the sub-contig spans from the first start to the last end among them
        uint32_t rlo = 0xFFFFFFFFu, rhi = 0, slo = zlo, shi = zhi;
        for (uint32_t r = 1; r < n_reads; ++r) {
            const np2_read_t &rd = reads[r];
            if (rd.flags & NP2_READ_DROPPED) continue;
            if (rd.aln_t_e < zlo || rd.aln_t_s >= zhi) continue;
            rlo = std::min(rlo, r);
            rhi = std::max(rhi, r + 1);
            slo = std::min(slo, rd.aln_t_s);
            shi = std::max(shi, rd.aln_t_e + 1);
        }
        if (rlo == 0xFFFFFFFFu) rlo = rhi = 1;
        p.read_lo = rlo;
        p.read_hi = rhi;
        p.sub_lo = k == 0 ? 0u : (slo & ~63u); // (64-aligned: the packed contig slice starts on a byte / word boundary)
        p.sub_hi = k + 1 == n_shards ? L : shi;
        p.zone_lo = zlo;
        p.zone_hi = zhi;
    }
    return NP2_OK;
}

int np2_shard_upload(np2_ctx_t *cx, const uint8_t *ref, uint32_t L, const np2_read_t *reads, uint32_t n_reads,
                     const uint8_t *nibbles, uint64_t nib_bytes, const np2_shard_plan_t *pl, np2_contig_t **out) {
    if (!cx || !ref || !reads || !nibbles || !pl || !out) return NP2_E_ARG;
    *out = nullptr;
    try {
        if (pl->sub_hi > L || pl->sub_lo >= pl->sub_hi || pl->read_hi > n_reads || pl->read_lo < 1 || pl->read_lo > pl->read_hi)
            throw Np2Error(NP2_E_ARG, "shard plan does not fit the contig");
        const uint32_t Ls = pl->sub_hi - pl->sub_lo, n = 1 + (pl->read_hi - pl->read_lo);
        // local read 0: the sub-contig aligned to itself, packed here (AlignSeq::new of main.rs:1732-1739 on the slice)
        const uint64_t ref_bytes = ((((uint64_t)Ls + 1) >> 1) + 1 + 15) & ~15ull;
        // nibble streams of the shard's reads: one contiguous piece of the contig's buffer (reads are stored in order)
        uint64_t lo = ~0ull, hi = 0;
        for (uint32_t r = pl->read_lo; r < pl->read_hi; ++r) {
            lo = std::min<uint64_t>(lo, reads[r].nib_off);
            hi = std::max<uint64_t>(hi, reads[r].nib_off + (((uint64_t)reads[r].n_cols + 1) >> 1) + 1);
        }
        if (lo == ~0ull) lo = hi = 0;
        if (hi + 16 > nib_bytes && hi) throw Np2Error(NP2_E_ARG, "nibble stream (plus 16 B tail padding) exceeds the buffer");
        const uint64_t piece = hi > lo ? ((hi - lo + 15) & ~15ull) : 0;
        const uint64_t total = ref_bytes + piece + 64;
        std::vector<uint8_t> host(total, 0);
        for (uint32_t i = 0; i < Ls; ++i) {
            const uint8_t code = ascii_to_code(ref[pl->sub_lo + i]);
            host[i >> 1] |= (i & 1) ? code : (uint8_t)(code << 4);
        }
        host[Ls >> 1] |= (Ls & 1) ? 0x0F : 0xFF;
        if (piece) memcpy(host.data() + ref_bytes, nibbles + lo, std::min<uint64_t>(piece, nib_bytes - lo));
        std::vector<np2_read_t> rds(n);
        memset(rds.data(), 0, n * sizeof(np2_read_t));
        rds[0].aln_t_s = 0, rds[0].aln_t_e = Ls - 1, rds[0].n_cols = Ls, rds[0].nib_off = 0;
        for (uint32_t i = 1; i < n; ++i) {
            const np2_read_t &g = reads[pl->read_lo + i - 1];
            np2_read_t &d = rds[i];
            d = g;
            const bool outside = (g.flags & NP2_READ_DROPPED) || g.aln_t_e < pl->zone_lo || g.aln_t_s >= pl->zone_hi;
            if (outside) { // keeps its index (like a read the clip filter emptied, main.rs:571)
                d.flags |= NP2_READ_DROPPED;
                d.n_cols = 0;
                d.aln_t_s = d.aln_t_e = 0;
                d.nib_off = ref_bytes; // (any valid 16-B aligned slot: never decoded)
                continue;
            }
            if (g.aln_t_s < pl->sub_lo || g.aln_t_e >= pl->sub_hi) throw Np2Error(NP2_E_ARG, "shard plan: read outside the sub-contig");
            d.aln_t_s = g.aln_t_s - pl->sub_lo;
            d.aln_t_e = g.aln_t_e - pl->sub_lo;
            d.nib_off = ref_bytes + (g.nib_off - lo);
        }
        return np2_contig_upload(cx, ref + pl->sub_lo, Ls, rds.data(), n, host.data(), total, out);
    } catch (const Np2Error &e) {
        return fail(cx, e);
    } catch (const std::exception &ex) {
        return fail(cx, Np2Error(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what()));
    }
}

// a block of the process-wide pinned pool for one read-back
struct PinnedBlock {
    void *p = nullptr;
    explicit PinnedBlock(size_t bytes) : p(pinned_pool().get(std::max<size_t>(bytes, 64))) {
        if (!p) throw Np2Error(NP2_E_NOMEM, "hipHostMalloc failed");
    }
    ~PinnedBlock() {
        if (std::uncaught_exceptions() > 0) (void)hipDeviceSynchronize(); // (a copy into it may still be in flight)
        pinned_pool().put(p);
    }
    PinnedBlock(const PinnedBlock &) = delete;
    PinnedBlock &operator=(const PinnedBlock &) = delete;
};

#define NP2_SHARD_TRY(cxp, ...)                                                                      \
    try {                                                                                            \
        __VA_ARGS__                                                                                  \
    } catch (const Np2Error &e) {                                                                    \
        (void)hipStreamSynchronize((cxp)->stream);                                                   \
        if ((cxp)->stream2) (void)hipStreamSynchronize((cxp)->stream2);                              \
        return fail((cxp), e);                                                                       \
    } catch (const std::exception &ex) {                                                             \
        (void)hipStreamSynchronize((cxp)->stream);                                                   \
        return fail((cxp), Np2Error(NP2_E_NOMEM, std::string("unexpected exception: ") + ex.what())); \
    }

int np2_shard_begin(np2_ctx_t *cx, np2_contig_t *c, const np2_shard_plan_t *pl, const np2_opts_t *opts, uint32_t verify,
                    np2_shard_run_t **out) {
    if (!cx || !c || !pl || !opts || !out) return NP2_E_ARG;
    *out = nullptr;
    ShardRun *sr = new ShardRun();
    sr->plan = *pl;
    sr->verify = verify;
    sr->run.cx = cx, sr->run.c = c, sr->run.o = *opts;
    sr->run.own_lo = pl->own_lo - pl->sub_lo;
    sr->run.own_hi = pl->own_hi - pl->sub_lo;
    sr->run.wide_votes = true; // (exported as np2_vote_t and merged with the other shards')
    try {
        if (c->L != pl->sub_hi - pl->sub_lo || c->R != 1 + (pl->read_hi - pl->read_lo))
            throw Np2Error(NP2_E_ARG, "contig is not the upload of this shard plan");
        run_begin(sr->run);
    } catch (const Np2Error &e) {
        delete sr;
        (void)hipStreamSynchronize(cx->stream);
        return fail(cx, e);
    }
    *out = (np2_shard_run_t *)sr;
    return NP2_OK;
}
void np2_shard_end(np2_shard_run_t *h) { delete (ShardRun *)h; }
int np2_shard_passes_left(np2_shard_run_t *h) {
    ShardRun *sr = (ShardRun *)h;
    return sr ? (int)(sr->run.o.iter_count - sr->run.pass) : 0;
}

int np2_shard_vote(np2_shard_run_t *h, np2_vote_t *out) {
    ShardRun *sr = (ShardRun *)h;
    if (!sr || !out || sr->run.final_pass()) return NP2_E_ARG;
    np2_ctx *cx = sr->run.cx;
    NP2_SHARD_TRY(cx, {
        VoteData vd;
        const np2_shard_plan_t &pl = sr->plan;
        // (local read i >= 1 is contig read read_lo + i - 1 and read 0 never enters a pair: one 64-bit add renumbers both
        // ends, and the keys stay sorted; the emitting kernel adds it — a chromosome's shard exports tens of millions of pairs)
        const uint64_t shift = pl.read_lo - 1, add = (shift << 32) | shift;
        sr->run.key_add = add;
        run_vote_pass(sr->run, vd);
        sr->v_key.clear(), sr->v_cnt.clear(), sr->v_read.clear(), sr->v_first.clear(), sr->v_refw.clear(), sr->v_flags.clear();
        const uint64_t *out_key = nullptr;
        const uint32_t *out_cnt = nullptr;
        size_t out_np = 0;
        if (vd.any) {
            // region index -> contig position: the merged key order of the contig's weight map is by descending region
            // start (regions are listed right to left, main.rs:1613-1620), read index within a region.  (Read back through
            // a staging block of its own: the pairs are still where the vote's read-back left them, in the context's.)
            std::vector<uint32_t> start(sr->run.n_reg);
            if (sr->run.n_reg) {
                PinnedBlock tmp((size_t)sr->run.n_reg * 4 + 64); // (NP2_E_NOMEM on failure; given back on every path)
                op_d2h(cx, tmp.p, cx->lq_start.p, (size_t)sr->run.n_reg * 4);
                op_sync(cx);
                memcpy(start.data(), tmp.p, (size_t)sr->run.n_reg * 4);
            }
            out_np = (size_t)vd.n_pairs();
            if (vd.view_key && vd.key_added) {
                // the pairs stay in the read-back staging: valid until this context's next read-back, i.e. the next np2_shard_*
                // call of this run (np2.h) — no host pass over them at all
                out_key = vd.view_key, out_cnt = vd.view_cnt;
            } else {
                const uint64_t *k = vd.keys();
                const uint32_t *cn = vd.cnts();
                sr->v_key.resize(out_np);
                for (size_t i = 0; i < out_np; ++i) sr->v_key[i] = k[i] + (vd.key_added ? 0 : add);
                sr->v_cnt.assign(cn, cn + out_np);
                out_key = sr->v_key.data(), out_cnt = sr->v_cnt.data();
            }
            for (uint32_t r = 0; r < vd.R; ++r) {
                const bool votes = vd.first_key[r] != 0xFFFFFFFFu;
                if (!votes && !vd.ref_seen[r] && !vd.bad[r]) continue;
                sr->v_read.push_back(shard_global_read(pl, r));
                sr->v_first.push_back(votes ? start[vd.first_key[r]] + pl.sub_lo : 0xFFFFFFFFu);
                sr->v_refw.push_back(vd.ref_w[r]);
                sr->v_flags.push_back((uint8_t)((votes ? 1 : 0) | (vd.ref_seen[r] ? 2 : 0) | (vd.bad[r] ? 4 : 0)));
            }
        }
        out->n_pairs = out_np;
        out->pair_key = out_key;
        out->pair_cnt = out_cnt;
        out->n_reads = (uint32_t)sr->v_read.size();
        out->read_id = sr->v_read.data();
        out->first_pos = sr->v_first.data();
        out->ref_w = sr->v_refw.data();
        out->flags = sr->v_flags.data();
        // The decision is taken elsewhere — votes gathered, merged, Louvain: ~90 ms for a diploid chromosome — and the device
        // has nothing to do meanwhile.  As in polish_impl the next pass starts at once on the reads the vote kernel has
        // flagged (main.rs:977: they are removed whatever else the decision says); np2_shard_apply compares: the decision
        // removes exactly those -> the pass under way is the right one; it removes others too (a read flagged by the
        // neighbouring shard, a conflicting community) -> they go as well and the pass is started again.  Only kernels are
        // issued here, no read-back: the exported pairs stay valid where they are.
        const bool no_spec = cx->hooks.no_speculate;
        sr->spec = false;
        if (vd.any && vd.d_bad && !sr->run.o.use_all_reads && !cx->trace && !no_spec) {
            size_t n_bad = 0;
            for (uint8_t b : vd.bad) n_bad += b;
            if (n_bad) {
                PolishRun &r = sr->run;
                launch_kill_flagged(cx->stream, vd.d_bad, r.c->R, cx->alive.p);
                ++r.pass; // (what run_apply_losers does)
                r.reuse = false;
                pass_front_issue(cx, r.c, r.T, (int)r.pass);
                r.front_issued = true;
                op_submit(cx);
                sr->spec = true;
                sr->spec_bad = vd.bad;
                sr->spec_n_bad = n_bad;
            }
        }
        if (cx->hooks.shard_spec_log)
            fprintf(stderr, "[np2 shard] vote: any %d, flags on the device %d, early start %d (%zu flagged)\n", (int)vd.any, vd.d_bad != nullptr, (int)sr->spec, sr->spec_n_bad);
    })
    return NP2_OK;
}

int np2_vote_decide(const np2_vote_t *votes, int n_votes, uint32_t n_reads_total, const np2_opts_t *opts, uint32_t *losers,
                    uint32_t *n_losers) {
    if (!votes || n_votes < 1 || !opts || !losers || !n_losers) return NP2_E_ARG;
    try {
        VoteData vd;
        vd.R = n_reads_total;
        vd.first_key.assign(n_reads_total, 0xFFFFFFFFu);
        vd.ref_w.assign(n_reads_total, 0);
        vd.ref_seen.assign(n_reads_total, 0);
        vd.bad.assign(n_reads_total, 0);
        std::vector<uint32_t> first_pos(n_reads_total, 0);
        std::vector<uint8_t> votes_any(n_reads_total, 0);
        // every shard's pairs arrive sorted by key (a << 32 | b): the contig's list is their merge, the counts of a pair
        // that shares regions of two shards added up before the weight rule is applied (main.rs:996-1002).  Neighbouring
        // shards share only the reads of their halos, so a merge step is mostly two block copies.
        std::vector<uint64_t> mk, tk;
        std::vector<uint32_t> mc, tc;
        for (int v = 0; v < n_votes; ++v) {
            const np2_vote_t &x = votes[v];
            for (uint32_t i = 0; i < x.n_reads; ++i) {
                const uint32_t r = x.read_id[i];
                if (r >= n_reads_total) return NP2_E_ARG;
                vd.any = true;
                vd.ref_w[r] += x.ref_w[i];
                if (x.flags[i] & 2) vd.ref_seen[r] = 1;
                if (x.flags[i] & 4) vd.bad[r] = 1;
                if (x.flags[i] & 1) { // the read's first vote = its rightmost HETE region over all shards
                    if (!votes_any[r] || x.first_pos[i] > first_pos[r]) first_pos[r] = x.first_pos[i];
                    votes_any[r] = 1;
                }
            }
            if (!x.n_pairs) continue;
            // (the keys arrive from other ranks: both endpoints index per-read arrays below; unsorted input is sorted here)
            std::vector<uint64_t> sk;
            std::vector<uint32_t> sc;
            const uint64_t *xk = x.pair_key;
            const uint32_t *xc = x.pair_cnt;
            bool sorted = true;
            { // (2 x 10^7 pairs for a chromosome: on all threads)
                std::vector<uint8_t> st(phase::host_threads(), 0); // bit 0: an endpoint outside the contig's reads, bit 1: unsorted
                phase::parallel_ranges((size_t)x.n_pairs, (size_t)1 << 18, [&](unsigned t, size_t lo, size_t hi) {
                    uint8_t f = 0;
                    for (size_t i = lo; i < hi; ++i) {
                        if ((uint32_t)(xk[i] >> 32) >= n_reads_total || (uint32_t)xk[i] >= n_reads_total) f |= 1;
                        if (i && !(xk[i - 1] < xk[i])) f |= 2;
                    }
                    st[t] = f;
                });
                for (uint8_t f : st) {
                    if (f & 1) return NP2_E_ARG;
                    if (f & 2) sorted = false;
                }
            }
            if (!sorted) {
                std::vector<std::pair<uint64_t, uint32_t>> tmp(x.n_pairs);
                for (uint64_t i = 0; i < x.n_pairs; ++i) tmp[i] = {xk[i], xc[i]};
                std::sort(tmp.begin(), tmp.end());
                for (auto &t : tmp) {
                    if (!sk.empty() && sk.back() == t.first) {
                        const uint32_t same = (sc.back() & 0xFFFFu) + (t.second & 0xFFFFu), neg = (sc.back() >> 16) + (t.second >> 16);
                        if (same > 0xFFFFu || neg > 0xFFFFu) return NP2_E_UNSUPPORTED;
                        sc.back() = same | (neg << 16);
                    } else {
                        sk.push_back(t.first);
                        sc.push_back(t.second);
                    }
                }
                xk = sk.data(), xc = sc.data();
            }
            const size_t nx = sorted ? (size_t)x.n_pairs : sk.size();
            if (n_votes == 1 && sorted) { // the caller's arrays as they are
                vd.view_key = xk, vd.view_cnt = xc, vd.view_n = nx;
                continue;
            }
            if (mk.empty()) {
                mk.assign(xk, xk + nx);
                mc.assign(xc, xc + nx);
                continue;
            }
            // merge (mk, mc) with (xk, xc): the part of mk below xk[0] and the part of xk above mk.back() are copied en bloc
            tk.clear(), tc.clear();
            tk.reserve(mk.size() + nx), tc.reserve(mk.size() + nx);
            size_t i = (size_t)(std::lower_bound(mk.begin(), mk.end(), xk[0]) - mk.begin()), j = 0;
            tk.insert(tk.end(), mk.begin(), mk.begin() + (long)i);
            tc.insert(tc.end(), mc.begin(), mc.begin() + (long)i);
            while (i < mk.size() && j < nx) {
                if (mk[i] < xk[j]) {
                    tk.push_back(mk[i]), tc.push_back(mc[i]), ++i;
                } else if (xk[j] < mk[i]) {
                    tk.push_back(xk[j]), tc.push_back(xc[j]), ++j;
                } else {
                    const uint32_t same = (mc[i] & 0xFFFFu) + (xc[j] & 0xFFFFu), neg = (mc[i] >> 16) + (xc[j] >> 16);
                    if (same > 0xFFFFu || neg > 0xFFFFu) return NP2_E_UNSUPPORTED;
                    tk.push_back(mk[i]), tc.push_back(same | (neg << 16)), ++i, ++j;
                }
            }
            tk.insert(tk.end(), mk.begin() + (long)i, mk.end());
            tc.insert(tc.end(), mc.begin() + (long)i, mc.end());
            tk.insert(tk.end(), xk + j, xk + nx);
            tc.insert(tc.end(), xc + j, xc + nx);
            mk.swap(tk), mc.swap(tc);
        }
        if (!vd.view_key) {
            vd.pair_key.swap(mk);
            vd.pair_cnt.swap(mc);
        }
        for (uint32_t r = 0; r < n_reads_total; ++r)
            if (votes_any[r]) vd.first_key[r] = 0xFFFFFFFEu - first_pos[r]; // ascending = right to left
        if (getenv("NP2_VOTE_COMPACT")) {
            // test hook (no device needed): the same vote in the compact row form the plain pipeline reads back from the
            // vote kernels, so that the row builder over that form is checked against recorded votes on the CPU
            const uint64_t *k = vd.keys();
            const uint32_t *cn = vd.cnts();
            const uint64_t np = vd.n_pairs();
            vd.c_off_own.assign((size_t)n_reads_total + 1, 0);
            vd.c_pairs_own.resize(np);
            for (uint64_t i = 0; i < np; ++i) {
                const uint32_t a = (uint32_t)(k[i] >> 32), b = (uint32_t)k[i], same = cn[i] & 0xFFFFu, neg = cn[i] >> 16;
                if (b <= a || b - a - 1 > 255u || same > VOTE_CNT_MAX || neg > VOTE_CNT_MAX || (i && k[i - 1] >= k[i]))
                    return NP2_E_UNSUPPORTED; // (such a vote takes the wide form)
                ++vd.c_off_own[a + 1];
                vd.c_pairs_own[i] = (b - a - 1) | (same << 8) | (neg << 20);
            }
            for (uint32_t r = 0; r < n_reads_total; ++r) vd.c_off_own[r + 1] += vd.c_off_own[r];
            vd.c_off = vd.c_off_own.data(), vd.c_pairs = vd.c_pairs_own.data();
        }
        std::vector<uint32_t> ls = vote_decide(nullptr, vd, opts->use_all_reads != 0);
        *n_losers = (uint32_t)ls.size();
        for (size_t i = 0; i < ls.size(); ++i) losers[i] = ls[i];
    } catch (const Np2Error &e) {
        return e.code;
    } catch (const std::exception &) {
        return NP2_E_NOMEM;
    }
    return NP2_OK;
}

int np2_shard_apply(np2_shard_run_t *h, const uint32_t *losers, uint32_t n) {
    ShardRun *sr = (ShardRun *)h;
    if (!sr || (n && !losers) || (sr->run.final_pass() && !sr->spec)) return NP2_E_ARG; // (a pass started early has advanced the counter)
    np2_ctx *cx = sr->run.cx;
    NP2_SHARD_TRY(cx, {
        std::vector<uint32_t> local;
        const np2_shard_plan_t &pl = sr->plan;
        for (uint32_t i = 0; i < n; ++i) {
            if (losers[i] == 0) throw Np2Error(NP2_E_REFPANIC, "reference would panic: the contig itself was voted out");
            if (losers[i] >= pl.read_lo && losers[i] < pl.read_hi) local.push_back(losers[i] - pl.read_lo + 1);
        }
        if (sr->spec) { // the pass is under way on the flagged reads (np2_shard_vote): is it the right one?
            sr->spec = false;
            std::vector<uint32_t> extra, spared;
            size_t n_flagged = 0;
            std::vector<uint8_t> removed(sr->run.c->R, 0);
            for (uint32_t id : local) {
                REFPANIC_IF(id >= sr->run.c->R, "index out of bounds: alignseqs[id]");
                removed[id] = 1;
                if (sr->spec_bad[id]) ++n_flagged; else extra.push_back(id);
            }
            // A decision that KEEPS a read the vote kernel flagged (a replayed or custom loser list; the reference's own
            // decision removes every flagged read, main.rs:977) is a misspeculation like any other: the read was alive
            // before the vote, it comes back, and the pass starts again.
            if (n_flagged != sr->spec_n_bad)
                for (uint32_t id = 0; id < sr->run.c->R; ++id)
                    if (sr->spec_bad[id] && !removed[id]) spared.push_back(id);
            if (cx->hooks.shard_spec_log) fprintf(stderr, "[np2 shard] apply: %zu removed here, %zu beyond the flagged ones, %zu flagged ones kept\n", local.size(), extra.size(), spared.size());
            if (!extra.empty() || !spared.empty() || cx->hooks.test_misspeculate) { // (test hook: the redo branch)
                if (!spared.empty()) {
                    cx->kill_ids.ensure(spared.size() + 1);
                    h2d_staged(cx, cx->kill_ids.p, spared.data(), spared.size() * 4);
                    launch_revive_reads(cx->stream, cx->kill_ids.p, (uint32_t)spared.size(), cx->alive.p);
                    op_sync(cx); // (the staging buffer is reused for the other list right away)
                }
                if (!extra.empty()) {
                    cx->kill_ids.ensure(extra.size() + 1);
                    h2d_staged(cx, cx->kill_ids.p, extra.data(), extra.size() * 4);
                    launch_kill_reads(cx->stream, cx->kill_ids.p, (uint32_t)extra.size(), cx->alive.p);
                }
                sr->run.reuse = false;
                sr->run.front_issued = false; // (run_pass_front starts the pass again, on the right reads)
            }
            return NP2_OK;
        }
        // (no identical-pass reuse across shards: a neighbour's removals are not visible here, the decision would
        // differ from shard to shard only in cost, never in result — but keep it simple)
        const uint32_t n_reg = sr->run.n_reg;
        run_apply_losers(sr->run, local);
        if (n) sr->run.reuse = false;
        (void)n_reg;
    })
    return NP2_OK;
}

int np2_shard_final(np2_shard_run_t *h, uint8_t **out_bases, uint32_t **out_pos, uint64_t *out_len) {
    ShardRun *sr = (ShardRun *)h;
    if (!sr || !out_bases || !out_pos || !out_len || !sr->run.final_pass()) return NP2_E_ARG;
    np2_ctx *cx = sr->run.cx;
    ResultOut r;
    struct PutBack { // the pinned result blocks go back to the pool unless they are handed to the caller
        ResultOut &r;
        bool keep = false;
        ~PutBack() {
            if (keep) return;
            if (r.bases) pinned_pool().put(r.bases);
            if (r.pos) pinned_pool().put(r.pos);
        }
    } guard{r};
    NP2_SHARD_TRY(cx, {
        run_final_pass(sr->run, r);
        const np2_shard_plan_t &pl = sr->plan;
        shard_check_stuck(sr);
        // keep the owned interval (+ the verification margin), in contig coordinates
        const uint32_t lo = pl.own_lo > sr->verify ? pl.own_lo - sr->verify : 0u;
        const uint64_t hi = (uint64_t)pl.own_hi + sr->verify;
        uint64_t w = 0;
        for (uint64_t i = 0; i < r.len; ++i) {
            const uint64_t gp = (uint64_t)r.pos[i] + pl.sub_lo;
            if (gp < lo || gp >= hi) continue;
            r.bases[w] = r.bases[i];
            r.pos[w] = (uint32_t)gp;
            ++w;
        }
        r.len = w;
    })
    guard.keep = true;
    *out_bases = r.bases;
    *out_pos = r.pos;
    *out_len = r.len;
    return NP2_OK;
}

void *np2_alloc_pinned(uint64_t bytes) { return pinned_pool().get((size_t)bytes + 1); }
void np2_trim_device_cache(void) { dev_cache().trim(0); }

int np2_shard_final_device(np2_shard_run_t *h, np2_shard_piece_t *out) {
    ShardRun *sr = (ShardRun *)h;
    if (!sr || !out || !sr->run.final_pass()) return NP2_E_ARG;
    np2_ctx *cx = sr->run.cx;
    memset(out, 0, sizeof *out);
    void *blocks[4] = {nullptr, nullptr, nullptr, nullptr};
    struct PutBack {
        void **b;
        bool keep = false;
        ~PutBack() {
            if (!keep)
                for (int i = 0; i < 4; ++i)
                    if (b[i]) pinned_pool().put(b[i]);
        }
    } guard{blocks};
    NP2_SHARD_TRY(cx, {
        ResultOut r;
        r.want_bases = false, r.want_pos = false; // the polished sub-contig stays on the device
        run_final_pass(sr->run, r);
        shard_check_stuck(sr);
        const np2_shard_plan_t &pl = sr->plan;
        const uint32_t v = sr->verify;
        // positions in sub-contig coordinates; the consensus is ordered by position
        auto sub = [&](uint64_t p) -> uint32_t { return p <= pl.sub_lo ? 0u : (uint32_t)std::min<uint64_t>(p - pl.sub_lo, 0xFFFFFFFFull); };
        const uint32_t t[6] = {sub(pl.own_lo > v ? pl.own_lo - v : 0u), sub(pl.own_lo), sub((uint64_t)pl.own_lo + v),
                               sub(pl.own_hi > v ? pl.own_hi - v : 0u), sub(pl.own_hi), sub((uint64_t)pl.own_hi + v)};
        cx->shard_bounds.ensure(8);
        const uint32_t *M_p = cx->mlen.p; // (a device copy of the final length: any buffer holding it will do)
        {
            // the final length is known on the host (fetch_result read it back): stage it next to the bounds
            const uint32_t Mh = (uint32_t)cx->last_len;
            cx->mlen.ensure(32);
            h2d_staged(cx, cx->mlen.p + 31, &Mh, 4);
            M_p = cx->mlen.p + 31;
        }
        launch_shard_bounds(cx->stream, cx->last_dpos, M_p, t, cx->shard_bounds.p);
        const std::vector<uint32_t> b = d2h(cx, cx->shard_bounds.p, 8);
        sr->own_off = b[1];
        sr->own_len = b[4] - b[1];
        sr->have_piece = true;
        out->own_len = sr->own_len;
        out->dev_bases = cx->last_dbase + b[1];
        out->dev_pos = cx->last_dpos + b[1];
        out->first_pos = b[6] + pl.sub_lo;
        out->last_pos = b[7] + pl.sub_lo;
        out->lo_len = b[2] - b[0];
        out->hi_len = b[5] - b[3];
        auto strip = [&](uint32_t i0, uint32_t n, uint8_t *&hb, uint32_t *&hp, int slot) {
            if (!n) return;
            hb = (uint8_t *)(blocks[slot] = pinned_pool().get((size_t)n + 1));
            hp = (uint32_t *)(blocks[slot + 1] = pinned_pool().get(((size_t)n + 1) * 4));
            if (!hb || !hp) throw Np2Error(NP2_E_NOMEM, "pinned strip allocation failed");
            op_d2h(cx, hb, cx->last_dbase + i0, n);
            op_d2h(cx, hp, cx->last_dpos + i0, (size_t)n * 4);
        };
        strip(b[0], out->lo_len, out->lo_bases, out->lo_pos, 0);
        strip(b[3], out->hi_len, out->hi_bases, out->hi_pos, 2);
        op_sync(cx);
        for (uint32_t i = 0; i < out->lo_len; ++i) out->lo_pos[i] += pl.sub_lo;
        for (uint32_t i = 0; i < out->hi_len; ++i) out->hi_pos[i] += pl.sub_lo;
        flush_timings(cx); // (np2_last_timings: the stage timers of the whole run, the dense pass of np2_shard_begin included)
    })
    guard.keep = true;
    return NP2_OK;
}

int np2_shard_fetch(np2_shard_run_t *h, uint8_t *dst_bases, uint32_t *dst_pos) {
    ShardRun *sr = (ShardRun *)h;
    if (!sr || !dst_bases || !sr->have_piece) return NP2_E_ARG;
    np2_ctx *cx = sr->run.cx;
    NP2_SHARD_TRY(cx, {
        HIPCHK(hipSetDevice(cx->device));
        if (!cx->last_dbase) throw Np2Error(NP2_E_ARG, "the shard's device result is gone (another call used its context)");
        if (sr->own_len) {
            op_d2h(cx, dst_bases, cx->last_dbase + sr->own_off, sr->own_len);
            if (dst_pos) op_d2h(cx, dst_pos, cx->last_dpos + sr->own_off, sr->own_len * 4);
            op_sync(cx);
            if (dst_pos) {
                const uint32_t add = sr->plan.sub_lo;
                if (add)
                    for (uint64_t i = 0; i < sr->own_len; ++i) dst_pos[i] += add;
            }
        }
    })
    return NP2_OK;
}
}
