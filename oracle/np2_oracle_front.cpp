// TEST INFRASTRUCTURE — CPU oracle for the read-admission / columnarisation front-end.
//
// Literal restatement of Nextomics/NextPolish2 v0.2.2 src/main.rs:1732-1817 (per-record filters,
// Alignment::fill_with_cigar 386-440, is_clip 1796-1797, Alignment::trim 447-513, AlignSeq::new 279-312,
// clip labelling 1806-1812) and filter_alignseqs_by_clip (531-574).  Input = the fields rust-htslib
// exposes for each BAM record of one contig, in file order; output = the boundary format of
// include/np2.h (np2_read_t + nibble streams), reads[0] = the contig aligned to itself.
// PARITY UNPINNED (no reference tests; see np2_oracle.cpp header).
#include "../include/np2.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
static const uint8_t SEQ_NUM[128] = {
    65, 67, 71, 84, 45, 78, 77, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4,  4,  4,  4,  4,  4,  4,  4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
    4,  4,  4,  4,  4,  4,  4,  4, 4, 4, 4, 4, 4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 6,
    5,  4,  4,  4,  4,  4,  3,  3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 0, 4, 1, 4, 4, 4, 2,
    4,  4,  4,  4,  4,  6,  5,  4, 4, 4, 4, 4, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};
static const uint32_t ALN_T_S_LABLE = 1u << 31; // main.rs:271

struct Alignment { // main.rs:353-362
    uint32_t shift = 0, aln_t_s = 0, aln_t_e = 0, aln_q_s = 0, aln_q_e = 0;
    std::string q_aln_str, t_aln_str;
    size_t aln_len() const { return t_aln_str.size() - shift; }
};

// BAM CIGAR op codes: M0 I1 D2 N3 S4 H5 P6 =7 X8
// returns false on an op the reference panics on ("Unknown cigar", main.rs:430-432)
bool fill_with_cigar(Alignment &a, const uint32_t *cigar, uint32_t n_cigar, const char *tseq, size_t tlen,
                     const char *qseq, size_t qlen) {
    uint32_t qs = 0, ts = 0;
    bool is_first = true;
    for (uint32_t i = 0; i < n_cigar; ++i) {
        const uint32_t l = cigar[i] >> 4, op = cigar[i] & 15;
        switch (op) {
        case 4: // SoftClip
            qs += l;
            if (is_first)
                a.aln_q_s = qs;
            else
                a.aln_q_e = qs - l;
            break;
        case 0: case 7: case 8: // Match / Equal / Diff
            for (uint32_t k = 0; k < l; ++k) {
                if (qs >= qlen) return false; // index out of bounds
                a.q_aln_str.push_back(qseq[qs]);
                qs += 1;
            }
            if ((size_t)ts + l > tlen) return false;
            a.t_aln_str.append(tseq + ts, l);
            ts += l;
            break;
        case 1: // Ins
            for (uint32_t k = 0; k < l; ++k) {
                if (qs >= qlen) return false;
                a.q_aln_str.push_back(qseq[qs]);
                qs += 1;
            }
            a.t_aln_str.append(l, '-');
            break;
        case 2: // Del
            a.q_aln_str.append(l, '-');
            if ((size_t)ts + l > tlen) return false;
            a.t_aln_str.append(tseq + ts, l);
            ts += l;
            break;
        case 5: // HardClip
            break;
        default:
            return false;
        }
        is_first = false;
    }
    if (a.aln_q_e == 0) a.aln_q_e = qs;
    a.aln_t_e = a.aln_t_s + ts;
    return true;
}

void trim(Alignment &a, uint32_t len) { // main.rs:447-513
    uint32_t j = 0;
    const std::string &t = a.t_aln_str, &q = a.q_aln_str;
    for (size_t i = 0; i < t.size(); ++i) {
        if (t[i] == q[i]) {
            j += 1;
            a.aln_t_s += 1;
            a.aln_q_s += 1;
        } else {
            if (t[i] != '-') a.aln_t_s += 1;
            if (q[i] != '-') a.aln_q_s += 1;
            j = 0;
        }
        if (j == len) {
            a.aln_t_s -= len;
            a.aln_q_s -= len;
            a.shift = (uint32_t)i + 1 - len;
            break;
        }
    }
    if (j == len) {
        j = 0;
        for (size_t i = t.size(); i-- > 0;) {
            if (t[i] == q[i]) {
                j += 1;
                a.aln_t_e -= 1;
                a.aln_q_e -= 1;
            } else {
                if (t[i] != '-') a.aln_t_e -= 1;
                if (q[i] != '-') a.aln_q_e -= 1;
                j = 0;
            }
            if (j == len) {
                a.aln_t_e += len;
                a.aln_q_e += len;
                const size_t new_len = i + len;
                if (new_len < t.size()) {
                    a.t_aln_str.resize(new_len);
                    a.q_aln_str.resize(new_len);
                }
                break;
            }
        }
    } else {
        a.shift = (uint32_t)t.size();
    }
}

struct AlignSeq {
    uint32_t aln_t_s, aln_t_e;
    std::vector<uint8_t> align_bases;
    uint32_t n_cols;
};
AlignSeq make_alignseq(const Alignment &aln) { // AlignSeq::new, main.rs:279-312
    const size_t n = aln.aln_len();
    const size_t len = (n + 1) >> 1;
    AlignSeq s;
    s.aln_t_s = aln.aln_t_s;
    s.aln_t_e = aln.aln_t_s;
    s.align_bases.assign(len + 1, 0);
    size_t i = 0;
    for (size_t c = aln.shift; c < aln.t_aln_str.size(); ++c) {
        uint8_t b = SEQ_NUM[(unsigned char)aln.q_aln_str[c] & 127];
        if (aln.t_aln_str[c] == '-')
            b |= 8;
        else if (i != 0)
            s.aln_t_e += 1;
        if ((i & 1) == 0) b <<= 4;
        s.align_bases[i >> 1] |= b;
        i += 1;
    }
    s.align_bases[i >> 1] |= (i & 1) == 0 ? 255 : 15;
    s.n_cols = (uint32_t)i;
    return s;
}

void filter_alignseqs_by_clip(std::vector<AlignSeq> &as) { // main.rs:531-574
    const uint32_t offset = 50;
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    uint32_t s = 0, e = 0;
    for (auto &x : as) {
        if (x.aln_t_s & ALN_T_S_LABLE) continue;
        const uint32_t ts = x.aln_t_s + offset, te = x.aln_t_e - offset; // u32 wrapping like the release build
        if (s == e) {
            s = ts;
            e = te;
        } else if (ts > e) {
            ranges.emplace_back(s, e);
            s = ts;
            e = te;
        } else if (e < te) {
            e = te;
        }
    }
    if (s != e) ranges.emplace_back(s, e);
    for (auto &x : as) {
        if (!(x.aln_t_s & ALN_T_S_LABLE)) continue;
        x.aln_t_s ^= ALN_T_S_LABLE;
        for (auto &r : ranges) {
            if (r.first <= x.aln_t_s && x.aln_t_e <= r.second) {
                x.align_bases.clear();
                break;
            } else if (x.aln_t_e < r.first) {
                break;
            }
        }
    }
}
} // namespace

extern "C" {

// One BAM record as rust-htslib exposes it (main.rs:1751-1797)
typedef struct np2o_bamrec {
    int32_t pos;        // reference_start
    uint16_t flag;
    uint8_t mapq;
    uint8_t pad;
    uint32_t n_cigar;
    uint64_t cigar_off; // into cigar[] (u32: len << 4 | op)
    uint32_t l_seq;
    uint64_t seq_off;   // into seq[] (ASCII, as Seq::index yields: "=ACMGRSVTWYHKDBN")
} np2o_bamrec_t;

typedef struct np2o_front_opts {
    uint32_t min_read_len;   // -l 1000
    uint32_t min_map_len;    // -a INT part, 500
    float min_map_fra;       // -a FRAC part, 0.5
    int16_t min_map_qual;    // -q 1
    uint32_t max_clip_len;   // -c 100
    uint8_t use_supplementary, use_secondary;
} np2o_front_opts_t;

// Returns 0, or NP2_E_REFPANIC.  Outputs are malloc'ed: reads (np2_read_t[n]), nibbles (16-B aligned slots).
int np2o_front_end(const char *tseq, uint32_t L, const np2o_bamrec_t *recs, uint32_t n_recs, const uint32_t *cigar,
                   const char *seq, const np2o_front_opts_t *o, np2_read_t **out_reads, uint32_t *out_n,
                   uint8_t **out_nib, uint64_t *out_nib_bytes) {
    std::vector<AlignSeq> as;
    {
        Alignment aln; // the contig vs itself (main.rs:1732-1739)
        aln.aln_t_e = L;
        aln.aln_q_e = L;
        aln.q_aln_str.assign(tseq, L);
        aln.t_aln_str.assign(tseq, L);
        as.push_back(make_alignseq(aln));
    }
    for (uint32_t i = 0; i < n_recs; ++i) {
        const np2o_bamrec_t &r = recs[i];
        const uint32_t *cg = cigar + r.cigar_off;
        // seq_len_from_cigar(true): M I S = X H ; reference_end - reference_start: M D N = X
        uint64_t rlen = 0;
        int64_t span = 0;
        for (uint32_t k = 0; k < r.n_cigar; ++k) {
            const uint32_t l = cg[k] >> 4, op = cg[k] & 15;
            if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8 || op == 5) rlen += l;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += l;
        }
        const bool secondary = r.flag & 0x100, supplementary = r.flag & 0x800;
        const int64_t need = std::max<int64_t>((int64_t)o->min_map_len, (int64_t)((float)rlen * o->min_map_fra));
        if ((r.flag & 0x404) != 0 || (int16_t)r.mapq <= o->min_map_qual || rlen <= o->min_read_len ||
            (secondary && !o->use_secondary) || (supplementary && !o->use_supplementary) || span < need)
            continue;
        // -S (main.rs:1775-1789): the caller passes, for a secondary record, the SEQ recovered from the read's primary
        // alignment (oracle/np2_oracle.py: secondary_seqs restates secondary.rs:82-148 and the strand rule)
        Alignment aln;
        aln.aln_t_s = (uint32_t)r.pos;
        if ((uint32_t)r.pos > L) return NP2_E_REFPANIC;
        if (!fill_with_cigar(aln, cg, r.n_cigar, tseq + r.pos, L - (uint32_t)r.pos, seq + r.seq_off, r.l_seq))
            return NP2_E_REFPANIC;
        const bool is_clip = aln.aln_q_e - aln.aln_q_s + o->max_clip_len < (uint32_t)rlen;
        trim(aln, 8);
        if (aln.aln_len() <= o->min_map_len) continue;
        AlignSeq s = make_alignseq(aln);
        if (is_clip) {
            if (L < 500000) continue;
            s.aln_t_s |= ALN_T_S_LABLE;
        }
        as.push_back(std::move(s));
    }
    filter_alignseqs_by_clip(as);
    // boundary format
    const uint32_t n = (uint32_t)as.size();
    np2_read_t *reads = (np2_read_t *)calloc(n, sizeof(np2_read_t));
    uint64_t off = 0;
    for (uint32_t i = 0; i < n; ++i) {
        reads[i].aln_t_s = as[i].aln_t_s;
        reads[i].aln_t_e = as[i].aln_t_e;
        reads[i].n_cols = as[i].n_cols;
        reads[i].nib_off = off;
        reads[i].flags = as[i].align_bases.empty() ? NP2_READ_DROPPED : 0;
        off += (((uint64_t)(as[i].n_cols + 1) >> 1) + 1 + 15) & ~15ull;
    }
    uint8_t *nib = (uint8_t *)calloc(off + 64, 1);
    for (uint32_t i = 0; i < n; ++i) {
        if (as[i].align_bases.empty()) { // keep a terminator so the slot stays well-formed
            nib[reads[i].nib_off] = 0xFF;
            reads[i].n_cols = 0;
            continue;
        }
        memcpy(nib + reads[i].nib_off, as[i].align_bases.data(), as[i].align_bases.size());
    }
    *out_reads = reads;
    *out_n = n;
    *out_nib = nib;
    *out_nib_bytes = off + 64;
    return NP2_OK;
}
void np2o_front_free(void *p) { free(p); }
}
