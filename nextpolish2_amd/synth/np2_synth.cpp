// Synthetic input generator for the NextPolish2 hot path (host-only C++; no GPU, no oracle).
//
// Emits inputs directly in the boundary format of include/np2.h — the reference's packed
// AlignSeq nibble streams (src/main.rs:279-312) and yak v2 bucket words
// (src/utils/kmer.rs:72-170) — following the recipe of SURVEY.md §8(d), whose parameters
// are borrowed from the reference's own simulation notes (doc/benchmark1.md:25,31):
//   genome i.i.d. uniform ACGT; optional second haplotype (SNP 0.5 %, indel 0.2 %);
//   assembly = hap1 + 1 error / 10 kb (50 % homopolymer +-1, 25 % SNV, 25 % 1-3 bp indel);
//   HiFi reads N(13000, 2000) >= 1000 bp at the requested depth, 0.2 % error (70 %
//   homopolymer indel, 15 % substitution, 15 % 1-3 bp indel), true alignments trimmed to
//   8-match anchors like Alignment::trim(8) (src/main.rs:447-513);
//   yak tables = canonical k-mer multiplicities of the true haplotypes x Poisson(lambda),
//   capped at 1023, hashed with yak_hash64 (kmer.rs:223-233), pre = 10.
#include "../../include/np2.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

struct Rng { // xoshiro256** seeded by splitmix64
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) {
        for (auto &v : s) v = splitmix(seed);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    double normal() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    // distance to the next event of per-site probability p (>= 1)
    uint64_t geometric(double p) {
        if (p <= 0) return ~0ULL >> 1;
        double u = uni();
        if (u < 1e-300) u = 1e-300;
        return 1 + (uint64_t)std::floor(std::log(u) / std::log1p(-p));
    }
    uint32_t poisson(double lam) {
        if (lam < 30) {
            double Lm = std::exp(-lam), p = 1.0;
            uint32_t k = 0;
            do {
                k++;
                p *= uni();
            } while (p > Lm);
            return k - 1;
        }
        double v = lam + std::sqrt(lam) * normal() + 0.5;
        return v < 0 ? 0 : (uint32_t)v;
    }
};

const char ACGT[4] = {'A', 'C', 'G', 'T'};

// a sequence expressed relative to hap1: per hap1 position a base code (0-3, 4 = deleted)
// plus sparse inserted strings *after* that position
struct Rendering {
    std::vector<uint8_t> base;
    std::unordered_map<uint32_t, std::string> ins; // codes 0-3 as chars 0..3
    const std::string *ins_at(uint32_t i) const {
        auto it = ins.find(i);
        return it == ins.end() ? nullptr : &it->second;
    }
};

uint32_t run_start(const std::vector<uint8_t> &h, uint32_t i) {
    while (i > 0 && h[i - 1] == h[i]) --i;
    return i;
}

struct Params {
    uint64_t seed;
    uint32_t L;
    uint32_t depth;
    uint32_t diploid;
    double snp_rate, hap_indel_rate, asm_err_rate, read_err_rate;
    double read_len_mean, read_len_sd;
    uint32_t read_len_min;
};

struct Synth {
    Params P;
    std::vector<uint8_t> hap1;
    Rendering hap2, asmr;
    std::string asm_seq;                // the contig to be polished (ASCII)
    std::string hap_seq[2];             // true haplotypes (ASCII)
    std::vector<uint32_t> asm_pos;      // assembly coordinate of hap1 position i (next base at/after)
    std::vector<np2_read_t> reads;
    std::vector<uint8_t> nibbles;
    // yak output scratch
    std::vector<uint64_t> yak_words, yak_off;
};

void add_ins(Rendering &r, uint32_t i, const std::string &s, bool front) {
    auto &dst = r.ins[i];
    if (front)
        dst.insert(0, s);
    else
        dst += s;
}

std::string render(const Rendering &r) {
    std::string out;
    out.reserve(r.base.size() + 16);
    for (uint32_t i = 0; i < r.base.size(); ++i) {
        if (r.base[i] < 4) out.push_back(ACGT[r.base[i]]);
        if (auto s = r.ins_at(i))
            for (char c : *s) out.push_back(ACGT[(int)c]);
    }
    return out;
}

void mutate_hap2(Synth &S, Rng &rng) {
    S.hap2.base = S.hap1;
    if (!S.P.diploid) return;
    uint32_t L = S.P.L;
    double p = S.P.snp_rate + S.P.hap_indel_rate;
    for (uint64_t i = rng.geometric(p) - 1; i < L; i += rng.geometric(p)) {
        if (rng.uni() < S.P.snp_rate / p) {
            S.hap2.base[i] = (uint8_t)((S.hap1[i] + 1 + rng.below(3)) & 3);
        } else if (rng.uni() < 0.5) {
            S.hap2.base[i] = 4;
        } else {
            uint32_t n = 1 + rng.below(3);
            std::string s;
            for (uint32_t j = 0; j < n; ++j) s.push_back((char)rng.below(4));
            add_ins(S.hap2, (uint32_t)i, s, true);
        }
    }
}

void mutate_asm(Synth &S, Rng &rng) {
    S.asmr.base = S.hap1;
    uint32_t L = S.P.L;
    double p = S.P.asm_err_rate;
    for (uint64_t i = rng.geometric(p) - 1 + 64; i + 64 < L; i += rng.geometric(p)) {
        double u = rng.uni();
        if (u < 0.5) { // homopolymer +-1 at the run start (left-aligned)
            uint32_t rs = run_start(S.hap1, (uint32_t)i);
            if (rng.uni() < 0.5) {
                S.asmr.base[rs] = 4;
            } else if (rs > 0) {
                add_ins(S.asmr, rs - 1, std::string(1, (char)S.hap1[rs]), false);
            }
        } else if (u < 0.75) {
            S.asmr.base[i] = (uint8_t)((S.hap1[i] + 1 + rng.below(3)) & 3);
        } else if (rng.uni() < 0.5) {
            uint32_t n = 1 + rng.below(3);
            for (uint32_t j = 0; j < n && i + j < L; ++j) S.asmr.base[i + j] = 4;
        } else {
            uint32_t n = 1 + rng.below(3);
            std::string s;
            for (uint32_t j = 0; j < n; ++j) s.push_back((char)rng.below(4));
            add_ins(S.asmr, (uint32_t)i, s, true);
        }
    }
}

struct Col {
    uint8_t t; // target code 0-3 or 4 ('-')
    uint8_t q; // query code 0-3 or 4 ('-')
};

struct Ev {
    uint32_t pos;
    uint8_t kind; // 0 sub, 1 del, 2 ins-after(front), 3 ins-after(back)
    std::string s;
};

void make_read(Synth &S, Rng &rng, int hap, uint32_t a, uint32_t b, std::vector<Col> &cols,
               uint32_t &t_start) {
    auto hbase = [&](uint32_t i) -> uint8_t { return hap == 0 ? S.hap1[i] : S.hap2.base[i]; };
    auto hins = [&](uint32_t i) -> const std::string * { return hap == 0 ? nullptr : S.hap2.ins_at(i); };

    // pre-sample read errors as events keyed by hap1 position
    std::vector<Ev> evs;
    double p = S.P.read_err_rate;
    for (uint64_t i = a + rng.geometric(p) - 1; i < b; i += rng.geometric(p)) {
        double u = rng.uni();
        Ev e;
        if (u < 0.70) {
            uint32_t rs = run_start(S.hap1, (uint32_t)i);
            if (rs <= a) continue;
            if (rng.uni() < 0.5) {
                e.pos = rs;
                e.kind = 1;
            } else {
                e.pos = rs - 1;
                e.kind = 3;
                e.s = std::string(1, (char)S.hap1[rs]);
            }
        } else if (u < 0.85) {
            e.pos = (uint32_t)i;
            e.kind = 0;
            e.s = std::string(1, (char)(1 + rng.below(3)));
        } else if (rng.uni() < 0.5) {
            e.pos = (uint32_t)i;
            e.kind = 1;
        } else {
            e.pos = (uint32_t)i;
            e.kind = 2;
            uint32_t n = 1 + rng.below(3);
            for (uint32_t j = 0; j < n; ++j) e.s.push_back((char)rng.below(4));
        }
        evs.push_back(e);
    }
    std::stable_sort(evs.begin(), evs.end(), [](const Ev &x, const Ev &y) { return x.pos < y.pos; });

    cols.clear();
    size_t ei = 0;
    std::string rins;
    for (uint32_t i = a; i < b; ++i) {
        uint8_t rb = hbase(i);
        rins.clear();
        if (auto s = hins(i)) rins = *s;
        while (ei < evs.size() && evs[ei].pos == i) {
            const Ev &e = evs[ei++];
            if (e.kind == 0) {
                if (rb < 4) rb = (uint8_t)((rb + e.s[0]) & 3);
            } else if (e.kind == 1) {
                rb = 4;
            } else if (e.kind == 2) {
                rins.insert(0, e.s);
            } else {
                rins += e.s;
            }
        }
        uint8_t ab = S.asmr.base[i];
        const std::string *ains = S.asmr.ins_at(i);
        if (ab < 4)
            cols.push_back({ab, rb});
        else if (rb < 4)
            cols.push_back({4, rb});
        size_t na = ains ? ains->size() : 0, nr = rins.size();
        size_t m = na < nr ? na : nr;
        for (size_t j = 0; j < m; ++j) cols.push_back({(uint8_t)(*ains)[j], (uint8_t)rins[j]});
        for (size_t j = m; j < na; ++j) cols.push_back({(uint8_t)(*ains)[j], 4});
        for (size_t j = m; j < nr; ++j) cols.push_back({4, (uint8_t)rins[j]});
    }
    t_start = S.asm_pos[a];
}

// Alignment::trim(8) equivalent on the column list; returns false if no anchor
bool trim8(std::vector<Col> &cols, uint32_t &t_start) {
    const int LEN = 8;
    int j = 0;
    size_t n = cols.size(), shift = n;
    uint32_t ts = t_start;
    for (size_t i = 0; i < n; ++i) {
        if (cols[i].t == cols[i].q) {
            j++;
            ts++;
        } else {
            if (cols[i].t != 4) ts++;
            j = 0;
        }
        if (j == LEN) {
            ts -= LEN;
            shift = i + 1 - LEN;
            break;
        }
    }
    if (shift == n) return false;
    j = 0;
    size_t end = n;
    for (size_t i = n; i-- > 0;) {
        if (cols[i].t == cols[i].q)
            j++;
        else
            j = 0;
        if (j == LEN) {
            end = i + LEN;
            break;
        }
    }
    if (end <= shift) return false;
    cols.erase(cols.begin() + (long)end, cols.end());
    cols.erase(cols.begin(), cols.begin() + (long)shift);
    t_start = ts;
    return true;
}

void pack_read(Synth &S, const std::vector<Col> &cols, uint32_t t_start) {
    np2_read_t r;
    memset(&r, 0, sizeof r);
    r.aln_t_s = t_start;
    r.aln_t_e = t_start;
    r.n_cols = (uint32_t)cols.size();
    size_t off = (S.nibbles.size() + 15) & ~(size_t)15;
    size_t nbytes = ((cols.size() + 1) >> 1) + 1; // AlignSeq::new: vec![0; len + 1]
    S.nibbles.resize(off + nbytes, 0);
    r.nib_off = off;
    uint8_t *dst = S.nibbles.data() + off;
    size_t i = 0;
    for (const Col &c : cols) {
        uint8_t b = c.q;
        if (c.t == 4)
            b |= 8;
        else if (i != 0)
            r.aln_t_e += 1;
        dst[i >> 1] |= (i & 1) ? b : (uint8_t)(b << 4);
        i++;
    }
    dst[i >> 1] |= (i & 1) ? 15 : 255;
    S.reads.push_back(r);
}

void generate(Synth &S) {
    const Params &P = S.P;
    Rng rng(P.seed);
    uint32_t L = P.L;
    S.hap1.resize(L);
    for (uint32_t i = 0; i < L; ++i) S.hap1[i] = (uint8_t)rng.below(4);
    Rng r2(P.seed ^ 0x1234567ULL), r3(P.seed ^ 0x89abcdefULL), r4(P.seed ^ 0x5555aaaaULL);
    mutate_hap2(S, r2);
    mutate_asm(S, r3);
    S.hap_seq[0].resize(L);
    for (uint32_t i = 0; i < L; ++i) S.hap_seq[0][i] = ACGT[S.hap1[i]];
    S.hap_seq[1] = render(S.hap2);
    S.asm_seq = render(S.asmr);
    S.asm_pos.resize(L + 1);
    {
        uint32_t p = 0;
        for (uint32_t i = 0; i < L; ++i) {
            S.asm_pos[i] = p;
            if (S.asmr.base[i] < 4) p++;
            if (auto s = S.asmr.ins_at(i)) p += (uint32_t)s->size();
        }
        S.asm_pos[L] = p;
    }
    // read 0 == the contig aligned to itself (main.rs:1732-1739)
    {
        std::vector<Col> cols(S.asm_seq.size());
        for (size_t i = 0; i < cols.size(); ++i) {
            uint8_t c = (uint8_t)(std::strchr("ACGT", S.asm_seq[i]) - "ACGT");
            cols[i] = {c, c};
        }
        pack_read(S, cols, 0);
    }
    uint64_t n_reads = (uint64_t)((double)P.depth * L / P.read_len_mean + 0.5);
    struct RS {
        uint32_t a, b;
        int hap;
    };
    std::vector<RS> rs;
    rs.reserve(n_reads);
    for (uint64_t r = 0; r < n_reads; ++r) {
        double len = P.read_len_mean + P.read_len_sd * r4.normal();
        if (len < P.read_len_min) len = P.read_len_min;
        int64_t l = (int64_t)len;
        int64_t a = (int64_t)(r4.uni() * ((double)L + (double)l)) - l;
        int64_t b = a + l;
        if (a < 0) a = 0;
        if (b > (int64_t)L) b = L;
        if (b - a < (int64_t)P.read_len_min) continue;
        int hap = P.diploid ? (int)(r4.next() >> 63) : 0;
        rs.push_back({(uint32_t)a, (uint32_t)b, hap});
    }
    std::stable_sort(rs.begin(), rs.end(), [](const RS &x, const RS &y) { return x.a < y.a; });
    std::vector<Col> cols;
    for (const RS &r : rs) {
        uint32_t ts;
        make_read(S, r4, r.hap, r.a, r.b, cols, ts);
        if (!trim8(cols, ts)) continue;
        if (cols.size() <= 500) continue; // aln_len() <= min_map_len (main.rs:1800)
        pack_read(S, cols, ts);
    }
    S.nibbles.resize(((S.nibbles.size() + 15) & ~(size_t)15) + 64, 0); // readable tail padding
}

inline uint64_t yak_hash64(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

void collect_kmers(const std::string &s, uint32_t k, std::vector<uint64_t> &out) {
    uint64_t mask = (1ULL << (2 * k)) - 1, shift = 2 * (k - 1), fw = 0, rv = 0;
    uint32_t l = 0;
    for (char ch : s) {
        const char *q = std::strchr("ACGT", ch);
        if (!q || !ch) {
            l = 0;
            continue;
        }
        uint64_t c = (uint64_t)(q - "ACGT");
        fw = (fw << 2 | c) & mask;
        rv = (rv >> 2) | (3 ^ c) << shift;
        if (++l >= k) out.push_back(yak_hash64(fw < rv ? fw : rv, mask));
    }
}

} // namespace

extern "C" {

typedef struct np2s_params {
    uint64_t seed;
    uint32_t L;
    uint32_t depth;
    uint32_t diploid;
    uint32_t read_len_min;
    double snp_rate, hap_indel_rate, asm_err_rate, read_err_rate;
    double read_len_mean, read_len_sd;
} np2s_params_t;

void *np2s_generate(const np2s_params_t *p) {
    Synth *S = new Synth();
    S->P.seed = p->seed;
    S->P.L = p->L;
    S->P.depth = p->depth;
    S->P.diploid = p->diploid;
    S->P.read_len_min = p->read_len_min;
    S->P.snp_rate = p->snp_rate;
    S->P.hap_indel_rate = p->hap_indel_rate;
    S->P.asm_err_rate = p->asm_err_rate;
    S->P.read_err_rate = p->read_err_rate;
    S->P.read_len_mean = p->read_len_mean;
    S->P.read_len_sd = p->read_len_sd;
    generate(*S);
    return S;
}
void np2s_free(void *h) { delete (Synth *)h; }

const char *np2s_ref(void *h, uint32_t *L) {
    Synth *S = (Synth *)h;
    *L = (uint32_t)S->asm_seq.size();
    return S->asm_seq.data();
}
const char *np2s_hap(void *h, int which, uint32_t *L) {
    Synth *S = (Synth *)h;
    *L = (uint32_t)S->hap_seq[which].size();
    return S->hap_seq[which].data();
}
const np2_read_t *np2s_reads(void *h, uint32_t *n) {
    Synth *S = (Synth *)h;
    *n = (uint32_t)S->reads.size();
    return S->reads.data();
}
const uint8_t *np2s_nibbles(void *h, uint64_t *nbytes) {
    Synth *S = (Synth *)h;
    *nbytes = S->nibbles.size();
    return S->nibbles.data();
}

// The reads of the contig (index >= 1) as BAM alignment records (CIGAR from the packed columns: M / I / D runs, 4-bit
// SEQ, missing qualities, MAPQ 60, alternating strand flag), coordinate-sorted and concatenated; rec_off[i] .. rec_off[i + 1] delimits record
// i, pos[i] / ref_len[i] are what an index needs.  Test / bench infrastructure: whole-assembly BAM files without a
// per-column Python loop.  Returns the number of records; buffers are malloc'ed (np2s_free_buf).
// pos0: added to every position (a contig laid out of several generated pieces: concat_pileups); name0: number of the first record's name
uint32_t np2s_bam_records_at(void *h, int32_t tid, uint32_t pos0, uint32_t name0, uint8_t **blob, uint64_t **rec_off, int32_t **pos,
                             uint32_t **ref_len);
uint32_t np2s_bam_records(void *h, int32_t tid, uint8_t **blob, uint64_t **rec_off, int32_t **pos, uint32_t **ref_len) {
    return np2s_bam_records_at(h, tid, 0, 0, blob, rec_off, pos, ref_len);
}
uint32_t np2s_bam_records_at(void *h, int32_t tid, uint32_t pos0, uint32_t name0, uint8_t **blob, uint64_t **rec_off, int32_t **pos,
                             uint32_t **ref_len) {
    Synth *S = (Synth *)h;
    const uint32_t n = S->reads.size() > 1 ? (uint32_t)S->reads.size() - 1 : 0;
    std::vector<uint8_t> out;
    uint64_t *off = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    int32_t *ps = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    uint32_t *rl = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n + 1));
    static const uint8_t enc4[8] = {1, 2, 4, 8, 0, 15, 3, 15}; // A C G T - N M
    auto reg2bin = [](int64_t beg, int64_t end) -> uint16_t {
        --end;
        if (beg >> 14 == end >> 14) return (uint16_t)(((1 << 15) - 1) / 7 + (beg >> 14));
        if (beg >> 17 == end >> 17) return (uint16_t)(((1 << 12) - 1) / 7 + (beg >> 17));
        if (beg >> 20 == end >> 20) return (uint16_t)(((1 << 9) - 1) / 7 + (beg >> 20));
        if (beg >> 23 == end >> 23) return (uint16_t)(((1 << 6) - 1) / 7 + (beg >> 23));
        if (beg >> 26 == end >> 26) return (uint16_t)(((1 << 3) - 1) / 7 + (beg >> 26));
        return 0;
    };
    auto put32 = [&](std::vector<uint8_t> &v, uint32_t x) {
        for (int k = 0; k < 4; ++k) v.push_back((uint8_t)(x >> (8 * k)));
    };
    std::vector<uint32_t> cig;
    std::vector<uint8_t> seq;
    std::vector<uint32_t> order(n); // coordinate-sorted, ties in read order
    for (uint32_t i = 0; i < n; ++i) order[i] = i + 1;
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t b) { return S->reads[a].aln_t_s < S->reads[b].aln_t_s; });
    for (uint32_t i = 0; i < n; ++i) {
        const np2_read_t &rd = S->reads[order[i]];
        const uint8_t *nb = S->nibbles.data() + rd.nib_off;
        cig.clear();
        seq.clear();
        uint32_t rlen = 0;
        for (uint32_t c = 0; c < rd.n_cols; ++c) {
            const uint8_t x = (c & 1) ? (nb[c >> 1] & 15) : (nb[c >> 1] >> 4);
            const uint8_t q = x & 7;
            const uint32_t op = (x & 8) ? 1u : (q == 4 ? 2u : 0u); // I, D, M
            if (op != 2) seq.push_back(enc4[q]);
            if (op != 1) ++rlen;
            if (!cig.empty() && (cig.back() & 15) == op) cig.back() += 16;
            else cig.push_back(16 | op);
        }
        char name[24];
        const int ln = snprintf(name, sizeof name, "r%u", name0 + i) + 1;
        off[i] = out.size();
        ps[i] = (int32_t)(pos0 + rd.aln_t_s);
        rl[i] = rlen;
        const uint32_t l_seq = (uint32_t)seq.size();
        const uint32_t body = 32 + (uint32_t)ln + 4 * (uint32_t)cig.size() + (l_seq + 1) / 2 + l_seq;
        put32(out, body);
        put32(out, (uint32_t)tid);
        put32(out, pos0 + rd.aln_t_s);
        out.push_back((uint8_t)ln);
        out.push_back(60);
        const uint16_t bin = reg2bin((int64_t)pos0 + rd.aln_t_s, (int64_t)pos0 + rd.aln_t_s + (rlen ? rlen : 1));
        out.push_back((uint8_t)bin), out.push_back((uint8_t)(bin >> 8));
        out.push_back((uint8_t)cig.size()), out.push_back((uint8_t)(cig.size() >> 8));
        const uint16_t flag = (i & 1) ? 16 : 0;
        out.push_back((uint8_t)flag), out.push_back((uint8_t)(flag >> 8));
        put32(out, l_seq);
        put32(out, 0xFFFFFFFFu), put32(out, 0xFFFFFFFFu), put32(out, 0);
        out.insert(out.end(), name, name + ln);
        for (uint32_t w : cig) put32(out, w);
        for (uint32_t k = 0; k < l_seq; k += 2) out.push_back((uint8_t)((seq[k] << 4) | (k + 1 < l_seq ? seq[k + 1] : 0)));
        out.insert(out.end(), l_seq, (uint8_t)0xFF);
    }
    off[n] = out.size();
    *blob = (uint8_t *)malloc(out.size() ? out.size() : 1);
    memcpy(*blob, out.data(), out.size());
    *rec_off = off, *pos = ps, *ref_len = rl;
    return n;
}
void np2s_free_buf(void *p) { free(p); }

// Build a yak v2 table (pre = 10) from the true haplotypes: count = min(1023, Poisson(lambda * m))
// where m is the canonical k-mer's multiplicity over the haplotype(s).  Zero counts are dropped.
int np2s_yak_build_multi(void **hs_in, uint32_t n_h, uint32_t k, double lambda, uint64_t seed, const uint64_t **words,
                         uint64_t *n_words, const uint64_t **bucket_off) {
    if (k < 2 || k >= 32 || n_h == 0) return -1;
    Synth *S = (Synth *)hs_in[0]; // the table is owned by the first generator
    std::vector<uint64_t> hs;
    for (uint32_t g = 0; g < n_h; ++g) { // one table over every contig of the assembly (what `yak count` produces)
        Synth *G = (Synth *)hs_in[g];
        collect_kmers(G->hap_seq[0], k, hs);
        if (G->P.diploid) collect_kmers(G->hap_seq[1], k, hs);
    }
    std::sort(hs.begin(), hs.end());
    Rng rng(seed ^ (0xabcdULL * k));
    std::vector<std::vector<uint64_t>> buckets(1024);
    for (size_t i = 0; i < hs.size();) {
        size_t j = i;
        while (j < hs.size() && hs[j] == hs[i]) ++j;
        uint32_t c = rng.poisson(lambda * (double)(j - i));
        if (c > 1023) c = 1023;
        if (c > 0) buckets[hs[i] & 1023].push_back((hs[i] >> 10) << 10 | c);
        i = j;
    }
    S->yak_words.clear();
    S->yak_off.assign(1025, 0);
    // yak dumps each bucket in its own hash-table order; emulate "unordered" by a keyed shuffle
    for (size_t b = 0; b < 1024; ++b) {
        auto &v = buckets[b];
        for (size_t i = v.size(); i > 1; --i) std::swap(v[i - 1], v[rng.below((uint32_t)i)]);
        S->yak_words.insert(S->yak_words.end(), v.begin(), v.end());
        S->yak_off[b + 1] = S->yak_words.size();
    }
    *words = S->yak_words.data();
    *n_words = S->yak_words.size();
    *bucket_off = S->yak_off.data();
    return 0;
}
int np2s_yak_build(void *h, uint32_t k, double lambda, uint64_t seed, const uint64_t **words,
                   uint64_t *n_words, const uint64_t **bucket_off) {
    return np2s_yak_build_multi(&h, 1, k, lambda, seed, words, n_words, bucket_off);
}

// The same table recipe on `n_threads` host threads, for chromosome-scale test inputs (a 248 Mb diploid contig is 5 x 10^8
// k-mers: 80-100 s through the serial builder).  Haplotypes are cut into pieces whose k-mers go to per-thread lists of the
// 1024 file buckets (hash & 1023); every bucket is then sorted, counted and drawn with a random stream of its own.  The
// table has the same distribution as the serial one but NOT the same words (the serial builder draws one stream over
// the globally sorted k-mers), so seeded small inputs keep using np2s_yak_build_multi.
int np2s_yak_build_multi_mt(void **hs_in, uint32_t n_h, uint32_t k, double lambda, uint64_t seed, uint32_t n_threads,
                            const uint64_t **words, uint64_t *n_words, const uint64_t **bucket_off) {
    if (k < 2 || k >= 32 || n_h == 0) return -1;
    if (n_threads < 1) n_threads = 1;
    Synth *S = (Synth *)hs_in[0];
    struct Piece {
        const std::string *s;
        size_t lo, hi; // k-mers ENDING in [lo, hi) — the scan starts k - 1 characters earlier
    };
    std::vector<Piece> pieces;
    const size_t PIECE = 1 << 20;
    for (uint32_t g = 0; g < n_h; ++g) {
        Synth *G = (Synth *)hs_in[g];
        for (int h = 0; h < (G->P.diploid ? 2 : 1); ++h) {
            const std::string &q = G->hap_seq[h];
            for (size_t lo = 0; lo < q.size(); lo += PIECE) pieces.push_back({&q, lo, std::min(q.size(), lo + PIECE)});
        }
    }
    const uint64_t mask = (1ULL << (2 * k)) - 1, shift = 2 * (k - 1);
    std::vector<std::vector<std::vector<uint64_t>>> part(n_threads, std::vector<std::vector<uint64_t>>(1024));
    std::atomic<size_t> next_piece{0};
    auto collect = [&](uint32_t t) {
        uint8_t code[256];
        memset(code, 4, sizeof code);
        code[(int)'A'] = 0, code[(int)'C'] = 1, code[(int)'G'] = 2, code[(int)'T'] = 3;
        auto &mine = part[t];
        for (;;) {
            const size_t pi = next_piece.fetch_add(1);
            if (pi >= pieces.size()) break;
            const Piece &pc = pieces[pi];
            const std::string &q = *pc.s;
            uint64_t fw = 0, rv = 0;
            uint32_t l = 0;
            for (size_t i = pc.lo >= k - 1 ? pc.lo - (k - 1) : 0; i < pc.hi; ++i) {
                const uint64_t c = code[(unsigned char)q[i]];
                if (c > 3) {
                    l = 0;
                    continue;
                }
                fw = (fw << 2 | c) & mask;
                rv = (rv >> 2) | (3 ^ c) << shift;
                if (++l >= k && i >= pc.lo) {
                    const uint64_t x = yak_hash64(fw < rv ? fw : rv, mask);
                    mine[x & 1023].push_back(x);
                }
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < n_threads; ++t) th.emplace_back(collect, t);
        for (auto &x : th) x.join();
    }
    std::vector<std::vector<uint64_t>> buckets(1024);
    std::atomic<uint32_t> next_bucket{0};
    auto reduce = [&]() {
        std::vector<uint64_t> v;
        for (;;) {
            const uint32_t b = next_bucket.fetch_add(1);
            if (b >= 1024) break;
            v.clear();
            for (uint32_t t = 0; t < n_threads; ++t) {
                v.insert(v.end(), part[t][b].begin(), part[t][b].end());
                std::vector<uint64_t>().swap(part[t][b]);
            }
            std::sort(v.begin(), v.end());
            Rng rng(seed ^ (0xabcdULL * k) ^ (0x9E3779B97F4A7C15ULL * (b + 1)));
            auto &out = buckets[b];
            for (size_t i = 0; i < v.size();) {
                size_t j = i;
                while (j < v.size() && v[j] == v[i]) ++j;
                uint32_t c = rng.poisson(lambda * (double)(j - i));
                if (c > 1023) c = 1023;
                if (c > 0) out.push_back((v[i] >> 10) << 10 | c);
                i = j;
            }
            for (size_t i = out.size(); i > 1; --i) std::swap(out[i - 1], out[rng.below((uint32_t)i)]);
        }
    };
    {
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < n_threads; ++t) th.emplace_back(reduce);
        for (auto &x : th) x.join();
    }
    S->yak_off.assign(1025, 0);
    for (size_t b = 0; b < 1024; ++b) S->yak_off[b + 1] = S->yak_off[b] + buckets[b].size();
    S->yak_words.resize(S->yak_off[1024]);
    {
        std::atomic<uint32_t> nb{0};
        auto place = [&]() {
            for (;;) {
                const uint32_t b = nb.fetch_add(1);
                if (b >= 1024) break;
                if (!buckets[b].empty()) memcpy(S->yak_words.data() + S->yak_off[b], buckets[b].data(), 8 * buckets[b].size());
                std::vector<uint64_t>().swap(buckets[b]);
            }
        };
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < n_threads; ++t) th.emplace_back(place);
        for (auto &x : th) x.join();
    }
    *words = S->yak_words.data();
    *n_words = S->yak_words.size();
    *bucket_off = S->yak_off.data();
    return 0;
}

// Pack explicit (target, query) gapped strings into the boundary format — the literal
// AlignSeq::new (src/main.rs:279-312) for hand-written test pileups.  `t`/`q` are ASCII
// gapped strings of equal length n; returns bytes written to `dst` (needs (n+1)/2+1).
uint64_t np2s_pack_alignment(const char *t, const char *q, uint32_t n, uint32_t aln_t_s,
                             uint8_t *dst, uint32_t *aln_t_e) {
    static const uint8_t SEQ_NUM[128] = {
        65, 67, 71, 84, 45, 78, 77, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
        4,  4,  4,  4,  4,  4,  4,  4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
        4,  4,  4,  4,  4,  4,  4,  4, 4, 4, 4, 4, 4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 6,
        5,  4,  4,  4,  4,  4,  3,  3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 0, 4, 1, 4, 4, 4, 2,
        4,  4,  4,  4,  4,  6,  5,  4, 4, 4, 4, 4, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};
    uint64_t nbytes = ((uint64_t)(n + 1) >> 1) + 1;
    memset(dst, 0, nbytes);
    uint32_t te = aln_t_s;
    uint32_t i = 0;
    for (; i < n; ++i) {
        uint8_t b = SEQ_NUM[(unsigned char)q[i] & 127];
        if (t[i] == '-')
            b |= 8;
        else if (i != 0)
            te += 1;
        dst[i >> 1] |= (i & 1) ? b : (uint8_t)(b << 4);
    }
    dst[i >> 1] |= (i & 1) ? 15 : 255;
    *aln_t_e = te;
    return nbytes;
}
}
