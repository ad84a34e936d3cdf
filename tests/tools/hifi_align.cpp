// Test infrastructure only: a small seed-chain-extend aligner for HiFi reads against a short assembly, used by
// tests/golden/make_ref_bundle.py to produce the sorted BAM that the reference's test/hh.sh:8 gets from
// `minimap2 -ax map-hifi | samtools sort` (neither tool exists here).  Not part of the product.
//
//   hifi_align <asm.fa.gz> <reads.fa.gz>  ->  stdout: name \t flag \t tid \t pos \t mapq \t CIGAR \t SEQ
//
// Exact 19-mer anchors unique in the assembly, the best diagonal band per strand, a colinear chain (LIS), greedy
// left-to-right exact-match extension, affine-gap global alignment between anchors and end-bonus extension (then soft
// clips) at the read ends.  Indels therefore come out right-aligned and identical across reads with the same context.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include <zlib.h>

namespace {

struct Rec {
    std::string name, seq;
};

std::vector<Rec> read_fasta(const char *path) {
    std::vector<Rec> out;
    gzFile f = gzopen(path, "rb");
    if (!f) {
        std::fprintf(stderr, "cannot open %s\n", path);
        std::exit(1);
    }
    static char buf[1 << 16];
    while (gzgets(f, buf, sizeof buf)) {
        size_t n = std::strlen(buf);
        while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) buf[--n] = 0;
        if (buf[0] == '>') {
            out.push_back({});
            char *sp = std::strpbrk(buf + 1, " \t");
            if (sp) *sp = 0;
            out.back().name = buf + 1;
        } else if (!out.empty()) {
            for (size_t i = 0; i < n; ++i) buf[i] = (char)std::toupper((unsigned char)buf[i]);
            out.back().seq.append(buf, n);
        }
    }
    gzclose(f);
    return out;
}

inline int code(char c) {
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    }
    return -1;
}
std::string revcomp(const std::string &s) {
    std::string r(s.rbegin(), s.rend());
    for (char &c : r) {
        const int k = code(c);
        if (k >= 0) c = "TGCA"[k];
    }
    return r;
}

constexpr int K = 19;
struct Hit {
    int32_t tid, pos;
};
using Index = std::unordered_map<uint64_t, std::vector<Hit>>;

template <class F> void each_kmer(const std::string &s, F f) {
    const uint64_t mask = (1ULL << (2 * K)) - 1;
    uint64_t w = 0;
    int l = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        const int c = code(s[i]);
        if (c < 0) {
            l = 0;
            continue;
        }
        w = ((w << 2) | (uint64_t)c) & mask;
        if (++l >= K) f((int32_t)(i + 1 - K), w);
    }
}

// ---- affine-gap alignment of q[0,lq) against r[0,lr); ops: M (match/mismatch), I (read only), D (assembly only)
struct Aln {
    std::string ops;
    int qi = 0, ri = 0; // consumed lengths
    bool ok = true;
};
constexpr int MA = 2, MI = -4, GO = 4, GE = 2, END_BONUS = 20;
constexpr int NEG = -(1 << 28);

Aln gotoh(const char *q, int lq, const char *r, int lr, bool extend) {
    Aln out;
    if ((int64_t)(lq + 1) * (lr + 1) > (int64_t)40 << 20) {
        out.ok = false;
        return out;
    }
    const int W = lr + 1;
    std::vector<uint8_t> tb((size_t)(lq + 1) * W, 0);
    std::vector<int> H(W), I(W, NEG);
    H[0] = 0;
    for (int j = 1; j <= lr; ++j) {
        H[j] = -GO - GE * j;
        tb[j] = 2 | (j > 1 ? 8 : 0);
    }
    int best = extend ? (lq == 0 ? END_BONUS : 0) : NEG, bi = 0, bj = 0;
    if (extend && lq == 0) bi = 0, bj = 0;
    for (int i = 1; i <= lq; ++i) {
        int diag = H[0];
        H[0] = -GO - GE * i;
        I[0] = H[0];
        tb[(size_t)i * W] = 1 | (i > 1 ? 4 : 0);
        int D = NEG;
        for (int j = 1; j <= lr; ++j) {
            uint8_t t = 0;
            // vertical (read-only) gap
            const int io = H[j] - GO - GE, ie = I[j] - GE;
            int iv = io;
            if (ie > io) iv = ie, t |= 4;
            I[j] = iv;
            const int dopen = H[j - 1] - GO - GE, de = D - GE;
            int dv = dopen;
            if (de > dopen) dv = de, t |= 8;
            D = dv;
            int h = diag + (q[i - 1] == r[j - 1] ? MA : MI), src = 0;
            if (iv > h) h = iv, src = 1;
            if (dv > h) h = dv, src = 2;
            diag = H[j];
            H[j] = h;
            tb[(size_t)i * W + j] = t | (uint8_t)src;
            if (extend) {
                const int s = h + (i == lq ? END_BONUS : 0);
                if (s > best || (s == best && i > bi)) best = s, bi = i, bj = j;
            }
        }
        if (extend && lr == 0) {
            const int s = H[0] + (i == lq ? END_BONUS : 0);
            if (s > best) best = s, bi = i, bj = 0;
        }
    }
    int i = extend ? bi : lq, j = extend ? bj : lr;
    out.qi = i, out.ri = j;
    int state = 0; // 0 H, 1 I, 2 D
    std::string ops;
    while (i > 0 || j > 0) {
        const uint8_t t = tb[(size_t)i * W + j];
        if (state == 0) {
            const int src = t & 3;
            if (src == 0) {
                ops.push_back('M');
                --i, --j;
            } else
                state = src;
        } else if (state == 1) {
            ops.push_back('I');
            if (!(t & 4)) state = 0;
            --i;
        } else {
            ops.push_back('D');
            if (!(t & 8)) state = 0;
            --j;
        }
    }
    std::reverse(ops.begin(), ops.end());
    out.ops = ops;
    return out;
}

std::string to_cigar(const std::string &ops) {
    std::string c;
    for (size_t i = 0; i < ops.size();) {
        size_t j = i;
        while (j < ops.size() && ops[j] == ops[i]) ++j;
        c += std::to_string(j - i);
        c.push_back(ops[i]);
        i = j;
    }
    return c;
}

} // namespace

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: hifi_align asm.fa.gz reads.fa.gz\n");
        return 1;
    }
    const std::vector<Rec> ctgs = read_fasta(argv[1]);
    const std::vector<Rec> reads = read_fasta(argv[2]);
    Index idx;
    for (size_t t = 0; t < ctgs.size(); ++t)
        each_kmer(ctgs[t].seq, [&](int32_t p, uint64_t w) { idx[w].push_back({(int32_t)t, p}); });

    size_t n_out = 0;
    for (const Rec &rd : reads) {
        struct A {
            int32_t q, r;
        };
        std::vector<A> bestc;
        int best_tid = -1, best_strand = 0;
        std::string best_seq;
        for (int strand = 0; strand < 2; ++strand) {
            const std::string s = strand ? revcomp(rd.seq) : rd.seq;
            std::vector<std::vector<A>> per(ctgs.size());
            each_kmer(s, [&](int32_t qp, uint64_t w) {
                auto it = idx.find(w);
                if (it == idx.end() || it->second.size() != 1) return; // anchors unique in the assembly
                per[it->second[0].tid].push_back({qp, it->second[0].pos});
            });
            for (size_t t = 0; t < ctgs.size(); ++t) {
                auto &h = per[t];
                if (h.size() < 20) continue;
                // densest diagonal band (512 wide, half-overlapping)
                std::unordered_map<int32_t, int> hist;
                for (const A &a : h) {
                    const int32_t d = (a.r - a.q + (1 << 28)) >> 8;
                    ++hist[d];
                    ++hist[d + 1];
                }
                int32_t bd = 0;
                int bc = 0;
                for (auto &kv : hist)
                    if (kv.second > bc || (kv.second == bc && kv.first < bd)) bc = kv.second, bd = kv.first;
                std::vector<A> sel;
                for (const A &a : h) {
                    const int32_t d = (a.r - a.q + (1 << 28)) >> 8;
                    if (d == bd || d + 1 == bd) sel.push_back(a);
                }
                // colinear chain: longest strictly increasing subsequence in r (sel is sorted by q)
                std::vector<int32_t> tail_r;
                std::vector<int> tail_i, prev(sel.size(), -1);
                for (size_t i = 0; i < sel.size(); ++i) {
                    const size_t pos = std::lower_bound(tail_r.begin(), tail_r.end(), sel[i].r) - tail_r.begin();
                    if (pos == tail_r.size()) {
                        tail_r.push_back(sel[i].r);
                        tail_i.push_back((int)i);
                    } else {
                        tail_r[pos] = sel[i].r;
                        tail_i[pos] = (int)i;
                    }
                    prev[i] = pos ? tail_i[pos - 1] : -1;
                }
                std::vector<A> chain;
                for (int i = tail_i.empty() ? -1 : tail_i.back(); i >= 0; i = prev[i]) chain.push_back(sel[i]);
                std::reverse(chain.begin(), chain.end());
                if (chain.size() > bestc.size()) bestc = chain, best_tid = (int)t, best_strand = strand, best_seq = s;
            }
        }
        if (best_tid < 0 || bestc.size() < 50) continue;
        const std::string &q = best_seq, &r = ctgs[best_tid].seq;
        const int lq = (int)q.size(), lr = (int)r.size();
        // left end: extend from the first anchor towards the read start
        std::string ops;
        int cq = bestc[0].q, cr = bestc[0].r;
        int clip_l = 0, pos = cr;
        {
            int eq = std::min(cq, 4000), er = std::min(cr, eq + 100);
            std::string qq(q.begin() + (cq - eq), q.begin() + cq), rr(r.begin() + (cr - er), r.begin() + cr);
            std::reverse(qq.begin(), qq.end());
            std::reverse(rr.begin(), rr.end());
            Aln a = gotoh(qq.data(), eq, rr.data(), er, true);
            std::reverse(a.ops.begin(), a.ops.end());
            while (!a.ops.empty() && a.ops[0] != 'M') { // never start on a gap
                if (a.ops[0] == 'I') --a.qi; else --a.ri;
                a.ops.erase(a.ops.begin());
            }
            ops = a.ops;
            clip_l = cq - a.qi;
            pos = cr - a.ri;
        }
        bool ok = true;
        for (const A &a : bestc) {
            if (a.q >= cq && a.r >= cr) {
                const int gq = a.q - cq, gr = a.r - cr;
                if (gq == 0 && gr == 0) {
                } else if (gq == 0) {
                    ops.append((size_t)gr, 'D');
                } else if (gr == 0) {
                    ops.append((size_t)gq, 'I');
                } else {
                    Aln g = gotoh(q.data() + cq, gq, r.data() + cr, gr, false);
                    if (!g.ok) {
                        ok = false;
                        break;
                    }
                    ops += g.ops;
                }
                ops.append((size_t)K, 'M');
                cq = a.q + K, cr = a.r + K;
            } else if (a.q - cq == a.r - cr && a.q + K > cq) {
                const int ext = a.q + K - cq;
                ops.append((size_t)ext, 'M');
                cq += ext, cr += ext;
            }
        }
        if (!ok) continue;
        int clip_r = 0;
        {
            int eq = std::min(lq - cq, 4000), er = std::min(lr - cr, eq + 100);
            Aln a = gotoh(q.data() + cq, eq, r.data() + cr, er, true);
            while (!a.ops.empty() && a.ops.back() != 'M') {
                if (a.ops.back() == 'I') --a.qi; else --a.ri;
                a.ops.pop_back();
            }
            ops += a.ops;
            clip_r = lq - cq - a.qi;
        }
        std::string cigar;
        if (clip_l) cigar += std::to_string(clip_l) + "S";
        cigar += to_cigar(ops);
        if (clip_r) cigar += std::to_string(clip_r) + "S";
        std::printf("%s\t%d\t%d\t%d\t60\t%s\t%s\n", rd.name.c_str(), best_strand ? 16 : 0, best_tid, pos, cigar.c_str(),
                    q.c_str());
        ++n_out;
    }
    std::fprintf(stderr, "hifi_align: %zu of %zu reads aligned\n", n_out, reads.size());
    return 0;
}
