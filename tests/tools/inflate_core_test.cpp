// Host twin of the GPU inflater (csrc/np2_inflate.hip): the SAME table builder, symbol decoder and stream loop
// (csrc/np2_inflate_core.hpp) driven by a one-lane machine, against zlib on streams zlib itself produced at every level and
// strategy (stored, fixed and dynamic blocks, several deflate blocks per stream, distances up to 32768, lengths up to 258).
// Built and run by tests/test_inflate_cpu.py; prints the number of streams checked.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <zlib.h>
#include "../../nextpolish2_amd/csrc/np2_inflate_core.hpp"

using namespace np2inf;

struct HostMachine {
    const uint8_t *in;
    uint32_t clen;
    std::vector<uint8_t> out;
    uint8_t lens_[384];
    uint32_t lt[1 << LBITS], dt[1 << DBITS];
    uint16_t ls[MAXL], ds[MAXD], sc[MAXBITS + 2];
    Code lc, dc;
    uint32_t uni(uint32_t v) { return v; }
    bool leader() { return true; }
    uint32_t lane() { return 0; }
    uint32_t lanes() { return 1; }
    void sync() {}
    uint32_t in32(uint32_t off) {
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i)
            if (off + i < clen) v |= (uint32_t)in[off + i] << (8 * i);
        return v;
    }
    void put(uint32_t o, uint32_t b) { out[o] = (uint8_t)b; }
    uint32_t lit_room(uint32_t o, uint32_t isize) { return isize - o < 64u - (o & 63u) ? isize - o : 64u - (o & 63u); } // (the device's rule)
    void put_fast(uint32_t o, uint32_t b) { out[o] = (uint8_t)b; }
    void copy(uint32_t o, uint32_t len, uint32_t dist) {
        for (uint32_t k = 0; k < len; ++k) out[o + k] = out[o - dist + (dist >= len ? k : k % dist)];
    }
    uint32_t fast(uint64_t &, uint32_t &, uint32_t &, uint32_t &, uint32_t, uint32_t) { return ST_OK; } // (the one-lane machine has no wide step)
    void tick(int) {}
    uint32_t slow(int mode, uint32_t bits) { return mode == MODE_LITLEN ? code_slow(lc, ls, bits, mode) : code_slow(dc, ds, bits, mode); }
    uint8_t *lens() { return lens_; }
    uint32_t *lit_table() { return lt; }
    uint32_t *dist_table() { return dt; }
    uint16_t *lit_sym() { return ls; }
    uint16_t *dist_sym() { return ds; }
    uint16_t *scratch16() { return sc; }
    Code &lit_code() { return lc; }
    Code &dist_code() { return dc; }
};

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &src, int level, int strategy, int memlevel) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, memlevel, strategy) != Z_OK) abort();
    std::vector<uint8_t> dst(deflateBound(&zs, src.size()) + 64);
    zs.next_in = const_cast<Bytef *>(src.data());
    zs.avail_in = (uInt)src.size();
    zs.next_out = dst.data();
    zs.avail_out = (uInt)dst.size();
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) abort();
    dst.resize(zs.total_out);
    deflateEnd(&zs);
    return dst;
}

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const int rounds = argc > 2 ? atoi(argv[2]) : 40;
    std::mt19937 rng(seed);
    long checked = 0;
    for (int r = 0; r < rounds; ++r) {
        // data of a few kinds: random bytes, packed nucleotides (a BAM's SEQ), long runs (its QUAL), text-like repeats,
        // repeats at distance ~32768
        for (int kind = 0; kind < 6; ++kind) {
            size_t n = kind == 5 ? 65280 : (size_t)(rng() % 65281);
            if (r == 0 && kind == 0) n = 0;
            if (r == 1 && kind == 0) n = 1;
            std::vector<uint8_t> src(n);
            switch (kind) {
            case 0: for (auto &b : src) b = (uint8_t)rng(); break;
            case 1: for (auto &b : src) b = (uint8_t)((1u << (rng() & 3)) << 4 | (1u << (rng() & 3))); break;
            case 2: { uint8_t v = 0xFF; for (size_t i = 0; i < n; ++i) { if (rng() % 500 == 0) v = (uint8_t)(rng() % 40); src[i] = v; } } break;
            case 3: { for (size_t i = 0; i < n; ++i) src[i] = (i >= 37 && rng() % 8) ? src[i - 37 + (rng() % 3)] : (uint8_t)(rng() % 90); } break;
            case 4: { for (size_t i = 0; i < n; ++i) src[i] = (uint8_t)("ACGT"[rng() & 3]); } break;
            default: { for (size_t i = 0; i < n; ++i) src[i] = i >= 32768 ? src[i - 32768 + (rng() % 64 == 0)] : (uint8_t)rng(); } break;
            }
            for (int level : {0, 1, 4, 6, 9})
                for (int strategy : {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED}) {
                    const int memlevel = (r + level) % 3 == 0 ? 1 : 8; // memLevel 1: many small deflate blocks in one stream
                    const std::vector<uint8_t> z = deflate_raw(src, level, strategy, memlevel);
                    HostMachine m;
                    m.in = z.data(), m.clen = (uint32_t)z.size();
                    m.out.assign(n + 8, 0xA5);
                    const uint32_t st = inflate_stream(m, (uint32_t)z.size(), (uint32_t)n);
                    if (st != ST_OK || memcmp(m.out.data(), src.data(), n) != 0 || m.out[n] != 0xA5) {
                        fprintf(stderr, "FAIL: round %d kind %d n %zu level %d strategy %d memlevel %d: status %u\n", r, kind, n, level, strategy, memlevel, st);
                        return 1;
                    }
                    ++checked;
                    // a damaged stream must end in an error or in different bytes, never in a crash or an overrun
                    if (z.size() > 8 && (rng() & 7) == 0) {
                        std::vector<uint8_t> bad = z;
                        bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() & 7));
                        HostMachine mb;
                        mb.in = bad.data(), mb.clen = (uint32_t)bad.size();
                        mb.out.assign(n + 8, 0xA5);
                        (void)inflate_stream(mb, (uint32_t)bad.size(), (uint32_t)n);
                        if (mb.out[n] != 0xA5) {
                            fprintf(stderr, "FAIL: damaged stream wrote past ISIZE\n");
                            return 1;
                        }
                    }
                }
        }
    }
    printf("%ld\n", checked);
    return 0;
}
