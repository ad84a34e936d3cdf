// Shared host/device definitions for the MI355X NextPolish2 hot path.
// Semantics mirror Nextomics/NextPolish2 v0.2.2 (file:line cited per item); the data
// layout (ref-diff sparse graph) is this project's own design, see DESIGN.md.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

#define NP2_HD __host__ __device__ __forceinline__

namespace np2 {

// ---- AlignBase / 3-column-mer node (src/main.rs:33-185) --------------------------------
struct AlignBase {
    uint32_t t_pos;
    uint16_t delta;
    uint8_t q;
    NP2_HD bool is_head() const { return q == 15; }
    NP2_HD bool eq(const AlignBase &o) const { return q == o.q && delta == o.delta && t_pos == o.t_pos; }
};
NP2_HD AlignBase ab_head(uint32_t t_pos, uint16_t delta) { return AlignBase{t_pos, delta, 15}; }

// Kmer::new (main.rs:84-102): bases = f14 | f12 | b1<<8 | b2<<4 | b3
NP2_HD uint16_t node_bases(const AlignBase &b1, const AlignBase &b2, const AlignBase &b3) {
    unsigned f = 0;
    if (b2.t_pos == b1.t_pos) f |= 0x4000u;
    if (b2.t_pos == b3.t_pos) f |= 0x1000u;
    return (uint16_t)(f | ((unsigned)b1.q << 8) | ((unsigned)b2.q << 4) | (unsigned)b3.q);
}
// Kmer::bases(p) (main.rs:105-184); u32/u16 wrap like the release build
NP2_HD void node_decode(uint16_t bases, uint16_t delta, uint32_t p, AlignBase &a, AlignBase &b, AlignBase &c) {
    a.q = (bases >> 8) & 0xF;
    b.q = (bases >> 4) & 0xF;
    c.q = bases & 0xF;
    if ((bases & 0x5000) == 0x5000) {
        a.t_pos = p, a.delta = delta;
        b.t_pos = p, b.delta = (uint16_t)(delta + 1);
        c.t_pos = p, c.delta = (uint16_t)(delta + 2);
    } else if (bases & 0x1000) {
        a.t_pos = p - 1, a.delta = delta;
        b.t_pos = p, b.delta = 0;
        c.t_pos = p, c.delta = 1;
    } else if (bases & 0x4000) {
        a.t_pos = p - 1, a.delta = delta;
        b.t_pos = p - 1, b.delta = (uint16_t)(delta + 1);
        c.t_pos = p, c.delta = 0;
    } else {
        a.t_pos = p - 2, a.delta = delta;
        b.t_pos = p - 1, b.delta = 0;
        c.t_pos = p, c.delta = 0;
    }
}
// delta of the 3rd column == the sort key of Msa::sort (main.rs:227-229)
NP2_HD uint16_t node_delta3(uint16_t bases, uint16_t delta) {
    if ((bases & 0x5000) == 0x5000) return (uint16_t)(delta + 2);
    if (bases & 0x1000) return 1;
    return 0;
}

// ---- yak hashing (src/utils/kmer.rs:223-233) ------------------------------------------------
NP2_HD uint64_t yak_hash64(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

// code -> ASCII for codes 0..6 (kmer.rs:14: A C G T - N M)
NP2_HD uint8_t code_to_ascii(uint8_t c) {
    const uint64_t tab = 0x004D4E2D54474341ULL; // 'A','C','G','T','-','N','M'
    return (uint8_t)(tab >> (8 * (c & 7)));
}
// ASCII -> code (kmer.rs:11-22 SEQ_NUM): A/a 0, C/c 1, G/g 2, T/t/U/u 3, N/n 5, M/m 6, else 4
NP2_HD uint8_t ascii_to_code(uint8_t ch) {
    switch (ch & 0x7F) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    case 'N': case 'n': return 5;
    case 'M': case 'm': return 6;
    default: return 4;
    }
}

static constexpr uint64_t INVALID_KMER = ~0ULL;   // main.rs:31
static constexpr uint32_t LQSEQ_MAX_CAN_COUNT = 60; // main.rs:30
static constexpr int64_t SCORE_NEG = INT64_MIN >> 1; // main.rs:1661

// consensus base classes for the LQ state machine (main.rs:1586-1625)
enum : uint8_t { CLS_HQ = 0, CLS_LQ = 1, CLS_RESET = 2 };
static constexpr uint32_t GROW_ERR = 0x1000u;   // device error word: a splice round would outgrow the GUESSED growth allowance (retried with the exact one)
static constexpr uint32_t LQ_LIST_ERR = 0x800u; // device error word: the list of low-quality bases overflowed its bound

// checkpoint grid: column index of the reference column at every CKPT-th contig position
static constexpr uint32_t CKPT_SHIFT = 5;
static constexpr uint32_t CKPT = 1u << CKPT_SHIFT;

// packed read nibble access (main.rs:314-322)
NP2_HD uint8_t nib_at(const uint8_t *bytes, uint32_t c) {
    uint8_t t = bytes[c >> 1];
    return (c & 1) ? (t & 15) : (t >> 4);
}


// XCD-aware order of a grid's workgroups (gfx950: workgroups go round the 8 XCDs in turn, each XCD has an L2 of its own).
// A kernel whose neighbouring blocks read the same lines — regions or tiles next to each other along the contig share
// reads, checkpoints, read lists, records — lets block b work on item xcd_order(b, n): the (b / 8)-th item of the (b % 8)-th
// eighth of the items, so that one XCD's L2 sees one stretch of the contig instead of all of it.  A bijection on [0, n).
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t xcd_order(uint32_t b, uint32_t n) {
    if (n < 64) return b;
    const uint32_t q = n >> 3, r = n & 7u, x = b & 7u;
    return x * q + (x < r ? x : r) + (b >> 3);
}
#endif

} // namespace np2
