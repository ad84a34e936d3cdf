// DEFLATE (RFC 1951) decoding pieces shared by the GPU inflater (np2_inflate.hip: one wavefront per BGZF block) and its
// host-side twin (tests/tools/inflate_core_test.cpp, which runs the same table builder and symbol decoder against zlib
// on this container's CPU).  Reference call sites this replaces on the input side: rust-htslib's bgzf reader behind
// bam::IndexedReader::fetch / records (main.rs:1745-1757) — the reference inflates on its reader thread.
//
// Decode tables: a primary table indexed by the next LBITS (DBITS) stream bits — LSB-first, so canonical codes are
// entered bit-reversed — whose entry is `symbol << 4 | code length` (0: the code is longer than the index or unassigned),
// and puff-style canonical arrays (codes per length + symbols sorted by (length, symbol)) for the rare longer codes.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define NP2_INF_HD __host__ __device__ __forceinline__
#else
#define NP2_INF_HD inline
#endif

namespace np2inf {

static constexpr int LBITS = 10;    // primary table bits, literal / length code
static constexpr int DBITS = 8;     // ... distance code
static constexpr int MAXBITS = 15;  // longest code
static constexpr int MAXL = 288;    // literal / length symbols
static constexpr int MAXD = 32;     // distance symbols (30 in use; 30, 31 never appear in a valid stream)

enum Status : uint32_t {
    ST_OK = 0,
    ST_BAD_BTYPE = 1,      // block type 3
    ST_BAD_STORED = 2,     // LEN != ~NLEN
    ST_BAD_LENGTHS = 3,    // code length repeat without a previous length / too many lengths
    ST_OVERSUBSCRIBED = 4, // a code set that is not prefix-free
    ST_BAD_SYMBOL = 5,     // an unassigned code, length symbol 286 / 287, distance symbol 30 / 31
    ST_BAD_DISTANCE = 6,   // distance reaches before the start of the output
    ST_OUT_OVERRUN = 7,    // more output than ISIZE
    ST_OUT_SHORT = 8,      // stream ended before ISIZE bytes
    ST_IN_OVERRUN = 9,     // stream ran past the block's compressed bytes
    ST_NO_END_CODE = 10,   // literal / length code without symbol 256
};
static constexpr uint32_t FAST_END_OF_BLOCK = 0xFFFFFFFEu; // Machine::fast consumed the end-of-block code

struct Code { // one Huffman code (in LDS on the device)
    uint16_t count[MAXBITS + 1]; // codes per length
    uint16_t start[MAXBITS + 2]; // first index of each length in sym[]
    uint16_t first[MAXBITS + 1]; // first canonical code of each length
};

// length / distance base values and extra bits (RFC 1951 3.2.5)
NP2_INF_HD uint32_t len_base(uint32_t s) { // s = symbol - 257, 0 .. 28
    const uint32_t eb = s < 8 ? 0u : (s - 4) >> 2;
    return s == 28 ? 258u : (s < 8 ? 3u + s : 3u + ((4u + (s & 3u)) << eb));
}
NP2_INF_HD uint32_t len_extra(uint32_t s) { return (s < 8 || s == 28) ? 0u : (s - 4) >> 2; }
NP2_INF_HD uint32_t dist_base(uint32_t s) { // s = 0 .. 29
    const uint32_t eb = s < 4 ? 0u : (s - 2) >> 1;
    return s < 4 ? 1u + s : 1u + ((2u + (s & 1u)) << eb);
}
NP2_INF_HD uint32_t dist_extra(uint32_t s) { return s < 4 ? 0u : (s - 2) >> 1; }

NP2_INF_HD uint32_t bit_reverse(uint32_t c, uint32_t n) { // the low n bits of c, reversed
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; ++i) r |= ((c >> i) & 1u) << (n - 1 - i);
    return r;
}

// Step 1 of a table build (one lane / the host): counts per length -> start[], first[]; returns ST_OK or
// ST_OVERSUBSCRIBED.  `len[0 .. n)` = code length of each symbol (0: unused).
NP2_INF_HD uint32_t code_prepare(Code &c, const uint8_t *len, uint32_t n) {
    for (int b = 0; b <= MAXBITS; ++b) c.count[b] = 0;
    for (uint32_t s = 0; s < n; ++s) ++c.count[len[s]];
    c.count[0] = 0;
    int32_t left = 1;
    uint32_t code = 0, at = 0;
    for (int b = 1; b <= MAXBITS; ++b) {
        left <<= 1;
        left -= (int32_t)c.count[b];
        if (left < 0) return ST_OVERSUBSCRIBED;
        code = (code + c.count[b - 1]) << 1;
        c.first[b] = (uint16_t)code;
        c.start[b] = (uint16_t)at;
        at += c.count[b];
    }
    c.start[MAXBITS + 1] = (uint16_t)at;
    c.first[0] = 0, c.start[0] = 0;
    return ST_OK;
}
// Step 2 (one lane / the host; after step 1): sym[] = the used symbols sorted by (length, symbol).  `fill[]`: scratch of
// MAXBITS + 1 running offsets.
NP2_INF_HD void code_sort(const Code &c, const uint8_t *len, uint32_t n, uint16_t *sym, uint16_t *fill) {
    for (int b = 0; b <= MAXBITS; ++b) fill[b] = c.start[b];
    for (uint32_t s = 0; s < n; ++s)
        if (len[s]) sym[fill[len[s]]++] = (uint16_t)s;
}
// Table entries (32 bits) carry everything the decode step needs, so that the step is one lookup and a few shifts — no
// length / distance arithmetic, no second lookup for the extra-bit counts:
//   bits 0-3 code length (0: the code is longer than the index, or unassigned), bits 4-7 number of extra bits,
//   bits 8-9 kind (literal / length code: 0 literal, 1 length, 2 end of block, 3 invalid symbol 286 / 287; distance code:
//   0 distance, 3 invalid symbol 30 / 31; code length code: 0), bits 16-31 value (the literal, the base length, the base
//   distance, the code length symbol).
enum { K_LIT = 0, K_LEN = 1, K_END = 2, K_BAD = 3 };
enum { MODE_PLAIN = 0, MODE_LITLEN = 1, MODE_DIST = 2 };
NP2_INF_HD uint32_t make_entry(uint32_t sym, uint32_t len, int mode) {
    uint32_t kind = K_LIT, extra = 0, val = sym;
    if (mode == MODE_LITLEN && sym >= 256) {
        if (sym == 256) kind = K_END, val = 0;
        else if (sym > 285) kind = K_BAD, val = 0;
        else kind = K_LEN, extra = len_extra(sym - 257), val = len_base(sym - 257);
    } else if (mode == MODE_DIST) {
        if (sym > 29) kind = K_BAD, val = 0;
        else extra = dist_extra(sym), val = dist_base(sym);
    }
    return len | (extra << 4) | (kind << 8) | (val << 16);
}
// Step 3 (lane `lane` of `nl`, or (0, 1) on the host; the table zeroed before): primary table entries of the codes no longer
// than TB bits.
NP2_INF_HD void code_table(const Code &c, const uint8_t *len, const uint16_t *sym, uint32_t *table, int TB, int mode, uint32_t lane, uint32_t nl) {
    const uint32_t used = c.start[MAXBITS + 1];
    for (uint32_t pos = lane; pos < used; pos += nl) {
        const uint32_t s = sym[pos], l = len[s];
        if ((int)l > TB) continue;
        const uint32_t code = (uint32_t)c.first[l] + (pos - c.start[l]);
        const uint32_t e = make_entry(s, l, mode);
        for (uint32_t i = bit_reverse(code, l); i < (1u << TB); i += 1u << l) table[i] = e;
    }
}
// A symbol whose code is longer than the primary index (or an unassigned code): canonical walk over the code lengths
// (puff.c's decode), on the stream bits `bits` (LSB first, at least MAXBITS valid).  Returns the symbol's entry, 0 if no
// code matches.
NP2_INF_HD uint32_t code_slow(const Code &c, const uint16_t *sym, uint32_t bits, int mode) {
    uint32_t code = 0;
    for (int l = 1; l <= MAXBITS; ++l) {
        code |= bits & 1u;
        bits >>= 1;
        const uint32_t cnt = c.count[l];
        if (code - (uint32_t)c.first[l] < cnt) return make_entry(sym[c.start[l] + (code - c.first[l])], (uint32_t)l, mode);
        code <<= 1;
    }
    return 0;
}

// order of the code length code lengths in a dynamic block header (RFC 1951 3.2.7)
NP2_INF_HD uint32_t clen_order(uint32_t i) {
    // 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 packed five bits each
    const uint64_t lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) |
                        (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
    const uint64_t hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
    return i < 12 ? (uint32_t)((lo >> (5 * i)) & 31u) : (uint32_t)((hi >> (5 * (i - 12))) & 31u);
}
// code lengths of the fixed codes (RFC 1951 3.2.6)
NP2_INF_HD uint8_t fixed_lit_len(uint32_t s) { return s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)); }


// ------------------------------------------------------------------------------------------------------------------
// One raw DEFLATE stream (a BGZF block's payload) -> exactly `isize` bytes.  Every lane of the decoding wavefront runs
// this function with the SAME values (the bit buffer, the positions, the symbols: `m.uni()` tells the compiler so —
// v_readfirstlane — and the arithmetic stays on the scalar unit); what differs from lane to lane is inside the machine
// `m`: staging input, copying a match, flushing output.  The host twin's machine has one lane.
//
// struct Machine {
//     uint32_t uni(uint32_t v);                   // a value that is the same in every lane
//     bool leader();                              // one lane (the writer of shared scalars)
//     uint32_t lane(), lanes();
//     void sync();                                // shared-memory writes of this wavefront visible to its lanes
//     uint32_t in32(uint32_t byte_off);           // 4 input bytes (little endian) at a multiple of 4; zeros past the end
//     void put(uint32_t out, uint32_t byte);      // output byte `out`
//     uint32_t lit_room(uint32_t out, uint32_t isize);   // bytes put_fast may take from `out` on without a check (0: none)
//     void put_fast(uint32_t out, uint32_t byte); // output byte `out` inside that allowance
//     void copy(uint32_t out, uint32_t len, uint32_t dist);   // output [out, out + len) = the bytes `dist` back
//     uint32_t fast(uint64_t &bb, uint32_t &bc, uint32_t &next, uint32_t &out, uint32_t isize, uint32_t in_limit);
//                                                 // optional wide step over the current block's symbols: ST_OK (go on with
//                                                 // the step below, state updated), FAST_END_OF_BLOCK, or an error
//     void tick(int phase);                       // optional phase clock (0: a block's tables begin, 1: they are ready)
//     uint32_t slow(int mode, uint32_t bits);     // code_slow on the literal / length (MODE_LITLEN) or distance code
//     uint8_t *lens();                            // 320 code lengths (shared)
//     uint32_t *lit_table(), *dist_table();       // shared decode tables
//     uint16_t *lit_sym(), *dist_sym(), *scratch16();   // shared
//     Code &lit_code(), &dist_code();             // shared
// };
// ------------------------------------------------------------------------------------------------------------------
template <class M> NP2_INF_HD uint32_t inflate_stream(M &m, uint32_t clen, uint32_t isize) {
    uint64_t bb = 0;          // bit buffer (LSB first)
    uint32_t bc = 0;          // valid bits in it
    uint32_t next = 0;        // next input byte to enter the buffer (a multiple of 4)
    uint32_t out = 0;
    const uint32_t in_limit = ((clen + 3u) & ~3u) + 8u; // a valid stream never asks for more
    uint32_t result = ST_OK; // (one way out: the device compiler lays a loop nest with many exits out as a maze of flag tests)
#define NP2_INF_FAIL(code)   \
    do {                     \
        result = (code);     \
        goto np2_inf_done;   \
    } while (0)
#define NP2_INF_NEED(n)                                            \
    do {                                                           \
        if (bc < (uint32_t)(n)) {                                  \
            if (next > in_limit) NP2_INF_FAIL(ST_IN_OVERRUN);             \
            bb |= (uint64_t)m.in32(next) << bc;                    \
            next += 4, bc += 32;                                   \
        }                                                          \
    } while (0)
#define NP2_INF_DROP(n) (bb >>= (n), bc -= (uint32_t)(n))
    for (;;) {
        NP2_INF_NEED(3);
        const uint32_t bfinal = (uint32_t)bb & 1u, btype = ((uint32_t)bb >> 1) & 3u;
        NP2_INF_DROP(3);
        if (btype == 3) NP2_INF_FAIL(ST_BAD_BTYPE);
        if (btype == 0) { // stored
            NP2_INF_DROP(bc & 7u);
            NP2_INF_NEED(32);
            const uint32_t len = (uint32_t)bb & 0xFFFFu, nlen = ((uint32_t)(bb >> 16)) & 0xFFFFu;
            NP2_INF_DROP(32);
            if ((len ^ 0xFFFFu) != nlen) NP2_INF_FAIL(ST_BAD_STORED);
            if (out + len > isize) NP2_INF_FAIL(ST_OUT_OVERRUN);
            for (uint32_t i = 0; i < len; ++i) {
                NP2_INF_NEED(8);
                m.put(out++, (uint32_t)bb & 0xFFu);
                NP2_INF_DROP(8);
            }
        } else {
            m.tick(0); // (a tool's clock: the block's tables begin)
            uint8_t *lens = m.lens();
            uint32_t hlit, hdist;
            if (btype == 1) {
                hlit = 288, hdist = 30;
                m.sync();
                for (uint32_t s = m.lane(); s < 288; s += m.lanes()) lens[s] = fixed_lit_len(s);
                for (uint32_t s = m.lane(); s < 32; s += m.lanes()) lens[288 + s] = 5;
                m.sync();
            } else {
                NP2_INF_NEED(14);
                hlit = ((uint32_t)bb & 31u) + 257u, hdist = (((uint32_t)bb >> 5) & 31u) + 1u;
                const uint32_t hclen = (((uint32_t)bb >> 10) & 15u) + 4u;
                NP2_INF_DROP(14);
                if (hlit > 286 || hdist > 30) NP2_INF_FAIL(ST_BAD_LENGTHS);
                // the code length code: lengths in lens[0, 19), its table in the distance table's place
                m.sync();
                for (uint32_t s = m.lane(); s < 19; s += m.lanes()) lens[s] = 0;
                m.sync();
                for (uint32_t i = 0; i < hclen; ++i) {
                    NP2_INF_NEED(3);
                    if (m.leader()) lens[clen_order(i)] = (uint8_t)((uint32_t)bb & 7u);
                    NP2_INF_DROP(3);
                }
                m.sync();
                uint32_t *ct = m.dist_table(); // 128 entries used
                uint32_t st = ST_OK;
                if (m.leader()) {
                    st = code_prepare(m.dist_code(), lens, 19);
                    if (st == ST_OK) code_sort(m.dist_code(), lens, 19, m.dist_sym(), m.scratch16());
                }
                st = m.uni(st);
                if (st != ST_OK) NP2_INF_FAIL(st);
                for (uint32_t i = m.lane(); i < 128; i += m.lanes()) ct[i] = 0;
                m.sync();
                code_table(m.dist_code(), lens, m.dist_sym(), ct, 7, MODE_PLAIN, m.lane(), m.lanes());
                m.sync();
                // the literal / length and distance code lengths, as one sequence (a repeat may cross from one to the other);
                // written behind the 19 they are decoded with: lens[32 ...)
                uint8_t *ll = lens + 32;
                const uint32_t total = hlit + hdist;
                uint32_t idx = 0, prev = 0;
                while (idx < total) {
                    NP2_INF_NEED(14);
                    const uint32_t e = m.uni(ct[(uint32_t)bb & 127u]);
                    if (!e) NP2_INF_FAIL(ST_BAD_SYMBOL);
                    NP2_INF_DROP(e & 15u);
                    const uint32_t s = e >> 16;
                    uint32_t rep = 1, val = s;
                    if (s == 16) {
                        if (idx == 0) NP2_INF_FAIL(ST_BAD_LENGTHS);
                        rep = 3u + ((uint32_t)bb & 3u), val = prev;
                        NP2_INF_DROP(2);
                    } else if (s == 17) {
                        rep = 3u + ((uint32_t)bb & 7u), val = 0;
                        NP2_INF_DROP(3);
                    } else if (s == 18) {
                        rep = 11u + ((uint32_t)bb & 127u), val = 0;
                        NP2_INF_DROP(7);
                    }
                    if (idx + rep > total) NP2_INF_FAIL(ST_BAD_LENGTHS);
                    if (m.leader())
                        for (uint32_t r = 0; r < rep; ++r) ll[idx + r] = (uint8_t)val;
                    idx += rep, prev = val;
                }
                m.sync();
                // into place: literal / length lengths at lens[0, 288) (zero beyond hlit), distance lengths at lens[288, 320)
                // (the two ranges overlap the staging area: through registers, all reads before all writes)
                uint8_t tmp[6]; // ceil(320 / 64) entries per lane (one lane on the host: see below)
                const uint32_t nl = m.lanes();
                if (nl >= 64) {
                    for (uint32_t j = 0; j < 5; ++j) {
                        const uint32_t s = m.lane() + j * nl;
                        uint8_t v = 0;
                        if (s < 288) v = s < hlit ? ll[s] : 0;
                        else if (s < 320) v = (s - 288) < hdist ? ll[hlit + (s - 288)] : 0;
                        tmp[j] = v;
                    }
                    m.sync();
                    for (uint32_t j = 0; j < 5; ++j) {
                        const uint32_t s = m.lane() + j * nl;
                        if (s < 320) lens[s] = tmp[j];
                    }
                } else { // (host: a plain copy through a stack array)
                    uint8_t all[320];
                    for (uint32_t s = 0; s < 320; ++s) all[s] = s < 288 ? (s < hlit ? ll[s] : 0) : ((s - 288) < hdist ? ll[hlit + (s - 288)] : 0);
                    for (uint32_t s = 0; s < 320; ++s) lens[s] = all[s];
                }
                (void)tmp;
                m.sync();
                if (m.uni(lens[256]) == 0) NP2_INF_FAIL(ST_NO_END_CODE);
            }
            // the two decode tables
            {
                uint32_t st = ST_OK;
                if (m.leader()) {
                    st = code_prepare(m.lit_code(), lens, 288);
                    if (st == ST_OK) code_sort(m.lit_code(), lens, 288, m.lit_sym(), m.scratch16());
                    if (st == ST_OK) st = code_prepare(m.dist_code(), lens + 288, 32);
                    if (st == ST_OK) code_sort(m.dist_code(), lens + 288, 32, m.dist_sym(), m.scratch16());
                }
                st = m.uni(st);
                if (st != ST_OK) NP2_INF_FAIL(st);
                uint32_t *lt = m.lit_table(), *dt = m.dist_table();
                for (uint32_t i = m.lane(); i < (1u << LBITS); i += m.lanes()) lt[i] = 0;
                for (uint32_t i = m.lane(); i < (1u << DBITS); i += m.lanes()) dt[i] = 0;
                m.sync();
                code_table(m.lit_code(), lens, m.lit_sym(), lt, LBITS, MODE_LITLEN, m.lane(), m.lanes());
                code_table(m.dist_code(), lens + 288, m.dist_sym(), dt, DBITS, MODE_DIST, m.lane(), m.lanes());
                m.sync();
            }
            m.tick(1); // (... are ready)
            const uint32_t *lt = m.lit_table(), *dt = m.dist_table();
            for (;;) {
                { // the machine's wide step, if it has one (the device: 64 bit offsets decoded at once, see DevMachine::fast)
                    const uint32_t fs = m.fast(bb, bc, next, out, isize, in_limit);
                    if (fs == FAST_END_OF_BLOCK) break;
                    if (fs != ST_OK) NP2_INF_FAIL(fs);
                }
                NP2_INF_NEED(32);
                // A run of literals, the common case (a BAM's SEQ bytes are all literals), without the general step's checks:
                // the machine says how many bytes may be put unconditionally (`room`: up to ISIZE, and on the device up to the
                // end of the 64-byte group whose bytes wait in the lanes' registers), the bit buffer is good for two codes
                // per refill, and anything that is not a literal of the primary table leaves the run.
                for (uint32_t room = m.lit_room(out, isize); room; room = m.lit_room(out, isize)) {
                    bool leave = false;
                    while (room && bc >= (uint32_t)MAXBITS) {
                        const uint32_t e1 = m.uni(lt[(uint32_t)bb & ((1u << LBITS) - 1u)]);
                        if ((e1 & 0x300u) != 0 || (e1 & 15u) == 0) { // not a literal (or not in the table)
                            leave = true;
                            break;
                        }
                        NP2_INF_DROP(e1 & 15u);
                        m.put_fast(out++, e1 >> 16);
                        --room;
                    }
                    if (leave) break;
                    NP2_INF_NEED(32);
                }
                NP2_INF_NEED(32);
                uint32_t e = m.uni(lt[(uint32_t)bb & ((1u << LBITS) - 1u)]);
                if (!(e & 15u)) {
                    e = m.uni(m.slow(MODE_LITLEN, (uint32_t)bb));
                    if (!e) NP2_INF_FAIL(ST_BAD_SYMBOL);
                }
                NP2_INF_DROP(e & 15u);
                const uint32_t kind = (e >> 8) & 3u;
                if (kind == K_LIT) {
                    if (out >= isize) NP2_INF_FAIL(ST_OUT_OVERRUN);
                    m.put(out++, e >> 16);
                    continue;
                }
                if (kind == K_END) break;
                if (kind == K_BAD) NP2_INF_FAIL(ST_BAD_SYMBOL);
                const uint32_t le = (e >> 4) & 15u;
                const uint32_t len = (e >> 16) + ((uint32_t)bb & ((1u << le) - 1u));
                NP2_INF_DROP(le);
                NP2_INF_NEED(32);
                e = m.uni(dt[(uint32_t)bb & ((1u << DBITS) - 1u)]);
                if (!(e & 15u)) {
                    e = m.uni(m.slow(MODE_DIST, (uint32_t)bb));
                    if (!e) NP2_INF_FAIL(ST_BAD_SYMBOL);
                }
                NP2_INF_DROP(e & 15u);
                if ((e >> 8) & 3u) NP2_INF_FAIL(ST_BAD_SYMBOL);
                const uint32_t de = (e >> 4) & 15u;
                const uint32_t dist = (e >> 16) + ((uint32_t)bb & ((1u << de) - 1u));
                NP2_INF_DROP(de);
                if (dist > out) NP2_INF_FAIL(ST_BAD_DISTANCE);
                if (out + len > isize) NP2_INF_FAIL(ST_OUT_OVERRUN);
                m.copy(out, len, dist);
                out += len;
            }
        }
        if (bfinal) break;
    }
    if ((uint64_t)next * 8u - bc > (uint64_t)clen * 8u) result = ST_IN_OVERRUN;
    else if (out != isize) result = ST_OUT_SHORT;
np2_inf_done:
#undef NP2_INF_NEED
#undef NP2_INF_DROP
#undef NP2_INF_FAIL
    return result;
}

} // namespace np2inf
