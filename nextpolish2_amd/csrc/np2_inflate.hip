// Read extraction on the GPU (gfx950, wave64): BGZF blocks -> inflated BAM stream -> the alignment records of one contig,
// without the host touching a payload byte.  Replaces, for a contig read with NP2_INFLATE=gpu (np2_io.cpp), the host pool's
// inflate + record walk + SEQ gather + SEQ upload; the reference does the same work on its reader thread through
// rust-htslib (bam::IndexedReader::fetch + records(), main.rs:1745-1757).
//
// k_bgzf_inflate — one wavefront per BGZF block (a complete raw DEFLATE stream of known inflated size, <= 64 KiB).
//   Huffman decoding is a serial chain, so the wavefront runs it ONCE: every lane executes np2inf::inflate_stream with the
//   same values (bit buffer, positions, symbols pinned to scalar registers by v_readfirstlane: the chain is scalar
//   arithmetic + one LDS table read per symbol), and the lanes differ only where the work is wide: staging 2 KiB of input
//   (32 bytes a lane), building the decode tables (a symbol per lane), copying a match (a byte per lane, all 64 lanes of a
//   258-byte QUAL run), flushing 4 KiB of output (64 bytes a lane, coalesced).  The last 8 KiB of output live in LDS as a
//   ring — a near back-reference costs an LDS round trip; a far one (beyond 8 KiB, up to DEFLATE's 32 KiB) reads the flushed
//   output back from L2 — next to a 4 KiB input window and the tables: 19.1 KB per wavefront, eight wavefronts per CU,
//   ~2000 blocks in flight.  (The whole 32 KiB window in LDS was four wavefronts per CU, one per SIMD, and a chain of LDS
//   and scalar latencies with nothing to hide them behind: 15.5 ms for the 4267 blocks of an E. coli-sized contig.)
// k_bam_chain_count / k_bam_chain_write — the record walk.  A BAM stream is a chain (a record's length says where the
//   next one starts), but the .bai linear index names a record START every 16 kb of reference: a thread per index entry
//   walks its ~30 records to the next entry — 15 k chains for a human chromosome — counting, then (after a scan of the
//   counts) writing np2_bamrec_t entries whose seq_off points INTO the inflated stream: the columnariser reads SEQ where
//   the inflater left it.  k_bam_cigars copies the CIGAR words, a wavefront per record.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "np2_inflate_core.hpp"
#include "np2_inflate.hpp"
#include "np2_blockscan.hpp"

namespace np2 {

static constexpr uint32_t INF_RING = 8192;   // output window kept in LDS (bytes, a power of two); matches further back read the
                                             // output where it was flushed to (every byte at least INF_RING back has been)
static constexpr uint32_t INF_IN = 4096;     // input window
static constexpr uint32_t INF_CHUNK = 2048;  // input staged per refill
static constexpr uint32_t INF_FLUSH = 2048;  // output written to memory per flush
static constexpr uint32_t INF_WIDE_SPAN = 2048; // most output one wide step may produce (a chain is cut where it would exceed it)
static_assert(INF_WIDE_SPAN >= 258 && INF_RING - INF_WIDE_SPAN >= INF_FLUSH + INF_WIDE_SPAN + 258, "far sources of a wide step are flushed");

static_assert(INF_FLUSH + 258 <= INF_RING, "a far match must find its sources flushed");
struct InfShared {
    uint8_t ring[INF_RING];
    uint8_t inw[INF_IN];
    uint8_t lens[384];
    uint32_t lt[1 << np2inf::LBITS], dt[1 << np2inf::DBITS];
    uint16_t ls[np2inf::MAXL], ds[np2inf::MAXD], sc[np2inf::MAXBITS + 2];
    np2inf::Code lc, dc;
};

__device__ __forceinline__ void inf_sync() { // LDS written by this wavefront is read by this wavefront
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// The rare paths — staging input, a flush, a far or overlapping match, a long code, the end — are real calls: inlined into
// the symbol loop they made it 36 KB of code whose every step was a string of taken branches.  They are free functions over
// plain values: a member function's `this` would pin the machine's scalars (positions, counts) to memory.
//
// the next INF_CHUNK input bytes (from `filled` on) into the window's free half, 32 bytes a lane (zeros beyond the payload)
__device__ __noinline__ void inf_stage(InfShared &S, const uint8_t *__restrict__ gin, uint32_t clen, uint32_t lane, uint32_t filled) {
    const uint32_t g = filled + lane * 32u;
    uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
    if (g + 32u <= clen) {
        __builtin_memcpy(&a, gin + g, 16);
        __builtin_memcpy(&b, gin + g + 16, 16);
    } else if (g < clen) {
        uint8_t t[32];
#pragma unroll
        for (uint32_t i = 0; i < 32; ++i) t[i] = g + i < clen ? gin[g + i] : (uint8_t)0;
        __builtin_memcpy(&a, t, 16);
        __builtin_memcpy(&b, t + 16, 16);
    }
    uint8_t *dst = S.inw + ((filled & (INF_IN - 1u)) + lane * 32u);
    *reinterpret_cast<uint4 *>(dst) = a;
    *reinterpret_cast<uint4 *>(dst + 16) = b;
    inf_sync();
}
// whole INF_FLUSH pieces of the ring to memory, INF_FLUSH / 64 bytes a lane; returns the new `flushed`
__device__ __noinline__ uint32_t inf_flush(InfShared &S, uint8_t *__restrict__ gout, uint32_t lane, uint32_t flushed, uint32_t out) {
    while (out - flushed >= INF_FLUSH) {
        inf_sync();
        constexpr uint32_t PER_LANE = INF_FLUSH / 64u;
        static_assert(PER_LANE % 16u == 0, "whole 16-byte pieces per lane");
        const uint8_t *src = S.ring + ((flushed & (INF_RING - 1u)) + lane * PER_LANE);
        uint8_t *dst = gout + flushed + lane * PER_LANE;
#pragma unroll
        for (uint32_t j = 0; j < PER_LANE / 16u; ++j) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + 16 * j);
            __builtin_memcpy(dst + 16 * j, &v, 16);
        }
        flushed += INF_FLUSH;
    }
    return flushed;
}
// A match that is not the usual one (one round inside the window, no wrap of the pattern).  Inside the LDS window: the
// sources all lie before `out` (an overlapping match repeats its first `dist` bytes), so a round's reads never meet this
// match's writes except through the ring's wrap-around, where the slot a lane reads is written by a LATER byte of the
// match: reads before writes.  A source further back (dist > INF_RING >= len) has left the window — and has been flushed
// (the flush lags by less than INF_FLUSH <= INF_RING - 258 bytes): it is read back from memory, past this CU's vector
// cache (the flush's stores went through to L2).
__device__ __noinline__ void inf_copy_general(InfShared &S, const uint8_t *gout, uint32_t lane, uint32_t out, uint32_t len, uint32_t dist, uint32_t near) {
    if (dist <= near) {
        for (uint32_t k0 = 0; k0 < len; k0 += 64) {
            const uint32_t k = k0 + lane;
            uint8_t v = 0;
            if (k < len) v = S.ring[(out - dist + (dist >= len ? k : k % dist)) & (INF_RING - 1u)];
            inf_sync();
            if (k < len) S.ring[(out + k) & (INF_RING - 1u)] = v;
        }
    } else {
        // The sources were written by THIS wavefront's flushes — plain stores through this CU's vector cache — and are read
        // back the same way: inside a workgroup the cache is coherent, so a workgroup-scope fence (wait for the stores) is all
        // it takes.  Measured on the way: an AGENT-scope release writes the L2 back (buffer_wbl2) — with eight blocks per CU
        // doing that for each of their ~280 far matches a block's copies went from 1.1 M to 9.8 M clocks —, and agent-scope
        // (cache-bypassing) LOADS behind a workgroup-scope fence raced with the stores still on their way through the cache.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t k0 = 0; k0 < len; k0 += 64) {
            const uint32_t k = k0 + lane;
            if (k < len) S.ring[(out + k) & (INF_RING - 1u)] = *reinterpret_cast<const volatile uint8_t *>(gout + (out - dist + k));
        }
    }
}
__device__ __noinline__ uint32_t inf_slow(InfShared &S, int mode, uint32_t bits) {
    return mode == np2inf::MODE_LITLEN ? np2inf::code_slow(S.lc, S.ls, bits, mode) : np2inf::code_slow(S.dc, S.ds, bits, mode);
}
// what is left in the ring at the end of the stream
__device__ __noinline__ void inf_finish(InfShared &S, uint8_t *__restrict__ gout, uint32_t lane, uint32_t flushed, uint32_t out) {
    inf_sync();
    for (uint32_t i = flushed + lane; i < out; i += 64) gout[i] = S.ring[i & (INF_RING - 1u)];
}

struct DevMachine {
    InfShared &S;
    const uint8_t *__restrict__ gin;
    uint8_t *__restrict__ gout;
    uint32_t clen, lane_;
    uint32_t filled = 0, flushed = 0; // input bytes staged / output bytes written to memory (the same in every lane)
    // phase clocks of a block (NP2_INF_PROF, a tool's switch: np2_bgzf_inflate_device prints their means): [0] the wide
    // step's decode of 64 offsets, [1] following the chain (literals included), [2] match copies, [3] tokens, [4] matches
    unsigned long long *prof = nullptr;
    uint32_t probe = 0; // (NP2_INF_PROBE, a tool's switch: bit 0 no match copies, bit 1 no literal writes, bit 2 no flushes — wrong output, timing only)
    unsigned long long pacc[7] = {0, 0, 0, 0, 0, 0, 0}; // ([5] table builds, [6] deflate blocks)
    unsigned long long t_tick = 0;
    __device__ __forceinline__ void tick(int phase) {
        if (!prof) return;
        if (phase == 0) t_tick = clock64();
        else pacc[5] += clock64() - t_tick, ++pacc[6];
    }

    __device__ __forceinline__ uint32_t uni(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    __device__ __forceinline__ bool leader() const { return lane_ == 0; }
    __device__ __forceinline__ uint32_t lane() const { return lane_; }
    __device__ __forceinline__ uint32_t lanes() const { return 64; }
    __device__ __forceinline__ void sync() const { inf_sync(); }
    __device__ __forceinline__ uint32_t in32(uint32_t off) {
        // (a chunk is staged over the half of the window that lies wholly behind `off`)
        if (off + INF_CHUNK > filled) {
            inf_stage(S, gin, clen, lane_, filled);
            filled += INF_CHUNK;
        }
        return uni(*reinterpret_cast<const uint32_t *>(S.inw + (off & (INF_IN - 1u))));
    }
    __device__ __forceinline__ void flush_upto(uint32_t out) {
        if (out - flushed >= INF_FLUSH) flushed = (probe & 4u) ? (out & ~(INF_FLUSH - 1u)) : inf_flush(S, gout, lane_, flushed, out);
    }
    // Literals of a run wait in the lanes' registers — lane (out & 63) holds output byte `out` — and go to the ring 64 at
    // a time (or when anything else needs the ring): per literal one compare-and-select instead of an LDS write under a
    // one-lane mask and a flush test.  pend_lo: first byte still waiting (pend_lo == out: none).
    uint32_t pend_lo = 0, pv = 0;
    __device__ __forceinline__ void commit(uint32_t out) {
        if (out != pend_lo) { // (the waiting bytes lie inside one 64-byte group)
            const uint32_t pos = ((out - 1u) & ~63u) + lane_;
            if (pos >= pend_lo && pos < out) S.ring[pos & (INF_RING - 1u)] = (uint8_t)pv;
            pend_lo = out;
            flush_upto(out);
        }
    }
    __device__ __forceinline__ uint32_t lit_room(uint32_t out, uint32_t isize) {
        if ((out & 63u) == 0) commit(out); // (a full group)
        const uint32_t a = isize - out, b = 64u - (out & 63u);
        return a < b ? a : b;
    }
    __device__ __forceinline__ void put_fast(uint32_t out, uint32_t byte) { pv = lane_ == (out & 63u) ? byte : pv; }
    __device__ __forceinline__ void put(uint32_t out, uint32_t byte) {
        commit(out);
        if (lane_ == 0) S.ring[out & (INF_RING - 1u)] = (uint8_t)byte;
        pend_lo = out + 1u;
        flush_upto(out + 1u);
    }
    __device__ __forceinline__ void copy(uint32_t out, uint32_t len, uint32_t dist) {
        commit(out);
        pend_lo = out + len;
        inf_sync();
        if (dist >= len && dist <= INF_RING && len <= 64u) { // the usual match: one round, no wrap of the pattern
            uint8_t v = 0;
            if (lane_ < len) v = S.ring[(out - dist + lane_) & (INF_RING - 1u)];
            inf_sync();
            if (lane_ < len) S.ring[(out + lane_) & (INF_RING - 1u)] = v;
        } else {
            inf_copy_general(S, gout, lane_, out, len, dist, INF_RING);
        }
        flush_upto(out + len);
    }
    // a match of the wide step: the chain's literals are already in the ring — also those BEHIND this match, up to
    // INF_WIDE_SPAN bytes ahead of it, which have taken the slots of the window's oldest bytes — so only a source nearer
    // than INF_RING - INF_WIDE_SPAN is read from the ring, anything further from the flushed output
    __device__ __forceinline__ void copy_at(uint32_t out, uint32_t len, uint32_t dist) {
        inf_sync();
        if (dist >= len && dist <= INF_RING - INF_WIDE_SPAN && len <= 64u) {
            uint8_t v = 0;
            if (lane_ < len) v = S.ring[(out - dist + lane_) & (INF_RING - 1u)];
            inf_sync();
            if (lane_ < len) S.ring[(out + lane_) & (INF_RING - 1u)] = v;
        } else {
            inf_copy_general(S, gout, lane_, out, len, dist, INF_RING - INF_WIDE_SPAN);
        }
    }
    __device__ __forceinline__ void finish(uint32_t out) {
        commit(out);
        inf_finish(S, gout, lane_, flushed, out);
    }
    // THE WIDE STEP.  A symbol at a time, every symbol is a chain — table read, shift, branch: ~1300 clocks, 3.5 ms a block —
    // that 63 lanes watch.  Here lane i decodes the TOKEN that would start at bit offset i of the stream's next 64 bits — a
    // literal, an end-of-block code, or a whole match (length code + extra bits + distance code + extra bits: at most 48
    // bits, and every lane has 64 of its own) — as if a token started there: two LDS gathers and some shifts for all 64
    // offsets at once.  Only one chain of those offsets is real: the one from offset 0, each token's bit count leading to
    // the next.  Following it is all that is left to do serially — a readlane and an add per token —, and it carries some
    // eight tokens per step.  A lane whose bits are no code of the primary tables stops the chain there: the symbol loop of
    // inflate_stream takes that one token (the long-code path) and comes back.
    __device__ __forceinline__ uint32_t fast(uint64_t &bb, uint32_t &bc, uint32_t &next, uint32_t &out, uint32_t isize, uint32_t in_limit) {
        uint32_t bp = next * 8u - bc; // bit offset of the first unconsumed bit
        uint32_t result = np2inf::ST_OK;
        for (;;) {
            if ((bp >> 3) > in_limit) {
                result = np2inf::ST_IN_OVERRUN;
                break;
            }
            const uint32_t byte_hi = (((bp + 63u) >> 5) << 2) + 12u; // the bytes lane 63 reads end here
            while (byte_hi > filled) {
                inf_stage(S, gin, clen, lane_, filled);
                filled += INF_CHUNK;
            }
            const unsigned long long tp0 = prof ? clock64() : 0ull;
            const uint32_t b = bp + lane_, w = (b >> 5) << 2, sh = b & 31u;
            const uint32_t d0 = *reinterpret_cast<const uint32_t *>(S.inw + (w & (INF_IN - 1u)));
            const uint32_t d1 = *reinterpret_cast<const uint32_t *>(S.inw + ((w + 4u) & (INF_IN - 1u)));
            const uint32_t d2 = *reinterpret_cast<const uint32_t *>(S.inw + ((w + 8u) & (INF_IN - 1u)));
            uint64_t x = (((uint64_t)d1 << 32) | d0) >> sh;
            if (sh) x |= (uint64_t)d2 << (64u - sh);
            const uint32_t e1 = S.lt[(uint32_t)x & ((1u << np2inf::LBITS) - 1u)];
            const uint32_t l1 = e1 & 15u, k1 = (e1 >> 8) & 3u;
            // the match reading of my bits (computed by every lane: a second gather whether or not it is used)
            const uint32_t le = (e1 >> 4) & 15u;
            uint64_t y = x >> l1;
            const uint32_t mlen = (e1 >> 16) + ((uint32_t)y & ((1u << le) - 1u));
            y >>= le;
            const uint32_t e2 = S.dt[(uint32_t)y & ((1u << np2inf::DBITS) - 1u)];
            const uint32_t l2 = e2 & 15u, de = (e2 >> 4) & 15u;
            const uint32_t mdist = (e2 >> 16) + ((uint32_t)(y >> l2) & ((1u << de) - 1u));
            uint32_t n = 0, kind = 3u, val = 0; // kind: 0 literal, 1 match, 2 end of block, 3 not decodable here (n = 0)
            if (l1) {
                if (k1 == np2inf::K_LIT) n = l1, kind = 0, val = e1 >> 16;
                else if (k1 == np2inf::K_END) n = l1, kind = 2u;
                else if (k1 == np2inf::K_LEN && l2 && ((e2 >> 8) & 3u) == 0) n = l1 + le + l2 + de, kind = 1u, val = mlen;
            }
            const uint32_t T0 = n | (kind << 8) | (val << 16);
            unsigned long long tp1 = 0;
            if (prof) {
                tp1 = (unsigned long long)__builtin_amdgcn_readfirstlane((int)(T0 + (uint32_t)mdist)) * 0ull + clock64(); // (after the gathers have landed)
                pacc[0] += tp1 - tp0;
            }
            // The chain from offset 0: which offsets start a token (a 64-bit scalar mask).  This loop is the serial rest of the
            // step, a dependent scalar instruction costs this machine ~20 clocks, so it is kept to a lane read, an OR and an
            // add per token: a lane whose token ends the chain (end of block, not decodable here) reads as 0 bits, and what
            // stopped it — and the step's output cap — are looked at afterwards.
            const uint32_t N = kind < 2u ? n : 0u;
            uint64_t members = 0;
            uint32_t cur = 0;
            for (;;) {
                const uint32_t tn = (uint32_t)__builtin_amdgcn_readlane((int)N, (int)cur);
                if (!tn) break;
                members |= 1ull << cur;
                cur += tn;
                if (cur >= 64u) break;
            }
            uint32_t stop_kind = 0; // 0: ran off the 64 offsets (or reached the step's output cap); 2: end-of-block code (consumed); 3: a token this step cannot decode
            if (cur < 64u) {
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)T0, (int)cur);
                stop_kind = (t0 >> 8) & 3u; // (2 or 3: what made the lane read as 0 bits)
                if (stop_kind == 2u) cur += t0 & 255u;
            }
            // output offsets of the chain's tokens (a literal is one byte, a match its length): one scan over the lanes
            bool mine = (members >> lane_) & 1ull;
            uint32_t olen = mine ? (kind == 0 ? 1u : val) : 0u;
            const uint32_t incl = wave_incl_scan<OpAdd>(olen);
            uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t ooff = incl - olen;
            if (total > INF_WIDE_SPAN) { // the step's output cap: the chain is cut before the first token that would exceed it
                const uint64_t over = __ballot(mine && incl > INF_WIDE_SPAN);
                const uint32_t j = (uint32_t)__builtin_ctzll(over);
                members &= (1ull << j) - 1ull;
                mine = mine && lane_ < j;
                olen = mine ? olen : 0u;
                total = (uint32_t)__builtin_amdgcn_readlane((int)ooff, (int)j);
                cur = j, stop_kind = 0;
            }
            if (out + total > isize) {
                result = np2inf::ST_OUT_OVERRUN;
                break;
            }
            if (total) {
                commit(out); // (literals of the symbol loop still waiting in registers)
                // every literal of the chain at once ...
                if (mine && kind == 0 && !(probe & 2u)) S.ring[(out + ooff) & (INF_RING - 1u)] = (uint8_t)val;
                // ... and its matches in order (a match may repeat what the tokens before it produced)
                uint64_t mm = (probe & 1u) ? 0ull : __ballot(mine && kind == 1u);
                const unsigned long long tc0 = prof ? clock64() : 0ull;
                while (mm) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(mm);
                    mm &= mm - 1ull;
                    const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)T0, (int)j) >> 16;
                    const uint32_t dist = (uint32_t)__builtin_amdgcn_readlane((int)mdist, (int)j);
                    const uint32_t at = out + (uint32_t)__builtin_amdgcn_readlane((int)ooff, (int)j);
                    if (dist > at) {
                        result = np2inf::ST_BAD_DISTANCE;
                        break;
                    }
                    copy_at(at, len, dist);
                    if (prof) ++pacc[4];
                }
                if (prof) pacc[2] += clock64() - tc0;
                if (result != np2inf::ST_OK) break;
                out += total;
                pend_lo = out;
                flush_upto(out);
            }
            if (prof) pacc[1] += clock64() - tp1, pacc[3] += (uint64_t)__builtin_popcountll(members);
            bp += cur;
            if (stop_kind == 2u) {
                result = np2inf::FAST_END_OF_BLOCK;
                break;
            }
            if (stop_kind == 3u) break;
        }
        // the symbol loop's bit buffer, from bit offset bp on
        next = (bp >> 5) << 2;
        const uint32_t sh = bp & 31u;
        bb = (uint64_t)(in32(next) >> sh);
        bc = 32u - sh;
        next += 4u;
        return result;
    }
    __device__ __forceinline__ uint32_t slow(int mode, uint32_t bits) { return inf_slow(S, mode, bits); }
    __device__ __forceinline__ uint8_t *lens() { return S.lens; }
    __device__ __forceinline__ uint32_t *lit_table() { return S.lt; }
    __device__ __forceinline__ uint32_t *dist_table() { return S.dt; }
    __device__ __forceinline__ uint16_t *lit_sym() { return S.ls; }
    __device__ __forceinline__ uint16_t *dist_sym() { return S.ds; }
    __device__ __forceinline__ uint16_t *scratch16() { return S.sc; }
    __device__ __forceinline__ np2inf::Code &lit_code() { return S.lc; }
    __device__ __forceinline__ np2inf::Code &dist_code() { return S.dc; }
};

__global__ __launch_bounds__(64) void k_bgzf_inflate(const InfBlock *__restrict__ blk, uint32_t n_blk, const uint8_t *__restrict__ comp,
                                                     uint8_t *__restrict__ out, uint32_t *__restrict__ status, uint32_t *__restrict__ n_bad,
                                                     unsigned long long *__restrict__ prof, uint32_t probe) {
    __shared__ __attribute__((aligned(16))) InfShared S;
    const uint32_t b = blockIdx.x;
    if (b >= n_blk) return;
    const InfBlock B = blk[b];
    DevMachine m{S, comp + B.in_off, out + B.out_off, B.clen, threadIdx.x};
    m.prof = prof;
    m.probe = probe;
    const unsigned long long t_begin = prof ? clock64() : 0ull;
    uint32_t st = np2inf::ST_OK;
    if (B.isize > 65536u || B.clen > 65536u) st = np2inf::ST_OUT_OVERRUN;
    else st = np2inf::inflate_stream(m, B.clen, B.isize);
    if (st == np2inf::ST_OK) m.finish(B.isize);
    if (prof && threadIdx.x == 0) {
        unsigned long long *q = prof + (size_t)b * 8;
        q[0] = clock64() - t_begin;
        for (int i = 0; i < 7; ++i) q[1 + i] = m.pacc[i];
    }
    if (threadIdx.x == 0) {
        status[b] = st;
        if (st != np2inf::ST_OK) atomicAdd(n_bad, 1u);
    }
}

// ---- the record walk ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// One thread per chain c: records from starts[c] to starts[c + 1] (the last chain: to the first record of another
// reference, or to `end`).  Counts the contig's records (refID == tid, pos < L: what fetch(tid, 0, L) returns) and their
// CIGAR words.  chain_info[c] = {records, cigar words}; flags: WALK_* bits.
// zone [zone_lo, zone_hi): the whole contig (0, L), or a shard's reference interval — then a record that starts before
// zone_lo is taken only if it reaches it (reference span from its CIGAR, like fetch_records), and the walk ends at the first
// record that starts at or beyond zone_hi.
template <bool WRITE>
__device__ __forceinline__ void bam_chain(const uint8_t *__restrict__ st, const uint64_t *__restrict__ starts, uint32_t n_chains, uint64_t end,
                                          int32_t tid, uint32_t L, uint32_t zone_lo, uint32_t zone_hi, uint32_t c, uint2 *__restrict__ chain_info,
                                          uint32_t *__restrict__ flags, unsigned long long *__restrict__ tail_at, const uint2 *__restrict__ chain_off,
                                          np2_bamrec_t *__restrict__ recs, uint64_t *__restrict__ cig_src) {
    uint64_t p = starts[c];
    const bool last = c + 1 == n_chains;
    const uint64_t stop = last ? end : starts[c + 1];
    uint32_t n_rec = 0, n_cig = 0;
    uint32_t ro = 0, co = 0;
    if (WRITE) ro = chain_off[c].x, co = chain_off[c].y;
    bool other = false;
    while (p < stop) {
        if (p + 4 > end) { // (the length field itself is cut off)
            if (!WRITE) atomicOr(flags, WALK_TAIL), atomicMin(tail_at, (unsigned long long)p);
            break;
        }
        const uint32_t bs = ld32(st + p);
        if (bs < 32) {
            if (!WRITE) atomicOr(flags, WALK_BAD);
            break;
        }
        if (p + 4 + bs > end) { // the record continues beyond the inflated range: the host inflates further and walks again
            if (!WRITE) atomicOr(flags, WALK_TAIL), atomicMin(tail_at, (unsigned long long)p);
            break;
        }
        const uint8_t *rec = st + p + 4;
        const int32_t refID = (int32_t)ld32(rec);
        if (refID != tid) {
            if (refID > tid || refID < 0) { // the contig's records are over
                other = true;
                break;
            }
            p += 4 + (uint64_t)bs;
            continue;
        }
        const int32_t pos = (int32_t)ld32(rec + 4);
        const uint32_t l_name = rec[8], nc = ld16(rec + 12), l_seq = ld32(rec + 16);
        if (pos >= 0 && (uint32_t)pos >= zone_hi) { // coordinate-sorted: nothing further starts inside the zone
            other = true;
            break;
        }
        if (pos < 0 || (uint32_t)pos >= L) { // (fetch(tid, 0, L) skips a record placed on the reference without a position)
            p += 4 + (uint64_t)bs;
            continue;
        }
        if ((uint64_t)32 + l_name + (uint64_t)nc * 4 + ((uint64_t)l_seq + 1) / 2 > bs) {
            if (!WRITE) atomicOr(flags, WALK_BAD);
            break;
        }
        if ((uint32_t)pos < zone_lo) { // the record must reach the zone
            const uint8_t *pc = rec + 32 + l_name;
            uint64_t span = 0;
            for (uint32_t k = 0; k < nc; ++k) {
                const uint32_t w = ld32(pc + 4ull * k), op = w & 15u;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += w >> 4;
            }
            if ((uint64_t)pos + span <= zone_lo) {
                p += 4 + (uint64_t)bs;
                continue;
            }
        }
        if (WRITE) {
            np2_bamrec_t r;
            r.pos = pos;
            r.flag = (uint16_t)ld16(rec + 14);
            r.mapq = rec[9];
            r.pad = (uint8_t)l_name; // (l_read_name: the record starts 36 + pad bytes before its CIGAR words)
            r.n_cigar = nc;
            r.cigar_off = co;
            r.l_seq = l_seq;
            r.seq_off = p + 4 + 32 + l_name + (uint64_t)nc * 4;
            recs[ro] = r;
            cig_src[ro] = p + 4 + 32 + l_name;
            ++ro, co += nc;
        }
        ++n_rec, n_cig += nc;
        p += 4 + (uint64_t)bs;
    }
    if (!WRITE) {
        chain_info[c] = make_uint2(n_rec, n_cig);
        // a chain must end exactly where the next one begins (the index names record starts), unless the contig ended in it
        if (!last && !other && p != stop && !(*flags & (WALK_TAIL | WALK_BAD))) atomicOr(flags, WALK_MISALIGNED);
        if (last && !other && p >= end) atomicOr(flags, WALK_AT_END); // ran into the end of the inflated range on a record boundary
        if (other && !last) atomicOr(flags, WALK_EARLY_END);          // (later chains then hold no record of the contig: fine)
    }
}
__global__ void k_bam_chain_count(const uint8_t *__restrict__ st, const uint64_t *__restrict__ starts, uint32_t n_chains, uint64_t end, int32_t tid,
                                  uint32_t L, uint32_t zone_lo, uint32_t zone_hi, uint2 *__restrict__ chain_info, uint32_t *__restrict__ flags,
                                  unsigned long long *__restrict__ tail_at) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_chains) bam_chain<false>(st, starts, n_chains, end, tid, L, zone_lo, zone_hi, c, chain_info, flags, tail_at, nullptr, nullptr, nullptr);
}
__global__ void k_bam_chain_write(const uint8_t *__restrict__ st, const uint64_t *__restrict__ starts, uint32_t n_chains, uint64_t end, int32_t tid,
                                  uint32_t L, uint32_t zone_lo, uint32_t zone_hi, const uint2 *__restrict__ chain_off, np2_bamrec_t *__restrict__ recs,
                                  uint64_t *__restrict__ cig_src) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_chains) bam_chain<true>(st, starts, n_chains, end, tid, L, zone_lo, zone_hi, c, nullptr, nullptr, nullptr, chain_off, recs, cig_src);
}
// CIGAR words of the records into one array (cigar_off), a wavefront per record
__global__ void k_bam_cigars(const uint8_t *__restrict__ st, const np2_bamrec_t *__restrict__ recs, const uint64_t *__restrict__ cig_src, uint32_t n_recs,
                             uint32_t *__restrict__ cigar) {
    const uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (r >= n_recs) return;
    const uint32_t n = recs[r].n_cigar;
    const uint64_t dst = recs[r].cigar_off;
    const uint8_t *src = st + cig_src[r];
    for (uint32_t k = lane; k < n; k += 64) cigar[dst + k] = ld32(src + 4ull * k);
}

void launch_bgzf_inflate(hipStream_t s, const InfBlock *blk, uint32_t n_blk, const uint8_t *comp, uint8_t *out, uint32_t *status, uint32_t *n_bad,
                         unsigned long long *prof, uint32_t probe) {
    if (n_blk) hipLaunchKernelGGL(k_bgzf_inflate, dim3(n_blk), dim3(64), 0, s, blk, n_blk, comp, out, status, n_bad, prof, probe);
}
void launch_bam_chain_count(hipStream_t s, const uint8_t *stream, const uint64_t *starts, uint32_t n_chains, uint64_t end, int32_t tid, uint32_t L,
                            uint32_t zone_lo, uint32_t zone_hi, uint2 *chain_info, uint32_t *flags, unsigned long long *tail_at) {
    if (n_chains) hipLaunchKernelGGL(k_bam_chain_count, dim3((n_chains + 63) / 64), dim3(64), 0, s, stream, starts, n_chains, end, tid, L, zone_lo, zone_hi, chain_info, flags, tail_at);
}
void launch_bam_chain_write(hipStream_t s, const uint8_t *stream, const uint64_t *starts, uint32_t n_chains, uint64_t end, int32_t tid, uint32_t L,
                            uint32_t zone_lo, uint32_t zone_hi, const uint2 *chain_off, np2_bamrec_t *recs, uint64_t *cig_src) {
    if (n_chains) hipLaunchKernelGGL(k_bam_chain_write, dim3((n_chains + 63) / 64), dim3(64), 0, s, stream, starts, n_chains, end, tid, L, zone_lo, zone_hi, chain_off, recs, cig_src);
}
void launch_bam_cigars(hipStream_t s, const uint8_t *stream, const np2_bamrec_t *recs, const uint64_t *cig_src, uint32_t n_recs, uint32_t *cigar) {
    if (n_recs) hipLaunchKernelGGL(k_bam_cigars, dim3((uint32_t)(((uint64_t)n_recs * 64 + 255) / 256)), dim3(256), 0, s, stream, recs, cig_src, n_recs, cigar);
}

} // namespace np2
