// GPU read extraction (np2_inflate.hip): BGZF inflate, one wavefront per block, and the record walk over the inflated BAM
// stream along the .bai linear index.  Host side: np2_io.cpp (GpuFetch).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
#include "../../include/np2_io.h"

namespace np2 {

struct InfBlock {     // one BGZF block
    uint64_t in_off;  // its raw DEFLATE payload in the staged file bytes
    uint64_t out_off; // where its inflated bytes go (the blocks of a range are contiguous: a BAM record may span blocks)
    uint32_t clen;    // payload bytes
    uint32_t isize;   // inflated bytes (the block's ISIZE field)
};

// flags of a record walk (k_bam_chain_count)
static constexpr uint32_t WALK_BAD = 1u;        // a record shorter than its fixed part, or whose fields overrun it
static constexpr uint32_t WALK_TAIL = 2u;       // a record continues beyond the inflated range (tail_at: where it starts)
static constexpr uint32_t WALK_MISALIGNED = 4u; // a chain did not end where the next one begins: the index is not trusted
static constexpr uint32_t WALK_AT_END = 8u;     // the last chain ran to the end of the range without meeting another reference
static constexpr uint32_t WALK_EARLY_END = 16u; // the contig's records ended before the last chain (informative)

// status[b] = np2inf::Status of block b; *n_bad += blocks that failed (both zeroed by the caller)
// prof (optional): 8 words per block — clocks of the block, of the wide step's decode, of its chain, of its match copies, tokens, matches
void launch_bgzf_inflate(hipStream_t s, const InfBlock *blk, uint32_t n_blk, const uint8_t *comp, uint8_t *out, uint32_t *status, uint32_t *n_bad,
                         unsigned long long *prof = nullptr, uint32_t probe = 0);
// starts[0 .. n_chains): stream offsets of record starts, ascending; the last chain ends at a record of another reference or at `end`
// zone [zone_lo, zone_hi): (0, L) for the whole contig, a shard's reference interval otherwise; cig_src[i] = stream offset of
// record i's CIGAR words (the record starts 36 + l_read_name bytes before: its BGZF virtual offset follows from that)
void launch_bam_chain_count(hipStream_t s, const uint8_t *stream, const uint64_t *starts, uint32_t n_chains, uint64_t end, int32_t tid, uint32_t L,
                            uint32_t zone_lo, uint32_t zone_hi, uint2 *chain_info, uint32_t *flags, unsigned long long *tail_at);
void launch_bam_chain_write(hipStream_t s, const uint8_t *stream, const uint64_t *starts, uint32_t n_chains, uint64_t end, int32_t tid, uint32_t L,
                            uint32_t zone_lo, uint32_t zone_hi, const uint2 *chain_off, np2_bamrec_t *recs, uint64_t *cig_src);
void launch_bam_cigars(hipStream_t s, const uint8_t *stream, const np2_bamrec_t *recs, const uint64_t *cig_src, uint32_t n_recs, uint32_t *cigar);

} // namespace np2
