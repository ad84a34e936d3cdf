"""Hand-derived known answers, second set: the stages that were pinned only by "polishes back to the truth" — the
low-quality (LQ) region state machine's close / pad / extend rules (main.rs:1586-1625), the decode limit of candidate
extraction and its start filter (main.rs:1460-1470, 1482-1497), is_valid_snp's homopolymer compression (main.rs:780-801),
the `dif <= -3` override of the read-pair weights (main.rs:996-1002) and the first-range quirk of the clip filter
(main.rs:531-574).  Every expectation is derived from the Rust text in the docstring of its test; tests/test_gpu_pinning.py
asks the same of the HIP path."""
import numpy as np

from nextpolish2_amd import Opts
from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import records_to_arrays
from nextpolish2_amd.synth import pileup_from_alignments
from oracle import np2_oracle as orc
from test_oracle_pinning import backbone, other, put, yak_counted


def _run(ref, alns, yaks, opts):
    o = orc.Oracle(yaks)
    o.set_trace(True)
    b, p = o.polish(pileup_from_alignments(ref, alns), opts)
    return o, b, p


def _cands(o, ps=0):
    co, so, seq = o.trace(ps, "cand.cand_off"), o.trace(ps, "cand.seq_off"), o.trace(ps, "cand.seq")
    return [[seq[so[i]:so[i + 1]].tobytes().decode() for i in range(co[g], co[g + 1])] for g in range(len(co) - 1)]


# ---- (i) the LQ state machine: pad, extension over a homopolymer, a close delayed by a homopolymer --------------------------
def test_lq_region_pad_extend_and_delayed_close():
    """generate_cns_from_best_score_lq, main.rs:1586-1625.  21 rows (the contig + 20 reads) on a sequence without equal
    neighbours; 2 reads carry another base at X = 100.  A node spans three columns, so the variant reads differ from the
    rest in the nodes of X, X + 1 and X + 2: there the best node has 19 of 21 rows, qv = 19 * 100 / 21 = 90 < 95: three LQ
    bases; everywhere else all 21 rows agree (one deviating row would still be 20 * 100 / 21 = 95: not below 95).
    The backtrack runs right to left; p counts emitted bases.  LQ bases at X + 2 (p = a), X + 1, X (p = a + 2): lq_s = a,
    lq_e = a + 2.  The first HQ base with p - lq_e > 4 is p = a + 7 (position X - 5); the two bases before it (X - 4,
    X - 3) differ in position and letter, so the region closes: lq_e = p - 2 (position X - 3), lq_s = a - 2 (position
    X + 4), no neighbour of equal letter to extend over:                                    region [97, 104].
    * positions 104, 105, 106 carrying one letter: `while ... base == base` walks lq_s from 104 over 105 to 106 -> [97, 106];
    * positions 96, 97 carrying one letter: at p = a + 7 the bases p - 1, p - 2 (96, 97) have the same letter: no close;
      at p = a + 8 (position 94) the bases 95, 96 differ: close with lq_e = p - 2 = position 96              -> [96, 104]."""
    X = 100
    base = backbone(220, 31)
    hp_right = put(put(base, 105, base[104]), 106, base[104])
    hp_left = put(base, 96, base[97])
    assert hp_right[107] != hp_right[106] and hp_right[103] != hp_right[104] and hp_left[95] != hp_left[96]
    for ref, want in ((base, (97, 104)), (hp_right, (97, 106)), (hp_left, (96, 104))):
        alt = other(ref[X], skip=(ref[X - 1], ref[X + 1]))
        alns = [(0, ref, ref)] * 18 + [(0, ref, put(ref, X, alt))] * 2
        o, b, _ = _run(ref, alns, [yak_counted([(ref, 50)], 21)], Opts(iter_count=1))
        assert (o.trace(0, "lq.start").tolist(), o.trace(0, "lq.end").tolist()) == ([want[0]], [want[1]])
        assert _cands(o)[0] == [ref[want[0]:want[1] + 1]] * 19 + [put(ref, X, alt)[want[0]:want[1] + 1]] * 2
        assert b.tobytes().decode() == ref


# ---- (ii) candidate extraction: the decode limit and the start filter -------------------------------------------------------
def test_decode_limit_invalid_kmer_and_columns_before_the_start():
    """generate_lqseqs_from_tags_kmer, main.rs:1460-1470: a read is decoded until the first column whose t_pos exceeds
    end + k of its rightmost region (that column included); 1482-1497: the first k-mer is made of the first k non-gap
    columns from the region's start among the decoded ones, INVALID_KMER if there are fewer; retrieve_kmer_count
    (main.rs:740-778): a string of at most k letters is scored by that k-mer, 0 if it is invalid.
    Region [97, 104] as in (i), k = 21: decoded columns reach t_pos 126.  A read that deletes the n bases 105 .. 104 + n
    has 8 + (126 - 104 - n) = 30 - n letters from 97 on: n = 9 -> 21: its k-mer is contig[97..105) + contig[114..127), the
    junction the deletion makes (in the table with count 33: k-score 33); n = 10 -> 20 < k: INVALID_KMER, k-score 0.
    1478: the candidate's columns start at column INDEX start - aln_t_s and are then filtered by t_pos >= start: a read
    with two inserted bases behind position 60 has its columns shifted by two, the filter drops the two columns of
    positions 95, 96 the index lands on: its candidate and k-mer are the contig's."""
    k, X = 21, 100
    ref = backbone(220, 31)
    alt = other(ref[X], skip=(ref[X - 1], ref[X + 1]))
    ins = other(ref[60], skip=(ref[61],)) + other(ref[61], skip=(ref[60], ref[62]))
    ins_read = (0, ref[:61] + "--" + ref[61:], ref[:61] + ins + ref[61:])
    for n_del, score in ((9, 33), (10, 0)):
        junction = ref[97:105] + ref[105 + n_del:105 + n_del + 13]
        del_read = (0, ref, ref[:105] + "-" * n_del + ref[105 + n_del:])
        alns = [(0, ref, ref)] * 16 + [(0, ref, put(ref, X, alt))] * 2 + [del_read, ins_read]
        o, _, _ = _run(ref, alns, [yak_counted([(ref, 50), (junction, 33)], k)], Opts(iter_count=1))
        assert (o.trace(0, "lq.start").tolist(), o.trace(0, "lq.end").tolist()) == ([97], [104])
        assert o.trace(0, "cand.order").tolist() == list(range(21))
        assert o.trace(0, "cand.kscore").tolist() == [50] * 17 + [0, 0, score, 50]
        km = o.trace(0, "cand.kmer")
        assert (int(km[19]) == 0xFFFFFFFFFFFFFFFF) == (n_del == 10) and km[20] == km[0]
        c = _cands(o)[0]
        assert c[19] == ref[97:105] and c[20] == ref[97:105]


# ---- (iii) is_valid_snp: two alleles that differ in the length of a homopolymer run are no marker ------------------------------
def test_homopolymer_length_difference_is_not_a_heterozygous_marker():
    """mark_hete_lqseqs / is_valid_snp, main.rs:916-946, 780-801.  10 reads of haplotype 1 (= the contig), 10 of haplotype 2,
    which has another base at 160 and a fourth A in the run AAA at 100..102.  Both regions hold 21 candidates, min_c = 3,
    the second group has 10 >= 3 members and >= max1_c / 2.  At 160 the strings differ at a letter: is_valid_snp is true,
    the region is HETE (0x40).  At the run the strings are AAACACT / AAAACACT (different lengths, but >= 6 candidates and
    max2_c >= max1_c / 2 lets them through to is_valid_snp): both cursors skip to the end of their A run after the first
    letter, the rest is equal, a string runs out: false -> not HETE.  Only the marker at 160 votes: the 10 reads that
    disagree with the contig there are removed (main.rs:977, no -r), the second pass polishes from haplotype 1 alone."""
    t = backbone(260, 33)
    for q in (100, 101, 102):
        t = put(t, q, "A")
    assert t[99] != "A" and t[103] != "A"
    alt = other(t[160], skip=(t[159], t[161]))
    hap2 = put(t, 160, alt)
    hap2 = hap2[:103] + "A" + hap2[103:]
    alns = [(0, t, t)] * 10 + [(0, t[:103] + "-" + t[103:], hap2)] * 10
    o, b, _ = _run(t, alns, [yak_counted([(t, 50), (hap2, 50)], 21)], Opts(iter_count=2))
    st = o.trace(0, "lq.start").tolist()
    assert len(st) == 2 and st[0] <= 160 <= o.trace(0, "lq.end")[0] and st[1] == 100  # (regions are listed right to left)
    assert sorted(set(_cands(o)[1])) == sorted({t[100:107], hap2[100:108]}) == ["AAAACACT", "AAACACT"]
    assert o.trace(0, "hete.lable").tolist() == [0x40, 0x00]
    assert o.trace(0, "invalid_ids").tolist() == list(range(11, 21))
    assert b.tobytes().decode() == t


# ---- (iv) read pairs that disagree at three markers are enemies whatever else they share -------------------------------------
def test_three_disagreements_override_the_summed_pair_weight():
    """phase_reads_by_lqseqs, main.rs:982-1002, with -r (reads that disagree with the contig stay in the graph).  Eight
    markers 25 apart; reads 1-8 carry the contig's alleles, reads 9-16 the others, read 17 the other allele at the first
    n markers and the contig's at the rest.  Pair weights: +1 per marker with equal strings, -1 otherwise, summed — but a
    pair with three or more -1's is ASSIGNED -(number of -1's) (dif <= -3, main.rs:996-1002).
    n = 3: read 17 against a read of the first group: 5 - 3 = +2 by the sum, but three disagreements -> -3; against the
      second group 3 - 5 -> -5.  Every edge of read 17 is negative: it stays a community of its own, in conflict with both
      groups.  Ranking against the contig's row (louvain.rs:294-316: +8 per read of the first group, -8 of the second,
      5 - 3 = +2 for read 17): first group (count 8) before {17} (count 1) before the second group (count -8); the first
      group invalidates every later community it has an edge to: reads 9-17 are removed.
    n = 2: 6 - 2 = +4 against the first group, two disagreements only: read 17 joins the first group: reads 9-16 go."""
    t = backbone(420, 35)
    sites = [60 + 25 * i for i in range(8)]
    alts = {q: other(t[q], skip=(t[q - 1], t[q + 1])) for q in sites}

    def hap(mask):
        s = t
        for q, on in zip(sites, mask):
            if on:
                s = put(s, q, alts[q])
        return s
    h_a, h_b = hap([0] * 8), hap([1] * 8)
    for n, removed in ((3, list(range(9, 18))), (2, list(range(9, 17)))):
        alns = [(0, t, h_a)] * 8 + [(0, t, h_b)] * 8 + [(0, t, hap([1] * n + [0] * (8 - n)))]
        o, _, _ = _run(t, alns, [yak_counted([(h_a, 50), (h_b, 50)], 21)], Opts(iter_count=2, use_all_reads=True))
        assert o.trace(0, "hete.lable").tolist() == [0x40] * 8 and int(o.trace(0, "hete.kscore").min()) == 50
        assert o.trace(0, "invalid_ids").tolist() == removed


# ---- (v) the clip filter's first range --------------------------------------------------------------------------------------
def clip_case():
    """Records for test_clip_filter_first_range_is_the_whole_contig (also fed to the GPU front end by test_gpu_pinning)."""
    rng = np.random.default_rng(77)
    ref = "".join("ACGT"[c] for c in rng.integers(0, 4, 500_100))
    L = len(ref)

    def rec(pos, n, clip_front=0, clip_back=0):
        cigar = ([("S", clip_front)] if clip_front else []) + [("M", n)] + ([("S", clip_back)] if clip_back else [])
        return dict(tid=0, pos=pos, mapq=60, flag=0, cigar=cigar, seq="A" * clip_front + ref[pos:pos + n] + "C" * clip_back)
    recs = [rec(10, 1600, clip_front=150),       # clipped, begins at 10 < 50: inside no range -> kept
            rec(400, 1600),                      # not clipped
            rec(1000, 1600, clip_back=150),      # clipped, [1000, 2599] inside (50, L - 51) -> emptied, index kept
            rec(3000, 1600, clip_front=150),     # the same
            rec(L - 1620, 1600, clip_back=150)]  # clipped, ends at L - 21 > L - 51 -> kept
    return ref, recs


def test_clip_filter_first_range_is_the_whole_contig():
    """main.rs:1796-1812: a read is clipped when aligned query length + max_clip_len(100) < read length (1600 + 100 <
    1750 here); on a contig of at least 500 000 bases it is pushed with a label instead of being dropped.
    filter_alignseqs_by_clip, main.rs:531-574: the ranges are built from the unlabelled entries (aln_t_s + 50,
    aln_t_e - 50) in order — and entry 0, the contig aligned to itself, gives (50, L - 51), which swallows every later
    one.  So a labelled read is emptied (align_bases = [], index retained: flag bit 0 at this boundary) iff
    50 <= aln_t_s and aln_t_e <= L - 51 — every clipped read that does not touch the first or last 50 bases."""
    ref, recs = clip_case()
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    pu = orc.front_end(ref.encode(), arr, cig, asc, asc_off, np2io.FrontOpts())
    L = len(ref)
    assert pu.reads["aln_t_s"].tolist() == [0, 10, 400, 1000, 3000, L - 1620]
    assert pu.reads["aln_t_e"].tolist() == [L - 1, 1609, 1999, 2599, 4599, L - 21]
    assert (pu.reads["flags"] & 1).tolist() == [0, 0, 0, 1, 1, 0]
