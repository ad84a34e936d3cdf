"""Hand-derived known answers, third set: the tie-breaks of the consensus DP (get_cns_from_align_tags, main.rs:1645-1687).
A node's best predecessor is replaced by a LATER one of EQUAL score unless that one starts with a gap
(`score > kmer_score || (score == kmer_score && base1.q_base != 4)`, main.rs:1670), nodes being visited in the order the
reads pushed them (the contig is read 0, main.rs:1732-1739) after a stable sort by delta (Msa::sort, main.rs:227-229);
at the last position the LAST node of maximal score wins (`kmer_score >= global_best_kmer.score`, main.rs:1680).
tests/test_gpu_pinning.py asks the same of the HIP path."""
import numpy as np

from nextpolish2_amd import Opts
from nextpolish2_amd.synth import pileup_from_alignments
from oracle import np2_oracle as orc
from test_oracle_pinning import backbone, other, put, yak_counted


def _raw(ref, alns):
    o = orc.Oracle([yak_counted([(ref, 50)], 21)])
    o.set_trace(True)
    o.polish(pileup_from_alignments(ref, alns), Opts(iter_count=1))
    return o.trace(0, "cns_raw.base").tobytes().decode(), o.trace(0, "cns_raw.pos").tolist()


def test_equal_paths_the_later_node_wins_unless_it_starts_with_a_gap():
    """Ten rows: the contig and 4 reads carry X at position P, 5 reads carry Y.  Coverage is 10 everywhere, so a node
    shared by all rows adds 10 * 10 - 4 * 10 = 60 and each of the three nodes that hold column P adds 10 * 5 - 40 = 10 on
    either path: the X path and the Y path reach the node of P + 3 — columns (P + 1, P + 2, P + 3), common again — with
    the same score.  Its predecessors are the two nodes of position P + 2 whose last two columns are (P + 1, P + 2):
    (X, P + 1, P + 2), pushed first by the contig, and (Y, P + 1, P + 2).  The first sets kmer_score; the second equals
    it and its first column is Y, not a gap: besti moves to it.  The backtrack therefore walks the Y nodes:
                                                                raw consensus = the contig with Y at P.
    With the 5 reads DELETING column P instead (a gap column, q_base = 4) the second predecessor is (gap, P + 1, P + 2):
    equal score, but base1.q_base == 4 — besti stays with the contig's node:     raw consensus = the contig, unchanged."""
    P = 120
    ref = backbone(260, 47)
    alt = other(ref[P], skip=(ref[P - 1], ref[P + 1]))
    snp = [(0, ref, ref)] * 4 + [(0, ref, put(ref, P, alt))] * 5
    base, pos = _raw(ref, snp)
    assert base == put(ref, P, alt) and pos == list(range(len(ref)))
    dele = [(0, ref, ref)] * 4 + [(0, ref, put(ref, P, "-"))] * 5
    base, pos = _raw(ref, dele)
    assert base == ref and pos == list(range(len(ref)))
    # (one read more on the contig's side: the X path is ahead by 3 * 10 and no tie arises, whatever the order)
    base, _ = _raw(ref, [(0, ref, put(ref, P, alt))] * 5 + [(0, ref, ref)] * 5)
    assert base == ref


def test_at_the_contig_end_the_last_node_of_maximal_score_wins():
    """The same 5 : 5 split at the LAST position: its two nodes, (L - 3, L - 2, X) pushed by the contig and (L - 3, L - 2, Y),
    have equal scores; `kmer_score >= global_best_kmer.score` lets the later one replace the earlier:
                                                                raw consensus = the contig with Y as its last base."""
    ref = backbone(200, 53)
    L = len(ref)
    alt = other(ref[L - 1], skip=(ref[L - 2],))
    base, pos = _raw(ref, [(0, ref, ref)] * 4 + [(0, ref, put(ref, L - 1, alt))] * 5)
    assert base == put(ref, L - 1, alt) and pos == list(range(L))


def test_an_insertion_carried_by_half_of_the_rows_is_taken():
    """Msa::coverage (main.rs:232-241: the counts of the nodes whose last column is NOT an insertion column, which sort()
    put first) and the score 10 * count - 4 * coverage (main.rs:1658-1669).  Ten rows; k reads carry one extra base after
    position P.  Their columns P - 1, P, ins, P + 1, P + 2 make the nodes (P - 1, P, ins) — filed under position P with
    delta 1, so it does not count towards coverage(P) = 10 —, (P, ins, P + 1) and (ins, P + 1, P + 2); the other rows go
    through (P - 1, P, P + 1) and (P, P + 1, P + 2), and both paths meet again in (P + 1, P + 2, P + 3).  Between the
    common nodes the insertion path collects 3 * (10 k - 40), the plain path 2 * (10 (10 - k) - 40):
        k = 4:   0 against  40 -> the contig as it is;
        k = 5:  30 against  20 -> the inserted base is emitted (at position P once more) although only half of the rows
                 carry it — three nodes against two.  (Had coverage(P) counted the insertion node as well, its score
                 would be 10 k - 4 (10 + k) and k = 5 would give 10 against 20: this case tells the two readings apart.)"""
    P = 90
    ref = backbone(220, 59)
    ins = other(ref[P], skip=(ref[P + 1],))
    t_ins = ref[:P + 1] + "-" + ref[P + 1:]
    q_ins = ref[:P + 1] + ins + ref[P + 1:]
    for k, want in ((4, ref), (5, q_ins)):
        base, pos = _raw(ref, [(0, ref, ref)] * (9 - k) + [(0, t_ins, q_ins)] * k)
        assert base == want
        assert pos == (list(range(len(ref))) if k == 4 else list(range(P + 1)) + [P] + list(range(P + 1, len(ref))))


def test_reads_that_start_at_position_one_move_the_start_of_the_consensus_position_two_does_not():
    """main.rs:1664-1668: a predecessor whose first column is a read's head sentinel is skipped once the node's middle
    column lies at t_pos >= 3 ("this can prevent the later backtracking algorithm from stopping at the start mapping
    position of reads"), and the backtrack ends at the first node whose middle column is a head sentinel (main.rs:1629).
    Nine reads begin at position S, the contig (one row) at 0.
    S = 1: the reads' nodes (h, h, 1) and (h, 1, 2) score 10 * 9 - 4 * 10 = 50 and 100; the contig's (h, h, 0), (h, 0, 1),
           (0, 1, 2) score 10 - 4 = 6, 6 + 10 - 40 = -24 and -54.  The node (1, 2, 3) looks at position 2 for predecessors
           ending in (1, 2): the contig's (0, 1, 2) at -54 and the reads' (h, 1, 2) at 100 — its middle column is at
           t_pos 2 < 3, the head predecessor is allowed and wins.  The backtrack stops in (h, h, 1):
                                                     the raw consensus begins at position 1, the contig's first base is lost.
    S = 2: (2, 3, 4) looks at position 3: the reads' (h, 2, 3) is skipped (middle column at t_pos 3), only the contig's
           (1, 2, 3) is left, whatever its score:    the raw consensus is the whole contig."""
    ref = backbone(200, 61)
    for S, first in ((1, 1), (2, 0)):
        base, pos = _raw(ref, [(S, ref[S:], ref[S:])] * 9)
        assert base == ref[first:] and pos == list(range(first, len(ref)))


# ---- read admission and trim(8), main.rs:1758-1801, 447-513 ------------------------------------------------------------------
def admission_case():
    """(contig, records, expected (aln_t_s, aln_t_e) of the admitted reads in order) — see the test below for the derivation."""
    rng = np.random.default_rng(79)
    ref = "".join("ACGT"[c] for c in rng.integers(0, 4, 500_100))
    comp = {"A": "C", "C": "G", "G": "T", "T": "A"}

    def rec(pos, n, mapq=60, flag=0, clip_back=0, mism=()):
        seq = list(ref[pos:pos + n])
        for i in mism:
            seq[i] = comp[seq[i]]
        cigar = [("M", n)] + ([("S", clip_back)] if clip_back else [])
        return dict(tid=0, pos=pos, mapq=mapq, flag=flag, cigar=cigar, seq="".join(seq) + "A" * clip_back)
    recs = [rec(100, 1500, mapq=1),                    # mapq <= min_map_qual (1)                       -> dropped
            rec(200, 1500, mapq=2),                    # mapq 2                                          -> kept
            rec(300, 1000),                            # rlen <= min_read_len (1000)                     -> dropped
            rec(400, 1001),                            # rlen 1001                                       -> kept
            rec(500, 1500, flag=0x4),                  # unmapped                                        -> dropped
            rec(600, 1500, flag=0x400),                # duplicate                                       -> dropped
            rec(700, 1500, flag=0x800),                # supplementary, use_supplementary off            -> dropped
            rec(800, 1500, flag=0x100),                # secondary, use_secondary off                    -> dropped
            rec(900, 1200, clip_back=1202),            # span 1200 < (2402 * 0.5) as i64 = 1201          -> dropped
            rec(1000, 1200, clip_back=1201),           # span 1200 >= (2401 * 0.5) as i64 = 1200         -> kept (clipped: labelled)
            rec(3000, 1500, mism=(3, 1490)),           # trim: first / last run of 8 matches             -> [3004, 4499]
            rec(5000, 1500, mism=(7, 1492)),           #                                                 -> [5008, 6491]
            rec(7000, 1500, mism=(8, 1491))]           # columns 0-7 and 1492-1499 match: nothing is cut  -> [7000, 8499]
    want = [(0, len(ref) - 1), (200, 1699), (400, 1400), (1000, 2199), (3004, 4499), (5008, 6491), (7000, 8499)]
    return ref, recs, want


def test_read_admission_boundaries_and_trim():
    """main.rs:1758-1771, defaults of option.rs (min_map_qual 1, min_read_len 1000, -a 500.5): a record is skipped when
    flags & 0x404, mapq <= 1, rlen <= 1000 (rlen = query length by CIGAR, clips included), secondary / supplementary
    without their switches, or reference span < max(500, (rlen as f32 * 0.5) as i64) — strict: 1200 against 2401 * 0.5
    = 1200.5 -> 1200 passes, against 2402 * 0.5 = 1201 does not.  trim(8), main.rs:447-513: the alignment begins with
    its first and ends with its last run of 8 matching columns — a mismatch in column 3 moves the start to column 4, in
    column 7 to column 8, in column 8 not at all (columns 0-7 match); from the back (n = 1500 columns), a mismatch in column
    n - 8 leaves only 7 matching columns behind it: the read ends with the 8 matches before it, in column n - 9; a
    mismatch in column n - 9 or n - 10 has 8 or 9 matches behind it and cuts nothing.  The read of 1200 + 1201 clipped bases is
    clipped (1200 + 100 < 2401): on a contig of 500 100 bases it is kept with a label, and emptied by the clip filter
    (inside (50, L - 51): tests/test_oracle_pinning2.py)."""
    from nextpolish2_amd import io as np2io
    from nextpolish2_amd.bamio import records_to_arrays
    ref, recs, want = admission_case()
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    pu = orc.front_end(ref.encode(), arr, cig, asc, asc_off, np2io.FrontOpts())
    assert list(zip(pu.reads["aln_t_s"].tolist(), pu.reads["aln_t_e"].tolist())) == want
    assert (pu.reads["flags"] & 1).tolist() == [0, 0, 0, 1, 0, 0, 0]


# ---- the 60-candidate cap, main.rs:1474 ------------------------------------------------------------------------------------
def test_a_region_keeps_its_first_sixty_candidates_in_read_order():
    """generate_lqseqs_from_tags_kmer, main.rs:1440-1474: the reads are visited in alignseqs order (the contig is entry 0)
    and a region that already holds LQSEQ_MAX_CAN_COUNT = 60 strings takes no more.  71 rows on the sequence of
    tests/test_oracle_pinning2.py's first case; 4 reads carry another base at X = 100: the best nodes of X, X + 1, X + 2
    have 67 of 71 rows, qv = 67 * 100 / 71 = 94 < 95 -> the region [97, 104] derived there.
    The 4 variant reads first:  the contig's string, the 4 variants, 55 more of the contig's = 60 strings (11 reads unseen);
    the 4 variant reads last:   60 times the contig's string — the variants, reads 67-70, never enter the table."""
    from test_oracle_pinning2 import _cands
    X = 100
    ref = backbone(220, 31)
    alt = other(ref[X], skip=(ref[X - 1], ref[X + 1]))
    var = put(ref, X, alt)
    for alns, want in (([(0, ref, var)] * 4 + [(0, ref, ref)] * 66, [ref[97:105]] + [var[97:105]] * 4 + [ref[97:105]] * 55),
                       ([(0, ref, ref)] * 66 + [(0, ref, var)] * 4, [ref[97:105]] * 60)):
        o = orc.Oracle([yak_counted([(ref, 50)], 21)])
        o.set_trace(True)
        o.polish(pileup_from_alignments(ref, alns), Opts(iter_count=1))
        assert (o.trace(0, "lq.start").tolist(), o.trace(0, "lq.end").tolist()) == ([97], [104])
        assert _cands(o)[0] == want


# ---- fill_with_cigar and is_clip, main.rs:386-440, 1796-1797 ---------------------------------------------------------------
def cigar_case():
    """(contig, records, expected [(aln_t_s, t_aln, q_aln) or None for a read the clip filter empties]) — derivation below."""
    rng = np.random.default_rng(83)
    ref = "".join("ACGT"[c] for c in rng.integers(0, 4, 500_100))
    comp = {"A": "C", "C": "G", "G": "T", "T": "A"}
    i1, i2 = comp[ref[2599]], comp[ref[2600]]  # two inserted bases (neither equal to its neighbours: no ambiguity to argue about)
    seq_a = ref[2000:2600] + i1 + i2 + ref[2600:3100] + ref[3103:3503]
    aln_a = (2000, ref[2000:2600] + "--" + ref[2600:3503], ref[2000:2600] + i1 + i2 + ref[2600:3100] + "---" + ref[3103:3503])
    x = comp[ref[6800]]
    seq_f = ref[6000:6800] + x + ref[6801:7601]
    recs = [dict(tid=0, pos=2000, mapq=60, flag=0, cigar=[("M", 600), ("I", 2), ("M", 500), ("D", 3), ("M", 400)], seq=seq_a),
            dict(tid=0, pos=5000, mapq=60, flag=0, cigar=[("H", 10), ("S", 150), ("M", 1600)], seq="A" * 150 + ref[5000:6600]),
            dict(tid=0, pos=6000, mapq=60, flag=0, cigar=[("=", 800), ("X", 1), ("=", 800)], seq=seq_f),
            dict(tid=0, pos=8000, mapq=60, flag=0, cigar=[("S", 150), ("M", 1600)], seq="A" * 150 + ref[8000:9600]),
            dict(tid=0, pos=10000, mapq=60, flag=0, cigar=[("M", 1600), ("S", 150), ("H", 10)], seq=ref[10000:11600] + "C" * 150)]
    want = [aln_a, (5000, ref[5000:6600], ref[5000:6600]), (6000, ref[6000:7601], seq_f), None, None]
    return ref, recs, want


def _read_bytes(pu, i):
    r = pu.reads[i]
    return (int(r["aln_t_s"]), int(r["aln_t_e"]), int(r["n_cols"]), pu.nibbles[int(r["nib_off"]):int(r["nib_off"]) + (int(r["n_cols"]) + 2) // 2].tobytes())


def check_cigar_case(pu, ref, want):
    assert pu.n_reads == 1 + len(want)
    assert (pu.reads["flags"] & 1).tolist() == [0] + [1 if w is None else 0 for w in want]
    exp = pileup_from_alignments(ref, [w for w in want if w is not None])
    k = 1
    for i, w in enumerate(want, start=1):
        if w is None:
            continue
        assert _read_bytes(pu, i) == _read_bytes(exp, k), i
        k += 1


def test_cigar_operations_and_what_counts_as_clipped():
    """fill_with_cigar, main.rs:386-440: M / = / X copy query and contig columns, I puts '-' into the contig string, D into
    the query string, H is skipped; 600M 2I 500M 3D 400M at 2000 therefore gives the two strings written out in cigar_case
    (1505 columns, aln_t_e = 2000 + 1503 - 1), and 800= 1X 800= is one mismatch column.  A soft clip sets aln_q_s if it is
    the FIRST operation and aln_q_e = qs - l otherwise (main.rs:395-402); aln_q_e == 0 is then taken for "not set"
    (main.rs:436-438).  is_clip = aln_q_e - aln_q_s + max_clip_len(100) < rlen with rlen counting soft AND hard clips:
      150S 1600M        aln_q_s = 150, aln_q_e = 1750:  1600 + 100 < 1750                      -> clipped;
      1600M 150S 10H    aln_q_e = 1600:                 1600 + 100 < 1760                      -> clipped;
      10H 150S 1600M    the soft clip is not first: aln_q_s stays 0, aln_q_e = 150 - 150 = 0 = "not set" -> 1750 at the end:
                        1750 + 100 < 1760 is false                                             -> NOT clipped.
    On this contig of 500 100 bases clipped reads are kept with a label and emptied by the clip filter (both lie inside
    (50, L - 51): tests/test_oracle_pinning2.py): flags bit 0."""
    from nextpolish2_amd import io as np2io
    from nextpolish2_amd.bamio import records_to_arrays
    ref, recs, want = cigar_case()
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    check_cigar_case(orc.front_end(ref.encode(), arr, cig, asc, asc_off, np2io.FrontOpts()), ref, want)


# ---- k-mer score of a candidate longer than k, main.rs:743-770 ---------------------------------------------------------------
def test_a_long_candidate_scores_the_rarest_of_its_kmers():
    """retrieve_kmer_count, main.rs:756-764: a candidate string longer than k scores the MINIMUM table count over all of
    its k-mers (a missing k-mer counts 0), a shorter one the count of its first k-mer.  The pileup of
    tests/test_oracle_pinning.py's long-indel case (the contig lacks 25 bases, reads 4 and 5 carry them); the table
    holds the true sequence at count 50 — and one 21-mer from the middle of the 25 inserted bases again, written later,
    at count c (a k-mer keeps the count of the last sequence that holds it).  The inserting candidates' strings hold
    that 21-mer, so their score is min(50, ..., c, ..., 50) = c:
        c = 7                          -> k-scores [0, 0, 0, 0, 7, 7, 0, 0, 0]  (the others' first k-mer spans the junction)
        c = 4 < min_kmer_count = 5     -> the k-mer is not retrieved at all (kmer.rs: `<`), reads as 0, the minimum is 0:
                                          no candidate with a k-mer in the table, nothing is inserted."""
    truth = backbone(260, 9)
    pos = 100
    ins = truth[pos:pos + 25]
    ref = truth[:pos] + truth[pos + 25:]
    t_aln = ref[:pos] + "-" * 25 + ref[pos:]
    q_aln = ref[:pos] + ins + ref[pos:]
    alns = [(0, ref, ref)] * 3 + [(0, t_aln, q_aln)] * 2 + [(0, ref, ref)] * 3
    rare = truth[pos + 2:pos + 23]
    assert len(rare) == 21 and rare in ins
    for c, want in ((7, [0, 0, 0, 0, 7, 7, 0, 0, 0]), (4, [0] * 9)):
        o = orc.Oracle([yak_counted([(truth, 50), (rare, c)], 21)])
        o.set_trace(True)
        b, _ = o.polish(pileup_from_alignments(ref, alns), Opts(max_indel_len=30, iter_count=1, min_kmer_count=5))
        assert o.trace(0, "cand.kscore").tolist() == want
        assert b.tobytes().decode() == (truth if c == 7 else ref)


# ---- fill_order_stat's first-come tie, main.rs:812-848, 862-899 -------------------------------------------------------------
def test_two_alleles_of_equal_support_the_first_in_read_order_seeds():
    """Nine rows at X: the contig and reads 7, 8 carry c0 (no k-mer of it in the table: k-score 0), reads 1-3 carry B,
    reads 4-6 carry A; the table holds both the A and the B haplotype at count 50.  fill_order_stat visits the candidates
    with a k-score > 0 in order: B (candidate 1) has 3 copies -> max1 = (3, 1); A (candidate 4) has 3 copies as well and
    `c > max1_c` is false -> it becomes max2, B stays: sudoseed = B's string.  order_stat = {1: 3, 4: 3}; the contig's
    string occurs 3 times (> 1), so order 0 is entered with min_c = get_min_count(9) = 3; retain_sort_seqs sorts stably
    by that count, descending — orders 0, 1, 4 (3 each) before the rest (0) — and cuts below min_c:
                          retained candidates [0, 1, 4], the seed is B: the spliced consensus carries B at X.
    With the A reads before the B reads the same reasoning seeds A."""
    X = 100
    ref = backbone(220, 31)
    a, b_ = [x for x in "ACGT" if x not in (ref[X], ref[X - 1], ref[X + 1])][:2]
    hap_a, hap_b = put(ref, X, a), put(ref, X, b_)
    for first, second in ((hap_b, hap_a), (hap_a, hap_b)):
        alns = [(0, ref, first)] * 3 + [(0, ref, second)] * 3 + [(0, ref, ref)] * 2
        o = orc.Oracle([yak_counted([(hap_a, 50), (hap_b, 50)], 21)])
        o.set_trace(True)
        o.polish(pileup_from_alignments(ref, alns), Opts(iter_count=1))
        assert (o.trace(0, "lq.start").tolist(), o.trace(0, "lq.end").tolist()) == ([97], [104])
        assert [k > 0 for k in o.trace(0, "cand.kscore").tolist()] == [False, True, True, True, True, True, True, False, False]
        assert o.trace(0, "seed.order").tolist() == [0, 1, 4]
        assert o.trace(0, "seed.sudo").tobytes().decode() == first[97:105]
        assert o.trace(0, "cns_succ.base").tobytes().decode() == first


# ---- mark_hete_lqseqs and the reads a marker removes, main.rs:916-946, 948-980 ------------------------------------------------
def test_a_marker_needs_min_c_reads_and_singletons_take_no_part():
    """Twenty rows at X (get_min_count(20) = 3); the table holds every allele used, so all candidates score > 0.
    mark_hete_lqseqs: a region is a marker when its second most frequent string has max2_c >= min_c copies (same length
    as the first, and a valid SNP); in a marker, candidates whose string has fewer than min_c copies lose their k-score.
    phase_reads_by_lqseqs without -r: a read whose candidate differs from the contig's (candidate 0) at a marker is
    removed (main.rs:972-980); candidates with k-score 0 are skipped.
      2 reads carry B:                        max2_c = 2 < 3: no marker                     -> nobody is removed;
      3 reads carry B (reads 17-19):          marker                                        -> reads 17, 18, 19 removed;
      9 reads carry B (10-18), read 19 a C:   marker; C has 1 copy < 3: its k-score is zeroed -> reads 10-18 removed, 19 stays."""
    X = 100
    ref = backbone(220, 31)
    b_, c_ = [x for x in "ACGT" if x not in (ref[X], ref[X - 1], ref[X + 1])][:2]
    hb, hc = put(ref, X, b_), put(ref, X, c_)
    yak = yak_counted([(ref, 50), (hb, 50), (hc, 50)], 21)
    for alns, removed in (([(0, ref, ref)] * 17 + [(0, ref, hb)] * 2, []),
                          ([(0, ref, ref)] * 16 + [(0, ref, hb)] * 3, [17, 18, 19]),
                          ([(0, ref, ref)] * 9 + [(0, ref, hb)] * 9 + [(0, ref, hc)], list(range(10, 19)))):
        o = orc.Oracle([yak])
        o.set_trace(True)
        o.polish(pileup_from_alignments(ref, alns), Opts(iter_count=2))
        assert (o.trace(0, "lq.start").tolist(), o.trace(0, "lq.end").tolist()) == ([97], [104])
        assert sorted(o.trace(0, "invalid_ids").tolist()) == removed
        assert (int(o.trace(0, "hete.lable")[0]) & 0x40 != 0) == bool(removed)  # LQSEQS_LABLE_HETE, main.rs:657


# ---- yak_hash64, kmer.rs:223-233 ---------------------------------------------------------------------------------------------
def test_yak_hash64_known_answers():
    """kmer.rs:223-233 with mask = 4^k - 1.  By hand for k = 2 (mask 15: every right shift gives 0, every left shift by
    >= 4 vanishes under the mask):  0 -> !0 & 15 = 15 -> 15 * 265 = 3975 = 7 (mod 16) -> 7 * 21 = 147 = 3 (mod 16) -> 3;
    5 -> !5 & 15 = 10 -> 2650 = 10 (mod 16) -> 210 = 2 (mod 16) -> 2.  For the k-mer sizes in use, a second restatement
    written here from the Rust text, step by step, must agree on random keys."""
    assert orc.yak_hash64(0, 2) == 3 and orc.yak_hash64(5, 2) == 2

    def h(key, mask):
        key = (~key + (key << 21)) & mask
        key ^= key >> 24
        key = (key + (key << 3) + (key << 8)) & mask
        key ^= key >> 14
        key = (key + (key << 2) + (key << 4)) & mask
        key ^= key >> 28
        return (key + (key << 31)) & mask
    rng = np.random.default_rng(5)
    for k in (17, 21, 27, 31):
        mask = (1 << (2 * k)) - 1
        for key in rng.integers(0, mask, 2000, dtype=np.uint64).tolist():
            assert orc.yak_hash64(key, k) == h(key, mask)


# ---- the recheck prefers the contig's string, main.rs:1366-1395 ---------------------------------------------------------------
def test_the_recheck_prefers_a_valid_contig_string_over_the_majority():
    """Nine rows at X: the contig and reads 1, 2 carry c0, reads 3-8 carry B; the table holds BOTH haplotypes.
    fill_seed_lqseqs: the contig's string has 3 copies -> max1 = (3, 0); B (first at candidate 3) has 6 > 3 -> max1 = (6, 3):
    sudoseed = B.  order_stat = {0: 3, 3: 6} (only the first copy of a string has an entry), min_c = get_min_count(9) = 3;
    retain_sort_seqs sorts by that count, descending, and cuts below 3:  retained candidates [3, 0], the spliced consensus
    carries B.  reupdate_consensus_with_lqseqs with the same table: both strings make k-mers that are in it; the choice
    loop `if c == 0 || seq.order == 0 { c = p + 1 }` (main.rs:1371-1376) takes the first valid string — B — and then
    moves to the contig's because its order is 0 ("in case ref is prefer"):
                          the recheck puts the contig's string back — the result is the contig, six reads of nine notwithstanding."""
    X = 100
    ref = backbone(220, 31)
    b_ = other(ref[X], skip=(ref[X - 1], ref[X + 1]))
    hb = put(ref, X, b_)
    o = orc.Oracle([yak_counted([(ref, 50), (hb, 50)], 21)])
    o.set_trace(True)
    b, _ = o.polish(pileup_from_alignments(ref, [(0, ref, ref)] * 2 + [(0, ref, hb)] * 6), Opts(iter_count=1))
    assert (o.trace(0, "lq.start").tolist(), o.trace(0, "lq.end").tolist()) == ([97], [104])
    assert o.trace(0, "seed.order").tolist() == [3, 0]
    assert o.trace(0, "cns_succ.base").tobytes().decode() == hb
    assert o.trace(0, "rech0.sudo").tobytes().decode() == ref[97:105]
    assert b.tobytes().decode() == ref


def test_two_valid_strings_go_on_to_the_next_table():
    """main.rs:1378-1395, 1421-1428: a region with two or more valid strings keeps its RECH label for the next (longer-k)
    table; a recheck that finds no valid string leaves the choice as it was unless it is the first one.  The pileup of
    test_two_alleles_of_equal_support_the_first_in_read_order_seeds (retained candidates [c0, B, A], c0 without k-mers in
    any table, seed = B):
      k21 {A, B}                  both valid, B is the first valid one in the retained order     -> B
      k21 {A, B}, k31 {A}         the region is checked again with k = 31: only A is valid       -> A
      k21 {A, B}, k31 {neither}   nothing valid in the second recheck (iter_count = 2)           -> stays B."""
    X = 100
    ref = backbone(220, 31)
    a, b_ = [x for x in "ACGT" if x not in (ref[X], ref[X - 1], ref[X + 1])][:2]
    hap_a, hap_b = put(ref, X, a), put(ref, X, b_)
    alns = [(0, ref, hap_b)] * 3 + [(0, ref, hap_a)] * 3 + [(0, ref, ref)] * 2
    k21 = yak_counted([(hap_a, 50), (hap_b, 50)], 21)
    for yaks, want in (([k21], hap_b), ([k21, yak_counted([(hap_a, 50)], 31)], hap_a),
                       ([k21, yak_counted([(backbone(220, 32), 50)], 31)], hap_b)):
        o = orc.Oracle(yaks)
        o.set_trace(True)
        b, _ = o.polish(pileup_from_alignments(ref, alns), Opts(iter_count=1))
        assert o.trace(0, "seed.order").tolist() == [0, 1, 4]
        assert o.trace(0, "rech0.sudo").tobytes().decode() == hap_b[97:105]
        assert b.tobytes().decode() == want


# ---- which haplotype goes: -r with the two models, louvain.rs:290-339, 72-117 ---------------------------------------------------
def test_with_all_reads_the_model_decides_which_haplotype_goes():
    """One marker; the contig and reads 1-3 carry c0, reads 4-9 carry B, both haplotypes in the table; -r (use_all_reads), so
    no read is removed just for disagreeing with the contig (main.rs:977) and all nine enter the graph: +1 inside a
    haplotype, -1 across.  first_stage (louvain.rs:72-117) visits the nodes in ascending order and moves a node to the
    neighbouring community of largest positive gain, ties to the smaller id: 1 -> community 2 (gain 1 with 2 and with 3),
    2 stays (its own community ties with 3 and is the smaller), 3 -> 2; 4 -> 5, 5 stays, 6-9 -> 5; nothing moves in the
    next sweep.  Two communities, {1, 2, 3} of internal weight 3 and {4, ..., 9} of weight 15, joined by -18: a conflict.
      model "len" (no reference row): ranked by weight, {4..9} first -> the conflicting {1, 2, 3} is removed;
      model "ref": ranked by agreement with the contig's candidate, (+3, 3) for {1, 2, 3} against (-6, -6) -> reads 4-9 go.
    Without -r the six B reads are removed before any graph is built, whatever the model."""
    X = 100
    ref = backbone(220, 31)
    b_ = other(ref[X], skip=(ref[X - 1], ref[X + 1]))
    hb = put(ref, X, b_)
    alns = [(0, ref, ref)] * 3 + [(0, ref, hb)] * 6
    yak = yak_counted([(ref, 50), (hb, 50)], 21)
    for opts, removed in ((Opts(iter_count=2, use_all_reads=True, model="len"), [1, 2, 3]),
                          (Opts(iter_count=2, use_all_reads=True, model="ref"), [4, 5, 6, 7, 8, 9]),
                          (Opts(iter_count=2, use_all_reads=False, model="len"), [4, 5, 6, 7, 8, 9])):
        o = orc.Oracle([yak])
        o.set_trace(True)
        o.polish(pileup_from_alignments(ref, alns), opts)
        assert int(o.trace(0, "hete.lable")[0]) & 0x40
        assert sorted(o.trace(0, "invalid_ids").tolist()) == removed
