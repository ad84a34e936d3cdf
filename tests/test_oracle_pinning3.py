"""Hand-derived known answers, third set: the tie-breaks of the consensus DP (get_cns_from_align_tags, main.rs:1643-1683).
A node's best predecessor is replaced by a LATER one of EQUAL score unless that one starts with a gap
(`score > kmer_score || (score == kmer_score && base1.q_base != 4)`, main.rs:1664), nodes being visited in the order the
reads pushed them (the contig is read 0, main.rs:1732-1739) after a stable sort by delta (Msa::sort, main.rs:227-229);
at the last position the LAST node of maximal score wins (`kmer_score >= global_best_kmer.score`, main.rs:1676).
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
    put first) and the score 10 * count - 4 * coverage (main.rs:1652-1663).  Ten rows; k reads carry one extra base after
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
    """main.rs:1660-1662: a predecessor whose first column is a read's head sentinel is skipped once the node's middle
    column lies at t_pos >= 3 ("this can prevent the later backtracking algorithm from stopping at the start mapping
    position of reads"), and the backtrack ends at the first node whose middle column is a head sentinel (main.rs:1628).
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
