"""More hand-derived known answers for the oracle (oracle/np2_oracle.cpp), on the parts of the final pass and of the
phasing vote whose outcome hangs on an ORDER: the Cartesian recheck of chained LQ regions (last writer wins,
main.rs:1319-1366), the seed rules of fill_seed_lqseqs (main.rs:862-914), and a tie between two conflicting
communities (louvain.rs:313-339, settled by hashbrown's bucket order).

The reference ships no golden vectors for this path and cannot be built here (no Rust toolchain): every expectation is
derived by hand from the Rust source and written out in the test, so that a reader can re-derive it."""
import numpy as np

from nextpolish2_amd import Opts
from nextpolish2_amd._types import Yak
from nextpolish2_amd.synth import pileup_from_alignments
from oracle import np2_oracle as orc

CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def yak_counted(seqs_counts, k):
    """yak table from [(sequence, count)]: every canonical k-mer of a sequence gets its count; a k-mer shared by several
    sequences keeps the count of the LAST one that holds it (kmer.rs:72-170 layout, pre = 10)."""
    words = {}
    mask = (1 << (2 * k)) - 1
    for s, count in seqs_counts:
        fw = rv = 0
        l = 0
        for ch in s:
            c = CODE[ch]
            fw = ((fw << 2) | c) & mask
            rv = (rv >> 2) | ((3 ^ c) << (2 * (k - 1)))
            l += 1
            if l >= k:
                words[orc.yak_hash64(min(fw, rv), k)] = count
    buckets = [[] for _ in range(1024)]
    for h, c in words.items():
        buckets[h & 1023].append(((h >> 10) << 10) | c)
    off = np.zeros(1025, np.uint64)
    flat = []
    for b in range(1024):
        flat.extend(buckets[b])
        off[b + 1] = len(flat)
    return Yak(k, np.array(flat, dtype=np.uint64), off)


def backbone(n, seed):
    """A sequence without two equal neighbours (no homopolymer runs: LQ regions do not grow past their padding,
    main.rs:1600-1611) and, at this length, without a repeated k-mer."""
    rng = np.random.default_rng(seed)
    s = [int(rng.integers(0, 4))]
    while len(s) < n:
        c = int(rng.integers(0, 4))
        if c != s[-1]:
            s.append(c)
    return "".join("ACGT"[c] for c in s)


def other(base, skip=()):
    """a base different from `base` and from everything in `skip` (deterministic)"""
    return next(b for b in "ACGT" if b != base and b not in skip)


def put(seq, pos, base):
    return seq[:pos] + base + seq[pos + 1:]


# ---- (i) chained regions: Cartesian product order and last-writer-wins ---------------------------------------------------
def test_three_chained_regions_cartesian_order_and_last_writer_wins():
    """reupdate_consensus_with_lqseqs, main.rs:1196-1206 (chain: next.start < prev.end + k), 1319-1366.

    Three sites A < B < C, 14 bases apart, two alleles each (0 = the contig's).  One k = 31 table holds three haplotypes:
        H0 = (0,0,0) count 50,   Hy = (0,1,1) count 20,   Hx = (1,1,0) count 10
    (written Hx, Hy, H0: a k-mer that covers no site, or only site A on allele 0, or only site C on allele 0, is shared
    and keeps H0's 50; with the sites 14 apart every 31-mer that covers B also covers A or C, so nothing else is shared).
    Reads: 6 x H0, 3 x Hy, 3 x Hx -> every region keeps two candidates, the contig's first (7..10 against 3..6 reads),
    all labelled RECH after the seed pass.  The recheck chains the three regions and walks the products of their kept
    candidates with the LAST region fastest (itertools multi_cartesian_product over regions left to right):
        000 -> H0, every k-mer 50: A0 = B0 = C0 = 50        001, 010 -> a 31-mer over all three sites is in no haplotype: 0
        011 -> Hy: 20: A0 = 20 (overwrites 50), B1 = C1 = 20  100, 101 -> 0
        110 -> Hx: 10: A1 = 10, B1 = 10 (overwrites 20), C0 = 10 (overwrites 50)     111 -> 0
    so the k-scores end as A: [20, 10], B: [50, 10], C: [10, 20].  (First writer wins would leave A0 = 50; the first region
    fastest would visit 110 before 011 and leave B1 = 20.)  Every candidate is valid, the contig's is preferred
    (main.rs:1371-1384): the sequence stays the contig's, the regions stay RECH (two valid candidates each)."""
    k = 31
    bb = backbone(260, 21)
    pa, pb, pc = 100, 114, 128
    alt = {p: other(bb[p], skip=(bb[p - 1], bb[p + 1])) for p in (pa, pb, pc)}  # (no new homopolymer next to a site)

    def hap(a, b, c):
        s = bb
        for p, on in ((pa, a), (pb, b), (pc, c)):
            if on:
                s = put(s, p, alt[p])
        return s
    h0, hy, hx = hap(0, 0, 0), hap(0, 1, 1), hap(1, 1, 0)
    yak = yak_counted([(hx, 10), (hy, 20), (h0, 50)], k)
    alns = [(0, h0, h0)] * 6 + [(0, h0, hy)] * 3 + [(0, h0, hx)] * 3
    o = orc.Oracle([yak])
    o.set_trace(True)
    b, _ = o.polish(pileup_from_alignments(h0, alns), Opts(iter_count=1))  # (the final pass alone: no phasing vote)
    st, en = o.trace(0, "lq.start").tolist(), o.trace(0, "lq.end").tolist()
    assert len(st) == 3 and st[0] <= pc <= en[0] and st[1] <= pb <= en[1] and st[2] <= pa <= en[2]  # listed right to left
    assert all(st[g] < en[g + 1] + k for g in range(2))  # chained: next.start < prev.end + k
    # kept candidates after the seed pass: the contig's, then the first read carrying the other allele
    assert o.trace(0, "seed.cand_off").tolist() == [0, 2, 4, 6]
    assert o.trace(0, "seed.order").tolist() == [0, 7, 0, 7, 0, 10]  # C: first Hy read, B: first Hy read, A: first Hx read
    assert o.trace(0, "rech0.kscore").tolist() == [10, 20, 50, 10, 20, 10]  # C0 C1 | B0 B1 | A0 A1
    assert b.tobytes().decode() == h0


# ---- (ii) fill_seed_lqseqs: the lone winner, and |len(sudoseed) - len(ref)| > max_indel_len ---------------------------------
def _site_case(reads_alleles, seed=5, pos=100):
    """contig with a wrong base at `pos`; reads carry the listed alleles there ('' = the base deleted, two letters = one
    base inserted behind it); the k-mer table holds the truth only"""
    truth = backbone(220, seed)
    wrong = other(truth[pos], skip=(truth[pos - 1], truth[pos + 1]))
    ref = put(truth, pos, wrong)
    alns = []
    for al in reads_alleles:
        if al == "":
            alns.append((0, ref, ref[:pos] + "-" + ref[pos + 1:]))
        elif len(al) == 2:
            alns.append((0, ref[:pos + 1] + "-" + ref[pos + 1:], ref[:pos] + al + ref[pos + 1:]))
        else:
            alns.append((0, ref, put(ref, pos, al)))
    return truth, ref, wrong, alns


def test_lone_winner_is_promoted_only_when_the_other_reads_all_differ():
    """fill_seed_lqseqs, main.rs:889-899.  Six candidates (the contig + 5 reads) -> min_c = 2 (main.rs:803-811).  Only
    the read carrying the true base has a k-mer in the table, so max1_c = 1 < min_c at max1_p != 0.
    * the four other reads all differ from each other (two other bases, a deletion, an insertion): no_dupseq_lqseq holds
      (main.rs:851-860: the contig's own candidate is skipped), the lone winner and the contig are both raised to min_c,
      the region stays RECH with [contig, winner], and the recheck (contig's candidate scores 0, the winner's > 0) takes
      the winner: the base is corrected;
    * the same with two reads sharing a wrong base: a duplicate among the reads, no promotion, only order_stat[0] = min_c
      (main.rs:897-899): retain_sort_seqs keeps the contig alone, sudoseed = the contig's sequence: the error stays."""
    truth = backbone(220, 5)
    t = truth[100]
    _, ref, wrong, _ = _site_case([t])
    x, y = [b for b in "ACGT" if b not in (t, wrong)]
    ins = t + other(t, skip=(truth[101],))
    for alleles, fixed in (([t, x, y, "", ins], True), ([t, x, x, "", ins], False)):
        truth, ref, wrong, alns = _site_case(alleles)
        o = orc.Oracle([yak_counted([(truth, 50)], 21)])
        o.set_trace(True)
        b, _ = o.polish(pileup_from_alignments(ref, alns), Opts(iter_count=1))
        assert len(o.trace(0, "lq.start")) == 1
        ks = o.trace(0, "cand.kscore").tolist()
        assert ks == [0, 50, 0, 0, 0, 0]  # contig, the true read, the four others
        if fixed:
            assert o.trace(0, "seed.order").tolist() == [0, 1] and o.trace(0, "seed.lable").tolist() == [0x80 | 0x20]
            assert o.trace(0, "rech0.kscore").tolist() == [0, 50]
            assert b.tobytes().decode() == truth
        else:
            assert o.trace(0, "seed.order").tolist() == [] and o.trace(0, "seed.lable").tolist() == [0x80]
            assert b.tobytes().decode() == ref


def test_long_indel_seed_is_refused_beyond_max_indel_len():
    """fill_seed_lqseqs, main.rs:903-912.  The contig lacks 25 bases; 2 of 8 reads carry them (the only candidates with
    k-mers in the table), 6 reads agree with the contig: 9 candidates, min_c = 3, max1_c = 2 at max1_p != 0 -> both the
    contig's candidate (shared by 7) and the winner are raised to min_c; the stable sort keeps the contig first, so
    seqs[0] is the contig's sequence and |len(sudoseed) - len(seqs[0])| = 25.
    * max_indel_len = 20 (the default): skip_long_lqseq -> sudoseed = the contig's sequence, RECH cleared: not inserted;
    * max_indel_len = 30: the region stays RECH, the recheck finds only the inserting candidate valid: inserted."""
    truth = backbone(260, 9)
    pos = 100
    ins = truth[pos:pos + 25]
    ref = truth[:pos] + truth[pos + 25:]
    t_aln = ref[:pos] + "-" * 25 + ref[pos:]
    q_aln = ref[:pos] + ins + ref[pos:]
    alns = [(0, ref, ref)] * 3 + [(0, t_aln, q_aln)] * 2 + [(0, ref, ref)] * 3
    for max_indel, inserted in ((20, False), (30, True)):
        o = orc.Oracle([yak_counted([(truth, 50)], 21)])
        o.set_trace(True)
        b, p = o.polish(pileup_from_alignments(ref, alns), Opts(max_indel_len=max_indel, iter_count=1))
        assert len(o.trace(0, "lq.start")) == 1
        ks = o.trace(0, "cand.kscore").tolist()
        assert [x > 0 for x in ks] == [False, False, False, False, True, True, False, False, False]
        if inserted:
            assert o.trace(0, "seed.order").tolist() == [0, 4]
            assert b.tobytes().decode() == truth
            assert (np.asarray(p) == o.trace(0, "lq.start")[0]).sum() == len(o.trace(0, "rech0.sudo"))
        else:
            assert o.trace(0, "seed.order").tolist() == [] and o.trace(0, "seed.lable").tolist() == [0x80]
            assert b.tobytes().decode() == ref


# ---- (iii) two conflicting communities that tie -----------------------------------------------------------------------------
def _two_pairs(a1, a2, b1, b2):
    e = []

    def add(a, b, w):
        e.append((a, b, w))
        e.append((b, a, w))
    add(a1, a2, 1.0)
    add(b1, b2, 1.0)
    for a in (a1, a2):
        for b in (b1, b2):
            add(a, b, -1.0)
    return e


def test_tie_between_conflicting_communities_is_settled_by_bucket_order():
    """louvain.rs:72-117, 119-195, 197-245, 313-339.  Two read pairs, +1 inside a pair, -1 across.  first_stage visits
    the reads in id order: the smaller read of a pair joins the larger one's community (its only positive neighbour), so
    the communities are named after the larger reads and have weight 1.0 each.  second_stage re-inserts the two ids into
    fresh maps (2 entries: 4 buckets, bucket = FxHash(id) & 3, FxHash(id) = id * 0x517cc1b727220a95 mod 2^64; no
    collision), get_communities lists them in bucket order, and phase_communities' STABLE sort leaves a tie on the
    weight (or on (count, weight) against the contig) in that order; the earlier community invalidates the later one.
      pairs {1,2}, {3,4}: ids 2 (bucket 2) and 4 (bucket 0)  -> order [4, 2] -> the reads of community 2 lose: [1, 2]
      pairs {1,2}, {6,7}: ids 2 (bucket 2) and 7 (bucket 3)  -> order [2, 7] -> the reads of community 7 lose: [6, 7]"""
    m = (1 << 64) - 1
    fx = lambda key: (key * 0x517CC1B727220A95) & m  # noqa: E731
    assert [fx(i) & 3 for i in (2, 4, 7)] == [2, 0, 3]
    assert sorted(orc.phase_communities(_two_pairs(1, 2, 3, 4), None)) == [1, 2]
    assert sorted(orc.phase_communities(_two_pairs(1, 2, 6, 7), None)) == [6, 7]
    # -m ref with the same support for both communities: count 1 and weight 1.0 each, the same tie, the same losers
    assert sorted(orc.phase_communities(_two_pairs(1, 2, 3, 4), {1: 1.0, 3: 1.0})) == [1, 2]
    assert sorted(orc.phase_communities(_two_pairs(1, 2, 6, 7), {2: 1.0, 6: 1.0})) == [6, 7]
    # ... and a real difference overrides the bucket order (Reverse((count, weight)): the better supported one first)
    assert sorted(orc.phase_communities(_two_pairs(1, 2, 3, 4), {1: 1.0, 2: 1.0, 3: 1.0})) == [3, 4]


def test_the_products_own_vote_settles_the_same_ties_the_same_way():
    """np2_phase_vote (csrc/np2_phase_host.hpp: the product's own Louvain + SwissTable order model) on the two tie cases
    derived by hand in test_oracle_pinning: the bucket order of a 4-bucket table decides."""
    from nextpolish2_amd.api import phase_vote

    def vote(a1, a2, b1, b2, ref=None):
        keys = sorted({a1, a2, b1, b2})
        edges = [(min(a1, a2), max(a1, a2), 1.0), (min(b1, b2), max(b1, b2), 1.0)]
        edges += [(min(a, b), max(a, b), -1.0) for a in (a1, a2) for b in (b1, b2)]
        return sorted(phase_vote(keys, sorted(edges), ref))
    assert vote(1, 2, 3, 4) == [1, 2]
    assert vote(1, 2, 6, 7) == [6, 7]
    assert vote(1, 2, 3, 4, {1: 1.0, 3: 1.0}) == [1, 2]
    assert vote(1, 2, 3, 4, {1: 1.0, 2: 1.0, 3: 1.0}) == [3, 4]
