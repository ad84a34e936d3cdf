"""GPU parity tests: the HIP path (through the C-ABI, include/np2.h) vs the CPU oracle.

Integer/byte work: the bar is bit-exact, at every stage and on the final consensus."""
import numpy as np
import pytest

from golden_util import golden_cases, load_golden
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd._types import Pileup, Yak
from nextpolish2_amd.api import Np2Error
from nextpolish2_amd.synth import Synth, pileup_from_alignments
from oracle import np2_oracle as orc

pytestmark = pytest.mark.gpu

STAGES = ["graph.off", "graph.bases", "graph.delta", "graph.count", "cns_raw.pos", "cns_raw.base", "lq.start", "lq.end",
          "cand.cand_off", "cand.order", "cand.seq_off", "cand.seq", "cand.kmer", "cand.kscore", "hete.lable",
          "hete.kscore", "invalid_ids", "seed.lable", "seed.sudo", "seed.cand_off", "seed.order", "cns_succ.pos",
          "cns_succ.base", "rech0.kscore", "rech0.lable", "rech0.sudo", "cns_rech0.pos", "cns_rech0.base",
          "rech1.kscore", "rech1.lable", "rech1.sudo", "cns_rech1.pos", "cns_rech1.base"]


def check_all_stages(pu, yaks, opts):
    o = orc.Oracle(yaks)
    o.set_trace(True)
    ob, op = o.polish(pu, opts)
    g = Polisher(yaks)
    g.set_trace(True)
    gb, gp = g.polish(pu, opts)
    for ps in range(opts.iter_count):
        for st in STAGES:
            a, b = o.trace(ps, st), g.trace(ps, st)
            assert (a is None) == (b is None), (ps, st)
            if a is not None:
                assert a.shape == b.shape and np.array_equal(a, b), f"pass {ps} stage {st} differs"
    assert np.array_equal(ob, gb) and np.array_equal(op, gp)
    return gb, gp


@pytest.mark.parametrize("name", golden_cases())
def test_golden_fixtures(name):
    pu, yaks, opts, exp_b, exp_p, _ = load_golden(name)
    gb, gp = Polisher(yaks).polish(pu, opts)
    assert np.array_equal(gb, exp_b) and np.array_equal(gp, exp_p)


@pytest.mark.parametrize("seed,diploid,ks", [(31, False, [21]), (32, True, [21, 31]), (33, True, [21]), (34, False, [17, 21, 31])])
def test_stage_parity_synthetic(seed, diploid, ks):
    s = Synth(50000, depth=30, seed=seed, diploid=diploid, read_len_mean=8000.0, read_len_sd=1500.0)
    check_all_stages(s.pileup, [s.yak(k) for k in ks], Opts())


def test_dp_two_stream_variant(monkeypatch):
    # NP2_DP_FORK: short-run and long-run DP kernels side by side on two streams, each classifying the runs itself
    # (the default lets the short kernel list what it leaves to the others)
    monkeypatch.setenv("NP2_DP_FORK", "1")
    s = Synth(60000, depth=30, seed=35, diploid=True, read_len_mean=8000.0, read_len_sd=1500.0, read_err_rate=0.01)
    check_all_stages(s.pileup, [s.yak(21)], Opts())


@pytest.mark.parametrize("opts", [Opts(iter_count=1), Opts(iter_count=3), Opts(model="len"), Opts(use_all_reads=True),
                                  Opts(min_kmer_count=60), Opts(max_indel_len=0)])
def test_stage_parity_options(opts, small_diploid):
    s, yaks = small_diploid
    check_all_stages(s.pileup, yaks, opts)


def test_high_error_reads_and_low_depth():
    s = Synth(30000, depth=8, seed=41, diploid=True, read_err_rate=0.02, read_len_mean=4000.0, read_len_sd=800.0)
    check_all_stages(s.pileup, [s.yak(21)], Opts())


def test_deep_pileup_hits_the_60_candidate_cap():
    s = Synth(20000, depth=120, seed=42, read_len_mean=5000.0, read_len_sd=800.0)
    gb, _ = check_all_stages(s.pileup, [s.yak(21)], Opts())
    assert gb.tobytes() == s.hap1


def test_dropped_reads_and_position_zero_starts():
    s = Synth(30000, depth=20, seed=43, read_len_mean=5000.0, read_len_sd=800.0)
    pu = s.pileup
    reads = pu.reads.copy()
    reads["flags"][5::7] = 1  # align_bases == [] but index retained (main.rs:571)
    check_all_stages(Pileup(pu.ref, reads, pu.nibbles), [s.yak(21)], Opts())


def test_hand_made_edge_cases():
    rng = np.random.default_rng(7)
    truth = "".join("ACGT"[i] for i in rng.integers(0, 4, 400))
    ref = truth[:150] + "T" + truth[150:]  # contig carries an extra base
    t_ins = ref
    q_del = truth[:150] + "-" + truth[150:]
    alns = [(0, t_ins, q_del)] * 6 + [(0, ref, ref)] * 2
    # a read with a long insertion and lower-case / N letters
    t2 = ref[:60] + "-----" + ref[60:300]
    q2 = ref[:60] + "acgNn" + ref[60:300]
    alns += [(0, t2, q2), (3, ref[3:], ref[3:]), (1, ref[1:390], ref[1:390])]
    pu = pileup_from_alignments(ref, alns)
    k = 21
    from test_oracle import yak_from_seqs
    check_all_stages(pu, [yak_from_seqs([truth], k)], Opts())


def wild_pileup(seed, L=9000, n_reads=36, ins_p=0.03, del_p=0.03, sub_p=0.03, long_ins=True):
    """Random alignments far outside HiFi statistics: dense indels, insertion runs up to 100 columns (crossing the
    32-column pieces and 4096-column chunks of the dense kernel), N/M letters, reads starting at positions 0..2."""
    rng = np.random.default_rng(seed)
    ref = "".join("ACGT"[i] for i in rng.integers(0, 4, L))
    alns = []
    for r in range(n_reads):
        s = int(rng.choice([0, 0, 1, 2, int(rng.integers(0, L // 2))]))
        e = int(min(L, s + rng.integers(1200, L)))
        t, q = [], []
        p = s
        while p < e:
            u = rng.random()
            edge = p < s + 8 or p >= e - 8  # 8-match anchors like trim(8) leaves them
            if edge or u > ins_p + del_p + sub_p:
                t.append(ref[p]); q.append(ref[p]); p += 1
            elif u < ins_p:
                n = int(rng.integers(1, 4))
                if long_ins and rng.random() < 0.15:
                    n = int(rng.integers(20, 100))
                for _ in range(n):
                    t.append("-"); q.append("ACGTNM"[int(rng.integers(0, 6 if rng.random() < 0.1 else 4))])
            elif u < ins_p + del_p:
                n = int(rng.integers(1, 4))
                for _ in range(n):
                    if p < e - 8:
                        t.append(ref[p]); q.append("-"); p += 1
            else:
                t.append(ref[p]); q.append("ACGT"[("ACGT".index(ref[p]) + int(rng.integers(1, 4))) % 4]); p += 1
        alns.append((s, "".join(t), "".join(q)))
    alns.sort(key=lambda a: a[0])
    return ref, pileup_from_alignments(ref, alns)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_wild_pileups_all_stages(seed):
    from test_oracle import yak_from_seqs
    ref, pu = wild_pileup(seed, ins_p=0.01 * seed, del_p=0.02, sub_p=0.02)
    yaks = [yak_from_seqs([ref], 21, count=30), yak_from_seqs([ref], 27, count=9)]
    check_all_stages(pu, yaks, Opts())
    check_all_stages(pu, yaks[:1], Opts(use_all_reads=True, model="len"))


def test_wild_pileup_long_reads_cross_chunks():
    from test_oracle import yak_from_seqs
    ref, pu = wild_pileup(11, L=30000, n_reads=24, ins_p=0.02, del_p=0.01, sub_p=0.01)
    assert int(pu.reads["n_cols"][1:].max()) > 3 * 4096
    check_all_stages(pu, [yak_from_seqs([ref], 21, count=30)], Opts())


@pytest.mark.parametrize("cap", ["1", "64"])
def test_device_wide_sort_fallback_matches_tile_sort(monkeypatch, cap):
    # a tile with more records than its bucket spills and takes the device-wide sort: force that with tiny buckets
    # (cap 1: nearly every record spills; cap 64: a mix of bucket-resident and spilled records)
    monkeypatch.setenv("NP2_TILE_CAP", cap)
    s = Synth(40000, depth=25, seed=77, diploid=True, read_len_mean=6000.0, read_len_sd=1000.0)
    check_all_stages(s.pileup, [s.yak(21)], Opts())
    from test_oracle import yak_from_seqs
    ref, pu = wild_pileup(3, ins_p=0.03, del_p=0.02, sub_p=0.02)
    check_all_stages(pu, [yak_from_seqs([ref], 21, count=30)], Opts())


def test_long_insertion_overflows_a_tile():
    # a 3 kb insertion carried by every read puts > TILE_CAP exception records into one contig tile
    rng = np.random.default_rng(5)
    L = 6000
    ref = "".join(rng.choice(list("ACGT"), L))
    ins = "".join(rng.choice(list("ACGT"), 3000))
    alns = []
    for i in range(12):
        t = ref[:3000] + "-" * len(ins) + ref[3000:]
        q = ref[:3000] + ins + ref[3000:]
        s0 = 10 * i
        cut = s0  # reads start at staggered positions, all span the insertion
        alns.append((s0, t[cut:], q[cut:]))
    pu = pileup_from_alignments(ref, alns)
    from test_oracle import yak_from_seqs
    truth = ref[:3000] + ins + ref[3000:]
    check_all_stages(pu, [yak_from_seqs([truth], 21, count=30)], Opts())


def test_very_long_reads_span_many_chunks_and_blocks():
    # reads of > 64 chunks (262 k columns): the dense pass sums the counts of a read's earlier chunks across several
    # 64-wide windows of status words and across thread blocks
    s = Synth(600000, depth=6, seed=91, read_len_mean=330000.0, read_len_sd=20000.0)
    assert int(s.pileup.reads["n_cols"][1:].max()) > 70 * 4096
    gb, _ = check_all_stages(s.pileup, [s.yak(21)], Opts())
    assert gb.tobytes() == s.hap1


def test_scaffold_gap_uncovered_stretch_and_soft_masking():
    # a contig with an N gap (reads stop before it and resume after it), an uncovered stretch (coverage 1: only the
    # contig's own row, main.rs:1586-1588 resets the LQ state there) and lower-case (soft-masked) letters
    from test_oracle import yak_from_seqs
    s1 = Synth(24000, depth=25, seed=71, read_len_mean=3000.0, read_len_sd=500.0)
    ref = bytearray(s1.pileup.ref.tobytes())
    L = len(ref)
    gap_a, gap_b = 9000, 9600      # N gap
    hole_a, hole_b = 15000, 16200  # no read covers this
    ref[gap_a:gap_b] = b"N" * (gap_b - gap_a)
    ref[3000:3300] = bytes(ref[3000:3300]).lower()
    alns = []
    recs = pileup_to_alignments(s1.pileup)
    for (ts, t, q) in recs:
        te = ts + sum(1 for c in t if c != "-") - 1
        if ts < gap_b + 20 and te > gap_a - 20:
            continue  # drop reads touching the gap
        if ts < hole_b and te > hole_a:
            continue  # and the uncovered stretch
        alns.append((ts, t, q))
    # rebuild target strings against the edited contig (case / N changes only affect reads we dropped or the mask)
    fixed = []
    for (ts, t, q) in alns:
        tt, p = [], ts
        for c in t:
            if c == "-":
                tt.append("-")
            else:
                tt.append(chr(ref[p])); p += 1
        fixed.append((ts, "".join(tt), q))
    pu = pileup_from_alignments(bytes(ref).decode(), fixed)
    yaks = [yak_from_seqs([s1.hap1.decode()], 21, count=40)]
    gb, gp = check_all_stages(pu, yaks, Opts())
    out = gb.tobytes()
    assert b"N" * (gap_b - gap_a) in out and out == out.upper()


def pileup_to_alignments(pu):
    """packed reads (index >= 1) -> [(aln_t_s, gapped target, gapped query)]"""
    code2 = "ACGT-NM"
    ref = pu.ref.tobytes().decode()
    out = []
    for r in range(1, pu.n_reads):
        rd = pu.reads[r]
        n = int(rd["n_cols"])
        b = pu.nibbles[int(rd["nib_off"]):int(rd["nib_off"]) + (n + 1) // 2 + 1]
        nib = np.empty(2 * len(b), dtype=np.uint8)
        nib[0::2] = b >> 4
        nib[1::2] = b & 15
        t, q, p = [], [], int(rd["aln_t_s"])
        for c in nib[:n]:
            if c & 8:
                t.append("-")
            else:
                t.append(ref[p]); p += 1
            q.append(code2[c & 7])
        out.append((int(rd["aln_t_s"]), "".join(t), "".join(q)))
    return out


def test_non_acgt_reference_letters():
    rng = np.random.default_rng(9)
    base = "".join("ACGT"[i] for i in rng.integers(0, 4, 300))
    ref = base[:100] + "R" + base[101:200] + "n" + base[201:]
    alns = [(0, ref, base)] * 5 + [(0, ref, ref)] * 2
    from test_oracle import empty_yak
    check_all_stages(pileup_from_alignments(ref, alns), [empty_yak()], Opts())


def test_score_strings_and_lookup_match_oracle(small_diploid):
    s, yaks = small_diploid
    rng = np.random.default_rng(3)
    h = s.hap1
    strs = []
    for _ in range(500):
        a = int(rng.integers(0, len(h) - 200))
        ln = int(rng.integers(0, 120))
        b = bytearray(h[a:a + ln])
        if ln and rng.random() < 0.3:
            b[int(rng.integers(0, ln))] = ord("N")
        if ln and rng.random() < 0.3:
            b[int(rng.integers(0, ln))] = ord("acgt"[int(rng.integers(0, 4))])
        strs.append(bytes(b))
    g, o = Polisher(yaks), orc.Oracle(yaks)
    for yi in range(len(yaks)):
        for mk in (1, 5, 40):
            assert np.array_equal(g.score_strings(yi, strs, mk), o.score_strings(yi, strs, mk))
        hashes = np.concatenate([yaks[yi].words[:2000] >> np.uint64(10) << np.uint64(10), rng.integers(0, 1 << 40, 500, dtype=np.uint64)])
        # reconstruct full hashes for stored words: bucket index is implied by the file position
        offs = yaks[yi].bucket_off
        bucket = np.searchsorted(offs, np.arange(2000), side="right") - 1
        hashes[:2000] |= bucket.astype(np.uint64)
        assert np.array_equal(g.lookup_hashes(yi, hashes, 5), o.lookup_hashes(yi, hashes, 5))


def test_bad_inputs_are_rejected_with_error_codes(small_haploid):
    s, yaks = small_haploid
    g = Polisher(yaks)
    pu = s.pileup
    reads = pu.reads.copy()
    reads["n_cols"][3] += 2  # descriptor disagrees with the stream
    with pytest.raises(Np2Error) as e:
        g.polish(Pileup(pu.ref, reads, pu.nibbles), Opts())
    assert e.value.code == -1
    reads = pu.reads.copy()
    reads["aln_t_s"][0] = 1  # reads[0] must be the contig itself
    with pytest.raises(Np2Error):
        g.polish(Pileup(pu.ref, reads, pu.nibbles), Opts())
    with pytest.raises(Np2Error):
        Polisher([Yak(33, np.zeros(0, np.uint64), np.zeros(1025, np.uint64))])
    # the context is still usable afterwards
    gb, _ = g.polish(pu, Opts())
    assert gb.tobytes() == s.hap1


def test_identical_pass_reuse_and_bases_only_output(small_haploid, small_diploid):
    # trace mode recomputes every pass; the default mode reuses a byte-identical pass (no read voted out)
    for s, yaks in (small_haploid, small_diploid):
        g = Polisher(yaks)
        c = g.upload(s.pileup)
        b1, p1 = g.polish_resident(c, Opts())
        g.set_trace(True)
        b2, p2 = g.polish_resident(c, Opts())
        g.set_trace(False)
        assert np.array_equal(b1, b2) and np.array_equal(p1, p2)
        b3, span = g.polish_resident(c, Opts(), want_pos=False)
        assert np.array_equal(b1, b3) and span == (int(p1[0]), int(p1[-1]))
        b4, _ = g.polish_resident(c, Opts(iter_count=4))
        ob, _ = orc.Oracle(yaks).polish(s.pileup, Opts(iter_count=4))
        assert np.array_equal(b4, ob)


def test_phasing_vote_at_moderate_scale():
    # thousands of reads in the signed read graph: the edge-driven product Louvain must pick exactly the reads
    # the oracle's literal (O(C^2)) restatement picks
    s = Synth(1_500_000, depth=30, seed=77, diploid=True)
    yaks = [s.yak(21), s.yak(31)]
    o = orc.Oracle(yaks)
    o.set_trace(True)
    ob, op = o.polish(s.pileup, Opts())
    g = Polisher(yaks)
    g.set_trace(True)
    gb, gp = g.polish(s.pileup, Opts())
    assert np.array_equal(o.trace(0, "invalid_ids"), g.trace(0, "invalid_ids"))
    assert len(o.trace(0, "invalid_ids")) > 1000
    assert np.array_equal(ob, gb) and np.array_equal(op, gp)
    g.set_trace(False)
    gb2, _ = g.polish(s.pileup, Opts(model="len"))
    ob2, _ = orc.Oracle(yaks).polish(s.pileup, Opts(model="len"))
    assert np.array_equal(ob2, gb2)


def test_full_size_properties():
    """BASELINE.json configs[1] scale (4.6 Mb, 30x, k21): size-independent properties — the polished
    sequence equals the simulated truth, positions are non-decreasing, and a second call is identical."""
    s = Synth(4_600_000, depth=30, seed=1)
    g = Polisher([s.yak(21)])
    c = g.upload(s.pileup)
    b1, p1 = g.polish_resident(c, Opts())
    b2, p2 = g.polish_resident(c, Opts())
    assert np.array_equal(b1, b2) and np.array_equal(p1, p2)
    assert np.all(np.diff(p1.astype(np.int64)) >= 0)
    assert b1.tobytes() == s.hap1
    assert s.pileup.ref.tobytes() != s.hap1
