"""Round-2 parity loose ends: --out_pos (main.rs:613-625), the f32 split of -a (option.rs:232,258-259), the documented
reference's walk back from its default node when no end node reaches a score >= 0 (main.rs:1651,1680), repeated keys in a
k-mer dump (kmer.rs:148-167), and a multi-context soak."""
import gzip
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd.api import Np2Error
from nextpolish2_amd.bamio import pileup_to_records, records_to_arrays, write_bam
from nextpolish2_amd.cli import split_map_len
from nextpolish2_amd.synth import Synth, pileup_from_alignments
from oracle import np2_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bundle(tmp_path, s, name="ctgA"):
    recs = pileup_to_records(s.pileup, tid=0, rng=np.random.default_rng(3), decorate=True)
    write_bam(str(tmp_path / "m.bam"), [(name, s.pileup.L)], recs)
    with gzip.open(tmp_path / "g.fa.gz", "wt") as f:
        f.write(f">{name} some description\n{s.pileup.ref.tobytes().decode()}\n>tiny\nacgtNN\n")
    np2io.write_yak(str(tmp_path / "k21.yak"), s.yak(21))
    return recs


def test_out_pos_table_matches_the_oracle(tmp_path):
    s = Synth(40000, depth=20, seed=901, diploid=True, read_len_mean=5000.0, read_len_sd=700.0)
    recs = _bundle(tmp_path, s)
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    pu = orc.front_end(s.pileup.ref.tobytes(), arr, cig, asc, asc_off, np2io.FrontOpts())
    b, p = orc.Oracle([s.yak(21)]).polish(pu, Opts())
    exp = b"".join(b"ctgA\t%c\t%d\n" % (bytes([x]), int(q)) for x, q in zip(b, p))
    exp += b"".join(b"tiny\t%c\t%d\n" % (bytes([x]), i) for i, x in enumerate(b"ACGTNN"))  # -u upper-cases pass-through
    r = subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", "--out_pos", "-u", "-L", "10000", str(tmp_path / "m.bam"),
                        str(tmp_path / "g.fa.gz"), str(tmp_path / "k21.yak")], capture_output=True,
                       env=dict(os.environ, PYTHONPATH=ROOT), timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout == exp
    assert len(set(p.tolist())) < len(p) or np.all(np.diff(p.astype(np.int64)) >= 0)


def test_min_map_len_is_split_in_single_precision(tmp_path):
    # option.rs:232 parses -a as f32: 500.3 -> (500, 0.29998779...), so (rlen as f32 * fra) as i64 = 2999 for rlen 10000
    # where double precision would give 3000 (2999.99... vs 3000.0000001)
    ml, fra = split_map_len("500.3")
    assert ml == 500 and abs(fra - 0.29998779296875) < 1e-12
    assert int(np.float32(10000) * np.float32(fra)) == 2999
    assert split_map_len("500.5") == (500, 0.5) and split_map_len("1000") == (1000, 0.0)
    # a read whose reference span sits exactly between the two thresholds is admitted with the f32 fraction, like the
    # oracle front end given the same f32 value, and would be rejected with 0.3
    s = Synth(30000, depth=12, seed=902, read_len_mean=10000.0, read_len_sd=1.0, read_len_min=9990)
    recs = pileup_to_records(s.pileup, tid=0, rng=np.random.default_rng(1), decorate=False)
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    pol = Polisher([s.yak(21)])
    for frac in (fra, 0.3, 0.29):
        fo = np2io.FrontOpts(min_map_len=ml, min_map_fra=frac)
        pu = orc.front_end(s.pileup.ref.tobytes(), arr, cig, asc, asc_off, fo)
        c = np2io.contig_from_records(pol, s.pileup.ref.tobytes(), arr, cig, seq4, fo)
        ex = np2io.export_contig(pol, c, s.pileup.ref.tobytes())
        assert np.array_equal(ex.reads, pu.reads)
        c.free()


def _negative_pileup(L, tail_clean, seed=7):
    """Three full-length reads that disagree with the contig and with each other at every column but the last
    `tail_clean`: every node of such a position has count 1 under coverage 4, i.e. 10 - 16 = -6 a step."""
    rng = np.random.default_rng(seed)
    ref = "".join(rng.choice(list("ACGT"), L))
    alns = []
    for k in range(3):
        q = "".join("ACGT"[("ACGT".index(c) + 1 + k) % 4] if i < L - tail_clean else c for i, c in enumerate(ref))
        alns.append((0, ref, q))
    return ref, pileup_from_alignments(ref, alns)


@pytest.mark.parametrize("tail_clean", [0, 1, 3, 4, 9])
@pytest.mark.parametrize("iters", [1, 2])
def test_negative_best_score_walks_back_from_the_default_node(tail_clean, iters):
    """main.rs:1651,1680: if no node at the last position reaches score >= 0 the reference backtracks from its default
    Kmer: an 'A' at L - 1 with count 0 (qv 0), then node 0 of position L - 2.  Rounds 2-5 refused such a pileup; product
    and oracle now both restate it.  tail_clean = 0 / 1: the last position is dirty (its run's path starts with the 'A');
    3: L - 1 clean, L - 2 dirty (the run ending there is entered at N0(L - 2) instead of N0(L - 1)'s best predecessor);
    4, 9: a clean tail (only the base and its quality class change)."""
    L = 600
    ref, pu = _negative_pileup(L, tail_clean)
    y = Synth(2000, seed=3).yak(21)
    ob, op = orc.Oracle([y]).polish(pu, Opts(iter_count=iters))
    assert chr(ob[-1]) == "A" and op[-1] == L - 1
    for env in ({}, {"NP2_FRONT_UNFUSED": "1"}):
        code = (
            "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_gpu_loose_ends import _negative_pileup\n"
            "from nextpolish2_amd import Opts, Polisher\nfrom nextpolish2_amd.synth import Synth\n"
            "ref, pu = _negative_pileup(%d, %d)\n"
            "g = Polisher([Synth(2000, seed=3).yak(21)])\n"
            "b, p = g.polish(pu, Opts(iter_count=%d))\n"
            "np.save(sys.argv[1], b); np.save(sys.argv[2], p)\n" % (ROOT, os.path.join(ROOT, "tests"), L, tail_clean, iters))
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            r = subprocess.run([sys.executable, "-c", code, d + "/b.npy", d + "/p.npy"], capture_output=True, text=True,
                               env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr[-2000:]
            gb, gp = np.load(d + "/b.npy"), np.load(d + "/p.npy")
        assert np.array_equal(gb, ob) and np.array_equal(gp, op), (tail_clean, iters, env)


def test_negative_best_score_on_short_contigs_and_through_the_batch_driver():
    """The same artefact where the re-walked run touches positions 0-2 (path starts are only accepted there,
    main.rs:1666-1668), and on the batch driver's recorded launches."""
    from nextpolish2_amd import BatchPolisher
    y = Synth(2000, seed=3).yak(21)
    g, o = Polisher([y]), orc.Oracle([y])
    pus = []
    for L, tail in ((5, 3), (6, 3), (7, 0), (9, 4), (40, 3), (40, 0), (300, 3)):
        for seed in (1, 2):
            pus.append(_negative_pileup(L, tail, seed)[1])
    want = [o.polish(pu, Opts(iter_count=1)) for pu in pus]
    for pu, (ob, op) in zip(pus, want):
        gb, gp = g.polish(pu, Opts(iter_count=1))
        assert np.array_equal(gb, ob) and np.array_equal(gp, op), len(pu.ref)
    cs = [g.upload(pu) for pu in pus]
    bp = BatchPolisher(g, 4)
    for (bb, pp), (ob, op) in zip(bp.polish(cs, Opts(iter_count=1), want_pos=True), want):
        assert np.array_equal(bb, ob) and np.array_equal(pp, op)
    bp.close()


def test_reads_starting_at_the_last_position_compete_on_absolute_scores():
    """main.rs:1659-1660, 1680: a read-start node's score is the absolute 10 count - 4 coverage, every other node carries the
    path's accumulated total.  Ten one-column reads with a foreign base at L - 1 over three full-length reads: 10 * 10 -
    4 * 14 = 44 loses to the contig's node (total ~ 24 a position) — the consensus is the contig's; over a pileup whose
    best path is negative everywhere the same 44 is the only end node that reaches 0: the consensus is that one base.
    (the kernels compared the 44 with scores relative to the run's left neighbour.)"""
    rng = np.random.default_rng(5)
    L = 200
    ref = "".join(rng.choice(list("ACGT"), L))
    x = "ACGT"[("ACGT".index(ref[-1]) + 1) % 4]
    y = Synth(2000, seed=3).yak(21)
    o = orc.Oracle([y])
    # (a) the contig wins
    alns = [(0, ref, ref)] * 3 + [(L - 1, ref[-1], x)] * 10
    pu = pileup_from_alignments(ref, alns)
    ob, op = o.polish(pu, Opts(iter_count=1))
    assert bytes(ob).decode() == ref
    # (b) the start node wins
    alns2 = [(0, ref, "".join("ACGT"[("ACGT".index(c) + 1 + k) % 4] for c in ref)) for k in range(3)] + [(L - 1, ref[-1], x)] * 10
    pu2 = pileup_from_alignments(ref, alns2)
    ob2, op2 = o.polish(pu2, Opts(iter_count=1))
    assert bytes(ob2).decode() == x and list(op2) == [L - 1]
    for env in ({}, {"NP2_FRONT_UNFUSED": "1"}):
        code = (
            "import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from nextpolish2_amd import Opts, Polisher\nfrom nextpolish2_amd.synth import Synth, pileup_from_alignments\n"
            "ref, x, L = %r, %r, %d\n"
            "g = Polisher([Synth(2000, seed=3).yak(21)])\n"
            "a1 = [(0, ref, ref)] * 3 + [(L - 1, ref[-1], x)] * 10\n"
            "a2 = [(0, ref, ''.join('ACGT'[('ACGT'.index(c) + 1 + k) %% 4] for c in ref)) for k in range(3)] + [(L - 1, ref[-1], x)] * 10\n"
            "b1, p1 = g.polish(pileup_from_alignments(ref, a1), Opts(iter_count=1))\n"
            "b2, p2 = g.polish(pileup_from_alignments(ref, a2), Opts(iter_count=1))\n"
            "print(bytes(b1).decode()); print(bytes(b2).decode(), list(map(int, p2)))\n" % (ROOT, ref, x, L))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        out = r.stdout.strip().splitlines()
        assert out[-2] == ref and out[-1] == "%s [%d]" % (x, L - 1), (env, out[-2:])


def test_repeated_key_in_a_dump_bucket_last_passing_word_wins(tmp_path):
    """kmer.rs:148-167: retrieve_kmers streams the dump and REPLACES a candidate's entry with every word whose count
    passes min_kmer_count, so of several words with one key the last passing one in file order is what get() returns
    (yak writes no such dump; rounds 2-5 refused it).  Boundary form and dump file, against the oracle."""
    from nextpolish2_amd._types import Yak
    base = Synth(3000, seed=5).yak(21)
    rng = np.random.default_rng(11)
    words, offs = [], [0]
    picked = []
    for b in range(1024):
        w = list(base.words[int(base.bucket_off[b]):int(base.bucket_off[b + 1])])
        if w and b % 3 == 0:
            k0 = int(w[0]) >> 10
            # the key of the bucket's first word four more times, counts 7, 2, 30, 1 - interleaved with the other words
            for cnt, at in ((7, 1), (2, len(w) // 2 + 1), (30, len(w) + 1), (1, len(w) + 3)):
                w.insert(min(at, len(w)), np.uint64((k0 << 10) | cnt))
            picked.append((b, k0))
        words += w
        offs.append(len(words))
    dup = Yak(21, np.array(words, np.uint64), np.array(offs, np.uint64))
    hashes = np.array([(k0 << 10) | b for b, k0 in picked] + [int(x) for x in rng.integers(0, 1 << 40, 200)], np.uint64)
    o = orc.Oracle([dup])
    g = Polisher([dup])
    np2io.write_yak(str(tmp_path / "dup.yak"), dup)
    gf = np2io.polisher_from_yak_files([str(tmp_path / "dup.yak")])
    for mk in (1, 2, 3, 8, 31, 40):
        want = o.lookup_hashes(0, hashes, mk)
        assert np.array_equal(g.lookup_hashes(0, hashes, mk), want), mk
        assert np.array_equal(gf.lookup_hashes(0, hashes, mk), want), mk
    # (the first word's own count also competes: with min 1 the last word (count 1) wins, with 8 the 30, with 31 nothing
    #  unless the first word's own count passes)
    assert set(o.lookup_hashes(0, hashes[:len(picked)], 1).tolist()) == {1}
    assert set(o.lookup_hashes(0, hashes[:len(picked)], 8).tolist()) == {30}


def test_soak_contexts_and_batches_share_one_gpu():
    """Four host threads, each with its own context (k_diff_reads spins on status words of other waves of ITS launch:
    interleaved launches of several contexts must not disturb that), plus a batch group, for a few hundred polishes."""
    from nextpolish2_amd import BatchPolisher
    ss = [Synth(60000 + 7000 * i, depth=25, seed=910 + i, diploid=bool(i & 1), read_len_mean=6000.0, read_len_sd=900.0)
          for i in range(4)]
    yaks = [Synth.yak_assembly(ss, 21)]
    pol = Polisher(yaks)
    ref = [pol.polish(s.pileup, Opts())[0].tobytes() for s in ss]
    bad = []

    def worker(w):
        p = pol.clone()
        c = p.upload(ss[w].pileup)
        for _ in range(60):
            if p.polish_resident(c, Opts(), want_pos=False)[0].tobytes() != ref[w]:
                bad.append(w)
        c.free()

    def batch():
        bp = BatchPolisher(pol, 4)
        cs = [pol.upload(s.pileup) for s in ss]
        for _ in range(30):
            out = bp.polish(cs, Opts())
            if [o[0].tobytes() for o in out] != ref:
                bad.append("batch")
        bp.close()
    ths = [threading.Thread(target=worker, args=(w,)) for w in range(4)] + [threading.Thread(target=batch)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not bad


def test_guessed_growth_allowance_and_its_retry(monkeypatch, small_diploid):
    """The final pass sizes its splice rounds with twice the phasing pass's growth bound instead of reading the exact bound
    back (one device round trip less); a round that would outgrow the guess splices nothing, raises GROW_ERR and the pass
    is repeated with the exact bound.  NP2_TEST_GROW_GUESS=0 forces that; NP2_EXACT_GROW=1 is the read-back path."""
    import numpy as np
    from nextpolish2_amd import Opts, Polisher
    from oracle.np2_oracle import Oracle
    s, yaks = small_diploid
    ob, op = Oracle(yaks).polish(s.pileup, Opts())
    gb, gp = Polisher(yaks).polish(s.pileup, Opts())
    assert np.array_equal(gb, ob) and np.array_equal(gp, op)
    monkeypatch.setenv("NP2_TEST_GROW_GUESS", "0")
    gb, gp = Polisher(yaks).polish(s.pileup, Opts())
    assert np.array_equal(gb, ob) and np.array_equal(gp, op)
    from nextpolish2_amd import BatchPolisher
    pol = Polisher(yaks)
    c = pol.upload(s.pileup)
    bp = BatchPolisher(pol, 2)
    for bb, pp in bp.polish([c, c], Opts(), want_pos=True):
        assert np.array_equal(bb, ob) and np.array_equal(pp, op)
    bp.close()


def test_next_pass_started_before_the_vote_is_decided(monkeypatch, small_diploid):
    """polish_impl starts the next pass on the reads the vote kernel flagged while the host still decides the vote; when
    the decision removes other reads as well the pass is started again (NP2_TEST_MISSPECULATE forces that branch);
    NP2_NO_SPECULATE... is read once per process, so the plain order is covered by the traced runs of every parity test
    (tracing switches the early start off).  All three must give the oracle's result — also through the batch driver,
    where the early start is a flush nobody waits for."""
    import numpy as np
    from nextpolish2_amd import BatchPolisher, Opts, Polisher
    from nextpolish2_amd.synth import Synth
    from oracle.np2_oracle import Oracle
    s, yaks = small_diploid
    s2 = Synth(90000, depth=30, seed=23, diploid=True, read_len_mean=9000.0, read_len_sd=1500.0)
    for opts in (Opts(), Opts(iter_count=3)):
        ob, op = Oracle(yaks).polish(s.pileup, opts)
        for env in (None, "1"):
            if env:
                monkeypatch.setenv("NP2_TEST_MISSPECULATE", env)
            else:
                monkeypatch.delenv("NP2_TEST_MISSPECULATE", raising=False)
            pol = Polisher(yaks)
            gb, gp = pol.polish(s.pileup, opts)
            assert np.array_equal(gb, ob) and np.array_equal(gp, op)
            c = pol.upload(s.pileup)
            bp = BatchPolisher(pol, 3)
            for bb, pp in bp.polish([c, c, c], opts, want_pos=True):
                assert np.array_equal(bb, ob) and np.array_equal(pp, op)
            bp.close()
    # contigs of different sizes in one batch: their pipelines reach the early start at different moments
    y2 = [Synth.yak_assembly([s, s2], k) for k in (21, 31)]
    o2 = Oracle(y2)
    exp = [o2.polish(x.pileup, Opts()) for x in (s, s2)]
    pol = Polisher(y2)
    cs = [pol.upload(x.pileup) for x in (s, s2)]
    bp = BatchPolisher(pol, 2)
    for (bb, pp), (ob, op) in zip(bp.polish(cs, Opts(), want_pos=True), exp):
        assert np.array_equal(bb, ob) and np.array_equal(pp, op)
    bp.close()


@pytest.mark.parametrize("env", [dict(NP2_DENSE_PRECOUNT="1"), dict(NP2_TEST_DENSE_LB_FAIL="1")], ids=["precount", "after-a-wait-that-gave-up"])
def test_dense_pass_with_its_chunk_counts_from_a_pre_pass(env):
    """run_diff (csrc/np2_host.cpp): the chunks' column counts from launch_chunk_counts — for a process that shares its GPU
    (NP2_DENSE_PRECOUNT) and as the second attempt after a chunk's wait gave up (test hook: the first attempt is declared
    failed, so cursors, scalars and the status epoch must start over cleanly).  Same results as ever: random contig mixes
    through the batch driver and plain contexts against the oracle, in a process of its own (the switches are read once)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "fuzz_batch.py"), "912", "3"], capture_output=True,
                       timeout=600, cwd=root, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout.decode().strip().splitlines()[-1].startswith("batch cases 3 bad 0"), r.stdout.decode()[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "fuzz_polish.py"), "913", "25"], capture_output=True,
                       timeout=600, cwd=root, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout.decode().strip().splitlines()[-1].startswith("cases 25 bad 0"), r.stdout.decode()[-2000:]


def test_dense_passes_of_two_streams_of_one_process_side_by_side():
    """k_diff_reads' chunks wait for the status words of lower-numbered blocks of the SAME launch (csrc/np2_dense.hip,
    np2_lookback.hpp): safe because the blocks of a launch are dispatched in index order, so whatever a resident block waits
    for is resident or finished — also when a second stream of the same process keeps the device filled with its own
    blocks (queues of one process are not preempted by draining; two PROCESSES on one device take the pre-counted path, test
    above).  Two contexts polish contigs with reads of many chunks at the same time, over and over; every result is the
    oracle's."""
    ss = [Synth(700000, depth=30, seed=951 + i, diploid=bool(i), read_len_mean=40000.0, read_len_sd=6000.0) for i in range(2)]
    yaks = [[s.yak(21)] for s in ss]
    want = [orc.Oracle(y).polish(s.pileup, Opts())[0] for s, y in zip(ss, yaks)]
    bad = []

    def work(i):
        g = Polisher(yaks[i])
        c = g.upload(ss[i].pileup)
        for _ in range(6):
            b, _ = g.polish_resident(c, Opts(), want_pos=False)
            if not np.array_equal(b, want[i]):
                bad.append(i)
        c.free()
        g.close()
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad


def test_candidate_strings_sized_by_a_read_back():
    """extract_candidates (csrc/np2_host.cpp): a chromosome-sized contig sizes its candidate strings by the exact byte count
    read back after the offsets scan instead of by the pileup's column count.  With the threshold lowered to nothing
    (NP2_CAND_EXACT_FROM, read once per process) random contig mixes through the batch driver and plain contexts are the
    oracle's as ever."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NP2_CAND_EXACT_FROM="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "fuzz_batch.py"), "922", "3"], capture_output=True,
                       timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout.decode().strip().splitlines()[-1].startswith("batch cases 3 bad 0"), r.stdout.decode()[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "fuzz_polish.py"), "923", "25"], capture_output=True,
                       timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout.decode().strip().splitlines()[-1].startswith("cases 25 bad 0"), r.stdout.decode()[-2000:]
