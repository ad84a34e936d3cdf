"""Reference-interval sharding of one contig (np2_shard_*, nextpolish2_amd.dist.polish_sharded): the shards' stitched
consensus must be byte-equal to the unsharded result and to the oracle, whatever the number of shards."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.api import ShardRun, shard_plan, vote_decide
from nextpolish2_amd.dist import ShardMismatch, polish_sharded_local, stitch_shards
from nextpolish2_amd.synth import Synth, concat_pileups
from oracle import np2_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n_shards,halo", [(2, 65536), (3, 20000), (5, 12000)])
def test_sharded_diploid_contig_equals_whole_and_oracle(n_shards, halo):
    s = Synth(500000, seed=810 + n_shards, diploid=True, read_len_mean=9000.0, read_len_sd=1500.0)
    yaks = [s.yak(21), s.yak(31)]
    pol = Polisher(yaks)
    b0, p0 = pol.polish(s.pileup, Opts())
    b1, p1 = polish_sharded_local(pol, s.pileup, Opts(), n_shards=n_shards, halo=halo)
    assert np.array_equal(b0, b1) and np.array_equal(p0, p1)
    ob, op = orc.Oracle(yaks).polish(s.pileup, Opts())
    assert np.array_equal(ob, b1) and np.array_equal(op, p1)


@pytest.mark.parametrize("opts", [Opts(iter_count=3), Opts(iter_count=1), Opts(model="len"), Opts(use_all_reads=True)])
def test_sharded_option_sets(opts):
    s = Synth(300000, seed=821, diploid=True, read_len_mean=8000.0, read_len_sd=1200.0)
    yaks = [s.yak(21)]
    pol = Polisher(yaks)
    b0, p0 = pol.polish(s.pileup, opts)
    b1, p1 = polish_sharded_local(pol, s.pileup, opts, n_shards=3, halo=24000)
    assert np.array_equal(b0, b1) and np.array_equal(p0, p1)


def test_votes_of_the_shards_merge_to_the_votes_of_the_contig():
    # the phasing decision on the merged shard votes removes exactly the reads the unsharded pass removes
    s = Synth(400000, seed=831, diploid=True)
    yaks = [s.yak(21)]
    pol = Polisher(yaks)
    pol.set_trace(True)
    pol.polish(s.pileup, Opts())
    whole = pol.trace(0, "invalid_ids")
    pol.set_trace(False)
    plans = shard_plan(s.pileup, 3, 30000)
    ctxs = [pol.clone() for _ in plans]
    runs = [ShardRun(c, s.pileup, pl, Opts()) for c, pl in zip(ctxs, plans)]
    votes = [r.vote() for r in runs]
    losers = vote_decide(votes, s.pileup.n_reads, Opts())
    assert np.array_equal(losers, whole) and len(whole) > 100
    # every HETE region was owned by exactly one shard: pair keys of different shards may coincide (a read pair sharing
    # regions on both sides of a cut), region votes may not be duplicated
    n_pairs = sum(len(v.pair_key) for v in votes)
    assert n_pairs >= len(np.unique(np.concatenate([v.pair_key for v in votes])))
    for r in runs:
        r.close()


def test_haploid_contig_without_votes_and_uneven_joint():
    # pieces laid end to end: coverage drops to the contig's own base at the joints (cov < 2 resets, no spanning read)
    parts = [Synth(120000, seed=841 + i, read_len_mean=7000.0, read_len_sd=1000.0) for i in range(3)]
    pu = concat_pileups([p.pileup for p in parts])
    yaks = [Synth.yak_assembly(parts, 21)]
    pol = Polisher(yaks)
    b0, p0 = pol.polish(pu, Opts())
    assert b0.tobytes() == b"".join(p.hap1 for p in parts)
    for n in (2, 4):
        b1, p1 = polish_sharded_local(pol, pu, Opts(), n_shards=n, halo=20000)
        assert np.array_equal(b0, b1) and np.array_equal(p0, p1)


def test_stitcher_rejects_disagreeing_neighbours():
    s = Synth(200000, seed=851, diploid=True)
    pol = Polisher([s.yak(21)])
    plans = shard_plan(s.pileup, 2, 30000)
    runs = [ShardRun(pol.clone(), s.pileup, pl, Opts(iter_count=1)) for pl in plans]
    pieces = [r.final() for r in runs]
    b, p = stitch_shards(pieces, plans, 1024)
    b0, p0 = pol.polish(s.pileup, Opts(iter_count=1))
    assert np.array_equal(b, b0) and np.array_equal(p, p0)
    bad = np.array(pieces[1][0]).copy()
    bad[5] = ord("A") if bad[5] != ord("A") else ord("C")
    with pytest.raises(ShardMismatch):
        stitch_shards([pieces[0], (bad, pieces[1][1])], plans, 1024)


def test_two_ranks_polish_the_halves_of_one_contig():
    """world_size 2 (gloo; both ranks on this box's one GPU): each rank uploads and polishes its half, votes are
    all-gathered and decided on every rank, the pieces all-gathered and stitched == one-rank result == oracle."""
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "tools", "shard_ranks.py")],
                       capture_output=True, cwd=ROOT, timeout=900, env=dict(os.environ, NP2_SHARD_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out == {"world": 2, "equal_single": True, "equal_oracle": True, "ranks_agree": True}


def test_cli_under_torchrun_shards_long_contigs_and_spreads_the_others(tmp_path):
    """nextPolish2 with two ranks (gloo, both on this box's GPU): the long contig is cut into two reference intervals, the
    others go whole to one rank each, rank 0 writes one FASTA in input order == the one-process CLI's output."""
    import gzip
    from nextpolish2_amd import io as np2io
    from nextpolish2_amd.bamio import pileup_to_records, write_bam
    ss = [Synth(260000, depth=20, seed=891, diploid=True, read_len_mean=7000.0, read_len_sd=900.0, name="long1"),
          Synth(50000, depth=20, seed=892, read_len_mean=5000.0, read_len_sd=700.0, name="ctgB"),
          Synth(40000, depth=20, seed=893, diploid=True, read_len_mean=5000.0, read_len_sd=700.0, name="ctgC")]
    recs = []
    for tid, s in enumerate(ss):
        recs += pileup_to_records(s.pileup, tid=tid, rng=np.random.default_rng(tid), decorate=True)
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    write_bam(str(tmp_path / "m.bam"), [(s.pileup.name, s.pileup.L) for s in ss] + [("tiny", 500)], recs)
    with gzip.open(tmp_path / "g.fa.gz", "wt") as f:
        for s in ss:
            f.write(f">{s.pileup.name}\n{s.pileup.ref.tobytes().decode()}\n")
        f.write(">tiny\nACGTACGTNNacgt\n")
    for k in (21, 31):
        np2io.write_yak(str(tmp_path / f"k{k}.yak"), Synth.yak_assembly(ss, k))
    args = ["-L", "10000", str(tmp_path / "m.bam"), str(tmp_path / "g.fa.gz"), str(tmp_path / "k21.yak"), str(tmp_path / "k31.yak")]
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", "-o", str(tmp_path / "one.fa")] + args,
                       capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "nextpolish2_amd.cli",
                        "--dist_backend", "gloo", "--device", "0", "--shard_min_len", "200000", "--shard_halo", "30000",
                        "-o", str(tmp_path / "two.fa")] + args, capture_output=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    one, two = (tmp_path / "one.fa").read_bytes(), (tmp_path / "two.fa").read_bytes()
    assert one == two and one.count(b">") == 4


def test_full_size_chr1_single_gpu_and_two_intervals():
    """BASELINE.json configs[3] workload (human chr1-sized contig: 248 Mb, 30x HiFi, k21 + k31) on ONE GPU, through
    size-independent properties: the polished sequence equals the simulated truth, positions never decrease, a second
    call is identical, and the contig cut into two reference intervals (the 2-GPU layout, run here one after the other)
    stitches to the byte-identical result.  The contig is generated as 16 pieces laid end to end (host threads)."""
    from concurrent.futures import ThreadPoolExecutor
    n_parts, L = 16, 248_000_000
    with ThreadPoolExecutor(n_parts) as ex:
        parts = list(ex.map(lambda i: Synth(L // n_parts, depth=30, seed=500 + i), range(n_parts)))
    pu = concat_pileups([p.pileup for p in parts], "chr1")
    from nextpolish2_amd._cpus import usable_cpus
    yaks = [Synth.yak_assembly(parts, k, threads=usable_cpus()) for k in (21, 31)]
    truth = b"".join(p.hap1 for p in parts)
    assert pu.ref.tobytes() != truth and pu.L > 247_000_000
    pol = Polisher(yaks)
    c = pol.upload(pu)
    b1, p1 = pol.polish_resident(c, Opts())
    assert b1.tobytes() == truth
    assert np.all(p1[1:] >= p1[:-1]) and int(p1[0]) == 0 and int(p1[-1]) == pu.L - 1
    b2, _ = pol.polish_resident(c, Opts(), want_pos=False)
    assert np.array_equal(b1, b2)
    c.free()
    b3, p3 = polish_sharded_local(pol, pu, Opts(), n_shards=2)
    assert np.array_equal(b1, b3) and np.array_equal(p1, p3)


def _edits(got, want, look=48, reach=16, max_edits=1000):
    """Number of local differences (any mix of substituted, inserted and deleted bases within `reach` positions) between
    `got` and `want`, found by walking both from the left and re-synchronising after every difference on `look` equal
    bases; -1 if they fall out of step or differ in more than max_edits places."""
    i = j = n_edits = 0
    block = 1 << 20
    cands = sorted(((di, dj) for di in range(reach + 1) for dj in range(reach + 1) if di + dj), key=lambda x: (x[0] + x[1], abs(x[0] - x[1])))
    while True:
        k = None
        while k is None:  # first difference at or after (i, j), a block at a time
            n = min(len(got) - i, len(want) - j, block)
            if n <= 0:
                return n_edits + (1 if (len(got) - i) != (len(want) - j) else 0)
            d = np.flatnonzero(got[i:i + n] != want[j:j + n])
            if len(d):
                k = int(d[0])
            else:
                i, j = i + n, j + n
        for di, dj in cands:
            a, b = got[i + k + di:i + k + di + look], want[j + k + dj:j + k + dj + look]
            m = min(len(a), len(b))
            if m == 0 or np.array_equal(a[:m], b[:m]):
                i, j, n_edits = i + k + di, j + k + dj, n_edits + 1
                break
        else:
            return -1
        if n_edits > max_edits:
            return -1


def test_full_size_chr1_diploid_whole_two_and_four_intervals(capsys):
    """BASELINE.json configs[3] with what the config name says — injected SNVs / indels — as a DIPLOID 248 Mb contig
    (15x + 15x reads of two haplotypes, SNP 0.5 %, indel 0.2 %): ~570 k reads, a heterozygous region every ~140 bp, the
    phasing vote with its host-side Louvain over ~280 k reads (main.rs:948-1015, louvain.rs:72-195, 290-356) at
    chromosome scale.  Whole contig == cut into 2 == cut into 4 reference intervals (the 2- and 4-GPU layouts, run here
    one after the other), byte for byte; a second call is identical; the polished sequence is the contig's own haplotype
    (hap1: the assembly is hap1 + errors and -m ref keeps the community that agrees with it) wherever that is
    unambiguous — checked piece by piece on the 16 generated pieces through the position array; the host side of the
    vote is logged."""
    from concurrent.futures import ThreadPoolExecutor
    from nextpolish2_amd._cpus import usable_cpus
    from nextpolish2_amd.api import ShardRun, shard_plan
    from nextpolish2_amd.dist import _run_local
    n_parts, L = 16, 248_000_000
    with ThreadPoolExecutor(n_parts) as ex:
        parts = list(ex.map(lambda i: Synth(L // n_parts, depth=30, seed=500 + i, diploid=True), range(n_parts)))
    pu = concat_pileups([p.pileup for p in parts], "chr1")
    yaks = [Synth.yak_assembly(parts, k, threads=usable_cpus()) for k in (21, 31)]
    haps = [p.hap1 for p in parts]
    part_len = [p.pileup.L for p in parts]
    del parts
    assert pu.L > 247_000_000 and pu.n_reads > 500_000
    pol = Polisher(yaks)
    c = pol.upload(pu)
    pol.set_timing(True)
    b1, p1 = pol.polish_resident(c, Opts())
    tm = pol.timings()
    pol.set_timing(False)
    assert np.all(p1[1:] >= p1[:-1]) and int(p1[0]) == 0 and int(p1[-1]) == pu.L - 1
    b2, _ = pol.polish_resident(c, Opts(), want_pos=False)
    assert np.array_equal(b1, b2)
    c.free()
    # the contig's own haplotype, piece by piece
    cuts = np.searchsorted(p1, np.cumsum([0] + part_len))
    edits = [_edits(b1[cuts[k]:cuts[k + 1]], np.frombuffer(h, dtype=np.uint8)) for k, h in enumerate(haps)]
    msg = (f"diploid 248 Mb contig: {pu.n_reads} reads; host vote (wall_louvain) {tm.get('wall_louvain', 0.0):.1f} ms, wall_vote "
           f"{tm.get('wall_vote', 0.0):.1f} ms; edits between the polished piece and hap1, per generated piece: {edits}")
    with capsys.disabled():
        print("\n  " + msg)
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        open(os.path.join(ROOT, "gpurun_out", "chr1_diploid_test.log"), "w").write(msg + "\n")
    # (a heterozygous site where the vote kept the other haplotype's reads is polished to hap2's allele)
    assert abs(len(b1) - sum(len(h) for h in haps)) <= 64 and all(0 <= e <= 64 for e in edits), msg
    whole = b1.tobytes()
    for ns in (2, 4):
        plans = shard_plan(pu, ns, 65536)
        ctxs = [pol.clone() for _ in range(ns)]
        runs = [ShardRun(ctxs[k], pu, plans[k], Opts(), 1024) for k in range(ns)]
        try:
            bs, ps = _run_local(runs, plans, pu.n_reads, Opts(), True)
        finally:
            for r in runs:
                r.close()
        assert bs.tobytes() == whole and np.array_equal(ps, p1), f"{ns} intervals differ from the whole contig"
        del runs, ctxs


@pytest.mark.parametrize("inflate", ["libdeflate", "gpu"])
def test_shards_read_straight_from_the_bam_number_their_reads_contig_wide(tmp_path, monkeypatch, capfd, inflate):
    """np2_shard_bam_*: every shard parses only the records overlapping its interval +- halo; the exchanged file offsets
    give every pushed record its contig-wide number; stitched result == whole-contig BAM path == oracle front end + polish.
    inflate = gpu: the interval's records extracted on the device (blocks from the linear index's entry for the zone's start
    on, the walk ending at the first record behind the zone, virtual offsets from the records' stream offsets)."""
    monkeypatch.setenv("NP2_INFLATE", inflate)
    monkeypatch.setenv("NP2_IO_PROFILE", "1")
    import gzip  # noqa: F401
    from nextpolish2_amd import io as np2io
    from nextpolish2_amd.bamio import pileup_to_records, records_to_arrays, write_bam
    from nextpolish2_amd.dist import polish_sharded_bam_local
    s = Synth(420000, depth=25, seed=895, diploid=True, read_len_mean=8000.0, read_len_sd=1200.0, name="long1")
    other = Synth(30000, depth=10, seed=896, name="other")
    recs = pileup_to_records(other.pileup, tid=0, rng=np.random.default_rng(5), decorate=True) + \
        pileup_to_records(s.pileup, tid=1, rng=np.random.default_rng(6), decorate=True)  # clips, secondaries, low mapq ...
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    bam_path = str(tmp_path / "m.bam")
    write_bam(bam_path, [("other", other.pileup.L), ("long1", s.pileup.L)], recs)
    yaks = [s.yak(21), s.yak(31)]
    pol = Polisher(yaks)
    ref = s.pileup.ref.tobytes()
    bam = np2io.Bam(bam_path)
    whole = np2io.contig_from_bam(pol, bam, "long1", ref)
    b0, p0 = pol.polish_resident(whole, Opts())
    whole_pu = np2io.export_contig(pol, whole, ref)
    whole.free()
    for n_shards, halo in ((2, 40000), (3, 25000)):
        # the shards' read lists, renumbered, are slices of the whole contig's read list
        cuts = np2io.shard_cuts(len(ref), n_shards)
        ctxs = [pol.clone() for _ in cuts]
        sbs = [np2io.ShardFromBam(ctxs[k], np2io.Bam(bam_path), "long1", ref, lo, hi, halo) for k, (lo, hi) in enumerate(cuts)]
        all_off = np.concatenate([sb.own_offsets for sb in sbs])
        assert len(all_off) == whole_pu.n_reads - 1 and np.all(np.diff(all_off.astype(np.int64)) > 0)
        for k, sb in enumerate(sbs):
            h, plan, n_total = sb.finish(all_off)
            assert n_total == whole_pu.n_reads and (plan.own_lo, plan.own_hi) == cuts[k]
            from nextpolish2_amd.io import _resident
            rc = _resident(ctxs[k], h, "long1", plan.sub_hi - plan.sub_lo)  # (owns the shard contig; freed below)
            sh = np2io.export_contig(ctxs[k], rc, ref[plan.sub_lo:plan.sub_hi])
            assert sh.n_reads == 1 + plan.read_hi - plan.read_lo
            for i in range(1, sh.n_reads):
                g = whole_pu.reads[plan.read_lo + i - 1]
                if sh.reads["flags"][i] & 1:
                    continue  # a hole / a read outside the zone / one the clip filter emptied (checked below)
                assert sh.reads["aln_t_s"][i] + plan.sub_lo == g["aln_t_s"] and sh.reads["aln_t_e"][i] + plan.sub_lo == g["aln_t_e"]
                assert sh.reads["n_cols"][i] == g["n_cols"] and not (g["flags"] & 1)
            held = [i for i in range(1, sh.n_reads) if not sh.reads["flags"][i] & 1]
            must = [r for r in range(plan.read_lo, plan.read_hi) if not whole_pu.reads["flags"][r] & 1
                    and whole_pu.reads["aln_t_e"][r] >= plan.zone_lo and whole_pu.reads["aln_t_s"][r] < plan.zone_hi]
            assert [plan.read_lo + i - 1 for i in held] == must
            rc.free()
        b1, p1 = polish_sharded_bam_local(pol, bam_path, "long1", ref, n_shards, Opts(), halo=halo)
        assert np.array_equal(b0, b1) and np.array_equal(p0, p1)
    arr, cig, seq4, asc, asc_off = records_to_arrays([r for r in recs if r["tid"] == 1])
    pu = orc.front_end(ref, arr, cig, asc, asc_off, np2io.FrontOpts())
    ob, op = orc.Oracle(yaks).polish(pu, Opts())
    assert np.array_equal(ob, b0) and np.array_equal(op, p0)
    err = capfd.readouterr().err
    # (the whole contig once; then 2 and 3 shards, each read once by ShardFromBam and once by polish_sharded_bam_local)
    assert (err.count("fetch_records_gpu:") >= 11) == (inflate == "gpu"), err[-1500:]
