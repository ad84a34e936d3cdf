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
