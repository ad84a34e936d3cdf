"""Host side of the phasing vote on a vote recorded from a GPU run (tools/vote_dump.py: 16 Mb diploid contig, 36.9 k reads,
1.08 M read pairs).  The expected losers are the ORACLE's: tests/golden/make_vote_fixture.py regenerates the same contig on
the CPU, runs the oracle's whole phasing pass (mark_hete_lqseqs, phase_reads_by_lqseqs, Louvain) and stores the reads it
removes.  The product's host code — adjacency rows and aggregation sums on several threads for a graph of this size —
must decide the recorded vote the same way, on one thread, on many, and with the vote cut into two shards."""
import os
import subprocess
import sys

import numpy as np

from nextpolish2_amd.api import Vote, vote_decide

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def load():
    z = np.load(os.path.join(HERE, "golden", "votes", "vote_16Mb_diploid.npz"))
    return Vote.unpack(z["packed"]), int(z["n_reads"][0]), z["losers"]


def test_recorded_vote_is_decided_the_same_on_several_threads():
    v, n_reads, want = load()
    assert len(v.pair_key) >= (1 << 18)  # large enough for the threaded paths
    got = vote_decide([v], n_reads)
    assert np.array_equal(got, want)
    # cut in two "shards" (pairs split in the middle, per-read records of both halves): the merge path
    h = len(v.pair_key) // 2
    a = Vote(pair_key=v.pair_key[:h], pair_cnt=v.pair_cnt[:h], read_id=v.read_id, first_pos=v.first_pos, ref_w=v.ref_w, flags=v.flags)
    b = Vote(pair_key=v.pair_key[h:], pair_cnt=v.pair_cnt[h:])
    assert np.array_equal(vote_decide([a, b], n_reads), want)


def test_one_thread_and_many_agree():
    code = ("import sys, zlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_vote_host_cpu import load\n"
            "from nextpolish2_amd.api import vote_decide\n"
            "v, n, want = load()\n"
            "print(zlib.crc32(vote_decide([v], n).tobytes()))\n" % (ROOT, HERE))
    outs = []
    # (NP2_VOTE_COMPACT: the vote in the 4-bytes-per-pair row form the plain pipeline reads back from the vote kernels)
    for t, extra in (("1", {}), ("8", {}), ("1", {"NP2_VOTE_COMPACT": "1"}), ("8", {"NP2_VOTE_COMPACT": "1"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, env=dict(os.environ, NP2_VOTE_THREADS=t, **extra), timeout=600)
        assert r.returncode == 0, r.stderr.decode()
        outs.append(int(r.stdout.strip()))
    import zlib
    assert set(outs) == {zlib.crc32(load()[2].tobytes())}  # == the oracle's decision, whatever the thread count and the form


def test_sweeps_decided_ahead_on_all_threads_change_nothing():
    """SignedLouvain::local_moving (csrc/np2_phase_host.hpp, louvain.rs:72-117): from the second sweep on a large graph's
    decisions are computed ahead on all threads and taken unless a neighbour moved earlier in the same sweep.  The recorded
    vote tiled three times along a contig (110 k reads: large enough for that path) and random signed graphs with the
    threshold lowered to nothing must be decided exactly as with the plain serial sweeps (NP2_VOTE_NO_AHEAD)."""
    code = ("import sys, zlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from test_vote_host_cpu import load\n"
            "from test_shard_cpu import _random_votes, _vote\n"
            "from nextpolish2_amd.api import Vote, vote_decide\n"
            "from nextpolish2_amd import Opts\n"
            "v0, R0, _ = load()\n"
            "span = int(v0.first_pos.max()) + 100000\n"
            "ks, cs, ids, fp, rw, fl = [], [], [], [], [], []\n"
            "for c in range(3):\n"
            "    sh = np.uint64(c * (R0 - 1))\n"
            "    ks.append(v0.pair_key + ((sh << np.uint64(32)) | sh)); cs.append(v0.pair_cnt)\n"
            "    ids.append(v0.read_id + np.uint32(c * (R0 - 1))); fp.append(v0.first_pos + np.uint32(c * span))\n"
            "    rw.append(v0.ref_w); fl.append(v0.flags)\n"
            "v = Vote(pair_key=np.concatenate(ks), pair_cnt=np.concatenate(cs), read_id=np.concatenate(ids),\n"
            "         first_pos=np.concatenate(fp), ref_w=np.concatenate(rw), flags=np.concatenate(fl))\n"
            "crc = zlib.crc32(vote_decide([v], 1 + 3 * (R0 - 1)).tobytes())\n"
            "rng = np.random.default_rng(5)\n"
            "for t in range(60):\n"
            "    n_reads = int(rng.integers(12, 60))\n"
            "    m = min(n_reads - 1, 40)\n"
            "    reads, pairs, first, refw, bad = _random_votes(rng, n_reads, int(rng.integers(5, m * (m - 1) // 4)))\n"
            "    for use_all in (False, True):\n"
            "        got = vote_decide([_vote(reads.tolist(), pairs, first, refw, bad, set(refw))], n_reads, Opts(use_all_reads=use_all))\n"
            "        crc = zlib.crc32(got.tobytes(), crc)\n"
            "print(crc)\n" % (ROOT, HERE))
    outs = []
    for extra in ({"NP2_VOTE_NO_AHEAD": "1"}, {"NP2_VOTE_AHEAD_MIN": "1"}, {"NP2_VOTE_AHEAD_MIN": "1", "NP2_VOTE_THREADS": "3"}, {}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, env=dict(os.environ, **({"NP2_VOTE_THREADS": "8"} | extra)), timeout=900)
        assert r.returncode == 0, r.stderr.decode()
        outs.append(int(r.stdout.strip()))
    assert len(set(outs)) == 1, outs


def test_components_swept_on_all_threads_equal_the_oracle():
    """SignedLouvain::local_moving, round 6: the first stage decomposes exactly over connected components (louvain.rs:84-96:
    a node's gain is the sum of its own edges into a neighbour's community; :103-105: a move touches two member sets), so
    the pieces of the read graph between cut points are swept to convergence on separate threads and their member-set
    histories concatenated.  With the thresholds lowered to nothing (NP2_VOTE_PIECES_MIN=1) random signed graphs made of
    many small components along the id axis — with ties, conflicts and negative communities that get declustered, whose
    member-set iteration order is the one thing a wrong history would change — must be decided exactly like the ORACLE's
    literal restatement decides them; and the recorded 16 Mb vote tiled eight times (eight components at least) like the
    serial sweeps (NP2_VOTE_NO_PIECES) decide it."""
    code = ("import sys, zlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from nextpolish2_amd.api import phase_vote, Np2Error\n"
            "from oracle import np2_oracle as orc\n"
            "rng = np.random.default_rng(int(sys.argv[1]))\n"
            "n_checked = 0\n"
            "for trial in range(150):\n"
            "    n_comp = int(rng.integers(2, 14))\n"
            "    pairs, base = {}, 1\n"
            "    for c in range(n_comp):\n"
            "        n = int(rng.integers(2, 18))\n"
            "        ids = np.sort(rng.choice(np.arange(base, base + 3 * n), size=n, replace=False))\n"
            "        base += 3 * n + int(rng.integers(0, 3))  # (sometimes the next component starts right behind: no gap in the ids)\n"
            "        for _ in range(int(rng.integers(n - 1, 3 * n))):\n"
            "            a, b = rng.choice(ids, size=2, replace=False)\n"
            "            a, b = int(min(a, b)), int(max(a, b))\n"
            "            same = (a %% 2) == (b %% 2)\n"
            "            w = float(rng.integers(1, 3)) * (1.0 if (same or rng.random() < 0.15) else -1.0)\n"
            "            if rng.random() < 0.08: w = -3.0\n"
            "            pairs[(a, b)] = pairs.get((a, b), 0.0) + w\n"
            "    plist = sorted((a, b, w) for (a, b), w in pairs.items()) if trial %% 2 else [(a, b, w) for (a, b), w in pairs.items()]\n"
            "    edges, keys, seen = [], [], set()\n"
            "    for a, b, w in plist:\n"
            "        edges.append((a, b, w)); edges.append((b, a, w))\n"
            "        for k in (a, b):\n"
            "            if k not in seen: seen.add(k); keys.append(k)\n"
            "    ref = None\n"
            "    if trial %% 3 == 0:\n"
            "        ref = {int(k): float(rng.choice([-1.0, 1.0, 2.0])) for k in rng.choice(keys, size=max(1, len(keys) // 3), replace=False)}\n"
            "    try:\n"
            "        exp = orc.phase_communities(edges, ref)\n"
            "    except orc.RefPanic:\n"
            "        try:\n"
            "            phase_vote(keys, plist, ref); raise SystemExit('product accepted what the reference panics on')\n"
            "        except Np2Error: continue\n"
            "    got = phase_vote(keys, plist, ref)\n"
            "    assert got == exp, (trial, got, exp)\n"
            "    n_checked += 1\n"
            "print(n_checked)\n" % (ROOT, HERE))
    for seed, threads in ((1, "4"), (2, "2"), (3, "7")):
        r = subprocess.run([sys.executable, "-c", code, str(seed)], capture_output=True, timeout=900,
                           env=dict(os.environ, NP2_VOTE_PIECES_MIN="1", NP2_VOTE_THREADS=threads))
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        assert int(r.stdout.strip()) > 100
    # the recorded vote, tiled: pieces on all threads == one piece
    code2 = ("import sys, zlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
             "import numpy as np\n"
             "from test_vote_host_cpu import load\n"
             "from nextpolish2_amd.api import Vote, vote_decide\n"
             "v0, R0, _ = load()\n"
             "span = int(v0.first_pos.max()) + 100000\n"
             "ks, cs, ids, fp, rw, fl = [], [], [], [], [], []\n"
             "for c in range(8):\n"
             "    sh = np.uint64(c * (R0 - 1))\n"
             "    ks.append(v0.pair_key + ((sh << np.uint64(32)) | sh)); cs.append(v0.pair_cnt)\n"
             "    ids.append(v0.read_id + np.uint32(c * (R0 - 1))); fp.append(v0.first_pos + np.uint32(c * span))\n"
             "    rw.append(v0.ref_w); fl.append(v0.flags)\n"
             "v = Vote(pair_key=np.concatenate(ks), pair_cnt=np.concatenate(cs), read_id=np.concatenate(ids),\n"
             "         first_pos=np.concatenate(fp), ref_w=np.concatenate(rw), flags=np.concatenate(fl))\n"
             "print(zlib.crc32(vote_decide([v], 1 + 8 * (R0 - 1)).tobytes()))\n" % (ROOT, HERE))
    outs = []
    for extra in ({"NP2_VOTE_NO_PIECES": "1"}, {}, {"NP2_VOTE_THREADS": "3"}):
        r = subprocess.run([sys.executable, "-c", code2], capture_output=True, env=dict(os.environ, **({"NP2_VOTE_THREADS": "8"} | extra)), timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        outs.append(int(r.stdout.strip()))
    assert len(set(outs)) == 1, outs
