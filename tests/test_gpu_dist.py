"""GPU tests of the multi-GPU plumbing that can run on one device: the device-resident result buffer handed to the
gatherer, and bench.py under torch.distributed.run with a single rank (RCCL process group, same code path as N > 1)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.dist import SequenceGatherer
from nextpolish2_amd.synth import Synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gather_from_the_device_result_buffer():
    import torch
    s = Synth(80000, depth=20, seed=71, read_len_mean=6000.0, read_len_sd=900.0)
    pol = Polisher([s.yak(21)])
    c = pol.upload(s.pileup)
    g = SequenceGatherer(s.pileup.L + 8192, torch.device("cuda", 0))
    for _ in range(3):  # back to back: the gather of one step overlaps the next polish
        bases, span = pol.polish_resident(c, Opts(), want_pos=False)
        ptr, n = pol.last_result_device()
        assert n == bases.shape[0]
        g.gather_device(ptr, n)
    torch.cuda.synchronize()
    assert g.to_host() == {0: bases.tobytes()}
    with pytest.raises(ValueError):
        SequenceGatherer(100, torch.device("cuda", 0)).gather_device(ptr, n)


def test_bench_under_torchrun_with_one_rank():
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "4", "--warmup", "1", "--scale", "0.05", "--cpu-threads", "8"],
                       capture_output=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["fasta_identical_to_oracle"] is True
    assert out["roofline"]["frac"] > 0 and out["cpu_baseline"]["value"] > 0
    assert out["config"]["contigs"] == 17 and out["oracle_checked_contigs"] == 17
    assert out["end_to_end"]["value"] > 0 and out["end_to_end"]["identical_to_resident_path"] is True


def _bench_line(args, env_extra=None, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, env=env, cwd=ROOT, timeout=timeout)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])


def test_bench_gpus_2_rehearsed_with_gloo_on_one_gpu():
    """`bench.py --gpus N` spawns its own ranks (spawn_ranks) and runs the N-rank branches of both scaling modes.  A box
    with one GPU cannot run two RCCL ranks, so NP2_BENCH_BACKEND=gloo lets both ranks share device 0 with the collectives
    going through host memory — the code paths (shard protocol over two ranks, staged assembly all-gather, max-over-ranks
    timing, the line with n_gpus == 2) are the ones the driver's SCALE run takes; results must equal the one-rank run."""
    gloo = {"NP2_BENCH_BACKEND": "gloo"}
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end"]
    strong = ["--scaling", "strong", "--contig-mb", "2", "--haploid"] + common
    one = _bench_line(["--gpus", "1"] + strong)
    two = _bench_line(["--gpus", "2"] + strong, gloo)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["polished_equals_truth"] and two["polished_equals_truth"]
    assert one["output_crc32"] == two["output_crc32"] and one["span"] == two["span"]
    # the default strong workload is DIPLOID (configs[3]: "chr1 with injected SNV / indel"): the vote is gathered, decided on
    # rank 0 and its decision broadcast
    dip = ["--scaling", "strong", "--contig-mb", "2"] + common
    one_d = _bench_line(["--gpus", "1"] + dip)
    two_d = _bench_line(["--gpus", "2"] + dip, gloo)
    assert one_d["config"]["diploid"] and one_d["output_crc32"] == two_d["output_crc32"] and one_d["span"] == two_d["span"]
    # (polished back to haplotype 1 like the haploid contig of the same seeds: the same bytes — what differs is the way there)
    weak = ["--scale", "0.05", "--strong-mb", "2", "--repeats", "2"] + common
    one = _bench_line(["--gpus", "1"] + weak)
    two = _bench_line(["--gpus", "2"] + weak, gloo)  # (asserts inside: the all-gathered bytes are rank 0's polished assembly)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert one["output_crc32"] == two["output_crc32"]
    assert two["config"]["assembly_bp"] == one["config"]["assembly_bp"] and two["value"] > 0
    assert one["ms_per_step_regions"]["n"] == 2 and "strong" not in one
    # ... and what a SCALE run of the driver's command line gets beside the weak figure: the same diploid contig over the N ranks
    st = two["strong"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 2 and st["output_crc32"] == one_d["output_crc32"] and st["span"] == one_d["span"]
    # one E. coli-sized contig per rank: the single-contig branch (deferred fetch + gather from the device result buffer)
    two = _bench_line(["--gpus", "2", "--workload", "ecoli", "--scale", "0.1"] + common, gloo)
    assert two["n_gpus"] == 2 and two["polished_equals_truth_contigs"] == 1


def test_deferred_output_fetch_overlaps_the_next_contig():
    """np2_result_fetch_begin / _end: the host copy of contig i is started after contig i and collected after contig
    i + 1; results equal the synchronous path."""
    sa = Synth(60000, depth=20, seed=72, read_len_mean=6000.0, read_len_sd=900.0)
    sb = Synth(90000, depth=20, seed=73, diploid=True, read_len_mean=6000.0, read_len_sd=900.0)
    ya, yb = sa.yak(21), sb.yak(21)
    pol = Polisher([ya])
    polb = Polisher([yb])
    ca, cb = pol.upload(sa.pileup), polb.upload(sb.pileup)
    exp_a, span_a = pol.polish_resident(ca, Opts(), want_pos=False)
    exp_a = exp_a.copy()
    # same context, back to back: A deferred, A again (different options) deferred, collect in order
    none, span = pol.polish_resident(ca, Opts(), want_pos=False, defer_output=True)
    assert none is None and span == span_a
    pol.fetch_begin()
    with pytest.raises(Exception):
        pol.fetch_begin()  # one fetch in flight per context
    exp2, span2 = Polisher([ya]).polish(sa.pileup, Opts(iter_count=1))
    _, s2 = pol.polish_resident(ca, Opts(iter_count=1), want_pos=False, defer_output=True)
    got1 = pol.fetch_end().copy()
    pol.fetch_begin()
    got2 = pol.fetch_end().copy()
    assert np.array_equal(got1, exp_a) and np.array_equal(got2, exp2) and s2 == (int(span2[0]), int(span2[-1]))
    with pytest.raises(Exception):
        pol.fetch_end()  # nothing in flight
    # and on another context with another contig
    eb, _ = polb.polish_resident(cb, Opts(), want_pos=False)
    polb.polish_resident(cb, Opts(), want_pos=False, defer_output=True)
    polb.fetch_begin()
    assert np.array_equal(polb.fetch_end(), eb)
