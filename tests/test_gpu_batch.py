"""Batch driver (np2_batch_*, csrc/np2_batch.cpp): several contigs polished with one launch per pipeline step.

Every contig keeps its own pipeline; only the launches are merged — so the results must be exactly those of
np2_polish_resident contig by contig, and those of the oracle."""
import threading

import numpy as np
import pytest

from nextpolish2_amd import BatchPolisher, Opts, Polisher
from nextpolish2_amd.synth import Synth
from oracle import np2_oracle as orc

pytestmark = pytest.mark.gpu


def _assembly():
    # deliberately uneven: diploid contigs of different sizes, one with very short reads (other kernels / branches),
    # one tiny one, so that the recorded command streams of the slots diverge
    specs = [dict(L=90000, seed=301), dict(L=30000, seed=302, read_len_mean=4000.0, read_len_sd=700.0),
             dict(L=150000, seed=303), dict(L=20000, seed=304, depth=12), dict(L=60000, seed=305, read_err_rate=0.01)]
    syn = [Synth(sp.pop("L"), diploid=True, **sp) for sp in specs]
    return syn, [Synth.yak_assembly(syn, 21), Synth.yak_assembly(syn, 31)]


def test_sinks_receive_the_polished_bases_on_the_device():
    """np2_batch_set_sink: a copy of every contig's polished bases lands, device to device and at its polished length, in the
    caller's buffer (a rank's all-gather staging buffer) by the time np2_batch_polish returns — with the bases delivered to
    the host as well, and with the sequences kept on the device; a sink taken away again receives nothing."""
    import torch
    syn, yaks = _assembly()
    pol = Polisher(yaks)
    contigs = [pol.upload(s.pileup) for s in syn]
    want = [pol.polish_resident(c, Opts())[0].tobytes() for c in contigs]
    bp = BatchPolisher(pol, len(contigs))
    cap = max(len(w) for w in want) + 4096
    buf = torch.full((len(contigs), cap), 7, dtype=torch.uint8, device="cuda")
    for slot in range(len(contigs)):
        bp.set_sink(slot, buf[slot].data_ptr(), cap)
    for keep in (False, True):
        buf.fill_(7)
        torch.cuda.synchronize()
        out = bp.polish(contigs, Opts(), keep_on_device=keep)
        host = buf.cpu().numpy()
        for i, w in enumerate(want):
            assert host[i, :len(w)].tobytes() == w and (host[i, len(w):] == 7).all()  # exactly the polished length
            if not keep:
                assert out[i][0].tobytes() == w
    bp.set_sink(1, 0, 0)
    buf.fill_(7)
    torch.cuda.synchronize()
    bp.polish(contigs, Opts())
    host = buf.cpu().numpy()
    assert (host[1] == 7).all() and host[0, :len(want[0])].tobytes() == want[0]
    bp.close()


@pytest.mark.parametrize("n_slots", [5, 2])
def test_batch_equals_per_contig_and_oracle(n_slots):
    syn, yaks = _assembly()
    pol = Polisher(yaks)
    contigs = [pol.upload(s.pileup) for s in syn]
    single = [pol.polish_resident(c, Opts()) for c in contigs]
    bp = BatchPolisher(pol, n_slots)  # 2 slots: three waves
    for _ in range(2):  # (second call: warm buffers, recycled pinned blocks)
        out = bp.polish(contigs, Opts(), want_pos=True)
        for (b, p), (sb, sp) in zip(out, single):
            assert np.array_equal(b, sb) and np.array_equal(p, sp)
    spans = bp.polish(contigs, Opts())
    assert [s[1] for s in spans] == [(int(p[0]), int(p[-1])) for _, p in single]
    o = orc.Oracle(yaks)
    for s, (b, p) in zip(syn, out):
        ob, op = o.polish(s.pileup, Opts())
        assert np.array_equal(ob, b) and np.array_equal(op, p)
    st = bp.stats()
    assert st["launches"] < st["commands"]  # launches really were merged
    bp.close()


def test_batch_with_haploid_and_regionless_contigs():
    # a contig without any LQ region leaves the wave early; the others must not wait for it
    a = Synth(40000, seed=311, asm_err_rate=0.0, read_err_rate=0.0)
    b = Synth(80000, seed=312, diploid=True)
    c = Synth(50000, seed=313)
    yaks = [Synth.yak_assembly([b, a, c], 21)]
    pol = Polisher(yaks)
    contigs = [pol.upload(s.pileup) for s in (a, b, c)]
    bp = BatchPolisher(pol, 3)
    out = bp.polish(contigs, Opts(), want_pos=True)
    o = orc.Oracle(yaks)
    for s, (gb, gp) in zip((a, b, c), out):
        ob, op = o.polish(s.pileup, Opts())
        assert np.array_equal(ob, gb) and np.array_equal(op, gp)
    assert out[0][0].tobytes() == a.hap1


def test_batch_reports_a_failing_contig_and_keeps_going():
    from nextpolish2_amd._types import Pileup
    from nextpolish2_amd.api import Np2Error
    good = Synth(40000, seed=321, diploid=True)
    yaks = [good.yak(21)]
    pol = Polisher(yaks)
    cg = pol.upload(good.pileup)
    bp = BatchPolisher(pol, 2)
    with pytest.raises(Np2Error):
        bp.polish([cg, cg], Opts(iter_count=0))  # every pipeline rejects iter_count 0
    out = bp.polish([cg, cg], Opts())  # the batch is still usable
    assert out[0][0].tobytes() == good.hap1 and np.array_equal(out[0][0], out[1][0])


def test_two_batches_on_two_host_threads():
    syn, yaks = _assembly()
    pol = Polisher(yaks)
    contigs = [pol.upload(s.pileup) for s in syn]
    ref = [pol.polish_resident(c, Opts(), want_pos=False)[0] for c in contigs]
    halves = [[0, 2, 4], [1, 3]]
    bps = [BatchPolisher(pol, len(h)) for h in halves]
    bps[0].set_priority(True)  # (what bench.py does: groups on alternating-priority streams)
    bps[1].set_priority(False)
    outs = [None, None]

    def run(k):
        for _ in range(3):
            outs[k] = bps[k].polish([contigs[i] for i in halves[k]], Opts())
    ths = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for k in range(2):
        for j, i in enumerate(halves[k]):
            assert np.array_equal(outs[k][j][0], ref[i])


def test_pair_votes_outside_the_band_take_the_sort_path(monkeypatch):
    # depth 400 with short reads: more than EDGE_BAND reads start inside one read's span, so some read pair lies outside
    # the banded accumulator and the whole contig falls back to sorting the raw pair votes
    s = Synth(12000, depth=400, seed=331, diploid=True, read_len_mean=3000.0, read_len_sd=300.0)
    yaks = [s.yak(21)]
    o = orc.Oracle(yaks)
    o.set_trace(True)
    ob, op = o.polish(s.pileup, Opts())
    g = Polisher(yaks)
    g.set_trace(True)
    gb, gp = g.polish(s.pileup, Opts())
    assert np.array_equal(o.trace(0, "invalid_ids"), g.trace(0, "invalid_ids"))
    assert np.array_equal(ob, gb) and np.array_equal(op, gp)
    # and the sort path forced on an ordinary contig gives what the band gives
    d = Synth(60000, seed=332, diploid=True)
    yd = [d.yak(21)]
    b1, p1 = Polisher(yd).polish(d.pileup, Opts())
    monkeypatch.setenv("NP2_EDGE_SORT", "1")
    b2, p2 = Polisher(yd).polish(d.pileup, Opts())
    assert np.array_equal(b1, b2) and np.array_equal(p1, p2)
    ob2, op2 = orc.Oracle(yd).polish(d.pileup, Opts())
    assert np.array_equal(ob2, b1) and np.array_equal(op2, p1)


def test_pileups_too_deep_for_the_on_chip_dp_take_the_long_run_kernels(monkeypatch):
    # the on-chip DP of short runs keeps coverages / counts in 16 bits and scores in 32: a pass with a position covered
    # 65536x or more hands every run to the eight-lane / per-thread kernels.  Force that at an ordinary depth — through
    # the plain context (two-stream variant: the kernels classify runs themselves) and through the batch driver (the
    # short kernel's list).
    syn, yaks = _assembly()
    o = orc.Oracle(yaks)
    want = [o.polish(s.pileup, Opts()) for s in syn[:3]]
    monkeypatch.setenv("NP2_TEST_DEEP_COV", "20")  # depth 30: most tiles have such a position, depth 12 contigs none
    monkeypatch.setenv("NP2_DP_FORK", "1")  # plain context: long-run kernels on a second stream, classifying runs themselves
    pol = Polisher(yaks)
    contigs = [pol.upload(s.pileup) for s in syn[:3]]
    for c, (ob, op) in zip(contigs, want):
        b, p = pol.polish_resident(c, Opts())
        assert np.array_equal(ob, b) and np.array_equal(op, p)
    bp = BatchPolisher(pol, 3)
    for (b, p), (ob, op) in zip(bp.polish(contigs, Opts(), want_pos=True), want):
        assert np.array_equal(ob, b) and np.array_equal(op, p)
    bp.close()


def test_full_size_yeast_assembly_through_the_batch_driver():
    """BASELINE.json configs[2] at full size (17 contigs, 12.16 Mb diploid, 30x, k21 + k31, phasing on): the batch
    driver's output per contig equals the one-contig-at-a-time path, is identical on a second run, recovers the
    haplotype of record on (nearly) every contig, and reads really were voted out."""
    from bench import YEAST, make_assembly
    syn = make_assembly(YEAST, 30, 1, True)
    yaks = [Synth.yak_assembly(syn, 21), Synth.yak_assembly(syn, 31)]
    pol = Polisher(yaks)
    contigs = [pol.upload(s.pileup) for s in syn]
    bp = BatchPolisher(pol, len(contigs))
    out1 = bp.polish(contigs, Opts(), want_pos=True)
    out2 = bp.polish(contigs, Opts(), want_pos=True)
    n_truth = 0
    for s, c, (b, p), (b2, p2) in zip(syn, contigs, out1, out2):
        assert np.array_equal(b, b2) and np.array_equal(p, p2)
        assert np.all(p[1:] >= p[:-1])
        sb, sp = pol.polish_resident(c, Opts())
        assert np.array_equal(b, sb) and np.array_equal(p, sp)
        n_truth += b.tobytes() == s.hap1
        assert b.tobytes() != s.pileup.ref.tobytes()
    assert n_truth >= 15  # (polishing recovers hap1 exactly unless a contig keeps an ambiguous site)
    # the largest contig against the oracle, with the reads the vote removed
    big = max(range(len(syn)), key=lambda i: syn[i].pileup.L)
    o = orc.Oracle(yaks)
    o.set_trace(True)
    ob, op = o.polish(syn[big].pileup, Opts())
    assert np.array_equal(ob, out1[big][0]) and np.array_equal(op, out1[big][1])
    assert len(o.trace(0, "invalid_ids")) > 1000


@pytest.mark.parametrize("env", [dict(NP2_WAIT="nap"), dict(NP2_WAIT="nap", LOCAL_WORLD_SIZE="8", NP2_BATCH_SPIN_US="0"),
                                 dict(NP2_WAIT="spin", NP2_BATCH_SPIN_US="30")], ids=["nap", "nap-8-ranks", "spin"])
def test_waiting_by_naps_or_by_spinning_changes_nothing(env):
    """np2_hostcpu.hpp: host threads that wait for the device nap when the process is short of CPUs (eight ranks of a node
    in a small container) and spin otherwise.  The policy is read once per process: a process of its own per setting runs
    the batch driver and a plain context on random contig mixes against the oracle (tests/tools/fuzz_batch.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "fuzz_batch.py"), "911", "3"], capture_output=True,
                       timeout=600, cwd=root, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    last = r.stdout.decode().strip().splitlines()[-1]
    assert last.startswith("batch cases 3 bad 0"), r.stdout.decode()[-2000:]
