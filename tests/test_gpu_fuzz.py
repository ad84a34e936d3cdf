"""Randomised parity: the HIP path against the oracle over a spread of contig sizes, depths, error rates, ploidy, yak
tables and options (seeded, so every run sees the same cases), plus the configurations an earlier, longer fuzz run
caught."""
import numpy as np
import pytest

from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.api import Np2Error
from nextpolish2_amd.synth import Synth
from oracle.np2_oracle import Oracle

pytestmark = pytest.mark.gpu


def run_case(L, depth, dip, rerr, aerr, rl, seed, ks, o):
    s = Synth(L, depth=depth, seed=seed, diploid=dip, read_err_rate=rerr, asm_err_rate=aerr, read_len_mean=min(rl, L / 2),
              read_len_sd=min(rl / 6, L / 12), read_len_min=min(1000, L // 4))
    yaks = [s.yak(k) for k in ks]
    opts = Opts(**o)
    try:
        ob, op = Oracle(yaks).polish(s.pileup, opts)
        oerr = None
    except Exception as e:  # the reference would panic: the product must report it too
        oerr = e
    try:
        gb, gp = Polisher(yaks).polish(s.pileup, opts)
        gerr = None
    except Np2Error as e:
        gerr = e
    assert (oerr is None) == (gerr is None), (oerr, gerr)
    if oerr is None:
        assert np.array_equal(ob, gb) and np.array_equal(op, gp)


# A run of dirty positions starting at contig position 1 or 2 competes with a read's head-sentinel start node: the
# run-relative scores need the absolute score of the position before the run (early_run_base in np2_kernels.hip).
EARLY_RUN_CASES = [
    dict(L=1500, depth=3, dip=False, rerr=0.01, aerr=0.001, rl=1200.0, seed=943221148, ks=[21],
         o=dict(min_kmer_count=5, max_indel_len=20, iter_count=3, model="len", use_all_reads=False)),
    dict(L=3000, depth=8, dip=True, rerr=0.01, aerr=0.005, rl=3000.0, seed=409677078, ks=[21, 31],
         o=dict(min_kmer_count=2, max_indel_len=20, iter_count=2, model="ref", use_all_reads=True)),
    dict(L=50000, depth=8, dip=True, rerr=0.03, aerr=0.0001, rl=3000.0, seed=438969798, ks=[21, 31],
         o=dict(min_kmer_count=2, max_indel_len=20, iter_count=1, model="ref", use_all_reads=False)),
    dict(L=120000, depth=100, dip=True, rerr=0.03, aerr=0.005, rl=8000.0, seed=984247457, ks=[21],
         o=dict(min_kmer_count=8, max_indel_len=5, iter_count=1, model="len", use_all_reads=False)),
]


@pytest.mark.parametrize("case", EARLY_RUN_CASES, ids=lambda c: f"L{c['L']}-seed{c['seed']}")
def test_runs_starting_next_to_the_contig_start(case):
    run_case(**case)


@pytest.mark.parametrize("stream", [101, 102, 103])
def test_random_configurations(stream):
    rng = np.random.default_rng(stream)
    for _ in range(40):
        run_case(L=int(rng.choice([1500, 3000, 8000, 20000, 50000, 120000])), depth=int(rng.choice([3, 8, 15, 30, 60, 100])),
                 dip=bool(rng.integers(0, 2)), rerr=float(rng.choice([0.0005, 0.002, 0.01, 0.03])),
                 aerr=float(rng.choice([1e-4, 1e-3, 5e-3])), rl=float(rng.choice([1200, 3000, 8000])),
                 seed=int(rng.integers(1, 1 << 30)), ks=[21] if rng.integers(0, 2) else [21, 31],
                 o=dict(min_kmer_count=int(rng.choice([2, 5, 8])), iter_count=int(rng.choice([1, 2, 3])),
                        model=str(rng.choice(["ref", "len"])), use_all_reads=bool(rng.integers(0, 2)),
                        max_indel_len=int(rng.choice([5, 20]))))


def test_one_context_many_contigs():
    """tests/tools/fuzz_reuse.py: one context polishing different contigs back to back, in all three output modes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "fuzz_reuse.py"), "9", "5"], capture_output=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    last = r.stdout.decode().strip().splitlines()[-1]
    assert last.startswith("reuse polishes") and last.endswith("bad 0"), r.stdout.decode()[-2000:]
