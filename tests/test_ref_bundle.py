"""BASELINE.json configs[0]: the reference's own test bundle (test/hh.sh) — real simulated HiFi reads of a 100 kb
diploid region, short-read k21 + k31 yak tables.  The inputs under tests/golden/ref_bundle/ are derived from the
reference's test data files by tests/golden/make_ref_bundle.py (which stands in for yak count / minimap2 / samtools);
expected.fa.gz is the oracle's output on them."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from nextpolish2_amd import Opts
from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import read_bam, records_to_arrays
from oracle import np2_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUNDLE = os.path.join(ROOT, "tests", "golden", "ref_bundle")
ASM = os.path.join(ROOT, "tests", "golden", "ref_test_asm.fa.gz")
BAM = os.path.join(BUNDLE, "hifi.map.sort.bam")


def bundle():
    (name, ref), = list(np2io.read_fasta(ASM))
    refs, recs = read_bam(BAM)
    assert refs == [(name, len(ref))]
    yaks = [np2io.load_yak(os.path.join(BUNDLE, "k21.yak")), np2io.load_yak(os.path.join(BUNDLE, "k31.yak"))]
    return name, ref, recs, yaks


def oracle_fasta(name, ref, recs, yaks, opts, fopts):
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    pu = orc.front_end(ref, arr, cig, asc, asc_off, fopts)
    b, p = orc.Oracle(yaks).polish(pu, opts)
    return pu, b">%s start:%d end:%d\n%s\n" % (name.encode(), p[0], p[-1], b.tobytes())


def test_oracle_reproduces_the_committed_output():
    name, ref, recs, yaks = bundle()
    assert len(recs) == 574 and [y.k for y in yaks] == [21, 31]
    pu, fa = oracle_fasta(name, ref, recs, yaks, Opts(), np2io.FrontOpts())
    assert pu.n_reads == 445  # the contig + 444 admitted reads (the clip and length filters drop the rest)
    assert fa == gzip.open(os.path.join(BUNDLE, "expected.fa.gz"), "rb").read()


@pytest.mark.gpu
def test_cli_on_the_reference_test_bundle(tmp_path):
    """hh.sh:12: nextPolish2 -t 5 hifi.map.sort.bam asm.fa.gz k21.yak k31.yak > asm.np2.fa"""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "nextpolish2_amd.cli", "-t", "5", BAM, ASM, os.path.join(BUNDLE, "k21.yak"),
           os.path.join(BUNDLE, "k31.yak")]
    # the literal hh.sh command: the 100 kb contig is shorter than the default -L 1000000, so the reference writes it
    # back unpolished (main.rs:1727-1730, option.rs:280)
    r = subprocess.run(cmd, capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    (name, ref), = list(np2io.read_fasta(ASM))
    assert r.stdout == b">%s start:0 end:%d\n%s\n" % (name.encode(), len(ref) - 1, ref)
    # with the length gate lowered the contig is polished
    r = subprocess.run(cmd[:3] + ["-L", "1000"] + cmd[3:], capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == gzip.open(os.path.join(BUNDLE, "expected.fa.gz"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("opts,fopts", [
    (dict(), dict()),
    (dict(min_kmer_count=2, iter_count=3), dict(max_clip_len=100000, min_map_qual=0)),
    (dict(model="len", use_all_reads=True, max_indel_len=5), dict(min_read_len=12800, min_map_fra=0.2)),
    (dict(iter_count=1, min_kmer_count=20), dict(max_clip_len=0)),
])
def test_resident_path_matches_oracle_on_the_bundle(opts, fopts):
    from nextpolish2_amd import Polisher
    from test_frontend_cpu import same_pileup
    name, ref, recs, yaks = bundle()
    o, fo = Opts(**opts), np2io.FrontOpts(**fopts)
    pu, fa = oracle_fasta(name, ref, recs, yaks, o, fo)
    pol = Polisher(yaks)
    bam = np2io.Bam(BAM)
    c = np2io.contig_from_bam(pol, bam, name, ref, fo)
    assert same_pileup(np2io.export_contig(pol, c, np.frombuffer(ref, dtype=np.uint8)), pu)
    b, p = pol.polish_resident(c, o)
    assert b">%s start:%d end:%d\n%s\n" % (name.encode(), p[0], p[-1], b.tobytes()) == fa
