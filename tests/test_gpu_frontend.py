"""GPU tests of the input side: GPU columnariser vs the oracle front end, BAM -> resident pileup, CLI end to end."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import pileup_to_records, records_to_arrays, write_bam
from nextpolish2_amd.synth import Synth
from oracle import np2_oracle as orc
from test_frontend_cpu import same_pileup

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_front(ref_bytes, recs, yaks, fopts=None):
    fopts = fopts or np2io.FrontOpts()
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    exp = orc.front_end(ref_bytes, arr, cig, asc, asc_off, fopts)
    pol = Polisher(yaks)
    c = np2io.contig_from_records(pol, ref_bytes, arr, cig, seq4, fopts)
    got = np2io.export_contig(pol, c, np.frombuffer(ref_bytes, dtype=np.uint8))
    assert same_pileup(got, exp)
    # and the polish of the resident pileup equals the oracle's polish of the oracle's pileup
    gb, gp = pol.polish_resident(c, Opts())
    ob, op = orc.Oracle(yaks).polish(exp, Opts())
    assert np.array_equal(gb, ob) and np.array_equal(gp, op)
    return exp


@pytest.mark.parametrize("seed,decorate,diploid", [(51, False, False), (52, True, False), (53, True, True)])
def test_columnariser_matches_oracle(seed, decorate, diploid):
    s = Synth(60000, depth=25, seed=seed, diploid=diploid, read_len_mean=7000.0, read_len_sd=1200.0)
    recs = pileup_to_records(s.pileup, rng=np.random.default_rng(seed), decorate=decorate)
    exp = check_front(s.pileup.ref.tobytes(), recs, [s.yak(21)])
    if not decorate:
        assert same_pileup(exp, s.pileup)


def test_soft_masked_contig_and_long_contig_clip_labels():
    # lower-case contig letters never equal the upper-case read in trim (main.rs:447-513): anchors must avoid them
    s = Synth(40000, depth=15, seed=54, read_len_mean=5000.0, read_len_sd=800.0)
    ref = bytearray(s.pileup.ref.tobytes())
    ref[5000:5400] = bytes(ref[5000:5400]).lower()
    ref[20000:20010] = bytes(ref[20000:20010]).lower()
    recs = pileup_to_records(s.pileup, rng=np.random.default_rng(1), decorate=True)
    check_front(bytes(ref), recs, [s.yak(21)])
    # options: -s, -q, -l, -a, -c
    check_front(bytes(ref), recs, [s.yak(21)], np2io.FrontOpts(use_supplementary=True, min_map_qual=0, min_read_len=2000,
                                                               min_map_len=1000, min_map_fra=0.9, max_clip_len=10))


def test_contig_from_bam_and_cli_end_to_end(tmp_path):
    s1 = Synth(50000, depth=20, seed=61, read_len_mean=6000.0, read_len_sd=900.0, name="ctgA")
    s2 = Synth(30000, depth=20, seed=62, diploid=True, read_len_mean=5000.0, read_len_sd=700.0, name="ctgB")
    recs = pileup_to_records(s1.pileup, tid=0, rng=np.random.default_rng(3), decorate=True) + \
        pileup_to_records(s2.pileup, tid=1, rng=np.random.default_rng(4), decorate=True)
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    write_bam(str(tmp_path / "m.bam"), [("ctgA", s1.pileup.L), ("ctgB", s2.pileup.L), ("tiny", 500)], recs)
    with gzip.open(tmp_path / "g.fa.gz", "wt") as f:
        for nm, s in (("ctgA", s1), ("ctgB", s2)):
            seq = s.pileup.ref.tobytes().decode()
            f.write(f">{nm} len={len(seq)}\n")
            for i in range(0, len(seq), 80):
                f.write(seq[i:i + 80] + "\n")
        f.write(">tiny\nACGTACGTNNacgt\n")
    # yak tables over both contigs' haplotypes: reuse s1's generator for k-mers of s1 only is not enough -> build from both
    from test_oracle import yak_from_seqs
    y21 = yak_from_seqs([s1.hap1.decode(), s2.hap1.decode(), s2.hap2.decode()], 21)
    y31 = yak_from_seqs([s1.hap1.decode(), s2.hap1.decode(), s2.hap2.decode()], 31)
    np2io.write_yak(str(tmp_path / "k31.yak"), y31)
    np2io.write_yak(str(tmp_path / "k21.yak"), y21)
    # expected: oracle front end + oracle polish per contig
    exp = b""
    o = orc.Oracle([y21, y31])
    for nm, s, tid in (("ctgA", s1, 0), ("ctgB", s2, 1)):
        rr = [r for r in recs if r["tid"] == tid]
        arr, cig, seq4, asc, asc_off = records_to_arrays(rr)
        pu = orc.front_end(s.pileup.ref.tobytes(), arr, cig, asc, asc_off, np2io.FrontOpts())
        b, p = o.polish(pu, Opts())
        exp += b">%s start:%d end:%d\n%s\n" % (nm.encode(), p[0], p[-1], b.tobytes())
    exp += b">tiny start:0 end:13\nACGTACGTNNacgt\n"
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = tmp_path / "out.fa"
    r = subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", "-L", "10000", "-o", str(out), str(tmp_path / "m.bam"),
                        str(tmp_path / "g.fa.gz"), str(tmp_path / "k31.yak"), str(tmp_path / "k21.yak")],
                       capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    assert out.read_bytes() == exp
    # -t 3: three np2 contexts on the same GPU, contigs in flight concurrently, output still in input order
    out3 = tmp_path / "out3.fa"
    r = subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", "-t", "3", "-L", "10000", "-o", str(out3),
                        str(tmp_path / "m.bam"), str(tmp_path / "g.fa.gz"), str(tmp_path / "k31.yak"),
                        str(tmp_path / "k21.yak")], capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    assert out3.read_bytes() == exp
    # API path
    pol = Polisher([y21, y31])
    bam = np2io.Bam(str(tmp_path / "m.bam"))
    c = np2io.contig_from_bam(pol, bam, "ctgB", s2.pileup.ref.tobytes())
    rr = [r for r in recs if r["tid"] == 1]
    arr, cig, seq4, asc, asc_off = records_to_arrays(rr)
    assert same_pileup(np2io.export_contig(pol, c, s2.pileup.ref), orc.front_end(s2.pileup.ref.tobytes(), arr, cig, asc, asc_off, np2io.FrontOpts()))
