"""CPU tests of the input side: oracle front end, BAM/FASTA/yak readers, CLI pass-through known answer."""
import gzip
import io as pyio
import os
import subprocess
import sys

import numpy as np
import pytest

from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import pileup_to_records, records_to_arrays, write_bam
from nextpolish2_amd.synth import Synth
from oracle import np2_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def nib_streams(pu):
    out = []
    for r in range(pu.n_reads):
        rd = pu.reads[r]
        n = int(rd["n_cols"])
        if rd["flags"] & 1:
            out.append(None)
        else:
            out.append(pu.nibbles[int(rd["nib_off"]):int(rd["nib_off"]) + ((n + 1) >> 1) + 1].tobytes())
    return out


def same_pileup(a, b):
    if a.n_reads != b.n_reads:
        return False
    for f in ("aln_t_s", "aln_t_e", "n_cols", "flags"):
        if not np.array_equal(a.reads[f], b.reads[f]):
            return False
    return nib_streams(a) == nib_streams(b)


def test_oracle_front_end_round_trips_a_clean_pileup():
    # packed reads -> CIGAR/SEQ records -> front end must give back the very same packed pileup
    s = Synth(30000, depth=12, seed=7, read_len_mean=4000.0, read_len_sd=600.0)
    recs = pileup_to_records(s.pileup, decorate=False)
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    pu = orc.front_end(s.pileup.ref.tobytes(), arr, cig, asc, asc_off, np2io.FrontOpts())
    assert same_pileup(pu, s.pileup)


def test_oracle_front_end_filters_and_trim_by_hand():
    ref = "ACGTTGCAAGCTTAGGCTAACGTAGCTAGGATCCGATTACGCTAGCTAGGCTTAAGCG" * 30
    L = len(ref)
    body = ref[100:1700]
    recs = [
        dict(tid=0, pos=100, mapq=60, flag=0, cigar=[("M", 1600)], seq=body),                      # kept
        dict(tid=0, pos=100, mapq=1, flag=0, cigar=[("M", 1600)], seq=body),                       # mapq <= 1
        dict(tid=0, pos=100, mapq=60, flag=0x400, cigar=[("M", 1600)], seq=body),                  # duplicate
        dict(tid=0, pos=100, mapq=60, flag=0x800, cigar=[("M", 1600)], seq=body),                  # supplementary
        dict(tid=0, pos=100, mapq=60, flag=0, cigar=[("M", 900)], seq=body[:900]),                 # rlen <= 1000
        dict(tid=0, pos=100, mapq=60, flag=0, cigar=[("S", 2000), ("M", 1600)], seq="A" * 2000 + body),  # span < rlen/2
        dict(tid=0, pos=100, mapq=60, flag=0, cigar=[("S", 150), ("M", 1600)], seq="A" * 150 + body),    # clipped, L < 500k
        # 3 mismatching columns in front: trimmed away by trim(8); aln_t_s moves to 103
        dict(tid=0, pos=100, mapq=60, flag=0, cigar=[("M", 1600)], seq="".join("ACGT"[("ACGT".index(c) + 1) % 4] for c in body[:3]) + body[3:]),
        # soft-masked (lower-case) contig never equals the upper-case read in trim -> covered in the GPU test
    ]
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    pu = orc.front_end(ref.encode(), arr, cig, asc, asc_off, np2io.FrontOpts())
    assert pu.n_reads == 3
    assert pu.reads["aln_t_s"].tolist() == [0, 100, 103]
    assert pu.reads["aln_t_e"].tolist() == [L - 1, 1699, 1699]
    assert pu.reads["n_cols"].tolist() == [L, 1600, 1597]


def test_bam_fasta_yak_readers(tmp_path):
    s = Synth(20000, depth=8, seed=9, read_len_mean=3000.0, read_len_sd=400.0)
    recs = pileup_to_records(s.pileup, decorate=True)
    write_bam(str(tmp_path / "a.bam"), [("ctg1", s.pileup.L), ("other", 1000)], recs)
    b = np2io.Bam(str(tmp_path / "a.bam"))
    assert b.refs() == [("ctg1", s.pileup.L), ("other", 1000)]
    with gzip.open(tmp_path / "g.fa.gz", "wt") as f:
        f.write(">ctg1 desc here\nACGT\nacgtn\n\n>c2\tx\nAC\n")
    assert list(np2io.read_fasta(str(tmp_path / "g.fa.gz"))) == [("ctg1", b"ACGTacgtn"), ("c2", b"AC")]
    y = s.yak(21)
    np2io.write_yak(str(tmp_path / "k.yak"), y)
    y2 = np2io.load_yak(str(tmp_path / "k.yak"))
    assert y2.k == 21 and np.array_equal(y.words, y2.words) and np.array_equal(y.bucket_off, y2.bucket_off)
    with open(tmp_path / "bad.yak", "wb") as f:
        f.write(b"NOPE" + b"\0" * 12)
    with pytest.raises(Exception):
        np2io.load_yak(str(tmp_path / "bad.yak"))


def test_cli_pass_through_known_answer(tmp_path):
    """The only known answer derivable from the reference without running it: with the default
    -L 1000000 the 100 kb contig of test/hh.sh takes the pass-through branch (src/main.rs:1727-1730)."""
    fa = os.path.join(HERE, "golden", "ref_test_asm.fa.gz")
    s = Synth(3000, depth=3, seed=1, read_len_mean=1500.0, read_len_sd=100.0)
    np2io.write_yak(str(tmp_path / "k21.yak"), s.yak(21))
    write_bam(str(tmp_path / "x.bam"), [("ptg000005l:21113231-21213230", 100000)], [])
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", "-t", "5", str(tmp_path / "x.bam"), fa,
                        str(tmp_path / "k21.yak")], capture_output=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    with gzip.open(fa, "rt") as f:
        lines = f.read().split("\n")
    assert lines[0].startswith(">ptg000005l:21113231-21213230")
    seq = "".join(lines[1:])
    assert len(seq) == 100000
    assert r.stdout.decode() == f">ptg000005l:21113231-21213230 start:0 end:99999\n{seq}\n"
    assert b"Real time" in r.stderr
    # -o refuses to overwrite (option.rs:312-316); -u upper-cases pass-through contigs
    out = tmp_path / "o.fa"
    out.write_text("x")
    r2 = subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", "-o", str(out), str(tmp_path / "x.bam"), fa,
                         str(tmp_path / "k21.yak")], capture_output=True, env=env, timeout=120)
    assert r2.returncode != 0 and b"already exists" in r2.stderr
    r3 = subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", "-u", str(tmp_path / "x.bam"), fa,
                         str(tmp_path / "k21.yak")], capture_output=True, env=env, timeout=120)
    assert r3.stdout.decode().split("\n")[1] == seq.upper()


def test_oracle_secondary_seq_recovery_rules():
    """secondary.rs:82-148 + main.rs:1775-1784 on a hand-made record list."""
    recs = [
        dict(name=b"a", flag=0x000, seq="AACGTN"),   # primary, forward, has a secondary
        dict(name=b"a", flag=0x100, seq=""),         # secondary forward  -> AACGTN
        dict(name=b"a", flag=0x110, seq=""),         # secondary reverse  -> revcomp = NACGTT
        dict(name=b"b", flag=0x010, seq="GGCAm"),    # primary on the reverse strand: stored in read orientation
        dict(name=b"b", flag=0x900, seq=""),         # secondary + supplementary counts as secondary for the id set
        dict(name=b"b", flag=0x800, seq="TTTT"),     # supplementary: never a source
        dict(name=b"c", flag=0x000, seq="ACGT"),     # no secondary: not collected
        dict(name=b"d", flag=0x100, seq=""),         # secondary without any primary
    ]
    sec = orc.secondary_seqs(recs)
    assert sec == {b"a": "AACGTN", b"b": "mTGCC"}
    out = orc.with_secondary_seq(recs, sec)
    assert [r["seq"] for r in out] == ["AACGTN", "AACGTN", "NACGTT", "GGCAm", "mTGCC", "TTTT", "ACGT", ""]
    with pytest.raises(AssertionError):
        orc.secondary_seqs(recs + [dict(name=b"a", flag=0, seq="AC")])



def test_generator_bam_records_equal_the_python_writer(tmp_path):
    # Synth.bam_records (C++) + bamio.write_bam_raw: whole-assembly BAM files for bench.py without the per-column Python
    # loop of pileup_to_records; same records, valid BGZF, an index the C++ reader accepts
    from nextpolish2_amd.bamio import pileup_to_records, read_bam, write_bam, write_bam_raw
    from nextpolish2_amd.synth import Synth
    ss = [Synth(30000, seed=5, diploid=True, name="a"), Synth(20000, seed=6, name="b")]
    refs = [(s.pileup.name, s.pileup.L) for s in ss]
    write_bam_raw(str(tmp_path / "x.bam"), refs, [s.bam_records(i) for i, s in enumerate(ss)], threads=2)
    recs = []
    for i, s in enumerate(ss):
        recs += pileup_to_records(s.pileup, tid=i)
    write_bam(str(tmp_path / "y.bam"), refs, recs)
    ra, xa = read_bam(str(tmp_path / "x.bam"))
    rb, xb = read_bam(str(tmp_path / "y.bam"))
    assert ra == rb and len(xa) == len(xb) == sum(s.pileup.n_reads - 1 for s in ss)
    for a, b in zip(xa, xb):
        assert all(a[k] == b[k] for k in ("tid", "pos", "mapq", "cigar", "seq"))
    bam = np2io.Bam(str(tmp_path / "x.bam"))  # header + .bai parse
    assert bam.refs() == refs
