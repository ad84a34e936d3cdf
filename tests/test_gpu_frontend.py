"""GPU tests of the input side: GPU columnariser vs the oracle front end, BAM -> resident pileup, CLI end to end."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd.api import Np2Error
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


def test_secondary_alignments_recover_seq_from_the_primary(tmp_path):
    """-S: secondary records carry no SEQ; it comes from the read's primary record (possibly on another contig and
    on the other strand), secondary.rs:82-148 + main.rs:1775-1789."""
    s1 = Synth(40000, depth=12, seed=71, read_len_mean=5000.0, read_len_sd=600.0, name="ctgA")
    s2 = Synth(30000, depth=10, seed=72, read_len_mean=4000.0, read_len_sd=500.0, name="ctgB")
    rng = np.random.default_rng(9)
    recs = []
    for tid, s in ((0, s1), (1, s2)):
        rr = pileup_to_records(s.pileup, tid=tid, rng=np.random.default_rng(5 + tid), decorate=False)
        for i, r in enumerate(rr):
            r["name"] = b"t%d_r%d" % (tid, i)
        recs += rr
    prim = list(recs)
    # secondaries: (a) a copy of a ctgA read's alignment flagged secondary (same strand or flipped), SEQ "*";
    # (b) ctgB reads whose *primary* lives on ctgA-style extra records: make the ctgB record secondary and add a primary
    #     elsewhere with the read sequence in the other orientation; (c) one secondary whose primary is missing but which
    #     fails the admission filters (mapq 0), so nobody may panic
    extra = []
    for r in prim[::5]:
        if r["tid"] != 0:
            continue
        sec = dict(r, flag=(r["flag"] & 0x10) | 0x100, seq="", name=r["name"])
        if rng.random() < 0.5:  # secondary on the opposite strand of its primary: SEQ must be reverse-complemented
            # the alignment columns describe the read as given, so flip the primary to keep the secondary sensible
            r["flag"] ^= 0x10
            r["seq"] = orc.reverse_complement(r["seq"])  # primary stored reverse-complemented, as an aligner would
            # (its own alignment to ctgA is now nonsense and gets trimmed / filtered like any bad read)
        extra.append(sec)
    for r in prim[3::7]:
        if r["tid"] != 1:
            continue
        # primary on ctgA at an arbitrary place (soft-clipped whole read would be filtered; use a short match), secondary on ctgB
        full = r["seq"]
        primary = dict(tid=0, pos=int(rng.integers(0, 30000)), mapq=60, flag=0x10, name=r["name"],
                       cigar=[("S", len(full) - 600), ("M", 600)], seq=orc.reverse_complement(full))
        extra.append(primary)
        r["flag"] = 0x100 | 0x0  # forward secondary: needs RC(RC(full)) == full
        r["seq"] = ""
    orphan = dict(prim[1], flag=0x100, mapq=0, seq="", name=b"orphan")
    extra.append(orphan)
    recs += extra
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    write_bam(str(tmp_path / "s.bam"), [("ctgA", s1.pileup.L), ("ctgB", s2.pileup.L)], recs)
    from test_oracle import yak_from_seqs
    y21 = yak_from_seqs([s1.hap1.decode(), s2.hap1.decode()], 21)
    sec = orc.secondary_seqs(recs)
    assert len(sec) >= 10
    fo = np2io.FrontOpts(use_secondary=True)
    pol = Polisher([y21])
    bam = np2io.Bam(str(tmp_path / "s.bam"))
    o = orc.Oracle([y21])
    for nm, s, tid in (("ctgA", s1, 0), ("ctgB", s2, 1)):
        rr = orc.with_secondary_seq([r for r in recs if r["tid"] == tid], sec)
        arr, cig, seq4, asc, asc_off = records_to_arrays(rr)
        want = orc.front_end(s.pileup.ref.tobytes(), arr, cig, asc, asc_off, fo)
        c = np2io.contig_from_bam(pol, bam, nm, s.pileup.ref.tobytes(), fo)
        assert same_pileup(np2io.export_contig(pol, c, s.pileup.ref), want), nm
        # the secondaries really made it into the pileup (more reads than without -S)
        c0 = np2io.contig_from_bam(pol, bam, nm, s.pileup.ref.tobytes(), np2io.FrontOpts())
        assert np2io.export_contig(pol, c, s.pileup.ref).n_reads > np2io.export_contig(pol, c0, s.pileup.ref).n_reads, nm
        gb, gp = pol.polish_resident(c, Opts())
        ob, op = o.polish(want, Opts())
        assert np.array_equal(gb, ob) and np.array_equal(gp, op)
    # np2_contig_from_records: the caller passes the recovered SEQ
    rr = orc.with_secondary_seq([r for r in recs if r["tid"] == 1], sec)
    arr, cig, seq4, asc, asc_off = records_to_arrays(rr)
    c = np2io.contig_from_records(pol, s2.pileup.ref.tobytes(), arr, cig, seq4, fo)
    assert same_pileup(np2io.export_contig(pol, c, s2.pileup.ref), orc.front_end(s2.pileup.ref.tobytes(), arr, cig, asc, asc_off, fo))
    # a secondary without a primary that passes the filters: the reference panics on the map lookup
    bad = [dict(r) for r in recs if r["tid"] == 1] + [dict(prim[-1], flag=0x100, seq="", name=b"nobody", tid=1)]
    bad.sort(key=lambda r: r["pos"])
    write_bam(str(tmp_path / "b.bam"), [("ctgA", s1.pileup.L), ("ctgB", s2.pileup.L)],
              [r for r in recs if r["tid"] == 0] + bad)
    with pytest.raises(Np2Error):
        np2io.contig_from_bam(pol, np2io.Bam(str(tmp_path / "b.bam")), "ctgB", s2.pileup.ref.tobytes(), fo)



def test_cli_many_contigs_concurrent_contexts(tmp_path):
    """16 contigs of mixed size and ploidy through 1 and 4 concurrent contexts: identical output, in input order."""
    rng = np.random.default_rng(7)
    syn, recs, refs = [], [], []
    for i in range(16):
        L = int(rng.choice([12000, 25000, 60000, 110000]))
        s = Synth(L, depth=int(rng.choice([10, 25])), seed=200 + i, diploid=bool(i % 3 == 0), read_len_mean=5000.0, read_len_sd=800.0,
                  name=f"c{i}")
        syn.append(s)
        refs.append((f"c{i}", s.pileup.L))
        recs += pileup_to_records(s.pileup, tid=i, rng=np.random.default_rng(300 + i), decorate=True)
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    write_bam(str(tmp_path / "m.bam"), refs, recs)
    with gzip.open(tmp_path / "g.fa.gz", "wt") as f:
        for i, s in enumerate(syn):
            f.write(f">c{i}\n{s.pileup.ref.tobytes().decode()}\n")
    from test_oracle import yak_from_seqs
    haps = []
    for s in syn:
        haps += [s.hap1.decode()] + ([s.hap2.decode()] if s.diploid else [])
    np2io.write_yak(str(tmp_path / "k21.yak"), yak_from_seqs(haps, 21))
    env = dict(os.environ, PYTHONPATH=ROOT)
    outs = []
    for t in ("1", "4", "4"):
        r = subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", "-t", t, "-L", "10000", str(tmp_path / "m.bam"),
                            str(tmp_path / "g.fa.gz"), str(tmp_path / "k21.yak")], capture_output=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr.decode()
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2]
    assert outs[0].count(b">") == 16
    # spot check against the oracle
    y21 = np2io.load_yak(str(tmp_path / "k21.yak"))
    o = orc.Oracle([y21])
    for tid in (0, 5, 15):
        rr = [r for r in recs if r["tid"] == tid]
        arr, cig, seq4, asc, asc_off = records_to_arrays(rr)
        pu = orc.front_end(syn[tid].pileup.ref.tobytes(), arr, cig, asc, asc_off, np2io.FrontOpts())
        b, p = o.polish(pu, Opts())
        assert b">c%d start:%d end:%d\n%s\n" % (tid, p[0], p[-1], b.tobytes()) in outs[0]


def test_cli_failing_contig_in_the_middle_of_an_assembly(tmp_path):
    """-S with a secondary record whose read has no primary on the SECOND of three contigs (the reference panics on the map
    lookup there): the run fails, and the first contig's record — polished in the same batch turn or not — is the one a
    run over that contig alone writes."""
    from test_oracle import yak_from_seqs
    syn = [Synth(30000, depth=15, seed=700 + i, read_len_mean=5000.0, read_len_sd=700.0, name=f"c{i}") for i in range(3)]
    recs = []
    for i, s in enumerate(syn):
        recs += pileup_to_records(s.pileup, tid=i, rng=np.random.default_rng(710 + i), decorate=False)
    good = sorted(recs, key=lambda r: (r["tid"], r["pos"]))
    prim1 = [r for r in recs if r["tid"] == 1][-1]
    bad = sorted(recs + [dict(prim1, flag=0x100, seq="", name=b"nobody")], key=lambda r: (r["tid"], r["pos"]))
    refs = [(f"c{i}", s.pileup.L) for i, s in enumerate(syn)]
    write_bam(str(tmp_path / "good.bam"), refs, good)
    write_bam(str(tmp_path / "bad.bam"), refs, bad)
    with open(tmp_path / "g.fa", "w") as f:
        for i, s in enumerate(syn):
            f.write(f">c{i}\n{s.pileup.ref.tobytes().decode()}\n")
    np2io.write_yak(str(tmp_path / "k21.yak"), yak_from_seqs([s.hap1.decode() for s in syn], 21))
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda bam, out: subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", "-S", "-t", "2", "-L", "10000", "-o", str(out), str(bam),
                                           str(tmp_path / "g.fa"), str(tmp_path / "k21.yak")], capture_output=True, env=env, timeout=600)
    r = run(tmp_path / "good.bam", tmp_path / "good.fa")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    want = (tmp_path / "good.fa").read_bytes()
    assert want.count(b">") == 3
    r = run(tmp_path / "bad.bam", tmp_path / "bad.fa")
    assert r.returncode != 0
    got = (tmp_path / "bad.fa").read_bytes() if (tmp_path / "bad.fa").exists() else b""
    first = want[:want.index(b">c1")]
    assert got == first, (len(got), len(first))


def test_command_line_on_a_whole_assembly_bam_equals_the_batch_driver(tmp_path):
    # bench.py's end_to_end_assembly leg at a small scale: BAM written from the generator's records, the CLI's code
    # path in process, output compared with the resident pileups polished through the batch driver
    from bench import end_to_end_assembly, make_assembly
    from nextpolish2_amd import BatchPolisher
    syn = make_assembly([60000, 45000, 30000], 30, 7, True)
    yaks = [Synth.yak_assembly(syn, 21), Synth.yak_assembly(syn, 31)]
    pol = Polisher(yaks)
    bp = BatchPolisher(pol, 3)
    out = bp.polish([pol.upload(s.pileup) for s in syn], Opts())
    r = end_to_end_assembly(syn, yaks, str(tmp_path), [np.array(o[0]) for o in out], [o[1] for o in out], workers=2)
    assert r["identical_to_resident_path"], r
    bp.close()


def test_tables_streamed_from_the_dump_files_equal_the_loaded_ones(tmp_path):
    """np2_ctx_create_from_files (dump -> device in pieces, table built from the file image) against
    np2_yak_load + np2_ctx_create: same counts for present, absent and below-threshold k-mers, tables ordered by k
    whatever the order of the paths; broken dumps are refused with the loader's messages."""
    s = Synth(60000, depth=20, seed=11, diploid=True)
    yaks = [s.yak(21), s.yak(31)]
    paths = []
    for y in yaks:
        paths.append(str(tmp_path / f"k{y.k}.yak"))
        np2io.write_yak(paths[-1], y)
    ref = Polisher(yaks)
    got = np2io.polisher_from_yak_files(paths[::-1])  # (k31 first: sorted by k inside)
    rng = np.random.default_rng(5)
    for i, y in enumerate(yaks):
        # a file word is (x >> 10) << 10 | count of the k-mer hash x with x & 1023 == its bucket: rebuild x for present keys
        pos = rng.integers(0, y.words.shape[0], 4000)
        b = (np.searchsorted(y.bucket_off, pos, side="right") - 1).astype(np.uint64)
        present = (y.words[pos] >> np.uint64(10) << np.uint64(10)) | b
        absent = rng.integers(0, 1 << 62, 4000, dtype=np.uint64)
        h = np.concatenate([present, absent])
        for mk in (1, 5, 200):
            a, c = ref.lookup_hashes(i, h, mk), got.lookup_hashes(i, h, mk)
            assert np.array_equal(a, c)
        assert (ref.lookup_hashes(i, present, 1) > 0).all()
    # and a polish through the streamed tables equals the one through the loaded tables
    c1, c2 = ref.upload(s.pileup), got.upload(s.pileup)
    b1, p1 = ref.polish_resident(c1, Opts())
    b2, p2 = got.polish_resident(c2, Opts())
    assert np.array_equal(b1, b2) and np.array_equal(p1, p2)
    with open(tmp_path / "bad.yak", "wb") as f:
        f.write(b"YAK\x01" + bytes(12))
    with pytest.raises(Np2Error, match="incompatible"):
        np2io.polisher_from_yak_files([str(tmp_path / "bad.yak")])
    raw = open(paths[0], "rb").read()
    with open(tmp_path / "cut.yak", "wb") as f:
        f.write(raw[: len(raw) // 2])
    with pytest.raises(Np2Error, match="Failed to parse"):
        np2io.polisher_from_yak_files([str(tmp_path / "cut.yak")])
    with pytest.raises(Np2Error, match="cannot open"):
        np2io.polisher_from_yak_files([str(tmp_path / "none.yak")])


def test_zlib_fallback_of_the_block_decoder_gives_the_same_records(tmp_path):
    """NP2_INFLATE=zlib (what a host without libdeflate's runtime library runs) against the default decoder: the command
    line's output is byte-identical."""
    s = Synth(60000, depth=20, seed=71, diploid=True, read_len_mean=6000.0, read_len_sd=900.0, name="ctgZ")
    recs = pileup_to_records(s.pileup, tid=0, rng=np.random.default_rng(9), decorate=True)
    write_bam(str(tmp_path / "m.bam"), [("ctgZ", s.pileup.L)], recs)
    with open(tmp_path / "g.fa", "wb") as f:
        f.write(b">ctgZ\n" + s.pileup.ref.tobytes() + b"\n")
    np2io.write_yak(str(tmp_path / "k21.yak"), s.yak(21))
    outs = []
    for mode in ("", "zlib"):
        env = dict(os.environ, PYTHONPATH=ROOT)
        if mode:
            env["NP2_INFLATE"] = mode
        out = tmp_path / f"out_{mode or 'default'}.fa"
        r = subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", "-L", "10000", "-o", str(out), str(tmp_path / "m.bam"),
                            str(tmp_path / "g.fa"), str(tmp_path / "k21.yak")], capture_output=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr.decode()
        outs.append(out.read_bytes())
    assert outs[0] == outs[1] and outs[0].startswith(b">ctgZ start:")
