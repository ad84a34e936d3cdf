"""Read extraction on the device (csrc/np2_inflate.hip): the BGZF inflate kernel against zlib, and a contig read through it —
blocks uploaded as they lie in the file, inflated one wavefront per block, records found along the .bai linear index, SEQ
read by the columnariser out of the inflated stream — against the host path (libdeflate / zlib pool + record walk) and the
oracle's front end."""
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd.bamio import pileup_to_records, records_to_arrays, write_bam
from nextpolish2_amd.synth import Synth
from oracle import np2_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bgzf_block(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, memlevel=8):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
    comp = co.compress(bytes(data)) + co.flush()
    assert len(comp) + 26 <= 65536
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25)
    return hdr + comp + struct.pack("<II", zlib.crc32(bytes(data)) & 0xFFFFFFFF, len(data))


def test_inflate_kernel_equals_zlib():
    rng = np.random.default_rng(5)
    pol = Polisher([Synth(2000, seed=3).yak(21)])
    blocks, want = [], []

    def add(data, **kw):
        blocks.append(bgzf_block(data, **kw))
        want.append(bytes(data))
    add(b"")                                   # the end-of-file marker's shape
    add(b"A")
    add(bytes(rng.integers(0, 256, 65280, dtype=np.uint8)), level=0)        # stored
    add(bytes(rng.integers(0, 256, 40000, dtype=np.uint8)), level=6)        # incompressible: literals only
    for n in (1, 2, 3, 257, 258, 259, 4095, 4096, 4097, 32767, 32768, 32769, 65280):
        add(b"\xff" * n)                       # runs: distance 1, lengths up to 258 (a BAM's QUAL without qualities)
    for lvl in (1, 4, 6, 9):
        for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            n = int(rng.integers(1, 65281))
            kind = int(rng.integers(0, 4))
            if kind == 0:    # packed nucleotides (a BAM's SEQ)
                d = ((1 << rng.integers(0, 4, n)) << 4 | (1 << rng.integers(0, 4, n))).astype(np.uint8)
            elif kind == 1:  # text with repeats
                d = np.frombuffer((b"GATTACA-%d-" % lvl) * (n // 8 + 2), dtype=np.uint8)[:n].copy()
                d[rng.integers(0, n, n // 50 + 1)] = rng.integers(0, 256, n // 50 + 1)
            elif kind == 2:  # a repeat at distance 32768
                base = rng.integers(0, 256, 32768, dtype=np.uint8)
                d = np.concatenate([base, base])[: max(n, 40000)][:65280]
            else:            # qualities: a few values in long runs
                d = np.repeat(rng.integers(0, 42, n // 40 + 1, dtype=np.uint8), 40)[:n]
            add(d.tobytes(), level=lvl, strategy=strat, memlevel=1 if (lvl + kind) % 2 else 8)
    data = b"".join(blocks)
    got, ms = np2io.bgzf_inflate_device(pol, data)
    exp = b"".join(want)
    assert len(got) == len(exp)
    assert got.tobytes() == exp
    # a damaged block is reported, not decoded into something else silently — and never written past its ISIZE
    bad = bytearray(blocks[3])
    bad[30] ^= 0x10
    with pytest.raises(Exception):
        g2, _ = np2io.bgzf_inflate_device(pol, bytes(bad))
        assert g2.tobytes() != want[3]
        raise RuntimeError("decoded to different bytes (acceptable: deflate carries no checksum of its own)")


def _contig(tmp_path, s, env):
    """export of the resident pileup np2_contig_from_bam builds, and the polished contig, in a fresh process"""
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from nextpolish2_amd import Opts, Polisher\nfrom nextpolish2_amd import io as np2io\n"
            "from nextpolish2_amd.synth import Synth\n"
            "pol = np2io.polisher_from_yak_files([sys.argv[3]])\n"
            "ref = open(sys.argv[2], 'rb').read()\n"
            "bam = np2io.Bam(sys.argv[1])\n"
            "c = np2io.contig_from_bam(pol, bam, 'ctgA', ref, np2io.FrontOpts())\n"
            "ex = np2io.export_contig(pol, c, np.frombuffer(ref, dtype=np.uint8))\n"
            "b, p = pol.polish_resident(c, Opts())\n"
            "np.savez(sys.argv[4], reads=ex.reads, nib=ex.nibbles, b=b, p=p)\n" % ROOT)
    out = str(tmp_path / ("out_%s.npz" % env.get("NP2_INFLATE", "host")))
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "m.bam"), str(tmp_path / "ref.txt"), str(tmp_path / "k21.yak"), out],
                       capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out), r.stderr


@pytest.mark.parametrize("L,depth,diploid,decorate", [(60000, 25, True, True), (250000, 30, False, False), (3000, 8, False, True)])
def test_contig_through_the_device_equals_the_host_path(tmp_path, L, depth, diploid, decorate):
    s = Synth(L, depth=depth, seed=1000 + L % 97, diploid=diploid, read_len_mean=min(9000.0, L / 3), read_len_sd=min(1500.0, L / 20),
              read_len_min=min(1200, L // 3))
    recs = pileup_to_records(s.pileup, tid=0, rng=np.random.default_rng(3), decorate=decorate)
    write_bam(str(tmp_path / "m.bam"), [("ctgA", s.pileup.L)], recs)
    (tmp_path / "ref.txt").write_bytes(s.pileup.ref.tobytes())
    np2io.write_yak(str(tmp_path / "k21.yak"), s.yak(21))
    host, _ = _contig(tmp_path, s, {"NP2_INFLATE": "libdeflate"})
    dev, err = _contig(tmp_path, s, {"NP2_INFLATE": "gpu", "NP2_IO_PROFILE": "1"})
    assert "fetch_records_gpu" in err, err[-2000:]  # (the device path really ran)
    for k in ("reads", "nib", "b", "p"):
        assert np.array_equal(host[k], dev[k]), k
    # ... and both equal the oracle's front end + polish
    arr, cig, seq4, asc, asc_off = records_to_arrays(recs)
    pu = orc.front_end(s.pileup.ref.tobytes(), arr, cig, asc, asc_off, np2io.FrontOpts())
    from nextpolish2_amd._types import Pileup
    from test_frontend_cpu import same_pileup
    assert same_pileup(Pileup(s.pileup.ref, dev["reads"], dev["nib"]), pu)
    ob, op = orc.Oracle([s.yak(21)]).polish(pu, Opts())
    assert np.array_equal(dev["b"], ob) and np.array_equal(dev["p"], op)


def _bundle4(tmp_path):
    """a BAM of four references — two with reads, one without any, one more with reads — plus FASTA and k-mer dumps"""
    import gzip
    from test_oracle import yak_from_seqs
    sA = Synth(50000, depth=20, seed=161, read_len_mean=6000.0, read_len_sd=900.0, name="ctgA")
    sB = Synth(30000, depth=20, seed=162, diploid=True, read_len_mean=5000.0, read_len_sd=700.0, name="ctgB")
    sD = Synth(20000, depth=15, seed=164, read_len_mean=4000.0, read_len_sd=600.0, name="ctgD")
    recs = []
    for tid, s, sd in ((0, sA, 3), (1, sB, 4), (3, sD, 5)):
        recs += pileup_to_records(s.pileup, tid=tid, rng=np.random.default_rng(sd), decorate=True)
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    refs = [("ctgA", sA.pileup.L), ("ctgB", sB.pileup.L), ("ctgC", 25000), ("ctgD", sD.pileup.L)]
    write_bam(str(tmp_path / "m.bam"), refs, recs)
    rngc = np.random.default_rng(7)
    ctgC = "".join(rngc.choice(list("ACGT"), 25000))
    with gzip.open(tmp_path / "g.fa.gz", "wt") as f:
        for nm, seq in (("ctgA", sA.pileup.ref.tobytes().decode()), ("ctgB", sB.pileup.ref.tobytes().decode()), ("ctgC", ctgC),
                        ("ctgD", sD.pileup.ref.tobytes().decode())):
            f.write(f">{nm}\n{seq}\n")
    y21 = yak_from_seqs([sA.hap1.decode(), sB.hap1.decode(), sB.hap2.decode(), sD.hap1.decode()], 21)
    np2io.write_yak(str(tmp_path / "k21.yak"), y21)
    return recs, (sA, sB, sD), ctgC, y21


@pytest.mark.parametrize("short_hint", [False, True])
def test_references_of_one_bam_through_the_device(tmp_path, short_hint):
    """The device path on a BAM of several references: a reference's records end inside a block the next one's begin in, a
    reference without records, the last reference running to the end-of-file marker — and (short_hint) an index that
    understates where a reference's records end, so that the range is extended until the walk meets another reference."""
    recs, (sA, sB, sD), ctgC, y21 = _bundle4(tmp_path)
    o = orc.Oracle([y21])
    exp = b""
    for nm, s, tid in (("ctgA", sA, 0), ("ctgB", sB, 1), ("ctgC", None, 2), ("ctgD", sD, 3)):
        if s is None:  # no read: polishing changes nothing (every position is the contig's own node)
            from nextpolish2_amd.synth import pileup_from_alignments
            b, p = o.polish(pileup_from_alignments(ctgC, []), Opts())
        else:
            rr = [r for r in recs if r["tid"] == tid]
            arr, cig, seq4, asc, asc_off = records_to_arrays(rr)
            b, p = o.polish(orc.front_end(s.pileup.ref.tobytes(), arr, cig, asc, asc_off, np2io.FrontOpts()), Opts())
        exp += b">%s start:%d end:%d\n%s\n" % (nm.encode(), p[0], p[-1], b.tobytes())
    outs = {}
    # (gpu: the whole file inflated on the device once, every reference a stretch of that stream; gpu_per_ref: a reference's
    # blocks uploaded and inflated by themselves — what a file too large to keep resident gets)
    for mode in ("gpu", "gpu_per_ref", "libdeflate"):
        env = dict(os.environ, PYTHONPATH=ROOT, NP2_INFLATE=mode.split("_")[0], NP2_IO_PROFILE="1")
        if mode == "gpu_per_ref":
            env["NP2_BAM_RESIDENT_MB"] = "0"
        if short_hint and mode.startswith("gpu"):
            env["NP2_TEST_FETCH_SHORT_HINT"] = "1"
        out = tmp_path / ("out_%s.fa" % mode)
        r = subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", "-L", "10000", "-o", str(out), str(tmp_path / "m.bam"),
                            str(tmp_path / "g.fa.gz"), str(tmp_path / "k21.yak")], capture_output=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        outs[mode] = out.read_bytes()
        if mode.startswith("gpu"):
            err = r.stderr.decode()
            assert err.count("fetch_records_gpu:") >= 3, err[-2000:]
            assert ("resident BAM:" in err and err.count("stretch of the resident stream") >= 3) == (mode == "gpu"), err[-2000:]
            if short_hint:
                assert "after extending the range" in err
    assert outs["gpu"] == outs["libdeflate"] == outs["gpu_per_ref"]
    assert outs["gpu"] == exp


def test_no_room_on_the_device_takes_the_host_pools_path(tmp_path):
    """The inflated stream of a reference does not fit on the device next to what else lives there (the allocation is refused:
    NP2_TEST_FETCH_NO_ROOM stands in for a full device): the contig goes through the host pool instead, with the same output."""
    _bundle4(tmp_path)
    outs = {}
    for mode in ("no_room", "libdeflate"):
        env = dict(os.environ, PYTHONPATH=ROOT, NP2_INFLATE="gpu" if mode == "no_room" else mode, NP2_IO_PROFILE="1")
        if mode == "no_room":
            env["NP2_TEST_FETCH_NO_ROOM"] = "1"
        out = tmp_path / ("out_%s.fa" % mode)
        r = subprocess.run([sys.executable, "-m", "nextpolish2_amd.cli", "-L", "10000", "-o", str(out), str(tmp_path / "m.bam"),
                            str(tmp_path / "g.fa.gz"), str(tmp_path / "k21.yak")], capture_output=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        outs[mode] = out.read_bytes()
        if mode == "no_room":
            assert "fetch_records_gpu:" not in r.stderr.decode()
    assert outs["no_room"] == outs["libdeflate"] and len(outs["no_room"]) > 100000
