"""The DEFLATE core of the GPU inflater (csrc/np2_inflate_core.hpp: table builder, symbol decoder, stream loop — the code
every lane of the decoding wavefront runs) as a one-lane host program against zlib: streams zlib itself wrote at every
level and strategy, stored / fixed / dynamic blocks, many blocks per stream, distances up to 32768, damaged streams."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_inflate_core_equals_zlib(tmp_path):
    exe = str(tmp_path / "inflate_core_test")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "tools", "inflate_core_test.cpp"), "-lz"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    for seed in ("11", "12"):
        r = subprocess.run([exe, seed, "8"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        assert int(r.stdout.strip()) == 8 * 6 * 25
