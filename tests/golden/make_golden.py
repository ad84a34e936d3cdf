"""Regenerates tests/golden/*.npz: small packed pileups + yak words (inputs) and the consensus the CPU
oracle produces for them (expected outputs), plus per-stage digests.

The reference (Rust + htslib) cannot be built or imported in this environment and ships no golden
vectors for this path, so these fixtures pin the oracle against itself over time (regression) and pin
the HIP path against the oracle on the GPU box; hand-derived expectations live in tests/test_oracle.py.
Usage: python tests/golden/make_golden.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from nextpolish2_amd import Opts  # noqa: E402
from nextpolish2_amd.synth import Synth  # noqa: E402
from oracle.np2_oracle import Oracle  # noqa: E402

CASES = {
    "haploid_k21": dict(L=6000, seed=101, diploid=False, ks=[21], opts={}),
    "diploid_k21_k31": dict(L=8000, seed=102, diploid=True, ks=[21, 31], opts={}),
    "diploid_len_model_allreads": dict(L=8000, seed=103, diploid=True, ks=[21], opts=dict(model="len", use_all_reads=True)),
}
STAGES = ["graph.bases", "graph.count", "cns_raw.base", "lq.start", "lq.end", "cand.order", "cand.kscore", "cand.seq"]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def main():
    for name, c in CASES.items():
        s = Synth(c["L"], depth=30, seed=c["seed"], diploid=c["diploid"], read_len_mean=2500.0, read_len_sd=400.0,
                  read_len_min=600)
        yaks = [s.yak(k) for k in c["ks"]]
        o = Oracle(yaks)
        o.set_trace(True)
        opts = Opts(**c["opts"])
        b, p = o.polish(s.pileup, opts)
        out = dict(ref=s.pileup.ref, reads=s.pileup.reads, nibbles=s.pileup.nibbles, out_bases=b, out_pos=p,
                   ks=np.array(c["ks"]), model_ref=np.array([opts.model == "ref"]),
                   use_all_reads=np.array([opts.use_all_reads]))
        for i, y in enumerate(yaks):
            out[f"yak{i}_words"] = y.words
            out[f"yak{i}_off"] = y.bucket_off
        dg = []
        for ps in range(2):
            for st in STAGES:
                t = o.trace(ps, st)
                dg.append(f"{ps}:{st}:{'-' if t is None else digest(t)}")
        out["stage_digests"] = np.array(dg)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, len(b), "bp", os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
