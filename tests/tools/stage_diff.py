"""Stage-by-stage comparison of the HIP path against the CPU oracle on synthetic pileups.

Run on a GPU box: python tools/stage_diff.py [L] [seed] [diploid] ; prints the first divergence per stage."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nextpolish2_amd import Opts, Polisher  # noqa: E402
from nextpolish2_amd.synth import Synth  # noqa: E402
from oracle.np2_oracle import Oracle  # noqa: E402

STAGES = ["graph.off", "graph.bases", "graph.delta", "graph.count", "cns_raw.pos", "cns_raw.base", "lq.start", "lq.end",
          "cand.cand_off", "cand.order", "cand.seq_off", "cand.seq", "cand.kmer", "cand.kscore", "hete.lable",
          "hete.kscore", "invalid_ids", "seed.lable", "seed.sudo", "seed.cand_off", "seed.order", "cns_succ.pos",
          "cns_succ.base", "rech0.kscore", "rech0.lable", "rech0.sudo", "cns_rech0.pos", "cns_rech0.base",
          "rech1.kscore", "rech1.sudo", "cns_rech1.pos", "cns_rech1.base"]


def compare(o, g, n_pass, log):
    ok = True
    for p in range(n_pass):
        for st in STAGES:
            a, b = o.trace(p, st), g.trace(p, st)
            if a is None and b is None:
                continue
            if a is None or b is None:
                log(f"pass {p} {st}: missing in {'oracle' if a is None else 'hip'} (other has {len(b if a is None else a)})")
                ok = False
                continue
            if a.shape != b.shape or not np.array_equal(a, b):
                ok = False
                n = min(len(a), len(b))
                d = np.nonzero(a[:n] != b[:n])[0]
                first = int(d[0]) if len(d) else n
                log(f"pass {p} {st}: MISMATCH len {len(a)} vs {len(b)}, first diff @{first}: "
                    f"oracle {a[max(0, first - 2):first + 4].tolist()} hip {b[max(0, first - 2):first + 4].tolist()} (#diff {len(d)})")
            else:
                log(f"pass {p} {st}: ok ({len(a)})")
    return ok


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    diploid = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    ks = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [21]
    log = print
    s = Synth(L, depth=30, seed=seed, diploid=diploid, read_len_mean=min(13000.0, L / 2), read_len_sd=min(2000.0, L / 10))
    yaks = [s.yak(k) for k in ks]
    o = Oracle(yaks)
    o.set_trace(True)
    t = time.time()
    ob, op = o.polish(s.pileup)
    log(f"oracle {time.time() - t:.3f}s len {len(ob)} stats {o.stats()}")
    g = Polisher(yaks)
    g.set_trace(True)
    t = time.time()
    try:
        gb, gp = g.polish(s.pileup)
    except Exception as e:  # still show the stages that completed
        log(f"HIP polish failed: {e}")
        compare(o, g, 2, log)
        return 1
    log(f"hip {time.time() - t:.3f}s len {len(gb)} timings {g.timings()}")
    ok = compare(o, g, 2, log)
    same = np.array_equal(ob, gb) and np.array_equal(op, gp)
    log(f"FINAL consensus identical: {same}")
    return 0 if (ok and same) else 1


if __name__ == "__main__":
    sys.exit(main())
