"""configs[3]-sized workload on one GPU: a 248 Mb contig (16 generated pieces laid end to end), 30x, k21 + k31; whole, and
cut into 2 / 4 reference intervals polished one after the other on the same device (the shard protocol of dist.py).
   python tools/chr1_probe.py [L] [--diploid]      (--diploid: 15x + 15x reads of two haplotypes: HETE regions, the
                                                    phasing vote and its host-side Louvain over ~570 k reads)"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.dist import polish_sharded_local
from nextpolish2_amd.synth import Synth, concat_pileups
args = [a for a in sys.argv[1:] if not a.startswith("--")]
diploid = "--diploid" in sys.argv
L = int(float(args[0])) if args else 248_000_000
NP = 16
t = time.time()
with ThreadPoolExecutor(NP) as ex:
    parts = list(ex.map(lambda i: Synth(L // NP, depth=30, seed=500 + i, diploid=diploid), range(NP)))
print(f"gen {time.time()-t:.1f}s diploid={diploid}", flush=True)
t = time.time(); pu = concat_pileups([p.pileup for p in parts], "chr1"); print(f"concat {time.time()-t:.1f}s L={pu.L} reads={pu.n_reads} cols={pu.n_columns()}", flush=True)
t = time.time()
yaks = [Synth.yak_assembly(parts, k) for k in (21, 31)]  # (one at a time: the table is built inside the first generator)
print(f"yak {time.time()-t:.1f}s words {[len(y.words) for y in yaks]}", flush=True)
truth = b"".join(p.hap1 for p in parts)
t = time.time(); pol = Polisher(yaks); print(f"ctx {time.time()-t:.1f}s", flush=True)
t = time.time(); c = pol.upload(pu); print(f"upload {time.time()-t:.2f}s", flush=True)
pol.set_timing(True)
for i in range(3):
    t = time.time(); b, span = pol.polish_resident(c, Opts(), want_pos=False); dt = time.time() - t
    tm = pol.timings()
    print(f"polish {dt*1e3:.1f} ms -> {pu.L/dt/1e6:.0f} Mbp/s span {span}; host vote (wall_louvain) {tm.get('wall_louvain', 0.0):.1f} ms; "
          f"wall_vote {tm.get('wall_vote', 0.0):.1f} wall_final {tm.get('wall_final', 0.0):.1f} wall_diff {tm.get('wall_diff', 0.0):.1f} "
          f"wall_graph {tm.get('wall_graph', 0.0):.1f} wall_cns_lq {tm.get('wall_cns_lq', 0.0):.1f} wall_extract {tm.get('wall_extract', 0.0):.1f}", flush=True)
pol.set_timing(False)
whole = b.tobytes()
print("equals truth:", whole == truth, len(b), len(truth), flush=True)
c.free()
from nextpolish2_amd.api import ShardRun, shard_plan
from nextpolish2_amd.dist import _run_local
for ns in (2, 4):
    plans = shard_plan(pu, ns, 65536)
    ctxs = [pol.clone() for _ in range(ns)]
    for rep in range(2):
        t = time.time(); runs = [ShardRun(ctxs[k], pu, plans[k], Opts(), 1024) for k in range(ns)]; t_up = time.time() - t
        t = time.time(); b2, _ = _run_local(runs, plans, pu.n_reads, Opts(), False); dt = time.time() - t
        for r in runs:
            r.close()
        print(f"sharded x{ns}, intervals one after the other on one GPU: shard uploads + dense passes {t_up:.2f} s; protocol (votes, "
              f"final passes, strips, slices fetched into one host array) {dt*1e3:.1f} ms == whole: {b2.tobytes() == whole}", flush=True)
