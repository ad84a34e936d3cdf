"""Probe: configs[2]-like workload (yeast-sized diploid assembly, 17 contigs, k21+k31) through 1..N contexts."""
import os, sys, time, threading
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.synth import Synth

YEAST = [230218, 813184, 316620, 1531933, 576874, 270161, 1090940, 562643, 439888, 745751, 666816, 1078177, 924431,
         784333, 1091291, 948066, 85779]
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
lens = [max(20000, int(l * scale)) for l in YEAST]
t = time.time()
with ThreadPoolExecutor(16) as ex:
    syn = list(ex.map(lambda a: Synth(a[1], depth=30, seed=100 + a[0], diploid=True, name=f"chr{a[0]+1}"), enumerate(lens)))
print(f"gen {time.time()-t:.1f}s total {sum(lens)} bp", flush=True)
t = time.time()
yaks = [Synth.yak_assembly(syn, 21), Synth.yak_assembly(syn, 31)]
print(f"yak {time.time()-t:.1f}s words {[len(y.words) for y in yaks]}", flush=True)
total = sum(lens)
res = [None] * len(syn)
for nctx in [int(x) for x in os.environ.get('NCTX', '1,2,4,8').split(',') if x]:
    pols = [Polisher(yaks) for _ in range(nctx)]
    contigs = [pols[i % nctx].upload(s.pileup) for i, s in enumerate(syn)]
    def work(w):
        for i in range(w, len(syn), nctx):
            b, span = pols[w].polish_resident(contigs[i], Opts(), want_pos=False)
            res[i] = bytes(b)
    for rep in range(3):
        t = time.time()
        ths = [threading.Thread(target=work, args=(w,)) for w in range(nctx)]
        [x.start() for x in ths]; [x.join() for x in ths]
        dt = time.time() - t
        print(f"nctx {nctx} rep {rep}: {dt*1e3:.1f} ms -> {total/dt/1e6:.0f} Mbp/s", flush=True)
    if nctx == 1:
        pols[0].set_timing(True)
        for i in (3, 9):
            pols[0].polish_resident(contigs[i], Opts(), want_pos=False)
            print(f"contig {i} L={lens[i]} timings", {k: round(v, 3) for k, v in pols[0].timings().items()}, flush=True)
        pols[0].set_timing(False)
        ok1 = sum(res[i] == syn[i].hap1 for i in range(len(syn)))
        ok2 = sum(res[i] == syn[i].hap2 for i in range(len(syn)))
        print("equals hap1:", ok1, "hap2:", ok2, "of", len(syn))
    for c in contigs: c.free()
    for p in pols: p.close()

# ---- batch driver: every contig of the assembly in one launch stream ----
from nextpolish2_amd import BatchPolisher
pol = Polisher(yaks)
contigs = [pol.upload(s.pileup) for s in syn]
for nslots in [int(x) for x in os.environ.get('NSLOTS', '17').split(',')]:
    bp = BatchPolisher(pol, nslots)
    for rep in range(4):
        t = time.time()
        out = bp.polish(contigs, Opts())
        dt = time.time() - t
        print(f"batch slots {nslots} rep {rep}: {dt*1e3:.1f} ms -> {total/dt/1e6:.0f} Mbp/s stats {bp.stats()}", flush=True)
    print("flush log (host, issue, wait ms):", bp.flush_log(), flush=True)
    ok = sum(bytes(out[i][0]) == res[i] for i in range(len(syn))) if res[0] is not None else -1
    okh = sum(bytes(out[i][0]) == syn[i].hap1 for i in range(len(syn)))
    print("batch == per-contig:", ok, "== hap1:", okh, "of", len(syn), flush=True)
    bp.close()

# ---- two half batches, phase-shifted: one group's host phases (Louvain) overlap the other group's kernels ----
order = sorted(range(len(syn)), key=lambda i: -lens[i])
halves = [order[0::2], order[1::2]]
bps = [BatchPolisher(pol, len(h)) for h in halves]
outs = [None, None]
def half(k):
    outs[k] = bps[k].polish([contigs[i] for i in halves[k]], Opts())
for rep in range(4):
    t = time.time()
    ths = [threading.Thread(target=half, args=(k,)) for k in range(2)]
    [x.start() for x in ths]; [x.join() for x in ths]
    dt = time.time() - t
    print(f"2 half batches rep {rep}: {dt*1e3:.1f} ms -> {total/dt/1e6:.0f} Mbp/s", flush=True)
okh = sum(bytes(outs[k][j][0]) == syn[i].hap1 for k in range(2) for j, i in enumerate(halves[k]))
print("half batches == hap1:", okh, "of", len(syn))
for k in range(2):
    print("flush log", k, bps[k].flush_log())
