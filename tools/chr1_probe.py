"""configs[3]-sized workload on one GPU: a 248 Mb contig (16 generated pieces laid end to end), 30x, k21 + k31."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.dist import polish_sharded_local
from nextpolish2_amd.synth import Synth, concat_pileups
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 248_000_000
NP = 16
t = time.time()
with ThreadPoolExecutor(NP) as ex:
    parts = list(ex.map(lambda i: Synth(L // NP, depth=30, seed=500 + i), range(NP)))
print(f"gen {time.time()-t:.1f}s", flush=True)
t = time.time(); pu = concat_pileups([p.pileup for p in parts], "chr1"); print(f"concat {time.time()-t:.1f}s L={pu.L} reads={pu.n_reads} cols={pu.n_columns()}", flush=True)
t = time.time()
yaks = [Synth.yak_assembly(parts, k) for k in (21, 31)]  # (one at a time: the table is built inside the first generator)
print(f"yak {time.time()-t:.1f}s words {[len(y.words) for y in yaks]}", flush=True)
truth = b"".join(p.hap1 for p in parts)
t = time.time(); pol = Polisher(yaks); print(f"ctx {time.time()-t:.1f}s", flush=True)
t = time.time(); c = pol.upload(pu); print(f"upload {time.time()-t:.2f}s", flush=True)
for i in range(3):
    t = time.time(); b, span = pol.polish_resident(c, Opts(), want_pos=False); dt = time.time() - t
    print(f"polish {dt*1e3:.1f} ms -> {pu.L/dt/1e6:.0f} Mbp/s span {span}", flush=True)
print("equals truth:", b.tobytes() == truth, len(b), len(truth), flush=True)
c.free()
for ns in (2, 4):
    t = time.time(); b2, p2 = polish_sharded_local(pol, pu, Opts(), n_shards=ns); dt = time.time() - t
    print(f"sharded x{ns} (sequential on one GPU, incl. uploads) {dt:.2f}s == whole: {np.array_equal(b2, b)}", flush=True)
