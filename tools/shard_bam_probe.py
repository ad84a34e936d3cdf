"""Reference-interval shards read straight from a BAM, at growing contig sizes (one process, one context per shard):
   python tools/shard_bam_probe.py L [n_shards] [halo]   -> stitched == whole-contig BAM path?"""
import os, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd import io as np2io
from nextpolish2_amd._cpus import usable_cpus
from nextpolish2_amd.bamio import write_bam_raw
from nextpolish2_amd.dist import polish_sharded_bam_local
from nextpolish2_amd.synth import Synth, concat_pileups
L = int(float(sys.argv[1])); ns = int(sys.argv[2]) if len(sys.argv) > 2 else 2; halo = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
NP = max(1, min(16, L // 2_000_000))
with ThreadPoolExecutor(NP) as ex:
    parts = list(ex.map(lambda i: Synth(L // NP, depth=30, seed=500 + i, diploid=True), range(NP)))
pu = concat_pileups([p.pileup for p in parts], "chr1")
yaks = [Synth.yak_assembly(parts, k, threads=usable_cpus()) for k in (21, 31)]
blobs, offs, poss, rls, pos0, name0, base = [], [np.zeros(1, dtype=np.uint64)], [], [], 0, 0, 0
for p in parts:
    b, o, ps, rl = p.bam_records(0, pos0, name0)
    blobs.append(b); offs.append(o[1:] + np.uint64(base)); poss.append(ps); rls.append(rl)
    base += len(b); pos0 += p.pileup.L; name0 += len(ps)
td = tempfile.mkdtemp(prefix="np2_sb_")
bam = os.path.join(td, "c.bam")
write_bam_raw(bam, [("chr1", pu.L)], [(b"".join(blobs), np.concatenate(offs), np.concatenate(poss), np.concatenate(rls))])
ref = pu.ref.tobytes()
pol = Polisher(yaks)
c = np2io.contig_from_bam(pol, np2io.Bam(bam), "chr1", ref)
b0, p0 = pol.polish_resident(c, Opts()); c.free()
print(f"L {pu.L} reads {pu.n_reads}: whole-contig BAM path ok", flush=True)
t = time.time()
try:
    b1, p1 = polish_sharded_bam_local(pol, bam, "chr1", ref, ns, Opts(), halo=halo)
    print(f"{ns} shards: {time.time() - t:.2f} s, stitched == whole: {np.array_equal(b0, b1) and np.array_equal(p0, p1)}", flush=True)
except Exception as e:
    print(f"{ns} shards FAILED: {e}", flush=True)
