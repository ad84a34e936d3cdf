"""kappa of SURVEY.md §8(d): k-mer table probes per polished bp that the reference algorithm makes on bench.py's
synthetic workloads (the oracle's kmer_probes stat: both passes, every yak table).  CPU only; writes profiles/r04_kappa.json.
   python tools/kappa.py"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nextpolish2_amd import Opts
from nextpolish2_amd.synth import Synth
from oracle.np2_oracle import Oracle

res = {}
for wl, (lengths, dip, ks) in {"yeast": (bench.YEAST, True, [21, 31]), "ecoli": ([4_600_000], False, [21])}.items():
    syn = bench.make_assembly(lengths, 30, 1, dip)
    yaks = [Synth.yak_assembly(syn, k) for k in ks]
    base = Oracle(yaks)
    small = min(range(len(syn)), key=lambda i: syn[i].pileup.L)
    base.polish(syn[small].pileup, Opts())  # (builds the shared tables before the workers clone the oracle)
    st = [None] * len(syn)
    def work(i):
        o = base.clone(5)
        o.polish(syn[i].pileup, Opts())
        st[i] = o.stats()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(syn))]
    t = time.time()
    [x.start() for x in ths]
    [x.join() for x in ths]
    L = sum(s.pileup.L for s in syn)
    cols = sum(int(s.pileup.n_columns()) for s in syn)
    tot = {k: sum(x[k] for x in st) for k in st[0]}
    res[wl] = dict(tot, assembly_bp=L, pileup_columns_incl_read0=cols, kappa=tot["kmer_probes"] / L,
                   bytes_per_bp=2 * 0.5 * cols / L + 2 + 8 * tot["kmer_probes"] / L)
    print(wl, res[wl], f"{time.time() - t:.1f} s", flush=True)
json.dump({"what": "k-mer table probes per polished bp (kappa of SURVEY.md 8(d)) made by the reference algorithm on bench.py's "
                   "synthetic workloads (oracle stat kmer_probes: both passes, all yak tables); bytes_per_bp = iter_count x 0.5 x "
                   "columns / bp + 2 + 8 x kappa; tools/kappa.py in the build container", "workloads": res},
          open(os.path.join(ROOT, "profiles", "r04_kappa.json"), "w"), indent=1)
