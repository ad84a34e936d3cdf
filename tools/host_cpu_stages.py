#!/usr/bin/env python3
"""Host CPU per stage of the polish loop: the yeast-sized diploid contigs (a sample of them) through a plain context with
stage timing on and the waits napping instead of spinning (NP2_WAIT=nap), so that the thread-CPU clock of a stage is the
work the host does in it, not the time it waits for the device.  Prints, per stage, wall ms and CPU ms summed over the
contigs, and CPU ms per Mb.      python tools/host_cpu_stages.py [n_contigs]"""
import os, sys
os.environ.setdefault("NP2_WAIT", "nap")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from nextpolish2_amd import Opts, Polisher
from nextpolish2_amd.synth import Synth
from bench import YEAST

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lens = sorted(YEAST)[-n:]
with ThreadPoolExecutor(8) as ex:
    syn = list(ex.map(lambda il: Synth(il[1], depth=30, seed=100 + il[0], diploid=True), enumerate(lens)))
yaks = [Synth.yak_assembly(syn, k) for k in (21, 31)]
pol = Polisher(yaks)
cs = [pol.upload(s.pileup) for s in syn]
opts = Opts()
for c in cs:  # warm
    pol.polish_resident(c, opts, want_pos=False)
pol.set_timing(True)
tot = {}
for rep in range(3):
    for c in cs:
        pol.polish_resident(c, opts, want_pos=False)
        for k, v in pol.timings().items():
            if k.startswith(("wall", "cpu")):
                tot[k] = tot.get(k, 0.0) + v / 3
mb = sum(lens) / 1e6
print(f"{len(cs)} contigs, {mb:.2f} Mb; per pass over them (ms):")
for k in sorted(tot, key=lambda k: (k.split('_', 1)[1], k)):
    print(f"  {k:24s} {tot[k]:9.3f}   {tot[k] / mb:7.3f} per Mb")
