"""Does the CPU baseline (the oracle run like the reference: one contig per worker thread) scale with the host's cores?
Polishes `jobs` contigs (the yeast-sized assembly's 17, replicated) on T threads for several T and prints Mbp/s with
the process's page faults and context switches per run — run it under different glibc malloc settings (environment
variables MALLOC_ARENA_MAX / MALLOC_MMAP_THRESHOLD_ / MALLOC_TRIM_THRESHOLD_ / MALLOC_TOP_PAD_, read at start-up).
   python tools/cpu_scaling_probe.py [jobs] [threads,threads,...]"""
import os, resource, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import YEAST, make_assembly
from nextpolish2_amd import Opts
from nextpolish2_amd.synth import Synth
from oracle.np2_oracle import Oracle

jobs_n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
threads = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [17, 32, 64, 128, 256]
syn = make_assembly(YEAST, 30, 1, True)
yaks = [Synth.yak_assembly(syn, k) for k in (21, 31)]
base = Oracle(yaks)
opts = Opts()
base.polish(syn[-1].pileup, opts)  # (the shared in-memory tables are built on first use: before the workers clone it)
print("malloc env:", {k: v for k, v in os.environ.items() if k.startswith("MALLOC_")}, "cores", os.cpu_count(), flush=True)
for n in threads:
    jobs = [i % len(syn) for i in range(max(n, jobs_n))]
    nxt, lock, done = [0], threading.Lock(), [0] * n

    def work(w):
        orc = base.clone(opts.min_kmer_count)
        while True:
            with lock:
                j = nxt[0]
                nxt[0] += 1
            if j >= len(jobs):
                return
            orc.polish(syn[jobs[j]].pileup, opts)
            done[w] += syn[jobs[j]].pileup.L
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    t = time.perf_counter()
    ths = [threading.Thread(target=work, args=(w,)) for w in range(n)]
    for x in ths:
        x.start()
    for x in ths:
        x.join()
    dt = time.perf_counter() - t
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    print(f"threads {n:4d}: {sum(done) / dt / 1e6:8.2f} Mbp/s in {dt:6.1f} s over {len(jobs)} contigs; cpu {r1.ru_utime - r0.ru_utime:.0f} s user "
          f"{r1.ru_stime - r0.ru_stime:.0f} s sys; minor faults {(r1.ru_minflt - r0.ru_minflt) / 1e6:.1f} M; "
          f"ctx switches vol {r1.ru_nvcsw - r0.ru_nvcsw} invol {r1.ru_nivcsw - r0.ru_nivcsw}", flush=True)
