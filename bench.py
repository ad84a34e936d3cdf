"""bench.py — polished reference Mbp/s of the NextPolish2 hot path on MI355X.

A "step" = one pass of the whole hot path (np2_polish_resident: dense diff -> sparse graph -> DP ->
LQ regions -> candidates -> yak scoring -> phasing vote -> second pass -> seed/recheck/splice) over one
HBM-resident synthetic contig.  Workload = BASELINE.json configs[1]: E. coli-sized 4.6 Mb contig,
30x simulated HiFi, k21 yak only.  With --gpus N every rank polishes its own contig (weak scaling, no
data-path collective) and the polished sequences are all-gathered over RCCL inside the timed step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
# HBM traffic of one k_diff_reads launch on the default workload, from rocprofv3 PMC passes
# (profiles/r01j_pmc_fetch_write.json: 2 x FETCH_SIZE (gfx950 half-count correction) + WRITE_SIZE)
PMC_TRAFFIC_DEFAULT_WORKLOAD = int((2 * 48559.0 + 31205.3) * 1024)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--length", type=int, default=4_600_000, help="contig length (bp); default = E. coli")
    ap.add_argument("--depth", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="bp of the same workload timed on the CPU oracle")
    ap.add_argument("--cpu-threads", type=int, default=32, help="host threads of the CPU baseline (one contig each)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from nextpolish2_amd import Opts, Polisher
    from nextpolish2_amd.dist import SequenceGatherer
    from nextpolish2_amd.synth import Synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = "RANK" in os.environ  # launched by torch.distributed.run (also with one rank: same code path)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    # synthetic inputs (SURVEY.md §8d recipe), one contig per rank
    syn = Synth(a.length, depth=a.depth, seed=1 + rank)
    yaks = [syn.yak(21)]
    pol = Polisher(yaks, device=local_rank)
    contig = pol.upload(syn.pileup)  # pileup resident in HBM before the timed region
    opts = Opts()
    gatherer = SequenceGatherer(a.length + a.length // 16 + 4096, dev) if distributed else None

    pending, last = [False], [None]

    def step():
        # FASTA output needs the sequence and the first/last position only (main.rs:627-632).  The sequence is fetched
        # deferred: the device-to-host copy of contig i runs on the context's output stream while contig i + 1 is
        # polished; drain() below waits for the last one inside the timed region.
        _, pos = pol.polish_resident(contig, opts, want_pos=False, defer_output=True)
        if distributed:
            # RCCL all-gather of the polished contigs straight from the context's result buffer in HBM; it runs on
            # torch's stream and overlaps the next contig's kernels (waited for by sync() at the end of the timed region)
            gatherer.gather_device(*pol.last_result_device())
        if pending[0]:
            last[0] = pol.fetch_end()  # the previous contig's sequence is on the host now
        pol.fetch_begin()
        pending[0] = True
        return pos

    def drain():
        if pending[0]:
            last[0] = pol.fetch_end()
            pending[0] = False
        return last[0]

    def sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    drain()
    diff_ms = []
    stage_ms = {}
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pos = step()
        # HIP events around k_diff_reads, recorded on the context's own stream inside the timed region
        diff_ms.append(pol.timings().get("diff_reads", 0.0))
    bases = drain()  # every polished sequence is on the host before the clock stops
    sync()
    dt = time.perf_counter() - t0
    bases = np.array(bases)  # (the view lives in the context's pinned buffer, reused by later fetches)
    # per-stage breakdown: a few extra, untimed steps with every stage timer armed (each timer adds event packets)
    pol.set_timing(True)
    for i in range(4):
        step()
        drain()
        if i == 0:
            continue  # the first step after arming the timers pays one-off event set-up in the runtime
        if os.environ.get("NP2_BENCH_DEBUG"):
            print("stage step", {k: round(v, 3) for k, v in pol.timings().items() if k.startswith("wall")}, file=sys.stderr)
        for k, v in pol.timings().items():
            stage_ms[k] = stage_ms.get(k, 0.0) + v / 3
    pol.set_timing(False)
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    L = syn.pileup.L
    total_bp = L
    if distributed:  # every rank polishes its own contig: sum their lengths
        tb = torch.tensor([L], dtype=torch.int64, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_bp = int(tb.item())
    value = total_bp * a.steps / dt / 1e6
    # roofline of the dominant kernel k_diff_reads: algorithmic bytes per launch =
    # 0.5 B per pileup column (packed nibbles, read once) + 0.5 B per contig base (nibble-packed contig)
    n_cols = syn.pileup.n_columns() - L  # read 0 (the contig itself) is not streamed
    alg_bytes = 0.5 * n_cols + 0.5 * L
    avg_ms = float(np.mean(diff_ms)) if diff_ms else 0.0
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0

    out = {
        "metric": "polished reference Mbp/s (whole node) at 30x HiFi + k21 yak; FASTA identical to oracle",
        "value": round(value, 3), "unit": "Mbp/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "E. coli-sized contig, 30x simulated HiFi, k21 yak only, 1 contig per MI355X",
                   "contig_bp": L, "depth": a.depth, "reads": syn.pileup.n_reads, "pileup_columns": int(n_cols),
                   "yak_k": [21], "iter_count": 2, "parallelism": f"contig-sharded x{world}",
                   "output": "polished sequence copied to the host per contig; the copy of contig i overlaps contig i+1"},
        "roofline": {"bound": "hbm", "kernel": "k_diff_reads", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": PMC_TRAFFIC_DEFAULT_WORKLOAD if (a.length == 4_600_000 and a.depth == 30) else None,
                     "alg_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(avg_ms, 4)},
        "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
    }

    if rank == 0 and not a.no_cpu_baseline:
        # CPU baseline: the oracle (a port of the reference algorithm; the Rust reference cannot be
        # built here) on a bounded sample of the same workload, one thread like one reference worker.
        from oracle.np2_oracle import Oracle
        sl = min(a.cpu_sample, a.length)
        s2 = syn if sl == a.length else Synth(sl, depth=a.depth, seed=1)
        y2 = yaks if s2 is syn else [s2.yak(21)]
        o = Oracle(y2)
        t1 = time.perf_counter()
        ob, op = o.polish(s2.pileup, opts)
        cpu_dt = time.perf_counter() - t1
        # the reference parallelises over contigs (one contig per rayon worker, main.rs:1726-1837): the same sample on C
        # host threads at once, one oracle instance each (the C++ oracle runs outside the GIL)
        import threading
        n_thr = max(1, min(a.cpu_threads, os.cpu_count() or 1))
        ths = [threading.Thread(target=lambda: Oracle(y2).polish(s2.pileup, opts)) for _ in range(n_thr)]
        t1 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        par_dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(n_thr * s2.pileup.L / par_dt / 1e6, 4), "unit": "Mbp/s", "cores": n_thr,
                               "kind": "port", "sample": f"{n_thr} contigs of {s2.pileup.L} bp (the same workload) polished "
                               f"concurrently, one contig per thread like the reference's workers; in-memory yak table",
                               "single_thread": round(s2.pileup.L / cpu_dt / 1e6, 4), "host_cores": os.cpu_count()}
        if s2 is syn:
            out["fasta_identical_to_oracle"] = bool(np.array_equal(ob, bases) and (int(op[0]), int(op[-1])) == pos)
        else:
            gb, gp = Polisher(y2, device=local_rank).polish(s2.pileup, opts)
            out["fasta_identical_to_oracle"] = bool(np.array_equal(ob, gb) and np.array_equal(op, gp))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        if rank == 0:
            # the gathered sequences really are the polished contigs (checked outside the timed region)
            got = gatherer.to_host()
            assert got[0] == bases.tobytes(), "all-gathered sequence differs from the polished contig"
            assert all(len(got[r]) > 0 for r in range(world))
        dist.barrier()  # (rank 0 spends a few seconds on the CPU baseline: leave together)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
