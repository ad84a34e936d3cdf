"""nextPolish2 command line mirror (src/utils/option.rs:45-292, src/main.rs:1689-1856).

Usage: nextPolish2 [OPTIONS] <HiFi.map.bam> <genome.fa[.gz]> <short.read.yak>...
Same positionals, flags, defaults and output format as the reference; contigs are polished on the GPU and written in
input order.  `-t N` (N >= 2) keeps two front ends (BGZF inflate + record walk on the host pool or the device, GPU
columnariser) and two polish workers going side by side; a worker polishes the contigs that are resident by its turn as
ONE batch (np2_batch_polish: one launch per pipeline step for all of them), and the host-side phases of one batch (the
phasing vote's Louvain) overlap the GPU phases of the other; the k-mer dumps are streamed into their HBM tables next to
the first front ends."""
import argparse
import os
import resource
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import io as np2io
from ._types import Opts
from .api import Polisher, fasta_record

VERSION = "np2-mi355x 0.1 (reference semantics: NextPolish2 v0.2.2)"


def _existing(path):
    p = os.path.abspath(path)
    if not os.path.exists(p):
        raise argparse.ArgumentTypeError(f"{p!r} does not exist!")
    return p


def _map_len(v):
    """-a INT.FLOAT parsed as f32 like the reference (option.rs:232 remove_one::<f32>)."""
    return np.float32(v)


def split_map_len(v):
    """(min_map_len, min_map_fra) = (f32 as usize, f32::fract()) (option.rs:258-259): both in single precision, so
    `-a 500.3` gives fra = 0.29998779 (not 0.3) and (rlen as f32 * fra) as i64 matches the reference to the unit."""
    f = np.float32(v)
    return int(f), float(np.float32(f - np.trunc(f)))


def build_parser():
    p = argparse.ArgumentParser(prog="nextPolish2", description="Repeat-aware polishing genomes assembled using HiFi long reads")
    p.add_argument("bam", type=_existing, metavar="HiFi.map.bam", help="HiFi-to-ref mapping file in sorted BAM format")
    p.add_argument("fa", type=_existing, metavar="genome.fa[.gz]", help="genome assembly file in [GZIP] FASTA format")
    p.add_argument("yak", type=_existing, nargs="+", metavar="short.read.yak", help="one or more k-mer dataset in yak format")
    p.add_argument("-o", "--out", default=None, metavar="FILE", help="output file [stdout]")
    p.add_argument("-u", "--uppercase", action="store_true", help="output in uppercase sequences")
    p.add_argument("--out_pos", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("-k", "--min_kmer_count", type=int, default=5)
    p.add_argument("-t", "--thread", type=int, default=1, help="contigs in flight on the GPU (1-4 np2 contexts)")
    p.add_argument("-i", "--iter_count", type=int, default=2)
    p.add_argument("-m", "--model", default="ref", type=str, help="ref|len (case-insensitive)")
    p.add_argument("-l", "--min_read_len", type=int, default=1000)
    p.add_argument("-L", "--min_ctg_len", type=int, default=1000000)
    p.add_argument("-n", "--max_indel_len", type=int, default=20)
    p.add_argument("-s", "--use_supplementary", action="store_true")
    p.add_argument("-S", "--use_secondary", action="store_true")
    p.add_argument("-a", "--min_map_len", type=_map_len, default=500.5, metavar="INT.FLOAT")
    p.add_argument("-q", "--min_map_qual", type=int, default=1)
    p.add_argument("-c", "--max_clip_len", type=int, default=100)
    p.add_argument("-r", "--use_all_reads", action="store_true")
    p.add_argument("--min_base_cov", type=int, default=1, help=argparse.SUPPRESS)
    p.add_argument("--device", type=int, default=None, help="HIP device index [LOCAL_RANK, else 0]")
    p.add_argument("--shard_min_len", type=int, default=32_000_000,
                   help="under torchrun: contigs at least this long are cut into one reference interval per rank "
                        "(shorter ones go to one rank each)")
    p.add_argument("--shard_halo", type=int, default=65536, help="reads overlapping a rank's interval widened by this are held")
    p.add_argument("--dist_backend", default=None, help="torch.distributed backend under torchrun [nccl]")
    p.add_argument("-V", "--version", action="version", version=VERSION)
    return p


def _cpu_seconds():
    ru = resource.getrusage(resource.RUSAGE_SELF)
    return ru.ru_utime + ru.ru_stime


def resource_str(t0, argv, cpu0=0.0):
    ru = resource.getrusage(resource.RUSAGE_SELF)
    return (f"[INFO] Version: {VERSION}\n[INFO] CMD: {' '.join(argv)}\n[INFO] Real time: {time.time() - t0:.3f} sec; "
            f"CPU: {ru.ru_utime + ru.ru_stime - cpu0:.3f} sec; Peak RSS: {ru.ru_maxrss / 1048576:.3f} GB")


def _record(a, name, b, first, last, pos=None):
    if a.uppercase:
        b = b.upper()
    if a.out_pos:
        return b"".join(b"%s\t%c\t%d\n" % (name.encode(), b[i:i + 1], int(pos[i])) for i in range(len(b)))
    return b">%s start:%d end:%d\n%s\n" % (name.encode(), first, last, b)


def _init_distributed(a):
    """torch.distributed process group of a torchrun launch, then the one decision that is rank 0's alone — may the
    output file be written (option.rs:312-316) — shared with every rank.  Returns rank 0's output stream (None elsewhere)."""
    import torch
    import torch.distributed as dist

    from .dist import all_gather_arrays
    rank = int(os.environ["RANK"])
    backend = a.dist_backend or "nccl"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dev_idx = a.device % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev_idx)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_idx))
        xdev = torch.device("cuda", dev_idx)
    else:
        dist.init_process_group(backend=backend)
        xdev = torch.device("cpu")
    out, verdict = None, b""
    if rank == 0:
        out = sys.stdout.buffer
        if a.out is not None and a.out != "stdout":
            path = os.path.abspath(a.out)
            if os.path.exists(path):
                verdict = f"Error: {path!r} already exists!".encode()
            else:
                out = open(path, "wb")
    got = all_gather_arrays(np.frombuffer(verdict, dtype=np.uint8), device=xdev)
    if len(got[0]):
        dist.destroy_process_group()
        raise SystemExit(got[0].tobytes().decode())
    return out


def _main_distributed(a, argv, t0, out, yaks, opts, fopts):
    """One process per GPU (torchrun): the assembly is polished by all ranks and written by rank 0 in input order.

    Contigs of at least --shard_min_len are cut into one reference interval per rank (np2_shard_*: votes all-gathered and
    decided contig-wide per phasing pass, pieces all-gathered and stitched); the others go whole to one rank each,
    longest first (the reference's unit of work, main.rs:1726-1837).  The only collectives are the small vote exchange
    and the all-gather of polished sequences (RCCL over xGMI with the default backend)."""
    import torch
    import torch.distributed as dist

    from .dist import ShardMismatch, all_gather_arrays, all_gather_sequences, assign_contigs, polish_sharded_bam
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = a.dist_backend or "nccl"
    dev_idx = a.device % max(1, torch.cuda.device_count())
    xdev = torch.device("cuda", dev_idx) if backend == "nccl" else torch.device("cpu")  # (group: _init_distributed)
    pol = Polisher(yaks, device=dev_idx)
    bam = np2io.Bam(a.bam)
    contigs = list(np2io.read_fasta(a.fa))  # every rank reads the assembly (cheap next to the BAM)
    records = {}
    for name, seq in contigs:  # main.rs:1707-1711, for every contig whichever way it is polished
        if len(seq) >= 0xFFFFFFFF:
            raise SystemExit(f"{name} is too long!")
    long_ones = [i for i, (_, seq) in enumerate(contigs) if len(seq) >= max(a.min_ctg_len, a.shard_min_len)]
    whole = [i for i, (_, seq) in enumerate(contigs) if len(seq) >= a.min_ctg_len and i not in set(long_ones)]
    # 1. long contigs: one reference interval per rank
    for i in long_ones:
        name, seq = contigs[i]
        try:  # every rank parses only the BAM records overlapping its interval +- halo; the pieces meet on rank 0
            b, p, span = polish_sharded_bam(pol, bam, name, seq, opts, fopts, halo=a.shard_halo, device=xdev,
                                            want_pos=a.out_pos, dst=0, with_span=True)
        except ShardMismatch as e:
            # raised on every rank alike (every exchange carries the ranks' status; the strips around the cuts are seen by
            # all): the shards disagree, a splice cursor got stuck inside one (NP2_E_UNSUPPORTED), or one failed for a
            # reason of its own -> rank 0 polishes the contig unsharded (and reports a genuine error itself)
            print(f"[WARN] {name}: {e}; polishing it unsharded", file=sys.stderr)
            b, fb_err = None, None
            if rank == 0:
                try:
                    c = np2io.contig_from_bam(pol, bam, name, seq, fopts)
                    try:
                        b, p = pol.polish_resident(c, opts, want_pos=a.out_pos)
                        span = (int(p[0]), int(p[-1])) if a.out_pos else (p[0], p[1])
                    finally:
                        c.free()
                except Exception as e2:  # noqa: BLE001 — the other ranks wait below: they must learn of it
                    fb_err = e2
            # a genuine error on rank 0 ends the run on every rank (nobody is left waiting in the next contig's collective)
            verdict = all_gather_arrays(np.array([0 if fb_err is None else 1], dtype=np.uint8), device=xdev)
            if int(verdict[0][0]):
                dist.destroy_process_group()
                raise SystemExit(f"Error: {name}: {fb_err if fb_err is not None else 'the unsharded fallback failed on rank 0'}")
        if rank == 0:
            records[i] = _record(a, name, np.asarray(b).tobytes(), int(span[0]), int(span[1]), p if a.out_pos else None)
    # 2. the other contigs: whole, one rank each, longest first
    mine = assign_contigs([len(contigs[i][1]) for i in whole], world)[rank]
    local = []
    for k in mine:
        i = whole[k]
        name, seq = contigs[i]
        c = np2io.contig_from_bam(pol, bam, name, seq, fopts)
        try:
            bases, pos = pol.polish_resident(c, opts, want_pos=a.out_pos)
        finally:
            c.free()
        b = bases.tobytes()
        local.append((i, _record(a, name, b, *( (int(pos[0]), int(pos[-1]), pos) if a.out_pos else (pos[0], pos[1], None)))))
    got = all_gather_sequences(local, device=xdev)
    if rank == 0:
        records.update(got)
        for i, (name, seq) in enumerate(contigs):
            if i in records:
                out.write(records[i])
            else:  # pass-through (main.rs:1727-1730)
                s_ = seq.upper() if a.uppercase else seq
                out.write(_record(a, name, s_, 0, len(seq) - 1, range(len(seq))))
        out.flush()
        if out is not sys.stdout.buffer:
            out.close()
        print(resource_str(t0, ["nextPolish2"] + argv), file=sys.stderr)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    t0 = time.time()
    cpu0 = _cpu_seconds()  # (main() may run inside a longer-lived process: report this call's CPU time, not the process's)
    a = build_parser().parse_args(argv)
    if a.model.lower() not in ("ref", "len"):
        raise SystemExit("error: invalid value for --model (ref|len)")
    for y in a.yak:  # (before the output file exists: a broken dump must not leave a partial output behind)
        try:
            np2io.check_yak_header(y)
        except (ValueError, OSError) as e:  # (a malformed dump, or one that cannot be read at all)
            raise SystemExit(f"Error: {e}")
    out = sys.stdout.buffer
    distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ
    if distributed:
        # under torchrun only rank 0 writes, and whether it may is settled together (a rank that left on its own before
        # the process group exists would leave the others waiting for it): _main_distributed opens the file
        out = None
    elif a.out is not None and a.out != "stdout":  # option.rs:76-79: the literal default "stdout" means stdout
        path = os.path.abspath(a.out)
        if os.path.exists(path):  # option.rs:312-316: refuse to overwrite
            raise SystemExit(f"Error: {path!r} already exists!")
        out = open(path, "wb")
    prof = os.environ.get("NP2_CLI_PROFILE")
    if a.device is None:
        a.device = int(os.environ.get("LOCAL_RANK", "0")) if distributed else 0
    if distributed:
        import torch
        from .dist import note_ranks_per_device
        note_ranks_per_device(int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))), torch.cuda.device_count())
        out = _init_distributed(a)  # process group first; the output-file verdict is rank 0's, shared with everyone
    # main.rs:1547 compares the raw option with "ref" (case-sensitive)
    opts = Opts(min_kmer_count=a.min_kmer_count, max_indel_len=a.max_indel_len, iter_count=a.iter_count,
                model=a.model, use_all_reads=a.use_all_reads)
    map_len, map_fra = split_map_len(a.min_map_len)
    fopts = np2io.FrontOpts(min_read_len=a.min_read_len, min_map_len=map_len,
                            min_map_fra=map_fra, min_map_qual=a.min_map_qual,
                            max_clip_len=a.max_clip_len, use_supplementary=a.use_supplementary,
                            use_secondary=a.use_secondary)

    def load_yaks():
        t_y = time.time()
        with ThreadPoolExecutor(max_workers=max(1, len(a.yak))) as ex:  # (the loader runs outside the GIL: one thread per dump)
            ys = sorted(ex.map(np2io.load_yak, a.yak), key=lambda y: y.k)  # option.rs:238
        if prof:
            print(f"[np2 profile] yak files loaded {time.time() - t_y:.3f} s (at +{time.time() - t0:.3f} s)", file=sys.stderr)
        return ys

    if distributed:
        return _main_distributed(a, argv, t0, out, load_yaks(), opts, fopts)

    # Three stages, each on its own threads, a contig moving through them in input order:
    #   front end  (a.thread capped at 2 threads, each with a table-less context and its own BAM handle): BGZF inflate +
    #              record walk (host pool or device), admission, H2D, GPU columnariser -> the contig's pileup resident in HBM;
    #   polish     (up to 2 workers, each with a batch driver over contexts sharing ONE copy of the k-mer tables): a worker
    #              takes the next contig in input order AND every following one whose pileup is already resident (up to 16
    #              contigs / 64 Mb) and polishes them as ONE batch — one launch per pipeline step for all of them
    #              (np2_batch_polish): the contigs of a many-contig assembly cost a few ms together instead of 2 - 3 ms each;
    #              a long contig goes alone;
    #   output     (this thread): records written in input order.
    # The k-mer dumps are loaded and their HBM tables built next to the first front ends — a resident pileup does not
    # depend on them —, so a run starts reading alignments at once instead of after the tables (main.rs:1698-1853: the
    # reference's reader / workers / writer threads around two bounded channels).
    # (two of each: a front end already spreads its inflate over the host pool, and two polish workers keep the GPU busy
    # through each other's host phases)
    # (NP2_CLI_FRONT / NP2_CLI_WORKERS: experiments with other splits; -t beyond 2 keeps 2 + 2, see above)
    n_workers = int(os.environ.get("NP2_CLI_WORKERS", max(1, min(2, a.thread))))
    n_front = int(os.environ.get("NP2_CLI_FRONT", max(1, min(2, a.thread))))
    BATCH_SLOTS, BATCH_BP = int(os.environ.get("NP2_CLI_BATCH", "16")), 64_000_000
    tls = threading.local()
    base, base_lock = [], threading.Lock()
    yak_pool = ThreadPoolExecutor(max_workers=1)
    base_future = []

    def build_base():
        # dumps -> HBM tables in one go (np2_ctx_create_from_files: each file streamed to the device as it is read, one
        # host thread per dump); the device is not touched before a contig needs polishing
        t_b = time.time()
        if prof:
            print(f"[np2 profile] k-mer dumps: start at +{t_b - t0:.3f} s", file=sys.stderr)
        pol = np2io.polisher_from_yak_files(a.yak, device=a.device)
        if prof:
            print(f"[np2 profile] k-mer dumps read + tables in HBM {time.time() - t_b:.3f} s (at +{time.time() - t0:.3f} s)", file=sys.stderr)
        return pol

    def front(name, seq):
        """the contig's pileup, resident in HBM (np2_contig_from_bam) — on this thread's table-less context"""
        if getattr(tls, "fpol", None) is None:
            t_c = time.time()
            tls.fpol = Polisher([], device=a.device)
            t_b = time.time()
            tls.bam = np2io.Bam(a.bam)
            if prof:
                print(f"[np2 profile] front-end thread: context {1e3 * (t_b - t_c):.1f} ms, BAM + index opened {1e3 * (time.time() - t_b):.1f} ms "
                      f"(at +{time.time() - t0:.3f} s)", file=sys.stderr)
        t_f = time.time()
        c = np2io.contig_from_bam(tls.fpol, tls.bam, name, seq, fopts)
        if prof:
            print(f"[np2 profile] {name}: front end {1e3 * (time.time() - t_f):.1f} ms (done at +{time.time() - t0:.3f} s)", file=sys.stderr)
        return c

    def record(name, bases, pos):
        b = bases.tobytes()
        if a.uppercase:
            b = b.upper()
        if a.out_pos:
            return b"".join(b"%s\t%c\t%d\n" % (name.encode(), b[i:i + 1], int(pos[i])) for i in range(len(b)))
        return b">%s start:%d end:%d\n%s\n" % (name.encode(), pos[0], pos[1], b)

    from collections import deque
    from concurrent.futures import Future
    from .api import BatchPolisher
    todo, todo_cv = deque(), threading.Condition()  # (name, length, front-end future, result future) in input order
    closing = []

    def ensure_batch():
        if getattr(tls, "batch", None) is None:
            b0 = base_future[0].result()
            with base_lock:  # one copy of the k-mer tables in HBM: the other worker's contexts share it
                tls.pol = b0 if not base else b0.clone()
                base.append(tls.pol)
            t_b = time.time()
            tls.batch = BatchPolisher(tls.pol, BATCH_SLOTS)
            if prof:
                print(f"[np2 profile] polish worker: batch driver with {BATCH_SLOTS} slot contexts {1e3 * (time.time() - t_b):.1f} ms "
                      f"(at +{time.time() - t0:.3f} s)", file=sys.stderr)

    def polish_worker():
        """takes the head of `todo` and every following contig that is already resident; one batch per turn"""
        while True:
            with todo_cv:
                while not todo and not closing:
                    todo_cv.wait()
                if not todo:
                    return
                items = [todo.popleft()]
            try:
                # this worker's batch driver first (it waits for the k-mer tables, not for a pileup: made while the first front
                # ends are still reading — 7-9 ms that used to sit between the first resident contig and its polish)
                ensure_batch()
            except BaseException:
                pass  # (reported below, by the polish that needs it)
            try:
                items[0][2].exception()  # (waits for the head's front end)
            except BaseException:
                pass
            with todo_cv:
                bp = items[0][1]
                while todo and len(items) < BATCH_SLOTS and todo[0][2].done() and bp + todo[0][1] <= BATCH_BP:
                    items.append(todo.popleft())
                    bp += items[-1][1]
            live, contigs = [], []
            for it in items:  # a contig whose front end failed reports that, in its place of the output order
                e = it[2].exception()
                if e is not None:
                    it[3].set_exception(e)
                else:
                    live.append(it)
                    contigs.append(it[2].result())
            try:
                if not live:
                    continue
                ensure_batch()
                t_p = time.time()
                try:
                    res = tls.batch.polish(contigs, opts, want_pos=a.out_pos)
                except Exception:
                    if len(contigs) == 1:
                        raise
                    # one contig of the batch failed: each on its own, so that the records before the failing one (in input
                    # order) are still written, as when every contig was polished by itself
                    res = []
                    for it, c in zip(live, contigs):
                        try:
                            res.append(tls.batch.polish([c], opts, want_pos=a.out_pos)[0])
                        except Exception as e1:
                            it[3].set_exception(e1)
                            res.append(None)
                if prof:
                    print(f"[np2 profile] {', '.join(it[0] for it in live)}: polished as one batch in {1e3 * (time.time() - t_p):.1f} ms "
                          f"(done at +{time.time() - t0:.3f} s)", file=sys.stderr)
                for it, r in zip(live, res):
                    if r is not None:
                        it[3].set_result(record(it[0], r[0], r[1]))
            except BaseException as e:  # (the writer re-raises it in input order)
                for it in live:
                    if not it[3].done():
                        it[3].set_exception(e)
            finally:
                for c in contigs:
                    c.free()

    workers = []
    try:
        with ThreadPoolExecutor(max_workers=n_front) as fpool:
            workers = [threading.Thread(target=polish_worker, name=f"np2-polish-{w}", daemon=True) for w in range(n_workers)]
            for w in workers:
                w.start()
            pending, pending_len = [], []  # records in input order: bytes or futures; the contigs' lengths

            def drain(keep):
                while len(pending) > keep:
                    rec = pending.pop(0)
                    pending_len.pop(0)
                    out.write(rec if isinstance(rec, bytes) else rec.result())

            try:
                for name, seq in np2io.read_fasta(a.fa):
                    if len(seq) >= 0xFFFFFFFF:
                        raise SystemExit(f"{name} is too long!")
                    if len(seq) < a.min_ctg_len:  # pass-through (main.rs:1727-1730)
                        s = seq.upper() if a.uppercase else seq
                        if a.out_pos:
                            pending.append(b"".join(b"%s\t%c\t%d\n" % (name.encode(), s[i:i + 1], i) for i in range(len(s))))
                        else:
                            pending.append(b">%s start:0 end:%d\n%s\n" % (name.encode(), len(seq) - 1, s))
                        pending_len.append(len(seq))
                    else:
                        if not base_future:  # the first contig to polish: tables into HBM next to its front end
                            base_future.append(yak_pool.submit(build_base))
                        res = Future()
                        with todo_cv:
                            todo.append((name, len(seq), fpool.submit(front, name, seq), res))
                            todo_cv.notify()
                        pending.append(res)
                        pending_len.append(len(seq))
                    # bounded look-ahead: that many contigs held in memory at most — and at most ~2 Gb of contig in flight
                    # (a resident 30x pileup is ~16 bytes per base of HBM: three chromosomes, not six); short contigs may
                    # queue up to fill the workers' batches
                    keep = n_workers * BATCH_SLOTS + n_front
                    while keep > 1 and sum(pending_len[-keep:]) > 2_000_000_000:
                        keep -= 1
                    drain(keep)
                drain(0)
            finally:
                with todo_cv:
                    closing.append(True)
                    todo_cv.notify_all()
                for w in workers:
                    w.join()
            out.flush()
            if prof:
                print(f"[np2 profile] last record written at +{time.time() - t0:.3f} s", file=sys.stderr)
        if not base_future:
            load_yaks()  # (an assembly of pass-through contigs only: a broken dump must still be reported)
        if prof:
            print(f"[np2 profile] contexts released at +{time.time() - t0:.3f} s", file=sys.stderr)
    finally:
        yak_pool.shutdown(wait=False)
        if out is not None and out is not sys.stdout.buffer:
            out.close()
    print(resource_str(t0, ["nextPolish2"] + argv, cpu0), file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
