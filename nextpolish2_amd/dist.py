"""Multi-GPU sharding of the hot path: one process per GPU, contigs sharded across ranks.

Contigs are fully independent in the reference (one contig per worker thread, src/main.rs:1726-1837),
so the data path needs no collective; the only exchange is the final all-gather of the polished
per-contig sequences (RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests)."""
import numpy as np
import torch
import torch.distributed as dist


def assign_contigs(lengths, world_size):
    """Longest-first greedy assignment of contigs to ranks -> list of contig-index lists per rank.

    Deterministic: ties broken by contig index, least-loaded rank with the smallest rank id wins."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(lengths[i])
    for lst in out:
        lst.sort()
    return out


def all_gather_sequences(local, device=None, group=None):
    """All-gather variable-length byte sequences.

    `local` is a list of (contig_index, bytes-like); every rank returns the same dict
    {contig_index: bytes}.  Payload = one padded uint8 all_gather + one int64 header all_gather."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return {int(i): bytes(s) for i, s in local}
    device = device or torch.device("cpu")
    n_local = len(local)
    hdr = torch.zeros(1, dtype=torch.int64, device=device)
    hdr[0] = n_local
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, hdr, group=group)
    max_n = max(int(c.item()) for c in counts)
    meta = torch.full((max(max_n, 1), 2), -1, dtype=torch.int64, device=device)
    for k, (i, s) in enumerate(local):
        meta[k, 0] = int(i)
        meta[k, 1] = len(s)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    total_local = sum(len(s) for _, s in local)
    tot = torch.tensor([total_local], dtype=torch.int64, device=device)
    tots = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(tots, tot, group=group)
    max_bytes = max(1, max(int(t.item()) for t in tots))
    buf = torch.zeros(max_bytes, dtype=torch.uint8, device=device)
    if total_local:
        cat = np.concatenate([np.frombuffer(bytes(s), dtype=np.uint8) for _, s in local if len(s)])
        buf[:total_local] = torch.from_numpy(cat.copy()).to(device)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    out = {}
    for r in range(world):
        off = 0
        m = metas[r].cpu().numpy()
        b = bufs[r].cpu().numpy()
        for k in range(int(counts[r].item())):
            idx, ln = int(m[k, 0]), int(m[k, 1])
            out[idx] = b[off:off + ln].tobytes()
            off += ln
    return out


# ---- reference-interval sharding of one long contig --------------------------------------------------------------
class ShardMismatch(RuntimeError):
    """The sharded attempt of a contig is abandoned — on every rank alike (the exchanges carry a status word): the
    shards disagree around a cut, or some rank's shard failed.  The caller polishes the contig unsharded."""


def stitch_shards(pieces, plans, verify):
    """Host-side stitcher for pieces from np2_shard_final [(bases, pos)], each covering [own_lo - verify, own_hi + verify)
    in contig coordinates (the form recorded fixtures hold; the product path keeps the pieces on the devices:
    np2_shard_final_device + check_strips + gather_slices)."""
    out_b, out_p = [], []
    for k, ((b, p), pl) in enumerate(zip(pieces, plans)):
        b, p = np.asarray(b), np.asarray(p)
        if k + 1 < len(pieces):
            nb, npos = np.asarray(pieces[k + 1][0]), np.asarray(pieces[k + 1][1])
            lo, hi = max(0, pl.own_hi - verify), pl.own_hi + verify
            m1 = (p >= lo) & (p < hi)
            m2 = (npos >= lo) & (npos < hi)
            if not (np.array_equal(b[m1], nb[m2]) and np.array_equal(p[m1], npos[m2])):
                raise ShardMismatch(f"shards {k} and {k + 1} disagree around position {pl.own_hi}")
        own = (p >= pl.own_lo) & (p < pl.own_hi)
        out_b.append(b[own])
        out_p.append(p[own])
    return np.concatenate(out_b), np.concatenate(out_p)


def all_gather_bytes(raw, device=None, group=None):
    """Every rank's byte string, in rank order."""
    return [x.tobytes() for x in all_gather_arrays(np.frombuffer(raw, dtype=np.uint8), device=device, group=group)]


def all_gather_arrays(arr, device=None, group=None):
    """Every rank's uint8 numpy array, in rank order (variable lengths: one length exchange + one padded all-gather
    from a device buffer with backend nccl = RCCL over xGMI; no Python byte strings on the way)."""
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [arr]
    device = device or torch.device("cpu")
    n = torch.tensor([arr.shape[0]], dtype=torch.int64, device=device)
    lens = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(lens, n, group=group)
    lens = [int(x.item()) for x in lens]
    cap = max(1, max(lens))
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    if arr.shape[0]:
        buf[:arr.shape[0]] = torch.from_numpy(arr).to(device)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    return [bufs[r][:lens[r]].cpu().numpy() for r in range(world)]


def _exchange(payload, failure, device, group, what):
    """All-gather one array per rank behind a status word.  `failure` (an exception or None) is this rank's outcome of
    the phase that produced `payload`; if ANY rank failed, every rank raises ShardMismatch — nobody is left waiting in
    the next collective for a rank that has gone."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    head = np.zeros(8, dtype=np.uint8)
    head[0] = 0 if failure is None else 1
    body = np.zeros(0, dtype=np.uint8) if failure is not None else np.ascontiguousarray(payload, dtype=np.uint8)
    got = all_gather_arrays(np.concatenate([head, body]), device=device, group=group)
    bad = [r for r, g in enumerate(got) if g[0] != 0]
    if bad:
        raise ShardMismatch(f"{what} failed on rank(s) {bad}" + (f": {failure}" if rank in bad else "")) from failure
    return [g[8:] for g in got]


def _decide_on_owner(payload, failure, n_reads_total, opts, device, group, owner=0):
    """The contig-wide decision of one phasing pass, taken ONCE: the ranks' votes are gathered onto `owner` (variable
    lengths: a tiny all-gather of status + length, then one padded gather — RCCL over xGMI with backend nccl), the owner
    merges and decides them (np2_vote_decide: the Louvain of louvain.rs:290-356 on the merged read graph, host only), and
    the reads it removes come back to every rank as a bitmap of n_reads_total bits in one broadcast.  (Every rank deciding
    the same all-gathered votes for itself — round 3 — put the ~130 ms of a chromosome's host vote on every rank's
    critical path and sent 370 MB of votes to eight ranks.)  Any rank's failure — its vote, or the owner's decision —
    reaches every rank as ShardMismatch in the same collectives."""
    from .api import Vote, vote_decide
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    device = device or torch.device("cpu")
    if world == 1:  # (the Vote itself: nothing to pack for nobody)
        if failure is not None:
            raise ShardMismatch(f"phasing vote failed: {failure}") from failure
        return vote_decide([payload if isinstance(payload, Vote) else Vote.unpack(payload)], n_reads_total, opts)
    body = np.zeros(0, dtype=np.uint8) if failure is not None else np.ascontiguousarray(payload, dtype=np.uint8)
    head = torch.tensor([0 if failure is None else 1, body.shape[0]], dtype=torch.int64, device=device)
    heads = [torch.zeros_like(head) for _ in range(world)]
    dist.all_gather(heads, head, group=group)
    heads = [h.cpu().numpy() for h in heads]
    bad = [r for r, h in enumerate(heads) if int(h[0]) != 0]
    if bad:
        raise ShardMismatch(f"phasing vote failed on rank(s) {bad}" + (f": {failure}" if rank in bad else "")) from failure
    lens = [int(h[1]) for h in heads]
    cap = max(1, max(lens))
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    if body.shape[0]:
        buf[:body.shape[0]] = torch.from_numpy(body).to(device)
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == owner else None
    dist.gather(buf, parts, dst=owner, group=group)
    n_bytes = (n_reads_total + 7) // 8
    out = torch.zeros(8 + n_bytes, dtype=torch.uint8, device=device)  # [status byte, 7 x pad][bitmap]
    err = None
    if rank == owner:
        try:
            votes = [Vote.unpack(parts[r][:lens[r]].cpu().numpy()) for r in range(world)]
            losers = vote_decide(votes, n_reads_total, opts)
            bits = np.zeros(n_bytes * 8, dtype=np.uint8)
            bits[losers] = 1
            host = np.concatenate([np.zeros(8, dtype=np.uint8), np.packbits(bits, bitorder="little")])
        except Exception as e:  # noqa: BLE001 — the other ranks wait in the broadcast below: tell them
            err = e
            host = np.concatenate([np.ones(8, dtype=np.uint8), np.zeros(n_bytes, dtype=np.uint8)])
        out.copy_(torch.from_numpy(host).to(device))
    dist.broadcast(out, src=owner, group=group)
    got = out.cpu().numpy()
    if got[0] != 0:
        raise ShardMismatch(f"the contig-wide vote failed on rank {owner}" + (f": {err}" if err is not None else "")) from err
    return np.flatnonzero(np.unpackbits(got[8:], bitorder="little")[:n_reads_total]).astype(np.uint32)


def check_strips(metas, plans):
    """Neighbouring shards computed the `verify` positions on either side of their common cut independently: the high
    strip of shard k must equal the low strip of shard k + 1 base for base and position for position, otherwise the halo
    was too small for this pileup and the contig is polished unsharded."""
    for k in range(len(metas) - 1):
        a, b = metas[k], metas[k + 1]
        if not (np.array_equal(a["hi_bases"], b["lo_bases"]) and np.array_equal(a["hi_pos"], b["lo_pos"])):
            raise ShardMismatch(f"shards {k} and {k + 1} disagree around position {plans[k].own_hi}")


def _piece_meta(pc):
    return {"own_len": pc.own_len, "first_pos": pc.first_pos, "last_pos": pc.last_pos, "lo_bases": pc.lo_bases,
            "lo_pos": pc.lo_pos, "hi_bases": pc.hi_bases, "hi_pos": pc.hi_pos}


def _run_local(runs, plans, n_reads_total, opts, want_pos):
    """Shard protocol of one process driving every shard (one context each): votes merged per phasing pass, final pass
    with the results left on the devices, strips checked, owned slices fetched straight into their places of ONE pinned
    host array."""
    from .api import pinned_array, vote_decide
    while runs[0].passes_left() > 1:
        losers = vote_decide([r.vote() for r in runs], n_reads_total, opts)
        for r in runs:
            r.apply(losers)
    pieces = [r.final_device() for r in runs]
    check_strips([_piece_meta(pc) for pc in pieces], plans)
    total = sum(pc.own_len for pc in pieces)
    out_b = pinned_array(total, np.uint8)
    out_p = pinned_array(total, np.uint32) if want_pos else None
    off = 0
    for r, pc in zip(runs, pieces):
        if pc.own_len:
            r.fetch(out_b[off:off + pc.own_len], out_p[off:off + pc.own_len] if want_pos else None)
        off += pc.own_len
    return out_b, out_p


def polish_sharded_local(polisher, pileup, opts=None, n_shards=2, halo=65536, verify=1024, want_pos=True):
    """Polish one contig as n_shards reference intervals inside this process, one np2 context per shard (clones of
    `polisher`: same device, shared k-mer tables): the single-process form of polish_sharded, used by tests and by a
    single-GPU run that wants the shard path.  Returns (bases, pos) of the whole contig (pos None unless want_pos)."""
    from .api import ShardRun, shard_plan
    plans = shard_plan(pileup, n_shards, halo)
    ctxs = [polisher.clone() for _ in range(n_shards)]  # a run keeps its state (and its result) in its context's scratch
    runs = [ShardRun(ctxs[k], pileup, plans[k], opts, verify) for k in range(n_shards)]
    try:
        return _run_local(runs, plans, pileup.n_reads, opts, want_pos)
    finally:
        for r in runs:
            r.close()


def _device_view(ptr, n, device, dtype=torch.uint8):
    """Zero-copy tensor over `n` elements of device memory at `ptr` (a shard's owned slice)."""
    typestr = {torch.uint8: "|u1", torch.int32: "<i4"}[dtype]
    iface = type("_Dev", (), {})()
    iface.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    with torch.cuda.device(device):
        return torch.as_tensor(iface, device=device)


def gather_slices(run, pc, lens, want_pos, device=None, group=None, dst=0):
    """The shards' owned slices, end to end, on rank `dst` (None: on every rank).  With a CUDA device the slices travel
    from the np2 result buffers on the devices (RCCL over xGMI: gather to `dst`, or all-gather), with a CPU device
    (gloo: tests) through host memory.  Returns (bases, pos) on the receiving rank(s), (None, None) elsewhere."""
    from .api import pinned_array
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    total, cap = sum(lens), max(1, max(lens))
    on_gpu = device is not None and torch.device(device).type == "cuda"
    sub_lo = int(run.plan.sub_lo)
    if world == 1:  # one rank: the owned slice goes straight from the device into the pinned result arrays
        b = pinned_array(total, np.uint8)
        p = pinned_array(total, np.uint32) if want_pos else None
        if pc.own_len:
            run.fetch(b, p)
        return b, p

    if on_gpu:
        from .api import lib
        lib().np2_trim_device_cache()  # (torch is about to allocate gather buffers: idle np2 blocks go back to the driver)

    def one(kind):
        dt_t, dt_n = (torch.uint8, np.uint8) if kind == "bases" else (torch.int32, np.uint32)
        n = pc.own_len
        if on_gpu:
            src = torch.zeros(cap, dtype=dt_t, device=device)
            if n:
                v = _device_view(pc.dev_bases if kind == "bases" else pc.dev_pos, n, device, dt_t)
                src[:n] = v if kind == "bases" else v + sub_lo  # (device positions are sub-contig coordinates)
        else:
            host = np.zeros(cap, dtype=dt_n)
            if n:
                hb = np.empty(n, dtype=np.uint8)
                hp = np.empty(n, dtype=np.uint32) if kind == "pos" else None
                run.fetch(hb, hp)
                host[:n] = hb if kind == "bases" else hp
            src = torch.from_numpy(host.view(np.int32) if kind == "pos" else host)
        want = dst is None or rank == dst
        if world == 1:
            parts = [src]
        elif dst is None:
            parts = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(parts, src, group=group)
        else:
            parts = [torch.empty_like(src) for _ in range(world)] if rank == dst else None
            dist.gather(src, parts, dst=dst, group=group)
        if not want:
            return None
        out = pinned_array(total, dt_n)
        off = 0
        for r in range(world):
            if lens[r]:
                t = torch.from_numpy(out[off:off + lens[r]].view(np.int32) if kind == "pos" else out[off:off + lens[r]])
                t.copy_(parts[r][:lens[r]])
            off += lens[r]
        return out
    b = one("bases")
    p = one("pos") if want_pos else None
    return b, p


PHASE_MS = {}  # wall clock per phase of the last _run_ranked on this rank (NP2_DIST_PROFILE=1: bench.py reports it)


def _run_ranked(run, plans, n_reads_total, opts, want_pos, device, group, dst):
    """Shard protocol of one rank among `world` (one process per GPU): every phase's exchange carries the ranks' status."""
    import os
    import time
    from .api import ShardPiece
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    prof = os.environ.get("NP2_DIST_PROFILE") is not None
    PHASE_MS.clear()
    t_last = [time.perf_counter()]

    def lap(name):
        if prof:
            now = time.perf_counter()
            PHASE_MS[name] = PHASE_MS.get(name, 0.0) + (now - t_last[0]) * 1e3
            t_last[0] = now
    apply_err = None
    # The number of phasing passes is fixed BEFORE the loop (it is opts.iter_count - 1 on every rank): a rank whose apply()
    # failed has not advanced its pass counter, and a loop on passes_left() would send it back into _decide_on_owner
    # (all-gather of a 2-word head) while the others enter the final _exchange (1-word length): mismatched collectives.
    for _ in range(max(0, run.passes_left() - 1)):
        err, payload = apply_err, None
        try:
            if err is None:
                payload = getattr(run, "vote_view", run.vote)()  # (borrowed arrays: packed or decided before the run goes on)
                lap("vote_pass")
                if world > 1:
                    payload = payload.pack()
                    lap("vote_pack")
        except Exception as e:  # noqa: BLE001 — any failure of this rank's shard ends the sharded attempt everywhere
            err = e
        losers = _decide_on_owner(payload, err, n_reads_total, opts, device, group)
        lap("vote_decide")
        try:
            run.apply(losers)
        except Exception as e:  # noqa: BLE001 — reported with the next exchange's status word: nobody waits for this rank
            apply_err = e
        lap("apply")
    err, pc = apply_err, None
    try:
        if err is None:
            pc = run.final_device()
    except Exception as e:  # noqa: BLE001
        err = e
    lap("final_pass")
    raws = _exchange(pc.strips() if pc is not None else None, err, device, group, "final pass")
    metas = [ShardPiece.unpack_strips(x) for x in raws]
    check_strips(metas, plans)  # (every rank sees the same strips: the same verdict everywhere)
    lap("strips")
    b, p = gather_slices(run, pc, [m["own_len"] for m in metas], want_pos, device=device, group=group, dst=dst)
    lap("gather_slices")
    full = [m for m in metas if m["own_len"]]
    span = (full[0]["first_pos"], full[-1]["last_pos"]) if full else (0, 0)
    return b, p, span


def polish_sharded(polisher, pileup, opts=None, halo=65536, verify=1024, device=None, group=None, want_pos=True, dst=None,
                   with_span=False, plans=None, resident=None):
    """Polish one contig cut into world_size reference intervals, one per rank (one process per GPU).

    Every rank holds the contig's host pileup (or at least its own shard's reads) and uploads only its shard.  Per
    phasing pass the ranks' votes (pair counts of the HETE regions they own, per-read vote records — a few MB per Mb of
    diploid contig) are gathered onto rank 0, which takes the contig-wide decision on the merged votes once (host only)
    and broadcasts the removed reads as a bitmap (_decide_on_owner); they are applied to every shard.  The final pass leaves each
    shard's polished sub-contig on its device; the ranks exchange the short strips around the cuts (checked on every
    rank) and the owned slices are gathered from the device buffers (RCCL over xGMI with backend nccl) onto rank `dst`
    (None: every rank).  Returns (bases, pos) there — pos None unless want_pos — and (None, None) on the other ranks."""
    from .api import ShardRun, shard_plan
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    plans = plans or shard_plan(pileup, world, halo)
    run, err = None, None
    try:  # (resident: this rank's shard is in HBM already — api.upload_shard — and stays there)
        run = ShardRun(polisher, pileup if resident is None else None, plans[rank], opts, verify, resident=resident,
                       own_contig=resident is None)
    except Exception as e:  # noqa: BLE001
        err = e
    try:
        _exchange(np.zeros(0, dtype=np.uint8), err, device, group, "shard upload")
        b, p, span = _run_ranked(run, plans, pileup.n_reads, opts, want_pos, device, group, dst)
        return (b, p, span) if with_span else (b, p)
    finally:
        if run is not None:
            run.close()


def polish_sharded_bam(polisher, bam, name, ref, opts=None, fopts=None, halo=65536, verify=1024, device=None, group=None,
                       want_pos=True, dst=None, with_span=False):
    """polish_sharded with the input side sharded too: every rank reads only the BAM records overlapping its interval
    +- halo (io.ShardFromBam), the ranks all-gather the file offsets of the pushed records starting in their own
    intervals (that list IS the contig's read numbering), then the shard protocol runs on the resident shards."""
    from . import io as np2io
    from .api import ShardRun
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    cuts = np2io.shard_cuts(len(ref), world)
    sb, err = None, None
    try:
        sb = np2io.ShardFromBam(polisher, bam, name, ref, cuts[rank][0], cuts[rank][1], halo, fopts)
    except Exception as e:  # noqa: BLE001
        err = e
    try:
        raws = _exchange(sb.own_offsets.view(np.uint8) if sb is not None else None, err, device, group, "shard input")
    except ShardMismatch:
        if sb is not None:
            sb.abort()
        raise
    all_off = np.concatenate([np.frombuffer(x, dtype=np.uint64) for x in raws]) if raws else np.zeros(0, dtype=np.uint64)
    run, err, plan, n_reads_total = None, None, None, 0
    try:
        h, plan, n_reads_total = sb.finish(all_off)
        run = ShardRun(polisher, None, plan, opts, verify, resident=h)
    except Exception as e:  # noqa: BLE001
        err = e
    try:
        plans_raw = _exchange(np.frombuffer(bytes(plan), dtype=np.uint8) if plan is not None else None, err, device, group, "shard setup")
        plans = [type(plan).from_buffer_copy(x.tobytes()) for x in plans_raw]
        b, p, span = _run_ranked(run, plans, n_reads_total, opts, want_pos, device, group, dst)
        return (b, p, span) if with_span else (b, p)
    finally:
        if run is not None:
            run.close()


def polish_sharded_bam_local(polisher, bam_path, name, ref, n_shards, opts=None, fopts=None, halo=65536, verify=1024,
                             want_pos=True):
    """The single-process form of polish_sharded_bam (one context + one BAM handle per shard): tests, single-GPU runs."""
    from . import io as np2io
    from .api import ShardRun
    cuts = np2io.shard_cuts(len(ref), n_shards)
    ctxs = [polisher.clone() for _ in range(n_shards)]
    bams = [np2io.Bam(bam_path) for _ in range(n_shards)]
    sbs = [np2io.ShardFromBam(ctxs[k], bams[k], name, ref, cuts[k][0], cuts[k][1], halo, fopts) for k in range(n_shards)]
    all_off = np.concatenate([sb.own_offsets for sb in sbs])
    fin = [sb.finish(all_off) for sb in sbs]
    plans = [f[1] for f in fin]
    n_reads_total = fin[0][2]
    runs = [ShardRun(ctxs[k], None, plans[k], opts, verify, resident=fin[k][0]) for k in range(n_shards)]
    try:
        return _run_local(runs, plans, n_reads_total, opts, want_pos)
    finally:
        for r in runs:
            r.close()


class _DeviceBytes:
    """Zero-copy view of `n` bytes at device address `ptr` for torch.as_tensor (array-interface protocol)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class SequenceGatherer:
    """Per-step all-gather of one polished contig — or one polished assembly — per rank, kept on the device.

    Buffers are allocated once; a step costs one padded uint8 all-gather (length word + the polished bytes) — on GPUs
    that is RCCL over xGMI with no host round trip.  The collectives are enqueued on torch's current stream and not
    waited for: they overlap the next contig's kernels, which run on the np2 context's own stream.  `gather_device`
    returns as soon as the context's result buffer has been copied out (device to device), so the context is free to
    start its next contig.

    An assembly of several contigs (the batch driver's output) is gathered from the device as well: every contig has a
    fixed slot in the rank's staging buffer (`set_slots`), `stage` copies a contig's result buffer (np2_last_result_device
    of its slot context) into its slot as soon as its batch group has delivered it — device to device, any thread —,
    and `gather_staged` sends the staging buffer with the table of the contigs' lengths in front.  `n_local` staging
    buffers let the groups stage step k + 1 while step k is still being sent.

    `collective_device`: where the tensors handed to torch.distributed live — the GPU for backend nccl (default), the
    host for gloo (several ranks on one GPU: tests; the staging buffer then crosses PCIe once per step)."""

    HDR = 8  # every rank's slot starts with its sequence length (int64), so one collective moves both

    def __init__(self, capacity, device, group=None, collective_device=None, n_local=1):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.device = torch.device(device)
        self.cdev = torch.device(collective_device) if collective_device is not None else self.device
        self.cap = (int(capacity) + 7) & ~7
        stride = self.stride = self.HDR + self.cap
        self.locals = [torch.zeros(stride, dtype=torch.uint8, device=self.device) for _ in range(max(1, n_local))]
        self.local = self.locals[0]
        self.flat = torch.zeros(self.world * stride, dtype=torch.uint8, device=self.cdev)
        self.bufs = [self.flat[r * stride + self.HDR:(r + 1) * stride] for r in range(self.world)]
        self.lens = [self.flat[r * stride:r * stride + self.HDR].view(torch.int64) for r in range(self.world)]
        self._len_host = torch.zeros(1, dtype=torch.int64)
        self._views = {}  # (device address, length) -> zero-copy tensor (the np2 result buffers are few and stable)
        self._slot_off = None
        if self.device.type == "cuda":
            self._len_host = self._len_host.pin_memory()
            self._copied = torch.cuda.Event()

    def _exchange(self, local=None):
        local = self.local if local is None else local
        if not dist.is_initialized():
            self.flat.copy_(local)
        elif self.cdev == self.device:
            dist.all_gather_into_tensor(self.flat, local, group=self.group)
        else:  # (gloo: through host memory)
            parts = [self.flat[r * self.stride:(r + 1) * self.stride] for r in range(self.world)]
            dist.all_gather(parts, local.to(self.cdev), group=self.group)
        return self.bufs, self.lens

    def gather_tensor(self, src):
        """src: 1-D uint8 tensor on this gatherer's device (this rank's polished contig)."""
        n = int(src.shape[0])
        if n > self.cap:
            raise ValueError("polished contig longer than the gather capacity")
        self._len_host[0] = n
        self.local[:self.HDR].view(torch.int64).copy_(self._len_host, non_blocking=True)
        self.local[self.HDR:self.HDR + n].copy_(src, non_blocking=True)
        if self.device.type == "cuda":
            self._copied.record()
        out = self._exchange()
        if self.device.type == "cuda":
            self._copied.synchronize()  # src (and the pinned length word) may be reused from here on
        return out

    def _view(self, ptr, n):
        view = self._views.get((ptr, n))
        if view is None:
            if len(self._views) > 256:
                self._views.clear()
            with torch.cuda.device(self.device):
                view = self._views[(ptr, n)] = torch.as_tensor(_DeviceBytes(ptr, n), device=self.device)
        return view

    def gather_device(self, ptr, n):
        """Gather straight from a device buffer (np2_last_result_device): no host round trip."""
        if n > self.cap:
            raise ValueError("polished contig longer than the gather capacity")
        return self.gather_tensor(self._view(ptr, n))

    # ---- an assembly of several contigs, staged on the device contig by contig ----
    def set_slots(self, capacities):
        """Contig i of this rank's assembly owns `capacities[i]` bytes of the staging buffer (behind the length table)."""
        caps = [(int(c) + 7) & ~7 for c in capacities]
        self._slot_cap = caps
        self._table = 8 * len(caps)
        self._slot_off = [self.HDR + self._table + int(x) for x in np.concatenate([[0], np.cumsum(caps)[:-1]])]
        if self._table + sum(caps) > self.cap:
            raise ValueError("slots exceed the gather capacity")
        self._tab_host = torch.zeros(len(caps), dtype=torch.int64)
        if self.device.type == "cuda":
            self._tab_host = self._tab_host.pin_memory()

    def slot_address(self, which, i):
        """(device address, capacity) of contig i's slot in staging buffer `which`: what np2_batch_set_sink takes — the batch
        driver then copies a contig's polished bases there itself, inside the polish call (no `stage`)."""
        return self.locals[which].data_ptr() + self._slot_off[i], self._slot_cap[i]

    def stage(self, which, items):
        """items: [(contig index, device address, length)] — copied device to device into staging buffer `which`; returns
        when the copies are done (the result buffers may be overwritten by the next polish call)."""
        local = self.locals[which]
        ev = None
        for i, ptr, n in items:
            if n > self._slot_cap[i]:
                raise ValueError("polished contig longer than its gather slot")
            if n:
                local[self._slot_off[i]:self._slot_off[i] + n].copy_(self._view(ptr, n), non_blocking=True)
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()

    def gather_staged(self, which, lengths):
        """Send staging buffer `which`: lengths[i] = polished length of contig i (the table in front of the slots)."""
        local = self.locals[which]
        self._len_host[0] = self._table + sum(self._slot_cap)
        self._tab_host.copy_(torch.as_tensor(np.asarray(lengths, dtype=np.int64)))
        local[:self.HDR].view(torch.int64).copy_(self._len_host, non_blocking=True)
        local[self.HDR:self.HDR + self._table].view(torch.int64).copy_(self._tab_host, non_blocking=True)
        if self.device.type == "cuda":
            self._copied.record()
        out = self._exchange(local)
        if self.device.type == "cuda":
            self._copied.synchronize()
        return out

    def gather(self, bases):
        """bases: 1-D uint8 numpy array on the host. Returns (list of device tensors, lengths)."""
        return self.gather_tensor(torch.from_numpy(np.ascontiguousarray(bases)).to(self.device))

    def to_host(self):
        """{rank: bytes} of the last gather (only for verification / output, not part of the hot loop): the rank's
        sequence — or, after gather_staged, its contigs' sequences end to end (slots are as on this rank: the ranks of a
        weak-scaling run hold assemblies of the same contig lengths)."""
        out = {}
        for r in range(self.world):
            raw = self.bufs[r][: int(self.lens[r].item())].cpu().numpy()
            if self._slot_off is None:
                out[r] = raw.tobytes()
                continue
            tab = raw[:self._table].view(np.int64)
            out[r] = b"".join(raw[o - self.HDR:o - self.HDR + int(n)].tobytes() for o, n in zip(self._slot_off, tab))
        return out


def note_ranks_per_device(n_local_ranks, n_devices):
    """Tell the native side that this process shares its GPU with other ranks (rehearsals of the multi-rank paths on a box
    with fewer GPUs than ranks): the dense pass then takes its chunks' column counts from a pre-pass instead of having
    chunks wait for lower-numbered blocks — a wait that a queue preempted in favour of another process can leave
    hanging (csrc/np2_lookback.hpp, launch_chunk_counts).  Must run before the first contig is polished."""
    import os
    if n_devices > 0 and n_local_ranks > n_devices:
        os.environ.setdefault("NP2_DENSE_PRECOUNT", "1")
