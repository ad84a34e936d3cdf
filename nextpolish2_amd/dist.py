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
def stitch_shards(pieces, plans, verify):
    """Join the shards' consensus pieces [(bases, pos)] (each covering [own_lo - verify, own_hi + verify) in contig
    coordinates) into the contig's consensus.  Neighbouring shards computed the `verify` positions on either side of
    their common boundary independently: they must agree base for base there, otherwise the halo was too small for this
    pileup and the caller falls back to the unsharded path."""
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


class ShardMismatch(RuntimeError):
    pass


def polish_sharded_local(polisher, pileup, opts=None, n_shards=2, halo=65536, verify=1024):
    """Polish one contig as n_shards reference intervals inside this process, one np2 context per shard (clones of
    `polisher`: same device, shared k-mer tables): the single-process form of polish_sharded, used by tests and by a
    single-GPU run that wants the shard path.  Returns (bases, pos) of the whole contig."""
    from .api import ShardRun, shard_plan, vote_decide
    plans = shard_plan(pileup, n_shards, halo)
    ctxs = [polisher.clone() for _ in range(n_shards)]  # a run keeps its state in its context's scratch
    runs = [ShardRun(ctxs[k], pileup, plans[k], opts, verify) for k in range(n_shards)]
    try:
        while runs[0].passes_left() > 1:
            votes = [r.vote() for r in runs]
            losers = vote_decide(votes, pileup.n_reads, opts)
            for r in runs:
                r.apply(losers)
        pieces = [r.final() for r in runs]
    finally:
        for r in runs:
            r.close()
    return stitch_shards(pieces, plans, verify)


def all_gather_bytes(raw, device=None, group=None):
    """Every rank's byte string, in rank order."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    got = all_gather_sequences([(rank, raw)], device=device, group=group)
    return [got[r] for r in range(len(got))]


def polish_sharded(polisher, pileup, opts=None, halo=65536, verify=1024, device=None, group=None):
    """Polish one contig cut into world_size reference intervals, one per rank (one process per GPU).

    Every rank holds the contig's host pileup (or at least its own shard's reads) and uploads only its shard.  Per
    phasing pass the ranks all-gather their votes (pair counts of the HETE regions they own, per-read vote records —
    a few MB per Mb of diploid contig) and each runs the contig-wide decision on the merged votes (host only,
    deterministic: no broadcast needed); the removed reads are applied to every shard.  The polished pieces are
    all-gathered (RCCL over xGMI with backend nccl) and stitched; every rank returns the whole contig's (bases, pos)."""
    from .api import ShardRun, Vote, shard_plan, vote_decide
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    plans = shard_plan(pileup, world, halo)
    run = ShardRun(polisher, pileup, plans[rank], opts, verify)
    try:
        while run.passes_left() > 1:
            raws = all_gather_bytes(run.vote().to_bytes(), device=device, group=group)
            losers = vote_decide([Vote.from_bytes(x) for x in raws], pileup.n_reads, opts)
            run.apply(losers)
        b, p = run.final()
    finally:
        run.close()
    hdr = np.array([len(b)], dtype=np.uint64).tobytes()
    raws = all_gather_bytes(hdr + np.asarray(b).tobytes() + np.asarray(p).tobytes(), device=device, group=group)
    pieces = []
    for x in raws:
        n = int(np.frombuffer(x[:8], dtype=np.uint64)[0])
        pieces.append((np.frombuffer(x[8:8 + n], dtype=np.uint8), np.frombuffer(x[8 + n:8 + 5 * n], dtype=np.uint32)))
    return stitch_shards(pieces, plans, verify)


def polish_sharded_bam(polisher, bam, name, ref, opts=None, fopts=None, halo=65536, verify=1024, device=None, group=None):
    """polish_sharded with the input side sharded too: every rank reads only the BAM records overlapping its interval
    +- halo (io.ShardFromBam), the ranks all-gather the file offsets of the pushed records starting in their own
    intervals (that list IS the contig's read numbering), then the shard protocol runs on the resident shards."""
    from . import io as np2io
    from .api import ShardRun, Vote, vote_decide
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    cuts = np2io.shard_cuts(len(ref), world)
    sb = np2io.ShardFromBam(polisher, bam, name, ref, cuts[rank][0], cuts[rank][1], halo, fopts)
    raws = all_gather_bytes(sb.own_offsets.tobytes(), device=device, group=group)
    all_off = np.concatenate([np.frombuffer(x, dtype=np.uint64) for x in raws]) if raws else np.zeros(0, dtype=np.uint64)
    h, plan, n_reads_total = sb.finish(all_off)
    run = ShardRun(polisher, None, plan, opts, verify, resident=h)
    plans_raw = all_gather_bytes(bytes(plan), device=device, group=group)
    plans = [type(plan).from_buffer_copy(x) for x in plans_raw]
    try:
        while run.passes_left() > 1:
            raws = all_gather_bytes(run.vote().to_bytes(), device=device, group=group)
            run.apply(vote_decide([Vote.from_bytes(x) for x in raws], n_reads_total, opts))
        b, p = run.final()
    finally:
        run.close()
    hdr = np.array([len(b)], dtype=np.uint64).tobytes()
    raws = all_gather_bytes(hdr + np.asarray(b).tobytes() + np.asarray(p).tobytes(), device=device, group=group)
    pieces = []
    for x in raws:
        n = int(np.frombuffer(x[:8], dtype=np.uint64)[0])
        pieces.append((np.frombuffer(x[8:8 + n], dtype=np.uint8), np.frombuffer(x[8 + n:8 + 5 * n], dtype=np.uint32)))
    return stitch_shards(pieces, plans, verify)


def polish_sharded_bam_local(polisher, bam_path, name, ref, n_shards, opts=None, fopts=None, halo=65536, verify=1024):
    """The single-process form of polish_sharded_bam (one context + one BAM handle per shard): tests, single-GPU runs."""
    from . import io as np2io
    from .api import ShardRun, vote_decide
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
        while runs[0].passes_left() > 1:
            losers = vote_decide([r.vote() for r in runs], n_reads_total, opts)
            for r in runs:
                r.apply(losers)
        pieces = [r.final() for r in runs]
    finally:
        for r in runs:
            r.close()
    return stitch_shards(pieces, plans, verify)


class _DeviceBytes:
    """Zero-copy view of `n` bytes at device address `ptr` for torch.as_tensor (array-interface protocol)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class SequenceGatherer:
    """Per-step all-gather of one polished contig per rank, kept on the device.

    Buffers are allocated once; a step costs one padded uint8 all-gather (length word + the polished bytes) — on GPUs that is RCCL over xGMI with no host round trip.  The collectives are enqueued on torch's
    current stream and not waited for: they overlap the next contig's kernels, which run on the np2 context's own
    stream.  `gather_device` returns as soon as the context's result buffer has been copied out (device to device), so
    the context is free to start its next contig."""

    HDR = 8  # every rank's slot starts with its sequence length (int64), so one collective moves both

    def __init__(self, capacity, device, group=None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.device = torch.device(device)
        self.cap = (int(capacity) + 7) & ~7
        stride = self.HDR + self.cap
        self.local = torch.zeros(stride, dtype=torch.uint8, device=self.device)
        self.flat = torch.zeros(self.world * stride, dtype=torch.uint8, device=self.device)
        self.bufs = [self.flat[r * stride + self.HDR:(r + 1) * stride] for r in range(self.world)]
        self.lens = [self.flat[r * stride:r * stride + self.HDR].view(torch.int64) for r in range(self.world)]
        self._len_host = torch.zeros(1, dtype=torch.int64)
        self._hdr = self.local[:self.HDR].view(torch.int64)
        self._views = {}  # (device address, length) -> zero-copy tensor (the np2 result buffers are few and stable)
        if self.device.type == "cuda":
            self._len_host = self._len_host.pin_memory()
            self._copied = torch.cuda.Event()

    def _exchange(self):
        if not dist.is_initialized():
            self.flat.copy_(self.local)
        else:
            dist.all_gather_into_tensor(self.flat, self.local, group=self.group)
        return self.bufs, self.lens

    def gather_tensor(self, src):
        """src: 1-D uint8 tensor on this gatherer's device (this rank's polished contig)."""
        n = int(src.shape[0])
        if n > self.cap:
            raise ValueError("polished contig longer than the gather capacity")
        self._len_host[0] = n
        self._hdr.copy_(self._len_host, non_blocking=True)
        self.local[self.HDR:self.HDR + n].copy_(src, non_blocking=True)
        if self.device.type == "cuda":
            self._copied.record()
        out = self._exchange()
        if self.device.type == "cuda":
            self._copied.synchronize()  # src (and the pinned length word) may be reused from here on
        return out

    def gather_device(self, ptr, n):
        """Gather straight from a device buffer (np2_last_result_device): no host round trip."""
        if n > self.cap:
            raise ValueError("polished contig longer than the gather capacity")
        view = self._views.get((ptr, n))
        if view is None:
            if len(self._views) > 16:
                self._views.clear()
            with torch.cuda.device(self.device):
                view = self._views[(ptr, n)] = torch.as_tensor(_DeviceBytes(ptr, n), device=self.device)
        return self.gather_tensor(view)

    def gather(self, bases):
        """bases: 1-D uint8 numpy array on the host. Returns (list of device tensors, lengths)."""
        return self.gather_tensor(torch.from_numpy(np.ascontiguousarray(bases)).to(self.device))

    def to_host(self):
        """{rank: bytes} of the last gather (only for verification / output, not part of the hot loop)."""
        return {r: self.bufs[r][: int(self.lens[r].item())].cpu().numpy().tobytes() for r in range(self.world)}
