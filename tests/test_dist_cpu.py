"""World-size-2 gloo test of the sharding helpers used by bench.py --gpus N (CPU, no GPU needed)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nextpolish2_amd.dist import SequenceGatherer, all_gather_sequences, assign_contigs


def test_assign_contigs_is_balanced_and_deterministic():
    lens = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57]
    a = assign_contigs(lens, 8)
    assert sorted(i for r in a for i in r) == list(range(len(lens)))
    loads = [sum(lens[i] for i in r) for r in a]
    assert max(loads) - min(loads) <= max(lens) // 2
    assert a == assign_contigs(lens, 8)
    assert assign_contigs([5], 4) == [[0], [], [], []]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lens = [30, 10, 20, 5, 7]
        mine = assign_contigs(lens, world)[rank]
        local = [(i, bytes([65 + i]) * lens[i]) for i in mine]
        got = all_gather_sequences(local, device=torch.device("cpu"))
        ok = sorted(got) == list(range(len(lens))) and all(got[i] == bytes([65 + i]) * lens[i] for i in got)
        # a rank with no contig still participates
        got2 = all_gather_sequences([(0, b"ACGT")] if rank == 0 else [], device=torch.device("cpu"))
        ok = ok and got2 == {0: b"ACGT"}
        import numpy as np
        g = SequenceGatherer(64, torch.device("cpu"))
        for step in range(2):
            mine_b = np.frombuffer((b"AC" if rank == 0 else b"GGTTA") * (step + 1), dtype=np.uint8)
            g.gather(mine_b)
            ok = ok and g.to_host() == {0: b"AC" * (step + 1), 1: b"GGTTA" * (step + 1)}
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_all_gather_sequences_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
