"""Multi-process tests of the image-parallel gradient exchange (gloo, world_size 2, CPU tensors): the N > 1 path of
bench.py / training is correct by construction on RCCL if it is correct here -- same torch.distributed calls."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diff_triangle_rasterization_2D.parallel import GradBucket, all_reduce_triangle_grads, reduce_render_stats, shard_views

        P, M = 37, 4
        g = torch.Generator().manual_seed(100 + rank)
        shapes = [(P, 3, 3), (P, M, 3), (P, 1), (P, 2)]
        grads = [torch.rand(s, generator=g) for s in shapes]
        bucket = GradBucket([torch.Size(s) for s in shapes], "cpu")
        out = [t.clone() for t in bucket.all_reduce(grads)]
        # reference result computed locally from both ranks' seeds
        expect = [torch.zeros(s) for s in shapes]
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            for e, s in zip(expect, shapes):
                e += torch.rand(s, generator=gr)
        ok = all(torch.allclose(a, b, atol=1e-6) for a, b in zip(out, expect))
        # None gradients count as zeros; mean mode divides by world size
        mb = GradBucket([torch.Size((5,))], "cpu", mean=True)
        m = mb.all_reduce([torch.full((5,), float(rank + 1)) if rank == 0 else None])[0]
        ok = ok and torch.allclose(m, torch.full((5,), 0.5))
        # parameter-level helper writes back into .grad
        p = torch.nn.Parameter(torch.zeros(P, 3, 3))
        p.grad = grads[0].clone()
        all_reduce_triangle_grads([p])
        ok = ok and torch.allclose(p.grad, expect[0], atol=1e-6)
        stats = reduce_render_stats({"radii": torch.tensor([1, 5, 2]) * (rank + 1), "visible_count": torch.tensor([1, 0, 1])})
        ok = ok and stats["radii"].tolist() == [2, 10, 4] and stats["visible_count"].tolist() == [2, 0, 2]
        views = shard_views(5, rank, world)
        ok = ok and views == ([0, 2, 4] if rank == 0 else [1, 3])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_image_parallel_grad_exchange_world2(hip_lib_built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}


def test_single_process_is_a_noop(hip_lib_built):
    from diff_triangle_rasterization_2D.parallel import GradBucket, shard_views
    b = GradBucket([torch.Size((3, 2))], "cpu")
    g = torch.arange(6.0).view(3, 2)
    assert torch.equal(b.all_reduce([g])[0], g)
    assert shard_views(4, 0, 1) == [0, 1, 2, 3]
