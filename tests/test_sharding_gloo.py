"""world_size-2 gloo test of the N > 1 path: ranks render disjoint ray ranges of one frame (here with the
oracle, since this container has no GPU) and reassemble them; the result must equal the unsharded render.
The forward path has no data-path collective - the all_gather is only the caller's output assembly."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_rays, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from _helpers import Scene
    from enarf_gan_amd import sharding
    sc = Scene(32, 1, "center_fixed", 20)
    coord = sc.raw["image_coord"][..., 32 * 14:32 * 14 + n_rays].contiguous()
    bins = torch.sort(torch.rand(1, n_rays, 16, generator=torch.Generator().manual_seed(3)), dim=-1)[0]
    sl = sharding.rays_for_rank(n_rays, rank, world)
    rc, rm, rd = sc.oracle_render(coord[..., sl], 12, 16, bins[:, sl], taps=False)
    full_mask = sharding.all_gather_rays(rm, n_rays)
    full_color = sharding.all_gather_rays(rc, n_rays)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)       # bench.py's max-over-ranks timing reduce
    if rank == 0:
        rc0, rm0, _ = sc.oracle_render(coord, 12, 16, bins, taps=False)
        ret["ok"] = bool(torch.equal(full_mask, rm0) and torch.equal(full_color, rc0) and float(t) == world)
        ret["hit"] = float((rm0 > 0).float().mean())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ray_sharding_matches_unsharded():
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, 45, ret), nprocs=2, join=True)      # 45 rays: ragged split 23 + 22
    assert ret["ok"] and ret["hit"] > 0.1


def test_split_range_properties():
    from enarf_gan_amd import sharding
    for total in (0, 1, 7, 16384, 16385):
        for world in (1, 2, 3, 8):
            spans = [sharding.split_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    with pytest.raises(ValueError):
        sharding.split_range(4, 2, 2)
    assert list(sharding.frames_for_rank(10, 1, 4)) == [3, 4, 5]
    # bench.py N > 1: the fixed C3 batch of 64 frames
    assert [sharding.batch_share(64, r, 8) for r in (0, 3, 7)] == [(0, 8), (24, 8), (56, 8)]
    assert sharding.batch_share(64, 1, 2) == (32, 32) and sharding.batch_share(64, 0, 1) == (0, 64)
    with pytest.raises(ValueError):
        sharding.batch_share(64, 0, 3)


def _grad_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from enarf_gan_amd import sharding
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2, 2)),
          torch.nn.Parameter(torch.zeros(4), requires_grad=False)]
    ps[0].grad = torch.full((3, 5), float(rank + 1))
    ps[1].grad = torch.arange(7.0) * (rank + 1)
    # ps[2] has no gradient on any rank -> zeros
    n = sharding.all_reduce_gradients(ps, bucket_bytes=64)        # 60 B + 28 B > 64 B -> two buckets, + one for ps[2]
    ok = torch.allclose(ps[0].grad, torch.full((3, 5), 1.5)) and torch.allclose(ps[1].grad, torch.arange(7.0) * 1.5) \
        and torch.equal(ps[2].grad, torch.zeros(2, 2)) and ps[3].grad is None and n == 2
    if rank == 0:
        ret["ok"], ret["n"] = bool(ok), n
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucketed_gradient_all_reduce():
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_grad_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["ok"], dict(ret)


def _accum_worker(rank, world, port, ret):
    """Two accumulation steps per rank through sharding.accumulate_and_reduce (the closure bench.py --train-step runs):
    asynchronous bucketed all-reduce per micro-batch, against the single-process sum over all four micro-batches."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from enarf_gan_amd import sharding
    g = torch.Generator().manual_seed(7)
    data = torch.randn(world * 2, 6, 5, generator=g)                # micro-batch k of the job: data[k]
    w0 = torch.randn(5, 3, generator=g)
    ps = [torch.nn.Parameter(w0.clone()), torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.ones(2, 2))]

    def local_grads(x):                                             # a toy "renderer": the third parameter gets no gradient
        loss = ((x @ ps[0] + ps[1]).tanh() ** 2).sum()
        g0, g1 = torch.autograd.grad(loss, ps[:2])
        return [g0, g1, None]

    red = sharding.GradientReducer(ps, world, bucket_bytes=64)      # 60 B | 12 B + 16 B -> two buckets per micro-batch
    mine = [data[2 * rank], data[2 * rank + 1]]
    sharding.accumulate_and_reduce(mine, local_grads, ps, red)
    n_coll = red.collectives
    # single process: the average over ranks of every micro-batch, summed over the accumulation steps
    want0 = sum(local_grads(data[k])[0] for k in range(2 * world)) / world
    want1 = sum(local_grads(data[k])[1] for k in range(2 * world)) / world
    ok = torch.allclose(ps[0].grad, want0, atol=1e-6) and torch.allclose(ps[1].grad, want1, atol=1e-6) \
        and torch.equal(ps[2].grad, torch.zeros(2, 2)) and n_coll == 0 and not red.pending
    # the same closure without a reducer (one process): plain accumulation
    qs = [torch.nn.Parameter(w0.clone()), torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.ones(2, 2))]
    sharding.accumulate_and_reduce(mine, local_grads, qs, None)
    ok = ok and torch.allclose(qs[0].grad, local_grads(mine[0])[0] + local_grads(mine[1])[0], atol=1e-6)
    if rank == 0:
        ret["ok"] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_accumulated_async_gradient_exchange():
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_accum_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["ok"], dict(ret)


def _stddev_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from enarf_gan_amd.libraries.custom_stylegan2.net import minibatch_stddev
    feats = [torch.randn(8, 6, 4, 4, generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
    out = minibatch_stddev(feats[rank], 4, ddp=True, world_size=world)           # the discriminator's ddp branch (net.py:665-667)
    if rank == 0:
        local = [minibatch_stddev(f, 4)[:, 6] for f in feats]                   # every rank's own statistic
        want = sum(local) / world
        ret["ok"] = bool(torch.allclose(out[:, 6], want, rtol=1e-6, atol=1e-7) and torch.equal(out[:, :6], feats[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_minibatch_stddev_is_averaged_over_the_ranks():
    """the only collective inside the discriminator: the stddev statistic is all-reduced and divided by world_size"""
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_stddev_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["ok"]
