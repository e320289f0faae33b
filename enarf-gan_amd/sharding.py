"""Sharding of the render path across the GPUs of one node.

Rays and frames are independent: the forward path has NO exchange step, so ranks only agree on who renders
what. Frames are dealt contiguously (each rank keeps its own tri-planes, poses and MLP packs); a single frame
is cut into contiguous ray ranges. The only collective is the one a caller wants for the outputs
(`all_gather_rays`), plus barriers / a MAX-reduce of wall time in bench.py.
"""
from __future__ import annotations

from typing import List, Tuple

import torch


def split_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of `rank` when `total` items are dealt contiguously, sizes differing by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(total, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def frames_for_rank(num_frames: int, rank: int, world: int) -> range:
    return range(*split_range(num_frames, rank, world))


def rays_for_rank(num_rays: int, rank: int, world: int) -> slice:
    return slice(*split_range(num_rays, rank, world))


def batch_share(global_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """(first frame, frames) of `rank` when a FIXED batch of `global_frames` frames is dealt evenly (bench.py N > 1: the C3
    batch of 64 frames -> 64 / N per rank). The reference's batch-global near / far planes (rendering.py:15-17) are then
    per rank-local batch - exactly what DistributedDataParallel does to the reference: every replica renders its own
    mini-batch with its own planes - so ranks exchange nothing in the forward."""
    if global_frames % world:
        raise ValueError(f"{global_frames} frames do not divide over {world} ranks")
    per = global_frames // world
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return rank * per, per


def all_gather_rays(local: torch.Tensor, num_rays: int, group=None) -> torch.Tensor:
    """Reassemble (..., n_local) per-rank ray outputs into (..., num_rays) on every rank (ragged ranges allowed)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [split_range(num_rays, r, world) for r in range(world)]
    maxn = max(b - a for a, b in sizes)
    pad = torch.zeros(*local.shape[:-1], maxn, dtype=local.dtype, device=local.device)
    pad[..., :local.shape[-1]] = local
    bufs: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][..., :sizes[r][1] - sizes[r][0]] for r in range(world)], dim=-1)


def all_reduce_gradients(params, world_size: int = None, group=None, bucket_bytes: int = 64 << 20, average: bool = True) -> int:
    """Data-parallel gradient exchange for the renderer's parameters (tri-plane, StyledMLP, producers upstream): the one
    collective of a training step (the reference wraps G and D in DistributedDataParallel, train_ENARF_GAN.py:203-206).

    Gradients are flattened into buckets of ~`bucket_bytes` and summed with one all-reduce per bucket (backend "nccl" is
    RCCL on ROCm). xGMI is point-to-point, 7 links x ~153 GB/s per GPU, so few large messages beat many small ones: the
    43 MB tri-plane gradient is one bucket; the 30 KB of MLP gradients ride along instead of paying a launch each.
    Returns the number of collectives issued. Parameters without a gradient get zeros (all ranks must agree on the layout)."""
    import torch.distributed as dist
    if world_size is None:
        world_size = dist.get_world_size(group)
    params = [p for p in params if p.requires_grad]
    n_coll, bucket, size = 0, [], 0

    def flush():
        nonlocal n_coll, bucket, size
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= world_size
        o = 0
        for p in bucket:
            k = p.numel()
            g = flat[o:o + k].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            o += k
        n_coll += 1
        bucket, size = [], 0

    for p in params:
        nbytes = p.numel() * p.element_size()
        if bucket and (size + nbytes > bucket_bytes or bucket[0].dtype != p.dtype):
            flush()
        bucket.append(p)
        size += nbytes
    flush()
    return n_coll


class GradientReducer:
    """Bucketed, ASYNCHRONOUS data-parallel gradient exchange with accumulation over micro-batches - what the reference's
    DistributedDataParallel wrapping does for `n_accum_step` backward passes per optimiser step (train_ENARF_GAN.py:113-126,
    :203-206: every loss_gen.backward() all-reduces, .grad accumulates).

    `launch(grads)` is called once per micro-batch, as soon as that micro-batch's backward kernels have been enqueued: the
    gradients are flattened into buckets of ~`bucket_bytes` on the current stream and one all-reduce per bucket is started
    with async_op=True - on RCCL ("nccl") it runs on the communicator's own stream behind the current stream's work, so the
    NEXT micro-batch's forward and backward run on the compute stream while the buckets travel over xGMI. `finish()` makes
    the current stream wait for every bucket and leaves, in `p.grad`, the sum over micro-batches of the rank-averaged
    gradients. xGMI is point-to-point (7 links x ~153 GB/s per GPU): few large buckets, not one message per tensor."""

    def __init__(self, params, world_size: int = None, group=None, bucket_bytes: int = 64 << 20, average: bool = True):
        import torch.distributed as dist
        self.dist, self.group, self.average = dist, group, average
        self.world = dist.get_world_size(group) if world_size is None else world_size
        self.params = list(params)
        self.buckets, cur, size = [], [], 0
        for i, p in enumerate(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (size + nbytes > bucket_bytes or self.params[cur[0]].dtype != p.dtype):
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(i)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self.pending = []          # (work handle, flat tensor, parameter indices) of every bucket in flight
        self.collectives = 0

    def launch(self, grads) -> None:
        """grads: one tensor (or None = zeros) per parameter, this micro-batch's local gradients."""
        if len(grads) != len(self.params):
            raise ValueError(f"{len(grads)} gradients for {len(self.params)} parameters")
        for idx in self.buckets:
            flat = torch.cat([(grads[i] if grads[i] is not None else torch.zeros_like(self.params[i])).reshape(-1) for i in idx])
            work = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append((work, flat, idx))
            self.collectives += 1

    def finish(self) -> int:
        """Wait for every bucket; p.grad = sum over the launched micro-batches of the (averaged) reduced gradients.
        Returns the number of collectives since the last finish()."""
        first = set()
        for work, flat, idx in self.pending:
            work.wait()
            if self.average:
                flat /= self.world
            o = 0
            for i in idx:
                p = self.params[i]
                g = flat[o:o + p.numel()].view_as(p)
                if i in first:
                    p.grad += g
                else:
                    p.grad = g.clone()
                    first.add(i)
                o += p.numel()
        n, self.pending, self.collectives = self.collectives, [], 0
        return n


def accumulate_and_reduce(micro_batches, backward_fn, params, reducer: "GradientReducer" = None) -> None:
    """One optimiser step's gradient work: for every micro-batch run `backward_fn(mb)` -> list of local gradients (one per
    parameter) and, with a reducer, start its exchange at once so that it overlaps the next micro-batch; without one
    (a single process) accumulate locally. Afterwards p.grad holds the step's gradient (bench.py --train-step)."""
    params = list(params)
    if reducer is None:
        for k, mb in enumerate(micro_batches):
            grads = backward_fn(mb)
            for p, g in zip(params, grads):
                if g is None:
                    g = torch.zeros_like(p)
                p.grad = g.clone() if k == 0 else p.grad + g
        return
    for mb in micro_batches:
        reducer.launch(backward_fn(mb))
    reducer.finish()
