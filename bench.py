#!/usr/bin/env python3
"""bench.py - rendered rays/s of the fused HIP ray march on synthetic 128x128 frames.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched with torch.distributed.run, one
rank per GPU. One step = one pass of the hot path over one batch of rays that is already resident in HBM:
enarf_render_step_fwd = one pre-march launch (part frames + modulated MLP weights, NCHW -> channel-last feature planes, ray
set-up) + the fused ray march (in-kernel importance sampling). Rank 0 prints ONE JSON line.

Workloads (BASELINE.json `configs`):
  N = 1   configs[1] "C1": one DSO-style 128x128 frame, Nc 48 + Nf 64 samples/ray, 24 SMPL joints (P = 23 parts), constant
          fp32 tri-plane - the configuration the headline metric is quoted on.
  N > 1   configs[3] "C3" shape: a FIXED batch of 64 GAN-style frames (one tri-plane per frame) of 128x128 rays, dealt
          64 / N per rank (`scaling: strong`, per-GPU work shrinks with N). The forward path has no exchange step (SURVEY.md
          8e): the only collectives are the barrier and the MAX-reduce of wall time - unless `--train-step` is given, which
          times forward + backward + the RCCL gradient all-reduce of the renderer's parameters (constant tri-plane 43 MB +
          StyledMLP), the collective a data-parallel training step has (train_ENARF_GAN.py:203-206).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# MI355X_MICROARCH.md: HBM3E 8 TB/s spec; L2 ~34.5 TB/s aggregate; 256 CUs
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0
NUM_CUS = 256
# what the CUs' vector-memory (texture) path sustains for the march's access shape - quads reading 2x2 footprints of 128-B
# texels, 12 waves per CU - MEASURED by tools/ub_ta.hip on this GPU model (pattern 1, all quads active): 32.06 TB/s in
# round 2 (profiles/r02_ub_ta.txt), 32.50 TB/s in round 3 (profiles/r03_ub_ta.txt); the lower one is the ceiling used
UB_TA_SHAPE_GBS = 32060.0
# an ASSUMPTION, not a figure of MI355X_MICROARCH.md: 64 B per clock per CU for the vector L1 (256 CUs x 64 B x 2.4 GHz)
L1_BYTES_PER_CLK_PER_CU_ASSUMED = 64.0
BF16_DENSE_TFLOPS = 2500.0
C3_GLOBAL_FRAMES = 64


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default: 1 at N = 1, 64 / N at N > 1)")
    ap.add_argument("--nc", type=int, default=48)
    ap.add_argument("--nf", type=int, default=64)
    ap.add_argument("--origin", default="center_fixed", choices=["center", "center_fixed", "center+head"])
    ap.add_argument("--mlp-mode", default="f16x3", choices=["f32", "f16x3", "bf16x3", "bf16"])
    ap.add_argument("--march", default="auto", choices=["auto", "ray", "task"],
                    help="which march kernel (ENARF_MARCH_*): auto picks by shape; both give the same bits")
    ap.add_argument("--drop-missed-rays", action="store_true",
                    help="measurement only: drop rays that hit no cube also for B > 1 (the reference does so for B == 1 only)")
    ap.add_argument("--style-dim", type=int, default=20)
    ap.add_argument("--early-stop-eps", type=float, default=0.0, help="opt-in early ray termination (0 = exact)")
    ap.add_argument("--cache-triplane", action="store_true", help="re-lay the (constant) tri-plane once instead of every step")
    ap.add_argument("--distinct-triplanes", action="store_true",
                    help="GAN style: one tri-plane per frame instead of one shared constant tri-plane (default at N > 1)")
    ap.add_argument("--group-frames", type=int, default=0,
                    help="frames per march launch for batches with per-frame tri-planes (enarf_render_args.group_frames): 0 = the "
                         "library's choice (8), >= batch = one launch (round 2's behaviour); same results bit for bit")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the steps alternate over, each with its own intermediates; with 2 the pre-march launch of "
                         "step i+1 fills the CUs the persistent march of step i frees in its tail, but event-bracketed kernel "
                         "times then include the overlap, so the default - and the roofline figure - is strictly serial steps")
    ap.add_argument("--shard-frame", action="store_true",
                    help="N > 1: cut ONE frame batch by rays (sharding.rays_for_rank) and all-gather the 5 floats per ray")
    ap.add_argument("--train-step", action="store_true",
                    help="time forward + backward (enarf_render_bwd + weight gradients) + the gradient all-reduce "
                         "(sharding.all_reduce_gradients: constant tri-plane + StyledMLP parameters) instead of the forward alone")
    ap.add_argument("--accum", type=int, default=2,
                    help="--train-step: micro-batches per step (n_accum_step of configs/enarfgan_train/*/config.yml: 2); every "
                         "micro-batch's gradient all-reduce overlaps the next one's forward and backward")
    ap.add_argument("--unfused", action="store_true", help="issue prepare / re-layout / render as the three separate C-ABI calls")
    ap.add_argument("--spinup-ms", type=float, default=40.0,
                    help="device spin-up during set-up, before the W warm-up steps (not part of W or K)")
    ap.add_argument("--no-p24", action="store_true", help="skip the extra timed pass with origin_location center+head (P = 24)")
    ap.add_argument("--no-f32", action="store_true", help="skip the extra timed pass with the exact fp32 MLP arithmetic")
    ap.add_argument("--two-streams", action="store_true",
                    help="add a timed pass over two HIP streams (off by default: +2 % at 300 steps on the build boxes, -3 % in the "
                         "driver's 20-step run of round 2 - within the noise of a 5 ms timed region)")
    ap.add_argument("--no-two-streams", action="store_true", help="accepted and ignored (the pass is opt-in since round 3)")
    ap.add_argument("--allow-variant", action="store_true", help="measurement only: permit --variant")
    ap.add_argument("--variant", default=None, help="another build of the same ABI (tools/build_variant.sh); needs --allow-variant")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=4096, help="rays of the same frame per CPU pass (middle band)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="repeat CPU passes until this much time is spent")
    return ap.parse_args()


def cpu_baseline(scene_cpu, Nc, Nf, n_rays, budget_s):
    """The oracle (a port of the reference's pure-PyTorch path, F.grid_sample form) on the host cores."""
    from oracle import enarf_oracle as O
    s = scene_cpu
    cores = min(os.cpu_count() or 1, 16)      # the GPU box's CPU share for one GPU is 16 cores
    torch.set_num_threads(cores)
    pose_p, bl_p = O.transform_pose(s["pose_to_camera"], s["bone_length"], s["origin_location"], s["parents"])
    cpose, cbl = O.register_canonical_pose(s["canonical_pose"], s["parents"], s["origin_location"])
    n = s["image_coord"].shape[-1]
    start = (n // 2) - n_rays // 2            # a horizontal band through the middle of the frame (hits and misses mixed)
    coord = s["image_coord"][:1, ..., start:start + n_rays].contiguous()
    g = torch.Generator().manual_seed(0)
    passes, t0 = 0, time.perf_counter()
    while True:
        O.render(coord, pose_p[:1], bl_p[:1], s["inv_intrinsics"][:1], cpose, cbl, s["tri_plane"][:1], s["mlp"], s["z_rend"][:1],
                 s["coordinate_scale"], Nc, Nf, generator=g, use_grid_sample=True)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or passes >= 64:
            break
    return {"value": passes * n_rays / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{n_rays} consecutive rays of one frame (middle band: hits and misses mixed), {passes} passes, "
                      f"{dt:.1f} s; oracle/enarf_oracle.py render() with F.grid_sample"}


def counter_fractions(workload_key, kernel_ms):
    """Counter-backed utilisation of the march from the committed rocprofv3 summary (profiles/r03_roofline.json, written
    by tools/pmc_summary.py from separate --pmc passes of this very command). Attached only when the profiled workload is
    the one being run; every entry names its counters and its source file."""
    path = os.path.join(ROOT, "profiles", "r03_roofline.json")
    if not os.path.exists(path):
        return None
    try:
        prof = json.load(open(path))
    except Exception:
        return None
    if prof.get("workload_key") != workload_key:
        return None
    return prof


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 with `python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path in enarf_gan_amd")
    # one rank per GPU; a rehearsal with more ranks than GPUs (ENARF_BENCH_BACKEND=gloo on a 1-GPU box) wraps around
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ENARF_BENCH_BACKEND", "nccl")     # nccl = RCCL over xGMI
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from enarf_gan_amd import _lib
    if os.environ.get("ENARF_LIB"):
        raise SystemExit("ENARF_LIB is set: the product ignores it; use --allow-variant --variant PATH for an A/B run")
    if args.variant:
        if not args.allow_variant:
            raise SystemExit("--variant needs --allow-variant (the default bench measures the in-tree library only)")
        _lib.use_variant(args.variant)
    from enarf_gan_amd import ops, sharding, synth
    from enarf_gan_amd.models.narf import TriPlaneNARF

    S, Nc, Nf = args.size, args.nc, args.nf
    shard = args.shard_frame and world > 1
    multi = world > 1 and not shard
    if args.batch is not None:
        B = args.batch
    elif multi:
        B = sharding.batch_share(C3_GLOBAL_FRAMES, rank, world)[1]
    else:
        B = 1
    distinct = (args.distinct_triplanes or multi) and B > 1       # C3 is GAN style: one tri-plane per frame
    # frames: rank r of the C3 batch renders frames [r B, (r + 1) B) of one global batch (pose seeds are per frame)
    first_frame = sharding.batch_share(world * B, rank, world)[0] if multi else 0
    sc = synth.make_scene(S, B, args.origin, args.style_dim, pose_seed=1234 + first_frame, shared_triplane=True)
    n_frame = S * S
    n = n_frame
    P = sc["num_parts"]
    # canonical buffers from the product's own register_canonical_pose (models/narf.py:84-120)
    cfg = synth.nerf_config(origin_location=args.origin, Nc=Nc, Nf=Nf)
    model = TriPlaneNARF(cfg, args.style_dim, 24, parent=sc["parents"], num_bone_param=23)
    model.register_canonical_pose(sc["canonical_pose"])
    cpose_d, cbl_d = model.canonical_pose.to(dev), model.canonical_bone_length.to(dev)
    if multi:      # this rank's rows of the global batch's latents (poses are seeded per frame already)
        sc["z_rend"] = synth.make_z_rend(world * B, args.style_dim)[first_frame:first_frame + B].contiguous()
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
    tri = sc["tri_plane"][:1].contiguous().to(dev)          # one constant tri-plane shared by the batch (DSO style)

    def frame_triplanes(base, first, count):
        """per-frame tri-planes: jittered copies of the constant one, made on the device, seeded per FRAME of the global batch"""
        out = torch.empty(count, *base.shape[1:], device=dev)
        g = torch.Generator(device=dev)
        for f in range(count):
            g.manual_seed(5 + first + f)
            out[f] = base[0] + 0.05 * torch.randn(*base.shape[1:], device=dev, generator=g)
        return out

    if distinct:
        tri = frame_triplanes(tri, first_frame, B)
    mlp = {k: v.to(dev) for k, v in sc["mlp"].items()}
    coord = d["image_coord"].reshape(B, 3, n).contiguous()
    if shard:                                   # this rank's contiguous range of every frame's rays
        coord = coord[..., sharding.rays_for_rank(n_frame, rank, world)].contiguous()
        n = coord.shape[-1]

    def gather_outputs(o):
        """all ranks end up with the whole frame: one collective of 5 floats per ray"""
        local = torch.cat([o.color, o.mask[:, None], o.disparity[:, None]], dim=1)
        if dist.get_backend() == "nccl":
            return sharding.all_gather_rays(local, n_frame)
        return sharding.all_gather_rays(local.cpu(), n_frame)        # rehearsal backends move host tensors

    n_streams = 1 if (args.unfused or args.train_step) else max(1, args.streams)
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(n_streams - 1)]
    sets = []                                      # per-stream intermediates: steps on different streams share only inputs
    for _ in range(n_streams):
        f = torch.empty(tri.shape[0], 3, 256, 256, 32, device=dev)
        ops.triplane_pack(tri, f)
        sets.append((f, torch.empty(B, P, 16, device=dev), torch.empty(B, ops.mlp_pack_bytes(), dtype=torch.uint8, device=dev)))
    feat_cl, parts, pack = sets[0]
    torch.cuda.synchronize()

    def bound_step(seed, count=False, k=0, mode=None, return_bins=False, frames=None):
        """one forward step bound to its arguments; frames: a slice of this rank's frames (micro-batches of --train-step)"""
        f, pa, pk = sets[k]
        fs = slice(None) if frames is None else frames
        ts = fs if tri.shape[0] > 1 else slice(None)          # a shared tri-plane is not sliced
        return ops.RenderStep(d["pose_to_camera"][fs], d["bone_length"][fs], cbl_d, d["z_rend"][fs], mlp, sc["parents"], args.origin,
                              3.0, coord[fs], d["inv_intrinsics"][fs], cpose_d, tri[ts], f[ts], Nc, Nf, parts_out=pa[fs],
                              pack_out=pk[fs], relayout=not args.cache_triplane, seed=seed, mlp_mode=mode or args.mlp_mode,
                              want_fine=True, count=count, early_stop_eps=args.early_stop_eps, return_bins=return_bins,
                              march=args.march, drop_invalid_rays=True if args.drop_missed_rays else None,
                              group_frames=args.group_frames)

    # ---- the training step (opt-in): n_accum micro-batches of forward + backward; every micro-batch's gradients start their
    # all-reduce (N > 1) as soon as its backward is enqueued and travel while the next micro-batch computes
    train_params = None
    if args.train_step:
        if Nf > 128:
            raise SystemExit("--train-step: the backward handles Nf <= 128")
        n_accum = max(1, min(args.accum, B))
        if B % n_accum:
            raise SystemExit(f"--accum {n_accum} does not divide {B} frames per GPU")
        tri_param = torch.nn.Parameter(tri.clone())
        mlp_params = {k: torch.nn.Parameter(v.clone()) for k, v in mlp.items() if "noise" not in k}
        # a per-frame tri-plane is an activation of the (un-vendored) synthesis network, not a parameter: its gradient stays
        # on the rank; a shared constant tri-plane (DSO style) is a parameter and is reduced with the StyledMLP's
        train_params = ([] if distinct else [tri_param]) + [mlp_params[k] for k in sorted(mlp_params)]
        g_color = torch.randn(B, 3, n, device=dev)
        g_mask = torch.randn(B, n, device=dev)
        reducer = sharding.GradientReducer(train_params, world) if dist is not None else None
        local_tri_grads = [None] * n_accum

    def train_step(i, ev=None, counters=None):
        """ev: per micro-batch (before the march, after the forward, after the backward) events"""
        per = B // n_accum

        def backward_of(mb):
            fs = slice(mb * per, (mb + 1) * per)
            ts = fs if distinct else slice(None)
            st = bound_step(99 + i, return_bins=True, frames=fs)
            st.run(ops.STEP_PRE)
            if ev:
                ev[mb][0].record()
            o = st.run(ops.STEP_MARCH)
            if ev:
                ev[mb][1].record()
            grad_tri, dW, db = ops.render_bwd(coord[fs], d["inv_intrinsics"][fs], st.parts, cpose_d, tri[ts], st.feat_cl, st.pack, Nf,
                                              o.taps["bins"], g_color[fs], g_mask[fs], counters=counters,
                                              group_frames=args.group_frames)
            pg, dz = ops.prepare_bwd(d["z_rend"][fs], mlp, dW)
            if ev:
                ev[mb][2].record()
            grads = [] if distinct else [grad_tri]
            local_tri_grads[mb] = grad_tri if distinct else None
            for kname in sorted(mlp_params):
                leaf = kname.split(".", 2)[2]
                layer = int(kname.split(".")[1])
                grads.append((db[layer] if leaf == "bias" else pg[kname]).reshape(mlp_params[kname].shape))
            return grads

        sharding.accumulate_and_reduce(range(n_accum), backward_of, train_params, reducer)

    def step(i, count=False):
        if args.train_step:
            return train_step(i)
        if not args.unfused:
            o = bound_step(99 + i, count).run()
            if shard:
                gather_outputs(o)
            return o
        ops.prepare(d["pose_to_camera"], d["bone_length"], cbl_d, d["z_rend"], mlp, sc["parents"], args.origin,
                    3.0, parts_out=parts, pack_out=pack)
        if not args.cache_triplane:
            ops.triplane_pack(tri, feat_cl)
        return ops.render_fwd(coord, d["inv_intrinsics"], parts, cpose_d, tri, feat_cl, pack, Nc, Nf,
                              seed=99 + i, mlp_mode=args.mlp_mode, want_fine=True, count=count,
                              early_stop_eps=args.early_stop_eps)

    # algorithmic work of one step, counted by the kernel itself in an untimed pass (same inputs, same seed)
    cnt = bound_step(99, count=True).run().counters
    torch.cuda.synchronize()
    V, tiles, rays_marched, rounds = [int(x) for x in cnt[:4].tolist()]
    watchdog_counter = int(cnt[7])
    if watchdog_counter != 0:
        raise SystemExit("the march's scheduler watchdog fired (counters[7] != 0): results are incomplete")
    bwd_cnt = None
    if args.train_step:                # the backward's own tallies (pairs, tiles, rays, 128-B lines added, mask adds), untimed
        bwd_cnt = torch.zeros(8, dtype=torch.int64, device=dev)
        train_step(0, counters=bwd_cnt)
        torch.cuda.synchronize()
        bwd_cnt = [int(x) for x in bwd_cnt.tolist()]

    # set-up: bring the device out of its idle clocks (reported in config.spinup_ms; not part of W or K)
    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
        for _ in range(8):
            bound_step(99).run()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        with torch.cuda.stream(streams[i % n_streams]):
            if args.unfused or args.train_step:
                step(i)
            else:
                o = bound_step(99 + i, k=i % n_streams).run()
                if shard:
                    gather_outputs(o)
    # per-frame tri-planes beyond one group of frames: the step is a sequence of (pre-march, march) pairs, see enarf_render.hip
    gsize = args.group_frames if args.group_frames > 0 else 8
    grouped = (not args.unfused) and tri.shape[0] > 1 and B > gsize
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    evt = [[[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_accum)] for _ in range(args.steps)] if args.train_step else None
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if args.train_step:
            ev0[i].record()
            train_step(i, ev=evt[i])
            ev1[i].record()
            continue
        if not args.unfused:      # same two launches as enarf_render_step_fwd(ENARF_STEP_ALL), with the march bracketed
            k = i % n_streams
            with torch.cuda.stream(streams[k]):
                st = bound_step(99, k=k)
                if grouped:       # a batch marched in groups: pre-march and march alternate group by group (one call)
                    ev0[i].record()
                    o = st.run()
                    ev1[i].record()
                else:
                    st.run(ops.STEP_PRE)
                    ev0[i].record()
                    o = st.run(ops.STEP_MARCH)
                    ev1[i].record()
                if shard:
                    gather_outputs(o)
            continue
        ops.prepare(d["pose_to_camera"], d["bone_length"], cbl_d, d["z_rend"], mlp, sc["parents"], args.origin,
                    3.0, parts_out=parts, pack_out=pack)
        if not args.cache_triplane:
            ops.triplane_pack(tri, feat_cl)
        ev0[i].record()
        ops.render_fwd(coord, d["inv_intrinsics"], parts, cpose_d, tri, feat_cl, pack, Nc, Nf, seed=99,
                       mlp_mode=args.mlp_mode, want_fine=True, early_stop_eps=args.early_stop_eps)
        ev1[i].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    if grouped and not args.train_step:   # the marches alone, from a few extra steps with the two phases issued apart
        ma, mb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for _ in range(3):
            st = bound_step(99)
            st.run(ops.STEP_PRE)
            ma.record()
            st.run(ops.STEP_MARCH)
            mb.record()
            torch.cuda.synchronize()
            tot += ma.elapsed_time(mb)
        kern_ms = tot / 3
    backward = None
    if args.train_step:
        step_ms = kern_ms
        # sums over the step's micro-batches: the marches alone (as in the forward-only bench), and the backward portions
        kern_ms = float(np.mean([sum(m[0].elapsed_time(m[1]) for m in st_) for st_ in evt]))
        bwd_ms = float(np.mean([sum(m[1].elapsed_time(m[2]) for m in st_) for st_ in evt]))
        fwd_ms = float(np.mean([ev0[i].elapsed_time(evt[i][0][1]) + sum(evt[i][k - 1][2].elapsed_time(evt[i][k][1]) for k in range(1, n_accum))
                                for i in range(args.steps)]))
        pairs, tiles, rays_b, lines, madds = bwd_cnt[:5]
        atomic_bytes = lines * 128 + madds * 4
        backward = {
            "what": "enarf_render_bwd + enarf_triplane_unpack_add + enarf_weight_grad + enarf_prepare_bwd, events on the launch stream, "
                    f"summed over the step's {n_accum} micro-batch(es)",
            "micro_batches": n_accum,
            "ms_per_step": bwd_ms, "forward_ms_per_step": fwd_ms, "collective_and_rest_ms_per_step": max(step_ms - fwd_ms - bwd_ms, 0.0),
            "valid_part_fine_sample_pairs": pairs, "mlp_backward_tiles_of_16": tiles, "rays": rays_b,
            "feature_gradient_lines_added_128B": lines, "part_probability_adds_4B": madds,
            "atomic_bytes_sent_per_step": atomic_bytes, "unmerged_atomic_bytes_per_step": pairs * 12 * (128 + 4),
            # MI355X_MICROARCH.md, Global float atomics: ~1.3 TB/s of added bytes chip-wide, executed at the memory side
            "atomic_ceiling": {"value": 1300.0, "unit": "GB/s", "achieved": atomic_bytes / (bwd_ms * 1e-3) / 1e9,
                               "frac": atomic_bytes / (bwd_ms * 1e-3) / 1e9 / 1300.0,
                               "floor_ms": atomic_bytes / 1300.0 / 1e6},
            # every added byte is read and written at the memory side; rows are written once and read once
            "hbm_frac_algorithmic": (2 * atomic_bytes + 2 * tiles * 16 * 144) / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "compact_row_bytes_per_step": tiles * 16 * 144,
        }

    def timed_variant(make_step):
        for _ in range(max(args.warmup, 1)):
            make_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            make_step()
        torch.cuda.synchronize()
        return time.perf_counter() - t1

    extras = world == 1 and not shard and not args.unfused and not args.train_step
    f32_mode = None
    if extras and args.mlp_mode != "f32" and not args.no_f32:
        dt = timed_variant(lambda: bound_step(99, mode="f32").run())
        f32_mode = {"workload": "the same step with the exact fp32 MLP arithmetic (v_mfma_f32_16x16x4_f32)",
                    "value": B * n * args.steps / dt, "unit": "rays/s", "ms_per_step": dt / args.steps * 1e3}
    two_streams = None
    if extras and n_streams == 1 and args.two_streams:
        # the same K steps alternating over two HIP streams with private intermediates: the next step's pre-march launch
        # (serial chains of a few blocks) runs in the tail of the persistent march. Reported beside `value`, which stays
        # the single-stream number the roofline figures are measured on.
        s2 = torch.cuda.Stream(dev)
        f2 = torch.empty_like(feat_cl)
        ops.triplane_pack(tri, f2)
        sets.append((f2, torch.empty(B, P, 16, device=dev), torch.empty(B, ops.mlp_pack_bytes(), dtype=torch.uint8, device=dev)))
        pair = [streams[0], s2]
        turn = [0]

        def one():
            k = turn[0] & 1
            turn[0] += 1
            with torch.cuda.stream(pair[k]):
                bound_step(99, k=k).run()
        torch.cuda.synchronize()
        dt = timed_variant(one)
        two_streams = {"workload": "the same steps alternating over two HIP streams (private intermediates per stream)",
                       "value": B * n * args.steps / dt, "unit": "rays/s", "ms_per_step": dt / args.steps * 1e3}
    p24 = None
    if extras and args.origin == "center_fixed" and not args.no_p24:
        # the same step with the head part added (center+head, P = 24): K timed steps after W warm-up steps, serial
        sc2 = synth.make_scene(S, B, "center+head", args.style_dim, pose_seed=1234, shared_triplane=True)
        m2 = TriPlaneNARF(synth.nerf_config(origin_location="center+head", Nc=Nc, Nf=Nf), args.style_dim, 24, parent=sc2["parents"],
                          num_bone_param=23)
        m2.register_canonical_pose(sc2["canonical_pose"])
        cp2, cb2 = m2.canonical_pose.to(dev), m2.canonical_bone_length.to(dev)
        d2 = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in sc2.items()}
        tri2 = sc2["tri_plane"][:1].contiguous().to(dev)
        mlp2 = {k: v.to(dev) for k, v in sc2["mlp"].items()}
        coord2 = d2["image_coord"].reshape(B, 3, n).contiguous()
        f2 = ops.triplane_pack(tri2)
        pa2 = torch.empty(B, sc2["num_parts"], 16, device=dev)
        pk2 = torch.empty(B, ops.mlp_pack_bytes(), dtype=torch.uint8, device=dev)
        dt = timed_variant(lambda: ops.RenderStep(d2["pose_to_camera"], d2["bone_length"], cb2, d2["z_rend"], mlp2, sc2["parents"],
                                                  "center+head", 3.0, coord2, d2["inv_intrinsics"], cp2, tri2, f2, Nc, Nf,
                                                  parts_out=pa2, pack_out=pk2, relayout=not args.cache_triplane, seed=99,
                                                  mlp_mode=args.mlp_mode, want_fine=True,
                                                  early_stop_eps=args.early_stop_eps).run())
        p24 = {"workload": f"the same step with origin_location center+head (P = {sc2['num_parts']})", "value": B * n * args.steps / dt,
               "unit": "rays/s", "ms_per_step": dt / args.steps * 1e3}

    one_gpu_same_job = None
    if multi and rank == 0 and not args.train_step and not args.unfused:
        # The N > 1 line is a different job from the N = 1 headline (C3's 64 frames against one C1 frame): here rank 0 alone
        # renders the WHOLE global batch, so that the line carries its own one-GPU reference (the other ranks wait at the
        # barrier below). Same frames (pose and tri-plane seeds are per frame of the global batch); one difference, named in
        # the record: one process reduces the near / far planes over all 64 frames, a rank over its own share - which is what
        # DistributedDataParallel does to the reference (every replica renders its own mini-batch, rendering.py:15-17).
        GB = world * B
        scg = synth.make_scene(S, GB, args.origin, args.style_dim, pose_seed=1234, shared_triplane=True)
        dg = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in scg.items()}
        trig = frame_triplanes(sc["tri_plane"][:1].contiguous().to(dev), 0, GB)
        fg = torch.empty(GB, 3, 256, 256, 32, device=dev)
        pag, pkg = torch.empty(GB, P, 16, device=dev), torch.empty(GB, ops.mlp_pack_bytes(), dtype=torch.uint8, device=dev)
        coordg = dg["image_coord"].reshape(GB, 3, n).contiguous()

        def whole():
            return ops.RenderStep(dg["pose_to_camera"], dg["bone_length"], cbl_d, dg["z_rend"], mlp, scg["parents"], args.origin, 3.0,
                                  coordg, dg["inv_intrinsics"], cpose_d, trig, fg, Nc, Nf, parts_out=pag, pack_out=pkg,
                                  relayout=not args.cache_triplane, seed=99, mlp_mode=args.mlp_mode, want_fine=True,
                                  early_stop_eps=args.early_stop_eps, march=args.march, group_frames=args.group_frames).run()
        k1 = max(2, min(args.steps, 8))
        for _ in range(2):
            whole()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k1):
            whole()
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        one_gpu_same_job = {"value": GB * n * k1 / dt1, "unit": "rays/s", "ms_per_step": dt1 / k1 * 1e3, "steps": k1,
                            "frames": GB, "where": "rank 0 of this run, alone, after the timed steps (other ranks at the barrier)",
                            "note": "same frames and tri-planes as the N-rank job; near / far planes reduced over all frames "
                                    "here, per rank share there (as DistributedDataParallel does to the reference)"}
        del trig, fg, pag, pkg, dg
    if rank == 0:
        rays_per_step = B * n_frame if shard else world * B * n
        value = rays_per_step * args.steps / elapsed
        # V*12*(C+1)*4 gathered texel bytes + each tri-plane once + outputs (+ fine side outputs)
        gather_bytes = V * 1584
        compulsory = tri.shape[0] * (96 + 3 * P) * 256 * 256 * 4 + B * n * 20 + B * n * (2 * Nf - 1) * 4
        t_k = kern_ms * 1e-3
        q = rays_marched * (Nc + Nf - 1)
        clock_ghz = 2.4
        l1_assumed = NUM_CUS * L1_BYTES_PER_CLK_PER_CU_ASSUMED * clock_ghz          # GB/s, an assumption (see above)
        workload_key = f"C1:{S}:{B}:{Nc}:{Nf}:{P}:{args.mlp_mode}:{int(distinct)}:{args.early_stop_eps}"
        if args.march != "auto":
            workload_key += f":{args.march}"
        if args.group_frames:
            workload_key += f":g{args.group_frames}"
        # which of the two march kernels ran (enarf_render.hip launch_render: ENARF_MARCH_AUTO picks by shape)
        spl = 2 if (Nc > 64 or Nf > 64) else 1
        task = args.march == "task" or (args.march == "auto" and spl == 2 and B == 1)
        mode_id = {"f32": 0, "bf16x3": 1, "bf16": 2, "f16x3": 3}[args.mlp_mode]
        kernel_name = f"enarf::march_kernel<{mode_id}, {spl}, 12>" if task else f"enarf::render_kernel<{mode_id}, {spl}>"
        roof = {
            # the texel gathers are L1/L2 hits (the tri-plane is read from HBM once): the roofline that bounds the march is
            # the CUs' vector-memory (texture) path, not HBM. `achieved` = algorithmic texel bytes / kernel time.
            "bound": "l1-texture-path", "kernel": kernel_name,
            "achieved": gather_bytes / t_k / 1e9, "peak": UB_TA_SHAPE_GBS, "unit": "GB/s", "frac": gather_bytes / t_k / 1e9 / UB_TA_SHAPE_GBS,
            "peak_note": "MEASURED ceiling of the CUs' texture path for this access shape (2x2 footprints of 128-B texels read by "
                         "quads, 12 waves per CU): tools/ub_ta.hip pattern 1, 16/16 quads, profiles/r02_ub_ta.txt (32.06 TB/s; "
                         "32.50 TB/s in profiles/r03_ub_ta.txt). MI355X_MICROARCH.md gives no L1 bandwidth figure",
            # for comparison with rounds 1-2, which divided by an ASSUMED 64 B per clock per CU (not a figure of the guide)
            "assumed_64B_per_clk_per_cu": {"value": l1_assumed, "unit": "GB/s", "frac": gather_bytes / t_k / 1e9 / l1_assumed,
                                           "note": f"{NUM_CUS} CUs x 64 B/clk x {clock_ghz} GHz: an assumption, kept so that rounds can be compared"},
            "kernel_ms": kern_ms, "gather_bytes_per_launch": gather_bytes, "compulsory_hbm_bytes_per_launch": compulsory,
            "hbm_frac_if_every_gather_missed": (gather_bytes + compulsory) / t_k / 1e9 / HBM_PEAK_GBS,
            "compulsory_hbm_frac": compulsory / t_k / 1e9 / HBM_PEAK_GBS,
            "valid_part_point_pairs": V, "rays_marched": rays_marched, "mlp_tiles_of_16": tiles,
            "gather_rounds": rounds, "gather_lane_utilisation": V / max(16 * rounds, 1),
            "mfma_eligible_tflops": q * 12800 / t_k / 1e12,
            "mfma_frac_of_bf16_dense_peak": q * 12800 / t_k / 1e12 / BF16_DENSE_TFLOPS,
            "traffic": None, "counters": None,
        }
        prof = counter_fractions(workload_key, kern_ms) if not args.variant else None
        if prof is not None:          # measured HBM traffic and unit utilisations of THIS workload (rocprofv3, profiles/)
            roof["traffic"] = prof.get("hbm_bytes_per_launch")
            if roof["traffic"]:      # the HBM roofline of the same launch, from the counters: far from the bound
                roof["hbm"] = {"achieved": roof["traffic"] / t_k / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": roof["traffic"] / t_k / 1e9 / HBM_PEAK_GBS}
            roof["counters"] = prof.get("fractions")
            roof["profiled_kernel_ms"] = prof.get("kernel_ms")
        mode_label = {"f32": "f32", "f16x3": "f32 (MLP products as 3-term split fp16 on MFMA, fp32 accumulate)",
                      "bf16x3": "f32 (MLP products as 3-term split bf16 on MFMA)", "bf16": "bf16 MLP operands, fp32 elsewhere"}
        if args.train_step:
            step_desc = ("forward (enarf_render_step_fwd) + enarf_render_bwd + enarf_weight_grad + enarf_prepare_bwd"
                         + ((" + all-reduce of the StyledMLP gradients" + ("" if distinct else " and of the constant tri-plane's"))
                            if world > 1 else ""))
        elif args.unfused:
            step_desc = "enarf_prepare + enarf_triplane_pack + enarf_render_fwd"
        else:
            step_desc = "enarf_render_step_fwd (pre-march launch: re-layout + prepare + ray set-up; then the march)"
        out = {
            "metric": "rendered rays/sec (128^2, 64 samples/ray, 24 bones)", "value": value, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if (shard or multi) else "weak", "vs_baseline": None,
            "dtype": mode_label[args.mlp_mode], "data": "synthetic",
            "config": {"workload": (f"{'C3 shape: ' + str(world * B) + ' GAN-style' if multi else 'C1: ' + str(B) + ' DSO-style'} "
                                    f"{S}x{S} frame(s), Nc {Nc} + Nf {Nf} samples/ray, 24 joints -> P={P} parts ({args.origin}), "
                                    f"{B} frame(s)/GPU/step, {'per-frame' if tri.shape[0] > 1 else 'constant'} fp32 tri-plane "
                                    f"256^2x(96+{3 * P}), in-kernel Philox importance sampling"),
                       "sharding": ("rays of one frame batch across ranks + all-gather of outputs" if shard else
                                    f"a fixed batch of {world * B} frames dealt {B} per rank" if multi else "one frame batch per rank"),
                       "library": _lib.library_info(), "env": {k: v for k, v in os.environ.items() if k.startswith("ENARF_")},
                       "spinup_ms": args.spinup_ms, "streams": n_streams, "mlp_arith": args.mlp_mode, "march": args.march,
                       "early_stop_eps": args.early_stop_eps, "triplane_relayout_in_step": not args.cache_triplane,
                       "group_frames": args.group_frames if args.group_frames else (8 if tri.shape[0] > 1 else 0),
                       "step": step_desc},
            "roofline": roof, "watchdog_counter": watchdog_counter,
        }
        if backward is not None:
            out["backward"] = backward
        if one_gpu_same_job is not None:
            out["one_gpu_same_job"] = one_gpu_same_job
            out["speedup_vs_one_gpu_same_job"] = value / one_gpu_same_job["value"]
        if f32_mode is not None:
            out["f32_mode"] = f32_mode
        if two_streams is not None:
            out["two_streams"] = two_streams
        if p24 is not None:
            out["p24"] = p24
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sc, Nc, Nf, min(args.cpu_rays, n), args.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
