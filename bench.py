#!/usr/bin/env python3
"""bench.py - rendered rays/s of the fused HIP ray march on synthetic 128x128 frames.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched with
torch.distributed.run, one rank per GPU. One step = one pass of the hot path over one batch of rays that
is already resident in HBM: enarf_render_step_fwd = one pre-march launch (part frames + modulated MLP weights,
NCHW -> channel-last feature planes, ray set-up) + the fused ray march (in-kernel importance sampling); with
--unfused the same work as enarf_prepare -> enarf_triplane_pack -> enarf_render_fwd. Rank 0 prints ONE JSON line. Weak scaling: every rank renders its own frame(s); the path
has no exchange step, so there is no data-path collective.

Workload = BASELINE.json configs[1]: 128x128 rays, Nc 48 + Nf 64 samples/ray, 24 SMPL joints (P = 23
parts, `center_fixed`), one frame per GPU per step, fp32 tri-plane, constant (DSO-style) tri-plane.
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

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BF16_DENSE_TFLOPS = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1, help="frames per GPU per step")
    ap.add_argument("--nc", type=int, default=48)
    ap.add_argument("--nf", type=int, default=64)
    ap.add_argument("--origin", default="center_fixed", choices=["center", "center_fixed", "center+head"])
    ap.add_argument("--mlp-mode", default="f16x3", choices=["f32", "f16x3", "bf16x3", "bf16"])
    ap.add_argument("--style-dim", type=int, default=20)
    ap.add_argument("--early-stop-eps", type=float, default=0.0, help="opt-in early ray termination (0 = exact)")
    ap.add_argument("--cache-triplane", action="store_true",
                    help="re-lay the (constant) tri-plane once instead of every step")
    ap.add_argument("--distinct-triplanes", action="store_true",
                    help="GAN style: one tri-plane per frame (generated on the device) instead of one shared constant tri-plane")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the steps alternate over, each with its own intermediates (part frames, MLP packs, "
                         "channel-last planes, workspace). With 2, the pre-march launch of step i+1 fills the CUs that the "
                         "persistent march of step i frees in its tail (+4 %% rays/s), but event-bracketed kernel times then "
                         "include the overlap, so the default - and the roofline figure - is strictly serial steps")
    ap.add_argument("--shard-frame", action="store_true",
                    help="N > 1: strong scaling of ONE frame batch - every rank marches its contiguous share of the rays "
                         "(sharding.rays_for_rank) and the 5 floats per ray are all-gathered on every rank each step "
                         "(SURVEY.md 8e, single DSO frame); default is weak scaling, one frame batch per rank")
    ap.add_argument("--unfused", action="store_true",
                    help="issue prepare / re-layout / render as the three separate C-ABI calls (4 launches) "
                         "instead of enarf_render_step_fwd (2 launches)")
    ap.add_argument("--spinup-ms", type=float, default=40.0,
                    help="device spin-up during set-up, before the W warm-up steps: the step is repeated until this much "
                         "wall time has passed, so that short runs (small K and W) are not timed at idle clocks")
    ap.add_argument("--no-p24", action="store_true",
                    help="skip the extra timed pass with origin_location center+head (P = 24), which SURVEY.md 8 asks to "
                         "report beside the shipping P = 23 configuration")
    ap.add_argument("--allow-variant", action="store_true", help="measurement only: permit --variant")
    ap.add_argument("--variant", default=None, help="path of another build of the same ABI (tools/build_variant.sh); "
                                                     "refused without --allow-variant")
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
    # a horizontal band through the middle of the frame (hits and misses mixed, like the full frame)
    start = (n // 2) - n_rays // 2
    coord = s["image_coord"][..., start:start + n_rays].contiguous()
    g = torch.Generator().manual_seed(0)
    passes, t0 = 0, time.perf_counter()
    while True:
        O.render(coord, pose_p, bl_p, s["inv_intrinsics"], cpose, cbl, s["tri_plane"], s["mlp"], s["z_rend"],
                 s["coordinate_scale"], Nc, Nf, generator=g, use_grid_sample=True)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or passes >= 64:
            break
    return {"value": passes * n_rays / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{n_rays} consecutive rays of the same frame (middle band: hits and misses mixed), {passes} passes, "
                      f"{dt:.1f} s; oracle/enarf_oracle.py render() with F.grid_sample"}


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
    from enarf_gan_amd import ops, synth
    from oracle import enarf_oracle as O   # only for the canonical-pose buffers of the synthetic scene and the cpu_baseline leg

    S, B, Nc, Nf = args.size, args.batch, args.nc, args.nf
    shard = args.shard_frame and world > 1
    sc = synth.make_scene(S, B, args.origin, args.style_dim, pose_seed=1234 + (0 if shard else 100 * rank), shared_triplane=True)
    n_frame = S * S
    n = n_frame
    P = sc["num_parts"]
    cpose, cbl = O.register_canonical_pose(sc["canonical_pose"], sc["parents"], args.origin)
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
    tri = sc["tri_plane"][:1].contiguous().to(dev)          # one constant tri-plane shared by the batch (DSO style)
    if args.distinct_triplanes and B > 1:                   # per-frame tri-planes: jittered copies, made on the device
        g = torch.Generator(device=dev).manual_seed(5)
        tri = (tri + 0.05 * torch.randn(B, *tri.shape[1:], device=dev, generator=g)).contiguous()
    mlp = {k: v.to(dev) for k, v in sc["mlp"].items()}
    cpose_d, cbl_d = cpose.to(dev), cbl.to(dev)
    coord = d["image_coord"].reshape(B, 3, n).contiguous()
    if shard:                                   # this rank's contiguous range of every frame's rays
        from enarf_gan_amd import sharding
        coord = coord[..., sharding.rays_for_rank(n_frame, rank, world)].contiguous()
        n = coord.shape[-1]

    def gather_outputs(o):
        """all ranks end up with the whole frame: one collective of 5 floats per ray"""
        local = torch.cat([o.color, o.mask[:, None], o.disparity[:, None]], dim=1)
        if dist.get_backend() == "nccl":
            return sharding.all_gather_rays(local, n_frame)
        return sharding.all_gather_rays(local.cpu(), n_frame)        # rehearsal backends move host tensors

    n_streams = 1 if args.unfused else max(1, args.streams)
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(n_streams - 1)]
    sets = []                                      # per-stream intermediates: steps on different streams share only inputs
    for _ in range(n_streams):
        f = torch.empty(tri.shape[0], 3, 256, 256, 32, device=dev)
        ops.triplane_pack(tri, f)
        sets.append((f, torch.empty(B, P, 16, device=dev), torch.empty(B, ops.mlp_pack_bytes(), dtype=torch.uint8, device=dev)))
    feat_cl, parts, pack = sets[0]
    torch.cuda.synchronize()

    def bound_step(seed, count=False, k=0):
        f, pa, pk = sets[k]
        return ops.RenderStep(d["pose_to_camera"], d["bone_length"], cbl_d, d["z_rend"], mlp, sc["parents"], args.origin,
                              3.0, coord, d["inv_intrinsics"], cpose_d, tri, f, Nc, Nf, parts_out=pa,
                              pack_out=pk, relayout=not args.cache_triplane, seed=seed, mlp_mode=args.mlp_mode,
                              want_fine=True, count=count, early_stop_eps=args.early_stop_eps)

    def step(i, count=False):
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
    cnt = step(0, count=True).counters
    torch.cuda.synchronize()
    V, tiles, rays_marched, rounds = [int(x) for x in cnt[:4].tolist()]

    # set-up: bring the device out of its idle clocks (reported in config.spinup_ms; not part of W or K)
    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
        for _ in range(8):
            step(0)
        torch.cuda.synchronize()

    for i in range(args.warmup):
        with torch.cuda.stream(streams[i % n_streams]):
            if args.unfused:
                step(i)
            else:
                o = bound_step(99 + i, k=i % n_streams).run()
                if shard:
                    gather_outputs(o)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if not args.unfused:      # same two launches as enarf_render_step_fwd(ENARF_STEP_ALL), with the march bracketed
            k = i % n_streams
            with torch.cuda.stream(streams[k]):
                st = bound_step(99, k=k)
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

    p24 = None
    if world == 1 and not shard and not args.unfused and args.origin == "center_fixed" and not args.no_p24:
        # the same step with the head part added (center+head, P = 24): K timed steps after W warm-up steps, serial
        sc2 = synth.make_scene(S, B, "center+head", args.style_dim, pose_seed=1234, shared_triplane=True)
        cp2, cb2 = O.register_canonical_pose(sc2["canonical_pose"], sc2["parents"], "center+head")
        d2 = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in sc2.items()}
        tri2 = sc2["tri_plane"][:1].contiguous().to(dev)
        mlp2 = {k: v.to(dev) for k, v in sc2["mlp"].items()}
        cp2, cb2 = cp2.to(dev), cb2.to(dev)
        coord2 = d2["image_coord"].reshape(B, 3, n).contiguous()
        f2 = ops.triplane_pack(tri2)
        pa2 = torch.empty(B, sc2["num_parts"], 16, device=dev)
        pk2 = torch.empty(B, ops.mlp_pack_bytes(), dtype=torch.uint8, device=dev)

        def step24():
            return ops.RenderStep(d2["pose_to_camera"], d2["bone_length"], cb2, d2["z_rend"], mlp2, sc2["parents"], "center+head",
                                  3.0, coord2, d2["inv_intrinsics"], cp2, tri2, f2, Nc, Nf, parts_out=pa2, pack_out=pk2,
                                  relayout=not args.cache_triplane, seed=99, mlp_mode=args.mlp_mode, want_fine=True,
                                  early_stop_eps=args.early_stop_eps).run()
        for _ in range(max(args.warmup, 1)):
            step24()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step24()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        p24 = {"workload": f"the same step with origin_location center+head (P = {sc2['num_parts']})", "value": B * n * args.steps / dt,
               "unit": "rays/s", "ms_per_step": dt / args.steps * 1e3}

    if rank == 0:
        rays_per_step = B * n_frame if shard else world * B * n
        value = rays_per_step * args.steps / elapsed
        # V*12*(C+1)*4 gathered texel bytes + each tri-plane once + outputs (+ fine side outputs)
        alg_bytes = V * 1584 + tri.shape[0] * (96 + 3 * P) * 256 * 256 * 4 + B * n * 20 + B * n * (2 * Nf - 1) * 4
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("render_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        q = rays_marched * (Nc + Nf - 1)
        out = {
            "metric": "rendered rays/sec (128^2, 64 samples/ray, 24 bones)", "value": value, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C1: DSO-style {S}x{S} frame, Nc {Nc} + Nf {Nf} samples/ray, 24 joints -> P={P} parts "
                                   f"({args.origin}), {B} frame/GPU/step, {'per-frame' if tri.shape[0] > 1 else 'constant'} fp32 tri-plane 256^2x(96+{3 * P}), "
                                   f"in-kernel Philox importance sampling",
                       "sharding": "rays of one frame batch across ranks + all-gather of outputs" if shard else "one frame batch per rank",
                       "library": _lib.library_info(), "spinup_ms": args.spinup_ms, "streams": n_streams, "mlp_arith": args.mlp_mode, "early_stop_eps": args.early_stop_eps, "triplane_relayout_in_step": not args.cache_triplane,
                       "step": ("enarf_prepare + enarf_triplane_pack + enarf_render_fwd" if args.unfused else
                                "enarf_render_step_fwd (pre-march launch: re-layout + prepare + ray set-up; then the march)")},
            "roofline": {"bound": "hbm", "kernel": "enarf::render_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         # tri-plane once + outputs: what would have to cross HBM with perfect caching
                         "compulsory_bytes_per_launch": alg_bytes - V * 1584,
                         "compulsory_GBps": (alg_bytes - V * 1584) / (kern_ms * 1e-3) / 1e9,
                         "valid_part_point_pairs": V, "rays_marched": rays_marched, "mlp_tiles_of_16": tiles,
                         "gather_rounds": rounds, "gather_lane_utilisation": V / max(16 * rounds, 1),
                         "mfma_eligible_tflops": q * 12800 / (kern_ms * 1e-3) / 1e12,
                         "mfma_frac_of_bf16_dense_peak": q * 12800 / (kern_ms * 1e-3) / 1e12 / BF16_DENSE_TFLOPS},
        }
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
