#!/usr/bin/env python3
"""One whole ENARF-GAN training iteration (train_ENARF_GAN.py:102-170) on the mirror modules - BASELINE configs[2] / [3]:
128 x 128, batch 32 in `n_accum_step` = 2 micro-batches (configs/enarfgan_train/SURREAL/config.yml), generator = HIP renderer
+ StyleGAN2 background network, residual discriminator, non-saturating loss + bone-guided mask loss, Adam(0, 0.99), R1 every
16th iteration. NOT the headline metric (bench.py is): this times the loop the renderer is invoked in.

The tri-planes come from the generator's own StyleGAN2-ADA synthesis network (libraries/stylegan2_ada/networks.py: this repo's
restatement of the un-vendored submodule); `--producer planes` replaces it by one learnable tri-plane per frame (SURVEY.md 8d,
C2: "GAN-style tri-plane per image"), which isolates the renderer's share. Data-parallel over WORLD_SIZE ranks as the reference's
DistributedDataParallel: the batch is dealt to the ranks, every micro-batch's generator gradients are all-reduced in buckets
while the next micro-batch runs (sharding.GradientReducer), the discriminator's after its backward.
  python tools/bench_gan_step.py [--batch 32 --accum 2 --size 128 --steps 6 --warmup 2]
  python -m torch.distributed.run --nproc-per-node N ... tools/bench_gan_step.py --gpus N"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from enarf_gan_amd import sharding, synth  # noqa: E402
from enarf_gan_amd.libraries.custom_stylegan2.net import Discriminator  # noqa: E402
from enarf_gan_amd.libraries.gan.loss import adv_loss_dis, adv_loss_gen, d_r1_loss  # noqa: E402
from enarf_gan_amd.models.generator import TriNARFGenerator  # noqa: E402
from enarf_gan_amd.models.loss import nerf_patch_loss  # noqa: E402


class Cfg(dict):
    __getattr__ = dict.__getitem__


def nerf_cfg(Nc, Nf):
    return Cfg(hidden_size=32, Nc=Nc, Nf=Nf, origin_location="center_fixed", coordinate_scale=3, render_bs=16384,
               no_ray_direction=True, multiply_density_with_triplane_wieght=False, clamp_mask=False, constant_triplane=False,
               constant_trimask=False, constant_trimask_lr_mul=1, deformation_field=False, selector_mlp=False, no_selector=False,
               time_conditional=True, pose_conditional=False)


def build(args, dev, frames):
    """generator, discriminator, the per-frame tri-planes standing in for the synthesis network, synthetic poses / images"""
    S, zd = args.size, 256
    sc = synth.make_scene(S, frames, "center_fixed", zd, shared_triplane=True)
    cfg = Cfg(z_dim=zd, background_ratio=0.7, crop_background=True, pretrained_background=False, nerf_params=nerf_cfg(args.nc, args.nf))
    gen = TriNARFGenerator(cfg, S, 24, sc["parents"], 23)
    gen.register_canonical_pose(sc["canonical_pose"])
    gen = gen.to(dev).train()
    g = torch.Generator(device=dev).manual_seed(5)
    base = sc["tri_plane"][:1].to(dev)
    tri = (base + 0.05 * torch.randn(frames, *base.shape[1:], device=dev, generator=g)).requires_grad_(True)
    dis = Discriminator(Cfg(minibatch_std=False), size=S).to(dev).train()          # SURREAL / AIST configs: minibatch_std False
    data = {k: sc[k].to(dev) for k in ("pose_to_camera", "bone_length", "inv_intrinsics")}
    data["real"] = torch.randn(frames, 3, S, S, device=dev, generator=g).clamp(-1, 1)
    data["bone_mask"] = (torch.rand(frames, S, S, device=dev, generator=g) > 0.97).float()
    return gen, dis, tri, data


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--accum", type=int, default=2)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--nc", type=int, default=48)
    ap.add_argument("--nf", type=int, default=64)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--r1-every", type=int, default=16)
    ap.add_argument("--producer", choices=("stylegan", "planes"), default="stylegan")
    ap.add_argument("--amp", action="store_true", help="opt-in, not the reference's arithmetic: torch.autocast(bf16) around the 2-D "
                    "networks' library convolutions (the HIP renderer and the HIP ops compute in fp32 either way)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo for a rehearsal)")
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    dev = torch.device("cuda", local % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = args.backend or "nccl"
        dist.init_process_group(backend, device_id=dev if backend == "nccl" else None)
    if args.batch % (world * args.accum):
        raise SystemExit(f"--batch {args.batch} must divide by ranks x micro-batches = {world * args.accum}")
    frames = args.batch // world                       # this rank's share
    mb = frames // args.accum
    torch.manual_seed(1234 + rank)
    gen, dis, tri, data = build(args, dev, frames)
    if args.producer == "planes":
        gen.nerf.tri_plane_gen = None          # drops the synthesis network's parameters from the generator
    gen_params = [p for p in gen.parameters() if p.requires_grad]
    dis_params = list(dis.parameters())
    lr_scale = args.batch / 32
    gen_opt = torch.optim.Adam(gen_params + [tri], lr=1e-3 * lr_scale, betas=(0.0, 0.99))
    dis_opt = torch.optim.Adam(dis_params, lr=2e-3 * lr_scale, betas=(0.0, 0.99))
    g_red = sharding.GradientReducer(gen_params, world) if dist is not None else None
    d_red = sharding.GradientReducer(dis_params, world) if dist is not None else None
    ev = {}
    t_w = time.perf_counter()
    import contextlib
    amp = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if args.amp else contextlib.nullcontext

    def mark(name, it):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.setdefault(it, []).append((name, e))

    def iteration(it, timed=False, talk=False):
        m = (lambda n: mark(n, it)) if timed else (lambda n: None)

        def say(what):          # the first iteration takes minutes (library warm-up): keep the log moving
            if talk and rank == 0:
                torch.cuda.synchronize()
                print(f"  [{time.perf_counter() - t_w:6.1f} s] {what}", file=sys.stderr, flush=True)
        m("start")
        # ---- generator step (train_ENARF_GAN.py:108-128)
        dis.requires_grad_(False)
        fakes = []

        def backward_of(k):
            sl = slice(k * mb, (k + 1) * mb)
            if args.producer == "planes":
                gen.nerf.tri_plane_gen = lambda z, enc, truncation_psi=1: tri[sl]
            z = torch.randn(mb, 4 * 256, device=dev)
            with amp():
                fake, mask, _, _ = gen(data["pose_to_camera"][sl], None, data["bone_length"][sl], z, data["inv_intrinsics"][sl])
                fake, mask = fake.float(), mask.float()
                logits = dis(fake, dist is not None, world).float()
            loss = adv_loss_gen(logits, "ce") + nerf_patch_loss(mask, data["bone_mask"][sl], gen.background_ratio)
            grads = torch.autograd.grad(loss, gen_params + [tri], allow_unused=True)       # tri: unused with the real producer
            if grads[-1] is not None:
                tri.grad = grads[-1] if (k == 0 or tri.grad is None) else tri.grad + grads[-1]
            fakes.append(fake.detach())
            say(f"generator micro-batch {k}")
            return list(grads[:-1])
        tri.grad = None
        sharding.accumulate_and_reduce(range(args.accum), backward_of, gen_params, g_red)
        m("generator forward + backward (+ exchange)")
        gen_opt.step()
        fake = torch.cat(fakes)
        # ---- discriminator step (:133-147)
        dis.requires_grad_(True)
        real = data["real"]
        with amp():
            d_real, d_fake = dis(real, dist is not None, world).float(), dis(fake, dist is not None, world).float()
        loss_d = adv_loss_dis(d_real, d_fake, "ce")
        sharding.accumulate_and_reduce([0], lambda _: list(torch.autograd.grad(loss_d, dis_params)), dis_params, d_red)
        dis_opt.step()
        say("discriminator step")
        m("discriminator step")
        if args.r1_every and it % args.r1_every == 0:      # :149-165
            x = real.detach().requires_grad_(True)
            with amp():
                pred = dis(x, dist is not None, world).float()
            r1 = 0.5 * d_r1_loss(pred, x) * 16 * 0.01 + 0 * pred[0].sum()
            sharding.accumulate_and_reduce([0], lambda _: list(torch.autograd.grad(r1, dis_params)), dis_params, d_red)
            dis_opt.step()
            m("R1 step")
        return fake

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t_w = time.perf_counter()
    iteration(0, talk=True)            # untimed: every phase once, R1 included (the convolution library picks its algorithms
    barrier()                          # on first use of a shape: minutes on a fresh box)
    if rank == 0:
        print(f"first iteration (library warm-up): {time.perf_counter() - t_w:.1f} s", file=sys.stderr, flush=True)
    for i in range(args.warmup):
        iteration(1 + i)               # no R1 here (iteration 0 of the timed part has it)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = iteration(i, timed=True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if (args.backend or "nccl") == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    phases = {}
    for it, marks in ev.items():
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            phases.setdefault(n1, []).append(e0.elapsed_time(e1))
    if rank == 0:
        n_r1 = len(phases.get("R1 step", []))
        print(json.dumps({
            "metric": "ENARF-GAN training iterations/s (generator + discriminator step; not the headline metric)",
            "value": args.steps / elapsed, "unit": "it/s", "frames_per_s": args.batch * args.steps / elapsed,
            "ms_per_iteration": elapsed / args.steps * 1e3, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "config": {"workload": f"{args.size}x{args.size}, batch {args.batch} = {world} rank(s) x {args.accum} micro-batch(es) x {mb} frames, "
                                   f"Nc {args.nc} + Nf {args.nf}, tri-planes from " + ("the StyleGAN2-ADA synthesis network, " if args.producer == "stylegan" else "one learnable tri-plane per frame, ") +
                                   f"R1 on {n_r1} of {args.steps} iterations", "backend": (args.backend or "nccl") if world > 1 else None},
            "dtype": "f32 (renderer MLP products as 3-term split fp16)" + ("; library convolutions of the 2-D networks under bf16 autocast (opt-in)" if args.amp else ""), "data": "synthetic",
            "phases_ms_mean_rank0": {k: sum(v) / len(v) for k, v in phases.items()},
            "fake_image_abs_mean": float(out.abs().mean()),
            "params_M": {"generator (renderer MLP + background network" + (" + tri-plane synthesis network)" if args.producer == "stylegan" else ")"): sum(p.numel() for p in gen_params) / 1e6,
                         "discriminator": sum(p.numel() for p in dis_params) / 1e6}}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
