"""The backward of the fused renderer at the sizes BASELINE.json quotes (C1 full frame, C3's per-GPU share of 8 frames with
per-frame tri-planes, C2's forward batch of 16, C4's 256^2 / Nf 96 shape) and enarf_query_bwd at 1 M points.

The oracle cannot run autograd over a full frame in test time, so every case combines
  * what is exact at any size: the backward is LINEAR in the output gradients and rays are independent, so the gradients of
    the whole launch equal the sum of the gradients of K disjoint ray subsets - up to the rounding of the float atomics
    (and of the split-K weight-gradient sums), whose order differs from run to run. Bound asserted: 2e-5 of each gradient
    tensor's largest magnitude (50x tighter than the parity bound; measured ~1e-6);
  * finite, non-zero gradients for every image; a permuted batch gives image b the same tri-plane gradient;
  * autograd through the oracle on a slice of <= 64 rays of EVERY image, with the output gradients of all other rays set
    to zero in a full-size launch (all rays are marched, every persistent workgroup runs, images change inside the launch):
    1e-3 of each gradient tensor's largest magnitude, the bound of tests/test_gpu_backward.py.
Reference: train_ENARF_GAN.py:113-126 (loss.backward() through the renderer), libraries/NeRF/activation.py:12-16."""
import json

import numpy as np
import pytest
import torch

from _helpers import DeviceScene, Scene, assert_close
from oracle import enarf_oracle as O
from test_gpu_configs import _body_rays

pytestmark = pytest.mark.gpu

ATOMIC_TOL = 2e-5       # of max |gradient|: re-ordered float sums (atomics, split-K partials)
PARITY_TOL = 1e-3       # vs autograd through the oracle


def _bwd(sc, ds, coord_d, Nf, bins_d, gc, gm, gd, tri=None, feat_cl=None, parts=None, pack=None, inv_K=None):
    """enarf_render_bwd + enarf_weight_grad + enarf_prepare_bwd -> dict of gradient tensors (device)."""
    from enarf_gan_amd import ops
    tri = ds.tri if tri is None else tri
    grad_tri, dW, db = ops.render_bwd(coord_d, ds.inv_K if inv_K is None else inv_K, ds.parts if parts is None else parts,
                                      ds.cpose, tri, ds.feat_cl if feat_cl is None else feat_cl,
                                      ds.pack if pack is None else pack, Nf, bins_d, gc, gm, gd)
    out = {"feat": grad_tri[:, :96], "mask": grad_tri[:, 96:]}
    for l in range(3):
        out[f"dW{l}"], out[f"db{l}"] = dW[l], db[l]
    return out


def _check_sum(whole, parts, what):
    worst = 0.0
    for k in whole:
        s = sum(p[k] for p in parts)
        worst = max(worst, assert_close(s.cpu(), whole[k].cpu(), f"{what}: {k}: sum over ray subsets vs whole launch", ATOMIC_TOL))
    return worst


def _subset_masks(B, n, K, seed, dev):
    g = torch.Generator().manual_seed(seed)
    owner = torch.randint(0, K, (B, n), generator=g).to(dev)
    return [(owner == k).float() for k in range(K)]


def _oracle_slice_grads(sc, coord, Nc, Nf, bins, gc, gm, gd, images=None):
    """autograd through the oracle; `images`: restate only these images of the batch (their near / far planes are the whole
    batch's, as in the launch) - the oracle's cost grows with the batch, not with the rays"""
    s = sc.raw
    pick = (lambda t: t) if images is None else (lambda t: t[images])
    tri = pick(s["tri_plane"]).clone().requires_grad_(True)
    mlp = {k: v.clone().requires_grad_(True) for k, v in s["mlp"].items() if "noise" not in k}
    z = pick(s["z_rend"]).clone().requires_grad_(True)
    planes = None if images is None else O.near_far(O.scale_pose_translation(sc.pose_parts, sc.cs))
    rc, rm, rd = O.render(coord, pick(sc.pose_parts), pick(sc.bl_parts), pick(s["inv_intrinsics"]), sc.cpose, sc.cbl, tri, mlp, z,
                          sc.cs, Nc, Nf, bins=bins, near_far_planes=planes)
    loss = (rc * gc).sum() + (rm * gm).sum() + (rd * gd).sum()
    keys = sorted(mlp)
    grads = torch.autograd.grad(loss, [tri, z] + [mlp[k] for k in keys])
    return (rc.detach(), rm.detach()), grads[0], grads[1], dict(zip(keys, grads[2:]))


def _run_case(S, B, Nc, Nf, style_dim, per_image, K=4, permute=False, label="", oracle_images=None):
    from enarf_gan_amd import ops
    sc = Scene(S, B, "center_fixed", style_dim)
    ds = DeviceScene(sc)
    dev = ds.dev
    n = S * S
    coord_d = sc.raw["image_coord"].to(dev).reshape(B, 3, n).contiguous()
    fwd = ds.render(coord_d, Nc, Nf, None, seed=31, mlp_mode="f32", return_bins=True, count=True)
    bins_d = fwd.taps["bins"]
    assert int(fwd.counters[7]) == 0
    g = torch.Generator(device=dev).manual_seed(S + B)
    gc, gm, gd = (torch.randn(B, 3, n, device=dev, generator=g), torch.randn(B, n, device=dev, generator=g),
                  torch.randn(B, n, device=dev, generator=g))
    whole = _bwd(sc, ds, coord_d, Nf, bins_d, gc, gm, gd)
    torch.cuda.synchronize()
    # ---- every gradient finite, and non-zero for every image
    for k, t in whole.items():
        assert torch.isfinite(t).all(), (label, k)
    nimg = ds.tri.shape[0]
    for b in range(nimg):
        assert float(whole["feat"][b].abs().max()) > 0 and float(whole["mask"][b].abs().max()) > 0, (label, b)
    for b in range(B):
        for l in range(3):
            assert float(whole[f"dW{l}"][b].abs().max()) > 0, (label, "dW", l, b)
    # ---- linearity over K disjoint ray subsets (the other rays' output gradients zeroed; same launch shape)
    masks = _subset_masks(B, n, K, 7, dev)
    parts = [_bwd(sc, ds, coord_d, Nf, bins_d, gc * m[:, None], gm * m, gd * m) for m in masks]
    worst = _check_sum(whole, parts, label)
    print(f"{label}: sum of {K} ray subsets vs whole launch: worst {worst:.2e} of max |grad| (bound {ATOMIC_TOL})")
    del parts
    # ---- the same rays as separate, smaller launches (another n, other queue lengths): two halves by ray index
    if B == 1:
        halves = []
        for lo, hi in ((0, n // 2 + 37), (n // 2 + 37, n)):
            halves.append(_bwd(sc, ds, coord_d[..., lo:hi].contiguous(), Nf, bins_d[:, lo:hi].contiguous(),
                               gc[..., lo:hi].contiguous(), gm[:, lo:hi].contiguous(), gd[:, lo:hi].contiguous()))
        _check_sum(whole, halves, label + " (two launches of half a frame)")
        del halves
    # ---- a permuted batch: image b keeps its gradients
    if permute:
        perm = torch.arange(B - 1, -1, -1, device=dev)
        s = sc.raw
        parts_p, pack_p = ops.prepare(s["pose_to_camera"].to(dev)[perm], s["bone_length"].to(dev)[perm], sc.cbl.to(dev),
                                      s["z_rend"].to(dev)[perm], ds.mlp, s["parents"], sc.ol, sc.cs)
        tri_p = ds.tri[perm].contiguous()
        wp = _bwd(sc, ds, coord_d[perm].contiguous(), Nf, bins_d[perm].contiguous(), gc[perm].contiguous(), gm[perm].contiguous(),
                  gd[perm].contiguous(), tri=tri_p, feat_cl=ops.triplane_pack(tri_p), parts=parts_p, pack=pack_p,
                  inv_K=ds.inv_K[perm].contiguous())
        for k in ("feat", "mask", "dW0", "dW1", "dW2"):
            assert_close(wp[k].cpu(), whole[k][perm].cpu(), f"{label}: {k} of a permuted batch", ATOMIC_TOL)
        del wp
    # ---- oracle autograd on a slice of every image; all other rays carry zero output gradient in a FULL-SIZE launch
    ids = _body_rays(sc, per_image, seed=B + S)                                      # (B, m) cpu
    uid = [torch.unique(ids[b]) for b in range(B)]                                   # two draws on one pixel count once
    m = min(len(u) for u in uid)
    ids = torch.stack([u[:m] for u in uid])
    sel = torch.zeros(B, n)
    sel.scatter_(1, ids, 1.0)
    img = None if oracle_images is None else torch.tensor(oracle_images)
    if img is not None:                       # the other images' rays carry no gradient either
        keep = torch.zeros(B, 1)
        keep[img] = 1.0
        sel = sel * keep
    sel_d = sel.to(dev)
    sl = _bwd(sc, ds, coord_d, Nf, bins_d, gc * sel_d[:, None], gm * sel_d, gd * sel_d)
    pg, dz = ops.prepare_bwd(sc.raw["z_rend"].to(dev), ds.mlp, [sl["dW0"], sl["dW1"], sl["dW2"]])
    take_all = lambda t, d: torch.gather(t.cpu(), d, ids.reshape([B] + [1] * (d - 1) + [m]).expand(*t.shape[:d], m))
    take = take_all if img is None else (lambda t, d: take_all(t, d)[img])
    nb = B if img is None else len(img)
    coord_s = take(sc.raw["image_coord"].reshape(B, 3, n), 2).reshape(nb, 1, 3, m)
    bins_s = torch.gather(bins_d.cpu(), 1, ids[:, :, None].expand(-1, -1, Nf)).contiguous()
    bins_s = bins_s if img is None else bins_s[img]
    (rc, rm), o_tri, o_z, o_mlp = _oracle_slice_grads(sc, coord_s, Nc, Nf, bins_s, take(gc, 2), take(gm, 1), take(gd, 1), images=img)
    if img is not None:                       # gradients of the restated images; the others got none
        rest = torch.ones(B, dtype=torch.bool)
        rest[img] = False
        assert float(sl["feat"][rest.to(dev)].abs().max()) == 0 and float(dz[rest.to(dev)].abs().max()) == 0
        sl = dict(sl, feat=sl["feat"][img.to(dev)], mask=sl["mask"][img.to(dev)])
        dz = dz[img.to(dev)]
    assert float(rm.max()) > 0.5, label
    assert_close(take(fwd.mask, 1), rm, f"{label}: forward mask on the slice")
    assert float(o_tri[:, :96].abs().max()) > 0 and float(o_tri[:, 96:].abs().max()) > 0
    assert_close(sl["feat"].cpu(), o_tri[:, :96], f"{label}: d feature planes vs oracle autograd", PARITY_TOL)
    assert_close(sl["mask"].cpu(), o_tri[:, 96:], f"{label}: d part-probability planes vs oracle autograd", PARITY_TOL)
    for l in range(3):
        assert_close(sl[f"db{l}"].cpu(), o_mlp[f"layers.{l}.bias"].reshape(-1), f"{label}: d bias {l}", PARITY_TOL)
        for leaf in ("conv.weight", "conv.modulation.weight", "conv.modulation.bias"):
            assert_close(pg[f"layers.{l}.{leaf}"].cpu(), o_mlp[f"layers.{l}.{leaf}"], f"{label}: d layers.{l}.{leaf}", PARITY_TOL)
    assert_close(dz.cpu(), o_z, f"{label}: d z_rend", PARITY_TOL)
    from enarf_gan_amd import _lib
    assert _lib.device_status(clear=False) == 0


def test_backward_c1_full_frame():
    """BASELINE C1: one 128^2 frame, Nc 48 + Nf 64, constant tri-plane - all 16 384 rays in one enarf_render_bwd launch."""
    _run_case(128, 1, 48, 64, 20, per_image=64, label="C1 128^2 B=1")


def test_backward_c3_share_8_frames_per_frame_triplanes():
    """BASELINE C3's per-GPU share: 8 GAN-style frames of 128^2 rays, one tri-plane PER FRAME, in one launch."""
    _run_case(128, 8, 48, 64, 256, per_image=24, permute=True, label="C3 share 128^2 B=8", oracle_images=[0, 3, 7])


def test_backward_c2_forward_batch_16():
    """BASELINE C2: forward_bs 16 frames of 128^2 rays with per-frame tri-planes (two such launches make the batch of 32,
    configs/enarfgan_train/SURREAL/config.yml:7 with n_accum_step 2). Finite / non-zero / linearity checks cover every image;
    the oracle's autograd restates four of the sixteen (its cost grows with the batch: 74 s for all of them)."""
    _run_case(128, 16, 48, 64, 256, per_image=12, K=2, label="C2 128^2 B=16", oracle_images=[0, 5, 10, 15])


def test_backward_c4_shape_256_nf96():
    """BASELINE C4's shape: 256^2 rays, Nc 72 + Nf 96 (two fine tiles per wave: the recompute path), 2 frames."""
    _run_case(256, 2, 72, 96, 256, per_image=24, K=2, permute=True, label="C4 256^2 Nf 96 B=2")


def test_query_backward_one_million_points():
    """enarf_query_bwd at 2^20 points (the size of one frame's fine pass): linearity over point subsets,
    finite non-zero gradients, and autograd through the oracle on a 2 048-point slice with every other point's output
    gradient zeroed in the full-size launch."""
    from enarf_gan_amd import ops
    B, N = 1, 1 << 20
    sc = Scene(64, B, "center+head", 20)
    ds = DeviceScene(sc)
    dev = ds.dev
    g = torch.Generator().manual_seed(3)
    jp = sc.pose_scaled[:, :, :3, 3]
    pick = torch.randint(0, jp.shape[1], (B, N), generator=g)
    pts = torch.gather(jp, 1, pick[..., None].expand(-1, -1, 3)).permute(0, 2, 1).contiguous()
    pts = pts + 0.3 * torch.randn(B, 3, N, generator=g)
    pts[:, :, ::7] += 40.0                                      # a seventh far away: no valid part, colour gradient only
    gD, gC = torch.randn(B, 1, N, generator=g), torch.randn(B, 3, N, generator=g)
    pts_d, gD_d, gC_d = pts.to(dev), gD.to(dev), gC.to(dev)

    def run(m=None):
        a, b = (gD_d, gC_d) if m is None else (gD_d * m[:, None], gC_d * m[:, None])
        grad_tri, dW, db = ops.query_bwd(pts_d, ds.parts, ds.cpose, ds.tri, ds.feat_cl, ds.pack, a, b)
        out = {"feat": grad_tri[:, :96], "mask": grad_tri[:, 96:]}
        for l in range(3):
            out[f"dW{l}"], out[f"db{l}"] = dW[l], db[l]
        return out

    whole = run()
    for k, t in whole.items():
        assert torch.isfinite(t).all() and float(t.abs().max()) > 0, k
    masks = _subset_masks(B, N, 4, 5, dev)
    worst = _check_sum(whole, [run(m) for m in masks], "query 2^20 points")
    print(f"query_bwd 2^20 points: sum of 4 subsets vs whole: worst {worst:.2e} of max |grad|")
    # oracle on a slice
    M = 2048
    ids = torch.randperm(N, generator=g)[:M].sort().values
    sel = torch.zeros(B, N)
    sel[:, ids] = 1.0
    sl = run(sel.to(dev))
    pg, dz = ops.prepare_bwd(sc.raw["z_rend"].to(dev), ds.mlp, [sl["dW0"], sl["dW1"], sl["dW2"]])
    s = sc.raw
    tri = s["tri_plane"].clone().requires_grad_(True)
    mlp = {k: v.clone().requires_grad_(True) for k, v in s["mlp"].items() if "noise" not in k}
    z = s["z_rend"].clone().requires_grad_(True)
    den, col, valid = O.query(pts[..., ids].contiguous(), sc.pose_scaled, sc.scale, sc.cpose, tri, O.modulated_weights(mlp, z))
    assert 0.05 < float(valid.any(dim=1).float().mean()) < 0.95
    keys = sorted(mlp)
    grads = torch.autograd.grad((den * gD[..., ids]).sum() + (col * gC[..., ids]).sum(), [tri, z] + [mlp[k] for k in keys])
    o_tri, o_z, o_mlp = grads[0], grads[1], dict(zip(keys, grads[2:]))
    assert_close(sl["feat"].cpu(), o_tri[:, :96], "query slice: d feature planes", PARITY_TOL)
    assert_close(sl["mask"].cpu(), o_tri[:, 96:], "query slice: d part-probability planes", PARITY_TOL)
    for l in range(3):
        assert_close(sl[f"db{l}"].cpu(), o_mlp[f"layers.{l}.bias"].reshape(-1), f"query slice: d bias {l}", PARITY_TOL)
        assert_close(pg[f"layers.{l}.conv.weight"].cpu(), o_mlp[f"layers.{l}.conv.weight"], f"query slice: d conv.weight {l}", PARITY_TOL)
    assert_close(dz.cpu(), o_z, "query slice: d z_rend", PARITY_TOL)


def test_bench_train_step_runs_in_process(monkeypatch, capsys):
    """`bench.py --train-step` (forward + enarf_render_bwd + weight gradients + enarf_prepare_bwd) at B = 2 for two timed
    steps, in this process (the native library is the one pytest has loaded): one JSON line with a `backward` object, the
    march's watchdog counter and the device status word stay clear."""
    import importlib
    import sys
    from enarf_gan_amd import _lib
    bench = importlib.import_module("bench")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--train-step", "--batch", "2", "--distinct-triplanes", "--steps", "2",
                                      "--warmup", "1", "--no-cpu-baseline", "--spinup-ms", "0"])
    bench.main()
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["value"] > 0 and r["steps"] == 2 and "backward" in r and r["backward"]["ms_per_step"] > 0
    assert r["watchdog_counter"] == 0
    assert _lib.device_status(clear=False) == 0
