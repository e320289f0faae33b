"""Backward of the fused renderer against autograd through the oracle.

The oracle is plain torch, so `torch.autograd` through it IS the reference's gradient (same ops, incl. MyReLU's
custom backward, libraries/NeRF/activation.py:12-16). Tolerance: 1e-3 of each gradient tensor's max magnitude
(float atomics and a different summation order; the forward bound of 1e-4 applies to values, not to gradients)."""
import numpy as np
import pytest
import torch

from _helpers import DeviceScene, Scene, assert_close
from oracle import enarf_oracle as O

pytestmark = pytest.mark.gpu


def _oracle_grads(sc, coord, Nc, Nf, bins, gc, gm, gd):
    s = sc.raw
    tri = s["tri_plane"].clone().requires_grad_(True)
    mlp = {k: v.clone().requires_grad_(True) for k, v in s["mlp"].items() if "noise" not in k}
    z = s["z_rend"].clone().requires_grad_(True)
    rc, rm, rd = O.render(coord, sc.pose_parts, sc.bl_parts, s["inv_intrinsics"], sc.cpose, sc.cbl, tri, mlp, z,
                          sc.cs, Nc, Nf, bins=bins)
    loss = (rc * gc).sum() + (rm * gm).sum() + (rd * gd).sum()
    keys = sorted(mlp)
    grads = torch.autograd.grad(loss, [tri, z] + [mlp[k] for k in keys])
    return grads[0], grads[1], dict(zip(keys, grads[2:]))


@pytest.mark.parametrize("B,size,style_dim,n_rays", [(1, 32, 20, 72), (2, 32, 256, 48)])
def test_render_backward_matches_oracle_autograd(B, size, style_dim, n_rays):
    from enarf_gan_amd import ops
    sc = Scene(size, B, "center_fixed", style_dim)
    ds = DeviceScene(sc)
    Nc, Nf = 48, 32
    n = size * size
    start = n // 2 - n_rays // 2                      # a band through the body
    coord = sc.raw["image_coord"][..., start:start + n_rays].contiguous()
    fwd = ds.render(coord, Nc, Nf, None, seed=7, debug=True, mlp_mode="f32")
    bins = fwd.taps["bins"].cpu()
    g = torch.Generator().manual_seed(5)
    gc, gm, gd = torch.randn(B, 3, n_rays, generator=g), torch.randn(B, n_rays, generator=g), torch.randn(B, n_rays, generator=g)
    o_tri, o_z, o_mlp = _oracle_grads(sc, coord, Nc, Nf, bins, gc, gm, gd)

    grad_tri, dW, db = ops.render_bwd(coord.cuda(), ds.inv_K, ds.parts, ds.cpose, ds.tri, ds.feat_cl, ds.pack, Nf,
                                      bins.cuda(), gc.cuda(), gm.cuda(), gd.cuda())
    pg, dz = ops.prepare_bwd(sc.raw["z_rend"].cuda(), ds.mlp, dW)
    # tri-plane: feature planes and part-probability planes separately (different magnitudes)
    assert float(o_tri[:, :96].abs().max()) > 0 and float(o_tri[:, 96:].abs().max()) > 0
    assert_close(grad_tri[:, :96].cpu(), o_tri[:, :96], "d loss / d feature planes", 1e-3)
    assert_close(grad_tri[:, 96:].cpu(), o_tri[:, 96:], "d loss / d part-probability planes", 1e-3)
    for l in range(3):
        assert_close(db[l].cpu(), o_mlp[f"layers.{l}.bias"].reshape(-1), f"d bias {l}", 1e-3)
        for leaf in ("conv.weight", "conv.modulation.weight", "conv.modulation.bias"):
            assert_close(pg[f"layers.{l}.{leaf}"].cpu(), o_mlp[f"layers.{l}.{leaf}"], f"d layers.{l}.{leaf}", 1e-3)
    assert_close(dz.cpu(), o_z, "d z_rend", 1e-3)


def test_model_forward_is_differentiable():
    """The mirror model under autograd: tri_plane (parameter), MLP parameters and z_rend receive the oracle's gradients."""
    from test_gpu_api import _model
    sc = Scene(32, 1, "center_fixed", 20)
    m = _model(sc, Nc=48, Nf=32, style_dim=20, mlp_mode="f32").train()
    n_rays = 64
    start = 32 * 14
    coord = sc.raw["image_coord"][..., start:start + n_rays].contiguous()
    s = sc.raw
    z = s["z_rend"].cuda().requires_grad_(True)
    with torch.no_grad():
        probe = m(1, coord.cuda(), s["pose_to_camera"].cuda(), s["inv_intrinsics"].cuda(), None, z.detach(),
                  s["bone_length"].cuda(), Nc=48, Nf=32, seed=3)
    bins = m.buffers_tensors["bins"].clone()
    color, mask = m(1, coord.cuda(), s["pose_to_camera"].cuda(), s["inv_intrinsics"].cuda(), None, z, s["bone_length"].cuda(),
                    Nc=48, Nf=32, bins=bins)
    assert torch.equal(color.detach(), probe[0])
    g = torch.Generator().manual_seed(1)
    gc, gm = torch.randn(1, 3, n_rays, generator=g), torch.randn(1, n_rays, generator=g)
    ((color * gc.cuda()).sum() + (mask * gm.cuda()).sum()).backward()
    o_tri, o_z, o_mlp = _oracle_grads(sc, coord, 48, 32, bins.cpu(), gc, gm, torch.zeros(1, n_rays))
    assert_close(m.tri_plane.grad.cpu(), o_tri, "tri_plane.grad", 1e-3)
    assert_close(z.grad.cpu(), o_z, "z_rend.grad", 1e-3)
    assert_close(m.mlp.layers[1].conv.weight.grad.cpu(), o_mlp["layers.1.conv.weight"], "layers.1.conv.weight.grad", 1e-3)
    assert_close(m.mlp.layers[2].bias.grad.cpu(), o_mlp["layers.2.bias"], "layers.2.bias.grad", 1e-3)


@pytest.mark.parametrize("name", ["grad_32_b1", "grad_32_b2"])
def test_render_backward_matches_reference_gradients(name):
    """The HIP backward against gradients recorded from the reference's own autograd (tests/golden/make_golden.py)."""
    import torch.nn.functional as F
    from _helpers import load_golden
    from enarf_gan_amd import ops
    g = load_golden(name)
    B = int(g["batch"])
    sc = Scene(int(g["size"]), B, str(g["origin_location"]), int(g["style_dim"]))
    ds = DeviceScene(sc)
    s0, nr, Nf = int(g["start"]), int(g["n_rays"]), int(g["Nf"])
    coord = sc.raw["image_coord"][..., s0:s0 + nr].contiguous()
    bins = torch.from_numpy(g["bins"])
    fwd = ds.render(coord, int(g["Nc"]), Nf, bins, mlp_mode="f32")
    assert_close(fwd.color.cpu(), g["color"], "colour vs reference")
    grad_tri, dW, db = ops.render_bwd(coord.cuda(), ds.inv_K, ds.parts, ds.cpose, ds.tri, ds.feat_cl, ds.pack, Nf, bins.cuda(),
                                      torch.from_numpy(g["g_color"]).cuda(), torch.from_numpy(g["g_mask"]).cuda(),
                                      torch.from_numpy(g["g_disp"]).cuda())
    pg, dz = ops.prepare_bwd(sc.raw["z_rend"].cuda(), ds.mlp, dW)
    assert_close((F.avg_pool2d(grad_tri, 16) * 256).cpu(), g["grad_tri_pool16"], "d tri-plane (16x16 sum-pooled)", 1e-3)
    assert abs(float(grad_tri.abs().sum()) - float(g["grad_tri_abs_sum"])) < 2e-3 * float(g["grad_tri_abs_sum"])
    assert_close(dz.cpu(), g["grad_z"], "d z_rend", 1e-3)
    for l in range(3):
        assert_close(db[l].cpu(), g[f"grad_layers.{l}.bias"].reshape(-1), f"d bias {l}", 1e-3)
        for leaf in ("conv.weight", "conv.modulation.weight", "conv.modulation.bias"):
            assert_close(pg[f"layers.{l}.{leaf}"].cpu(), g[f"grad_layers.{l}.{leaf}"], f"d layers.{l}.{leaf}", 1e-3)


@pytest.mark.parametrize("B,style_dim,N", [(1, 20, 1000), (2, 256, 333)])
def test_query_backward_matches_oracle_autograd(B, style_dim, N):
    """enarf_query_bwd (a9) against autograd through the oracle's query: points inside and outside the part cubes (the
    outside ones still send their colour gradient to the MLP through a zero feature), ragged N, batch 2."""
    from enarf_gan_amd import ops
    sc = Scene(32, B, "center+head", style_dim)
    ds = DeviceScene(sc)
    s = sc.raw
    g = torch.Generator().manual_seed(11)
    # points around the joints (mostly inside some cube) plus a fifth far away (no valid part)
    jp = sc.pose_scaled[:, :, :3, 3]                                   # (B, P, 3)
    pick = torch.randint(0, jp.shape[1], (B, N), generator=g)
    pts = torch.gather(jp, 1, pick[..., None].expand(-1, -1, 3)).permute(0, 2, 1).contiguous()
    pts = pts + 0.25 * torch.randn(B, 3, N, generator=g)
    pts[:, :, ::5] += 50.0
    gD, gC = torch.randn(B, 1, N, generator=g), torch.randn(B, 3, N, generator=g)

    tri = s["tri_plane"].clone().requires_grad_(True)
    mlp = {k: v.clone().requires_grad_(True) for k, v in s["mlp"].items() if "noise" not in k}
    z = s["z_rend"].clone().requires_grad_(True)
    den, col, valid = O.query(pts, sc.pose_scaled, sc.scale, sc.cpose, tri, O.modulated_weights(mlp, z))
    assert 0.05 < float(valid.any(dim=1).float().mean()) < 0.95
    keys = sorted(mlp)
    grads = torch.autograd.grad((den * gD).sum() + (col * gC).sum(), [tri, z] + [mlp[k] for k in keys])
    o_tri, o_z, o_mlp = grads[0], grads[1], dict(zip(keys, grads[2:]))

    grad_tri, dW, db = ops.query_bwd(pts.cuda(), ds.parts, ds.cpose, ds.tri, ds.feat_cl, ds.pack, gD.cuda(), gC.cuda())
    pg, dz = ops.prepare_bwd(s["z_rend"].cuda(), ds.mlp, dW)
    assert_close(grad_tri[:, :96].cpu(), o_tri[:, :96], "d loss / d feature planes", 1e-3)
    assert_close(grad_tri[:, 96:].cpu(), o_tri[:, 96:], "d loss / d part-probability planes", 1e-3)
    for l in range(3):
        assert_close(db[l].cpu(), o_mlp[f"layers.{l}.bias"].reshape(-1), f"d bias {l}", 1e-3)
        for leaf in ("conv.weight", "conv.modulation.weight", "conv.modulation.bias"):
            assert_close(pg[f"layers.{l}.{leaf}"].cpu(), o_mlp[f"layers.{l}.{leaf}"], f"d layers.{l}.{leaf}", 1e-3)
    assert_close(dz.cpu(), o_z, "d z_rend", 1e-3)
    # colour gradient alone, density gradient alone
    only_c = ops.query_bwd(pts.cuda(), ds.parts, ds.cpose, ds.tri, ds.feat_cl, ds.pack, None, gC.cuda())
    only_d = ops.query_bwd(pts.cuda(), ds.parts, ds.cpose, ds.tri, ds.feat_cl, ds.pack, gD.cuda(), None)
    assert_close((only_c[0] + only_d[0]).cpu(), o_tri, "linearity in the output gradients", 1e-3)


def test_model_query_entry_point_is_differentiable():
    """calc_density_and_color_from_camera_coord_v2 of the mirror model under autograd (enarf_query_bwd behind it): the
    tri-plane parameter and z_rend receive the oracle's gradients."""
    from test_gpu_api import _model
    sc = Scene(32, 1, "center_fixed", 20)
    m = _model(sc, Nc=48, Nf=32, style_dim=20, mlp_mode="f32").train()
    s = sc.raw
    g = torch.Generator().manual_seed(2)
    jp = sc.pose_scaled[:, :, :3, 3]
    pick = torch.randint(0, jp.shape[1], (1, 500), generator=g)
    pts = (torch.gather(jp, 1, pick[..., None].expand(-1, -1, 3)).permute(0, 2, 1) + 0.2 * torch.randn(1, 3, 500, generator=g)).contiguous()
    gD, gC = torch.randn(1, 1, 500, generator=g), torch.randn(1, 3, 500, generator=g)
    z = s["z_rend"].cuda().requires_grad_(True)
    mi = {"z": None, "z_rend": z, "bone_length": sc.bl_parts.cuda(), "truncation_psi": 1}
    den, col = m.calc_density_and_color_from_camera_coord_v2(pts.cuda(), sc.pose_scaled.cuda(), None, mi)
    ((den * gD.cuda()).sum() + (col * gC.cuda()).sum()).backward()
    tri = s["tri_plane"].clone().requires_grad_(True)
    mlp = {k: v.clone().requires_grad_(True) for k, v in s["mlp"].items() if "noise" not in k}
    zc = s["z_rend"].clone().requires_grad_(True)
    oden, ocol, _ = O.query(pts, sc.pose_scaled, sc.scale, sc.cpose, tri, O.modulated_weights(mlp, zc))
    assert_close(den.detach().cpu(), oden.detach(), "density")
    o_tri, o_z, o_b = torch.autograd.grad((oden * gD).sum() + (ocol * gC).sum(), [tri, zc, mlp["layers.2.bias"]])
    assert_close(m.tri_plane.grad.cpu(), o_tri, "tri_plane.grad", 1e-3)
    assert_close(z.grad.cpu(), o_z, "z_rend.grad", 1e-3)
    assert_close(m.mlp.layers[2].bias.grad.cpu(), o_b, "layers.2.bias.grad", 1e-3)


def test_model_with_deformation_field_producer():
    """nerf_params.deformation_field (models/narf.py:40-58) with a user-supplied flow generator: rendering without
    gradients takes the direct channel-last route (enarf_triplane_warp_fwd), training the differentiable NCHW route; both
    against the oracle on a tri-plane warped with F.grid_sample, and the gradients of the tri-plane parameter and of the
    flow generator's parameter against autograd through the oracle."""
    import torch.nn.functional as F
    from enarf_gan_amd.models.narf import TriPlaneNARF
    from test_host_cpu import _nerf_cfg
    sc = Scene(32, 2, "center_fixed", 20)
    s = sc.raw
    m = TriPlaneNARF(_nerf_cfg(origin_location="center_fixed", Nc=24, Nf=32, mlp_mode="f32", constant_triplane=False,
                               deformation_field=True), 20, 24, parent=s["parents"], num_bone_param=23)
    m.register_canonical_pose(s["canonical_pose"])
    m.load_state_dict({f"mlp.{k}": v for k, v in s["mlp"].items()}, strict=False)
    with torch.no_grad():
        m.tri_plane.copy_(s["tri_plane"][:1])
    m = m.cuda()

    class ToyFlow(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(8)
            self.register_buffer("field", F.avg_pool2d(torch.randn(2, 6, 256, 256, generator=g), 9, 1, 4))
            self.amp = torch.nn.Parameter(torch.tensor(6.0))

        def forward(self, z, bone_length, truncation_psi=1):
            return self.amp * self.field[:bone_length.shape[0]]

    fg = ToyFlow().cuda()
    m.flow_generator = fg
    n_rays = 48
    coord = s["image_coord"][..., 32 * 14:32 * 14 + n_rays].contiguous()
    g = torch.Generator().manual_seed(6)
    bins = torch.rand(2, n_rays, 32, generator=g).sort(-1).values
    gc, gm = torch.randn(2, 3, n_rays, generator=g), torch.randn(2, n_rays, generator=g)

    def oracle(tri_param, amp):
        flow = amp * fg.field.cpu()
        gx = (torch.arange(256) + 0.5 + flow[:, 0::2]) / 128 - 1
        gy = (torch.arange(256)[:, None] + 0.5 + flow[:, 1::2]) / 128 - 1
        grid = torch.stack([gx, gy], dim=-1).reshape(6, 256, 256, 2)
        src = tri_param[:, :96].reshape(1, 3, 32, 256, 256).expand(2, -1, -1, -1, -1).reshape(6, 32, 256, 256)
        warped = F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=False).reshape(2, 96, 256, 256)
        tri = torch.cat([warped, tri_param[:, 96:].expand(2, -1, -1, -1)], dim=1)
        return O.render(coord, sc.pose_parts, sc.bl_parts, s["inv_intrinsics"], sc.cpose, sc.cbl, tri, s["mlp"], s["z_rend"],
                        sc.cs, 24, 32, bins=bins)

    tp, amp = s["tri_plane"][:1].clone().requires_grad_(True), torch.tensor(6.0, requires_grad=True)
    rc, rm, _ = oracle(tp, amp)
    o_tri, o_amp = torch.autograd.grad((rc * gc).sum() + (rm * gm).sum(), [tp, amp])

    args = (2, coord.cuda(), s["pose_to_camera"].cuda(), s["inv_intrinsics"].cuda(), None, s["z_rend"].cuda(), s["bone_length"].cuda())
    m.eval()
    with torch.no_grad():
        color, mask = m(*args, Nc=24, Nf=32, bins=bins.cuda())
    assert m.buffers_tensors["tri_plane_feature"] is None            # the direct channel-last route ran
    assert_close(color.cpu(), rc.detach(), "colour, warp producer (no grad)")
    assert_close(mask.cpu(), rm.detach(), "mask, warp producer (no grad)")
    m.train()
    color, mask = m(*args, Nc=24, Nf=32, bins=bins.cuda())
    assert_close(color.detach().cpu(), rc.detach(), "colour, warp producer (autograd route)")
    ((color * gc.cuda()).sum() + (mask * gm.cuda()).sum()).backward()
    assert_close(m.tri_plane.grad.cpu(), o_tri, "tri_plane.grad through the warp", 1e-3)
    assert abs(float(fg.amp.grad) - float(o_amp)) < 2e-3 * max(abs(float(o_amp)), 1e-3), (float(fg.amp.grad), float(o_amp))


def _oracle_grads_modes(sc, coord, Nc, Nf, bins, gc, gm, gd, **modes):
    s = sc.raw
    tri = s["tri_plane"].clone().requires_grad_(True)
    mlp = {k: v.clone().requires_grad_(True) for k, v in s["mlp"].items() if "noise" not in k}
    z = s["z_rend"].clone().requires_grad_(True)
    rc, rm, rd = O.render(coord, sc.pose_parts, sc.bl_parts, s["inv_intrinsics"], sc.cpose, sc.cbl, tri, mlp, z,
                          sc.cs, Nc, Nf, bins=bins, **modes)
    loss = (rc * gc).sum() + (rm * gm).sum() + (rd * gd).sum()
    keys = sorted(mlp)
    grads = torch.autograd.grad(loss, [tri, z] + [mlp[k] for k in keys])
    return (rc, rm, rd), grads[0], grads[1], dict(zip(keys, grads[2:]))


@pytest.mark.parametrize("Nf,modes", [
    (96, {}),                                                   # BASELINE config C4's fine count: two tiles per wave, recompute
    (128, {}),
    (72, dict(multiply_density_with_weight=True)),              # narf.py:271-272: gradient into the max part probability
    (32, dict(multiply_density_with_weight=True, clamp_mask=True)),
    (32, dict(no_selector=True, multiply_density_with_weight=True)),
    (48, dict(clamp_mask=True)),
])
def test_render_backward_fine_counts_and_density_modes(Nf, modes):
    """enarf_render_bwd beyond round 1's limits: 64 < Nf <= 128 (each wave owns two fine tiles and recomputes them for the
    backward), multiply_density_with_triplane_wieght, clamp_mask (straight-through) and no_selector - forward values and
    every gradient against autograd through the oracle with the same switches."""
    from enarf_gan_amd import ops
    sc = Scene(32, 1, "center_fixed", 20)
    if modes.get("clamp_mask"):
        sc.raw["tri_plane"] = sc.raw["tri_plane"].clone()
        sc.raw["tri_plane"][:, 96:] *= 3.0                    # push plane samples beyond [-2, 5]
    ds = DeviceScene(sc)
    Nc, n_rays = 48, 40
    start = 32 * 15
    coord = sc.raw["image_coord"][..., start:start + n_rays].contiguous()
    kflags = dict(multiply_density_with_weight=modes.get("multiply_density_with_weight", False),
                  clamp_mask=modes.get("clamp_mask", False), uniform_part_weight=modes.get("no_selector", False))
    fwd = ds.render(coord, Nc, Nf, None, seed=9, mlp_mode="f32", return_bins=True, **kflags)
    bins = fwd.taps["bins"].cpu()
    g = torch.Generator().manual_seed(Nf)
    gc, gm, gd = torch.randn(1, 3, n_rays, generator=g), torch.randn(1, n_rays, generator=g), torch.randn(1, n_rays, generator=g)
    (rc, rm, rd), o_tri, o_z, o_mlp = _oracle_grads_modes(sc, coord, Nc, Nf, bins, gc, gm, gd, **modes)
    assert float(rm.detach().max()) > 0.2
    assert_close(fwd.color.cpu(), rc.detach(), "forward colour")
    assert_close(fwd.mask.cpu(), rm.detach(), "forward mask")
    grad_tri, dW, db = ops.render_bwd(coord.cuda(), ds.inv_K, ds.parts, ds.cpose, ds.tri, ds.feat_cl, ds.pack, Nf, bins.cuda(),
                                      gc.cuda(), gm.cuda(), gd.cuda(), **kflags)
    pg, dz = ops.prepare_bwd(sc.raw["z_rend"].cuda(), ds.mlp, dW)
    assert_close(grad_tri[:, :96].cpu(), o_tri[:, :96], "d loss / d feature planes", 1e-3)
    if modes.get("no_selector"):
        assert float(o_tri[:, 96:].abs().max()) == 0.0 and float(grad_tri[:, 96:].abs().max()) == 0.0
    else:
        assert float(o_tri[:, 96:].abs().max()) > 0
        assert_close(grad_tri[:, 96:].cpu(), o_tri[:, 96:], "d loss / d part-probability planes", 1e-3)
    for l in range(3):
        assert_close(db[l].cpu(), o_mlp[f"layers.{l}.bias"].reshape(-1), f"d bias {l}", 1e-3)
        assert_close(pg[f"layers.{l}.conv.weight"].cpu(), o_mlp[f"layers.{l}.conv.weight"], f"d layers.{l}.conv.weight", 1e-3)
    assert_close(dz.cpu(), o_z, "d z_rend", 1e-3)


@pytest.mark.parametrize("name", ["query_b1_clamp_multw", "query_b1_noselector"])
def test_query_modes_forward_and_backward(name):
    """clamp_mask / multiply_density_with_triplane_wieght / no_selector in the point query: forward against the REFERENCE's
    outputs (tests/golden, the reference run with those nerf_params), backward against autograd through the oracle."""
    from _helpers import load_golden
    from enarf_gan_amd import ops
    g = load_golden(name)
    sc = Scene(64, 1, "center_fixed", 20)
    sc.raw["tri_plane"] = sc.raw["tri_plane"].clone()
    sc.raw["tri_plane"][:, 96:] *= float(g["mask_scale"])
    ds = DeviceScene(sc)
    kflags = dict(multiply_density_with_weight=bool(g["mult_w"]), clamp_mask=bool(g["clamp_mask"]), uniform_part_weight=bool(g["no_selector"]))
    pts = torch.from_numpy(g["points"])
    den, col, vb = ds.query(pts, mlp_mode="f32", need_valid=True, **kflags)
    same = vb.cpu().numpy().view(np.uint32) == g["valid"]
    assert (~same).sum() <= 2
    assert_close(den.cpu().numpy()[:, 0][same], g["density"][:, 0][same], "density vs reference")
    assert_close(col.cpu().numpy().transpose(0, 2, 1)[same], g["color"].transpose(0, 2, 1)[same], "colour vs reference")
    # backward on a subset
    N = 600
    p = pts[..., :N].contiguous()
    gen = torch.Generator().manual_seed(2)
    gd, gc = torch.randn(1, 1, N, generator=gen), torch.randn(1, 3, N, generator=gen)
    tri = sc.raw["tri_plane"].clone().requires_grad_(True)
    oden, ocol, _ = O.query(p, sc.pose_scaled, sc.scale, sc.cpose, tri, sc.weights(), multiply_density_with_weight=bool(g["mult_w"]),
                            clamp_mask=bool(g["clamp_mask"]), no_selector=bool(g["no_selector"]))
    (o_tri,) = torch.autograd.grad((oden * gd).sum() + (ocol * gc).sum(), [tri])
    grad_tri, dW, db = ops.query_bwd(p.cuda(), ds.parts, ds.cpose, ds.tri, ds.feat_cl, ds.pack, gd.cuda(), gc.cuda(), **kflags)
    assert_close(grad_tri[:, :96].cpu(), o_tri[:, :96], "query: d feature planes", 1e-3)
    if not bool(g["no_selector"]):
        assert_close(grad_tri[:, 96:].cpu(), o_tri[:, 96:], "query: d part-probability planes", 1e-3)
