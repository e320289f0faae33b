"""Pin the oracle (oracle/enarf_oracle.py) against outputs of the reference's own Python code.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py, which imports the
reference unmodified in the build container. The reference itself ships no tests or golden vectors
(SURVEY.md §4), so these are the only pins that exist for this path.

Tolerances: 1e-4 relative to the tensor's scale for floating point (the north-star bound); validity
bit masks must agree exactly except for (part, point) pairs lying within a few ulp of a cube face,
where the reference's torch.matmul summation order is backend-defined (count reported, capped).
"""
import os

import numpy as np
import pytest
import torch

from enarf_gan_amd import synth
from oracle import enarf_oracle as O

RTOL = 1e-4


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False))


def _scene_for(g, size=None):
    return synth.make_scene(int(g["size"]) if size is None else size, int(g["batch"]),
                            str(g["origin_location"]), int(g["style_dim"]))


def _assert_close(ours, ref, what, rtol=RTOL, frac_ok=0.0):
    ours = np.asarray(ours, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = max(np.abs(ref).max(), 1e-6)
    err = np.abs(ours - ref) / scale
    bad = (err > rtol).mean()
    assert bad <= frac_ok, f"{what}: max rel err {err.max():.3e} (scale {scale:.3g}), {bad * 100:.3f}% above {rtol}"
    return err.max()


def test_sampler_matches_reference_grid_sample(golden_dir):
    g = _load(golden_dir, "sampler_b2")
    inp = torch.from_numpy(g["input"])
    pos = torch.from_numpy(g["position"])
    grid = pos.permute(0, 2, 1)[:, :, None, :].contiguous()          # (B,n,1,3), sampling.py:25
    out = O.triplane_sampler_forward(inp, grid)[..., 0]
    _assert_close(out, g["output"], "sampler fwd", 1e-5)
    gi, gg = O.triplane_sampler_backward(torch.from_numpy(g["grad_output"])[..., None], inp, grid)
    _assert_close(gi, g["grad_input"], "sampler grad_input", 1e-5)
    _assert_close(gg[:, :, 0].permute(0, 2, 1), g["grad_position"], "sampler grad_grid", 1e-5)


@pytest.mark.parametrize("name", ["query_b2_p23", "query_b1_p24"])
def test_query_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    B, ol, sd = int(g["batch"]), str(g["origin_location"]), int(g["style_dim"])
    scene = synth.make_scene(64, B, ol, sd)
    pose_p, bl_p = O.transform_pose(scene["pose_to_camera"], scene["bone_length"], ol, scene["parents"])
    cpose, cbl = O.register_canonical_pose(scene["canonical_pose"], scene["parents"], ol)
    pose_s = O.scale_pose_translation(pose_p, 3.0)
    scale = O.canonical_scale(cbl, bl_p, 3.0)
    weights = O.modulated_weights(scene["mlp"], scene["z_rend"])
    NP = 1536                                    # points are independent: a prefix keeps the CPU suite short
    g = {k: (v[..., :NP] if k in ("points", "density", "color", "valid", "weight") else v) for k, v in g.items()}
    pts = torch.from_numpy(g["points"])
    den, col, valid, taps = O.query(pts, pose_s, scale, cpose, scene["tri_plane"], weights, return_taps=True)
    ours_bits = (valid.numpy().astype(np.uint32) <<
                 np.arange(valid.shape[1], dtype=np.uint32)[None, :, None]).sum(axis=1).astype(np.uint32)
    mism = int((ours_bits != g["valid"]).sum())
    assert mism <= 2, f"validity bit masks differ from the reference on {mism} points"
    same = ours_bits == g["valid"]
    _assert_close(taps["canonical"][:, :, :, :256], g["canonical"], "canonical", 1e-5)
    _assert_close(taps["weight"].numpy()[np.broadcast_to(same[:, None], taps["weight"].shape)],
                  g["weight"][np.broadcast_to(same[:, None], g["weight"].shape)], "part prob")
    _assert_close(den.numpy()[:, 0][same], g["density"][:, 0][same], "density")
    _assert_close(col.numpy().transpose(0, 2, 1)[same], g["color"].transpose(0, 2, 1)[same], "color")
    # the explicit bilinear restatement and F.grid_sample are the same arithmetic
    den2, col2, _ = O.query(pts, pose_s, scale, cpose, scene["tri_plane"], weights, use_grid_sample=True)
    _assert_close(den2, den, "density explicit vs grid_sample", 1e-5)


@pytest.mark.parametrize("name", ["query_b1_clamp_multw", "query_b1_noselector"])
def test_query_modes_match_reference(golden_dir, name):
    """clamp_mask (sampling.py:46-47), multiply_density_with_triplane_wieght (narf.py:271-272) and no_selector (narf.py:133-134)
    in the oracle against the reference run with those nerf_params."""
    g = _load(golden_dir, name)
    scene = synth.make_scene(64, 1, "center_fixed", 20)
    scene["tri_plane"][:, 96:] *= float(g["mask_scale"])
    pose_p, bl_p = O.transform_pose(scene["pose_to_camera"], scene["bone_length"], "center_fixed", scene["parents"])
    cpose, cbl = O.register_canonical_pose(scene["canonical_pose"], scene["parents"], "center_fixed")
    pts = torch.from_numpy(g["points"])
    den, col, valid, taps = O.query(pts, O.scale_pose_translation(pose_p, 3.0), O.canonical_scale(cbl, bl_p, 3.0), cpose,
                                    scene["tri_plane"], O.modulated_weights(scene["mlp"], scene["z_rend"]), return_taps=True,
                                    multiply_density_with_weight=bool(g["mult_w"]), clamp_mask=bool(g["clamp_mask"]),
                                    no_selector=bool(g["no_selector"]))
    bits = (valid.numpy().astype(np.uint32) << np.arange(valid.shape[1], dtype=np.uint32)[None, :, None]).sum(axis=1).astype(np.uint32)
    same = bits == g["valid"]
    assert (~same).sum() <= 2
    _assert_close(taps["weight"].numpy()[np.broadcast_to(same[:, None], taps["weight"].shape)],
                  g["weight"][np.broadcast_to(same[:, None], g["weight"].shape)], "part weight")
    _assert_close(den.numpy()[:, 0][same], g["density"][:, 0][same], "density")
    _assert_close(col.numpy().transpose(0, 2, 1)[same], g["color"].transpose(0, 2, 1)[same], "colour")


RENDER_CASES = ["render_c0_64_b1", "render_c1_128_b1_p23", "render_c1_128_b1_p24", "render_gan_32_b2",
                "render_c4s_32_b2"]       # the last one: BASELINE config C4's sample counts, Nc 72 / Nf 96 (> 64)


@pytest.mark.parametrize("name", RENDER_CASES)
def test_render_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    scene = _scene_for(g)
    B, ol = int(g["batch"]), str(g["origin_location"])
    Nc, Nf = int(g["Nc"]), int(g["Nf"])
    pose_p, bl_p = O.transform_pose(scene["pose_to_camera"], scene["bone_length"], ol, scene["parents"])
    cpose, cbl = O.register_canonical_pose(scene["canonical_pose"], scene["parents"], ol)
    idx = torch.from_numpy(g["ray_idx"].astype(np.int64))                       # (B,m)
    coord = torch.gather(scene["image_coord"], 3, idx[:, None, None, :].expand(-1, 1, 3, -1))
    bins = torch.from_numpy(g["bins"])
    # B == 1 ray dropping is decided per ray, so rendering a subset of rays is exact
    rc, rm, rd, taps = O.render(coord, pose_p, bl_p, scene["inv_intrinsics"], cpose, cbl,
                                scene["tri_plane"], scene["mlp"], scene["z_rend"], 3.0, Nc, Nf,
                                bins=bins, return_taps=True)
    # (the fixture's near/far are render()'s un-updated arguments 0.3 / 5, rendering.py:205-213; the
    # batch-global planes of rendering.py:15-17 are pinned through depth_min / depth_max below)
    assert np.array_equal(taps["ray_validity"].numpy(), g["ray_validity"]), "ray validity differs"
    _assert_close(taps["depth_min"], g["depth_min"], "depth_min", 1e-6)
    _assert_close(taps["depth_max"], g["depth_max"], "depth_max", 1e-6)
    live = g["ray_validity"] if B == 1 else np.ones_like(g["ray_validity"])
    fv = taps["fine_valid"].numpy().astype(np.uint32)                             # (B,P,m,Nf)
    bits = (fv << np.arange(fv.shape[1], dtype=np.uint32)[None, :, None, None]).sum(axis=1).astype(np.uint32)
    mism = int((bits != g["fine_valid"])[live].sum())
    assert mism <= 2, f"fine-sample validity masks differ from the reference on {mism} samples"
    _assert_close(taps["fine_depth"].numpy()[live], g["fine_depth"][live], "fine_depth", 1e-6)
    frac = 2e-3 if mism else 0.0
    _assert_close(taps["coarse_density"].numpy()[live], g["coarse_density"][live], "coarse density", frac_ok=frac)
    _assert_close(taps["fine_density"].numpy()[live], g["fine_density"][live], "fine density", frac_ok=frac)
    _assert_close(taps["fine_weights"].numpy()[live], g["fine_weights"][live], "fine weights", frac_ok=frac)
    _assert_close(rc, g["color"], "color", frac_ok=frac)
    _assert_close(rm, g["mask"], "mask", frac_ok=frac)
    _assert_close(rd, g["disparity"], "disparity", frac_ok=frac)
    # the integer foreground mask (ENARF_GAN_demo.py:79): uint8 quantisation agrees wherever the float
    # mask is not within 1e-4 of a quantisation step
    ours_u8 = (rm.numpy() * 255).astype(np.uint8)
    ref_u8 = (g["mask"] * 255).astype(np.uint8)
    frac_part = (g["mask"].astype(np.float64) * 255) % 1.0
    safe = (frac_part > 1e-2) & (frac_part < 1 - 1e-2)
    assert np.array_equal(ours_u8[safe], ref_u8[safe])


def test_fullframe_integer_outputs_match_reference(golden_dir):
    """Every ray of the 2 x 32^2 GAN fixture (deterministic sampler, bins rebuilt from the fixture's uint8 indices): the
    oracle's ray-validity map and count equal the reference's, and its uint8 foreground mask equals the reference's on
    every pixel except those whose float values straddle a quantisation step (listed, each within 1e-4 of the other)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _helpers import assert_u8_mask_matches, fullframe_case
    g, rv, bins = fullframe_case("full_gan_32_b2")
    scene = _scene_for(g)
    ol = str(g["origin_location"])
    pose_p, bl_p = O.transform_pose(scene["pose_to_camera"], scene["bone_length"], ol, scene["parents"])
    cpose, cbl = O.register_canonical_pose(scene["canonical_pose"], scene["parents"], ol)
    rc, rm, rd, taps = O.render(scene["image_coord"], pose_p, bl_p, scene["inv_intrinsics"], cpose, cbl, scene["tri_plane"],
                                scene["mlp"], scene["z_rend"], 3.0, int(g["Nc"]), int(g["Nf"]), bins=bins, return_taps=True)
    assert np.array_equal(taps["ray_validity"].numpy(), rv)
    assert np.array_equal(taps["ray_validity"].sum(dim=1).numpy(), g["n_valid_rays"])
    _assert_close(rm, g["mask"], "mask, every ray")
    bad = assert_u8_mask_matches(rm.numpy(), g["mask"], "oracle vs reference, 2 x 32^2")
    ours_sum = int((rm.numpy() * 255).astype(np.uint8).astype(np.int64).sum())
    assert abs(ours_sum - int(g["mask_u8"].astype(np.int64).sum())) <= len(bad)


def test_linspace_formula_is_torch_linspace():
    for (a, b, n) in [(0.0, 1.0, 49), (0.3, 12.75, 32), (6.123, 14.9, 32), (0.0, 1.0, 73)]:
        ours = O.linspace_sym(a, b, n)
        ref = torch.linspace(a, b, n)
        assert torch.allclose(ours, ref, rtol=0, atol=2e-7 * max(abs(a), abs(b), 1.0))


@pytest.mark.parametrize("name", ["grad_32_b1", "grad_32_b2"])
def test_oracle_autograd_matches_reference_gradients(golden_dir, name):
    """Autograd through the oracle (incl. its MyReLU restatement) against the reference's own autograd."""
    import torch.nn.functional as F
    g = _load(golden_dir, name)
    B, ol, sd = int(g["batch"]), str(g["origin_location"]), int(g["style_dim"])
    scene = synth.make_scene(int(g["size"]), B, ol, sd)
    pose_p, bl_p = O.transform_pose(scene["pose_to_camera"], scene["bone_length"], ol, scene["parents"])
    cpose, cbl = O.register_canonical_pose(scene["canonical_pose"], scene["parents"], ol)
    s0, nr = int(g["start"]), int(g["n_rays"])
    coord = scene["image_coord"][..., s0:s0 + nr].contiguous()
    tri = scene["tri_plane"].clone().requires_grad_(True)
    mlp = {k: v.clone().requires_grad_(True) for k, v in scene["mlp"].items() if "noise" not in k}
    z = scene["z_rend"].clone().requires_grad_(True)
    rc, rm, rd = O.render(coord, pose_p, bl_p, scene["inv_intrinsics"], cpose, cbl, tri, mlp, z, 3.0, int(g["Nc"]),
                          int(g["Nf"]), bins=torch.from_numpy(g["bins"]))
    _assert_close(rc.detach(), g["color"], "colour")
    loss = (rc * torch.from_numpy(g["g_color"])).sum() + (rm * torch.from_numpy(g["g_mask"])).sum() + \
        (rd * torch.from_numpy(g["g_disp"])).sum()
    keys = sorted(mlp)
    grads = torch.autograd.grad(loss, [tri, z] + [mlp[k] for k in keys])
    _assert_close(F.avg_pool2d(grads[0], 16) * 256, g["grad_tri_pool16"], "d tri-plane (16x16 sum-pooled)", 1e-3)
    assert abs(float(grads[0].abs().sum()) - float(g["grad_tri_abs_sum"])) < 1e-3 * float(g["grad_tri_abs_sum"])
    _assert_close(grads[1], g["grad_z"], "d z_rend", 1e-3)
    for k, gk in zip(keys, grads[2:]):
        _assert_close(gk, g["grad_" + k], "d " + k, 1e-3)
