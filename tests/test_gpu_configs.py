"""GPU parity at the sizes BASELINE.json quotes (configs C1-C4) and bit-exact checks of the integer outputs.

* full frames against the reference with a deterministic sampler (tests/golden/full_*.npz): ray-validity map and count
  bit-exact, the uint8 foreground mask equal on every pixel except listed straddlers of a quantisation step;
* the marched-ray counter of every render fixture against the reference's `n_valid_rays`;
* C2 (16 and 32 frames with per-frame tri-planes, 128^2), C3's per-GPU share (8 frames) and C4 (256^2, Nc 72 / Nf 96, bf16
  MLP + early termination): size-independent properties on the whole batch plus an oracle slice through EVERY image;
* the split-fp16 MLP arithmetic on feature planes scaled over eight decades.
"""
import numpy as np
import pytest
import torch

from _helpers import (DeviceScene, Scene, assert_close, assert_u8_mask_matches, fullframe_case, load_golden, rel_err)
from test_gpu_parity import RENDER_CASES

pytestmark = pytest.mark.gpu


def _cpu(t):
    return t.detach().cpu()


# ------------------------------------------------------------------------------------ integer outputs, every ray of a frame
@pytest.mark.parametrize("name", ["full_c1_128_b1_p23", "full_gan_32_b2"])
@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_fullframe_integer_outputs_match_reference(name, mode):
    g, rv, bins = fullframe_case(name)
    B, S, Nc, Nf = int(g["batch"]), int(g["size"]), int(g["Nc"]), int(g["Nf"])
    sc = Scene(S, B, str(g["origin_location"]), int(g["style_dim"]))
    ds = DeviceScene(sc)
    out = ds.render(sc.raw["image_coord"], Nc, Nf, bins, mlp_mode=mode, debug=True, count=True)
    # ray validity: the map and the count, bit-exact
    ours_rv = _cpu(out.taps["ray_validity"]).numpy().astype(bool)
    assert np.array_equal(ours_rv, rv), "ray-validity map differs from the reference"
    assert np.array_equal(ours_rv.sum(axis=1), g["n_valid_rays"])
    marched = int(_cpu(out.counters)[2])
    assert marched == (int(g["n_valid_rays"].sum()) if B == 1 else B * S * S), "rays marched != the reference's valid rays"
    # float mask within the parity bound, then the integer mask pixel by pixel
    mask = _cpu(out.mask).numpy()
    assert_close(mask, g["mask"], "mask vs reference, every ray")
    bad = assert_u8_mask_matches(mask, g["mask"], f"{name} [{mode}]")
    ours_u8 = (mask * 255).astype(np.uint8)
    assert int(np.abs(ours_u8.astype(np.int64) - g["mask_u8"].astype(np.int64)).sum()) == len(bad)
    away = [b for b in bad if min(b[1], b[2]) < 0.999]          # not at the saturation step (see assert_u8_mask_matches)
    print(f"{name} [{mode}]: {len(bad)} of {mask.size} pixels straddle a uint8 step ({len(bad) - len(away)} at 254|255): "
          f"{[(i, a, b) for i, a, b in away]}")
    # dropped rays are exact zeros in the integer image as well
    if B == 1:
        assert int(ours_u8[~rv].max(initial=0)) == 0
    # the colour image as the demo stores it (ENARF_GAN_demo.py:75-77, on a black background): at most one level off
    col_u8 = np.clip(_cpu(out.color).numpy() * 127.5 + 127.5, 0, 255).astype(np.uint8)
    diff = np.abs(col_u8.astype(np.int64) - g["color_u8"].astype(np.int64))
    assert int(diff.max()) <= 1 and float((diff != 0).mean()) < 0.01
    assert abs(float(_cpu(out.disparity).double().sum()) - float(g["disparity_sum"])) < 1e-4 * abs(float(g["disparity_sum"]))


@pytest.mark.parametrize("name", RENDER_CASES)
def test_marched_ray_count_equals_reference_valid_rays(name):
    """`n_valid_rays` of every render fixture (the reference's decide_frustrum_range on the FULL frame) against the
    kernel's own count of marched rays and its validity map, on the full frame."""
    g = load_golden(name)
    B, S = int(g["batch"]), int(g["size"])
    sc = Scene(S, B, str(g["origin_location"]), int(g["style_dim"]))
    ds = DeviceScene(sc)
    out = ds.render(sc.raw["image_coord"], int(g["Nc"]), int(g["Nf"]), None, seed=1, debug=True, count=True, mlp_mode="f16x3")
    rv = _cpu(out.taps["ray_validity"]).numpy().astype(bool)
    assert np.array_equal(rv.sum(axis=1), g["n_valid_rays"])
    assert int(_cpu(out.counters)[2]) == (int(g["n_valid_rays"].sum()) if B == 1 else B * S * S)
    idx = g["ray_idx"].astype(np.int64)
    assert np.array_equal(np.take_along_axis(rv, idx, axis=1), g["ray_validity"])


@pytest.mark.parametrize("mode", ["f32", "f16x3", "bf16x3", "bf16"])
def test_march_is_bit_reproducible_in_every_mlp_mode(mode):
    """The forward march has no atomics on its data path: the same seed must give the same bits, at full frame with every
    CU holding three workgroups. (Round 1's split-precision modes did not - chained MFMAs that were not accumulated in place,
    profiles/r02_mfma_chain_hazard.md - and the small-subset tests could not see it.) Also: no mode may disagree with the
    exact one on more than a handful of (part, sample) pairs - the pair count is geometry plus the importance draw."""
    sc = Scene(128, 1, "center_fixed", 20)
    ds = DeviceScene(sc)
    coord = sc.raw["image_coord"]
    ref = ds.render(coord, 48, 64, None, seed=99, mlp_mode=mode, count=True, return_bins=True)
    for _ in range(3):
        again = ds.render(coord, 48, 64, None, seed=99, mlp_mode=mode, count=True, return_bins=True)
        assert torch.equal(again.color, ref.color) and torch.equal(again.mask, ref.mask)
        assert torch.equal(again.taps["bins"], ref.taps["bins"]) and torch.equal(again.counters, ref.counters)
    exact = ds.render(coord, 48, 64, None, seed=99, mlp_mode="f32", count=True)
    pairs, pairs_exact = int(_cpu(ref.counters)[0]), int(_cpu(exact.counters)[0])
    assert abs(pairs - pairs_exact) <= (2000 if mode == "bf16" else 40), (pairs, pairs_exact)


# ------------------------------------------------------------------------------------ BASELINE configs C2, C3 share, C4
def _body_rays(sc, per_image, seed):
    """ray ids per image, mostly through the body (around the projected root joint) plus a few anywhere"""
    S = sc.raw["size"]
    rs = np.random.RandomState(seed)
    ids = []
    for b in range(sc.B):
        root = sc.raw["pose_to_camera"][b, 0, :3, 3].numpy()
        u = 1.2 * S * root[0] / root[2] + S / 2
        v = 1.2 * S * root[1] / root[2] + S / 2
        x = np.clip(np.round(u + rs.normal(0, S / 10, per_image)), 0, S - 1).astype(np.int64)
        y = np.clip(np.round(v + rs.normal(0, S / 5, per_image)), 0, S - 1).astype(np.int64)
        ids.append(np.sort(y * S + x))
    return torch.from_numpy(np.stack(ids))


def _check_batch_properties(out, B, n, Nf, exact=True):
    m, c, d = _cpu(out.mask), _cpu(out.color), _cpu(out.disparity)
    assert torch.isfinite(m).all() and torch.isfinite(c).all() and torch.isfinite(d).all()
    assert float(m.min()) >= -1e-4 and float(m.max()) <= 1.0 + 1e-4
    assert float(c.abs().max()) <= 1.0 + 1e-4
    assert m.shape == (B, n) and c.shape == (B, 3, n)
    if exact:
        assert_close(_cpu(out.fine_weights)[:, 0].sum(-1), m, "sum of fine weights == mask", 1e-5)
    assert int(_cpu(out.counters)[2]) == B * n, "a batch drops no ray: every ray of every image is marched exactly once"
    for b in range(B):          # every image has a visible body
        assert float(m[b].max()) > 0.2 and 0.02 < float((m[b] > 0.05).float().mean()) < 0.9, b


def _oracle_slices(sc, ds, out_bins, Nc, Nf, per_image=20, seed=0, images=None, **render_kw):
    """Oracle on `per_image` rays of EVERY image with the bins the kernel drew (near / far are batch-global, so the
    oracle sees the whole batch of poses); returns the HIP outputs on the same rays."""
    ids = _body_rays(sc, per_image, seed)                                     # (B, m)
    coord = torch.gather(sc.raw["image_coord"], 3, ids[:, None, None, :].expand(-1, 1, 3, -1)).contiguous()
    bins = torch.gather(_cpu(out_bins), 1, ids[:, :, None].expand(-1, -1, Nf)).contiguous()
    sub = ds.render(coord, Nc, Nf, bins, **render_kw)
    if images is None:
        rc, rm, rd = sc.oracle_render(coord, Nc, Nf, bins, taps=False)
    else:          # the oracle on some images of a large batch only (its cost grows with the batch); HIP outputs cut to the same ones
        img = torch.tensor(images)
        rc, rm, rd = sc.oracle_render(coord[img], Nc, Nf, bins[img], taps=False, images=img)
    return ids, (rc, rm, rd), sub


@pytest.mark.parametrize("B,label", [(8, "C3 share: 8 frames per GPU"), (16, "C2: forward batch 16"), (32, "C2: batch 32")])
def test_gan_batches_with_per_image_triplanes_at_128(B, label):
    """BASELINE C2 / C3: B frames of 128^2 rays, Nc 48 + Nf 64, one tri-plane PER IMAGE (GAN style), in-kernel sampling."""
    S, Nc, Nf = 128, 48, 64
    sc = Scene(S, B, "center_fixed", 256)
    ds = DeviceScene(sc)
    n = S * S
    out = ds.render(sc.raw["image_coord"], Nc, Nf, None, seed=17, mlp_mode="f16x3", count=True, return_bins=True)
    _check_batch_properties(out, B, n, Nf)
    # oracle on a slice of EVERY image (same bins); the sub-render on those rays is bit-identical to the full launch
    images = None if B <= 8 else sorted({0, B // 3, 2 * B // 3, B - 1})          # larger batches: four of the frames restated
    ids, (rc, rm, rd), sub = _oracle_slices(sc, ds, out.taps["bins"], Nc, Nf, per_image=20, seed=B, images=images, mlp_mode="f16x3")
    assert float(rm.max()) > 0.5
    full_m = torch.gather(_cpu(out.mask), 1, ids)
    assert torch.equal(_cpu(sub.mask), full_m), "rays are independent: a subset rendered alone gives the same bits"
    cut = (lambda t: _cpu(t)) if images is None else (lambda t: _cpu(t)[torch.tensor(images)])
    assert_close(cut(sub.color), rc, f"{label}: colour vs oracle")
    assert_close(cut(sub.mask), rm, f"{label}: mask vs oracle")
    assert_close(cut(sub.disparity), rd, f"{label}: disparity vs oracle")
    # images differ (per-image tri-planes and poses really are used)
    assert not torch.equal(_cpu(out.mask)[0], _cpu(out.mask)[B - 1])
    # permutation of the batch permutes the outputs bit for bit (same bins): no state leaks between images
    perm = torch.arange(B - 1, -1, -1)
    s = sc.raw
    from enarf_gan_amd import ops
    d = ds.dev
    parts_p, pack_p = ops.prepare(s["pose_to_camera"][perm].to(d), s["bone_length"][perm].to(d), sc.cbl.to(d),
                                  s["z_rend"][perm].to(d), ds.mlp, s["parents"], sc.ol, sc.cs)
    tri_p = ds.tri[perm.to(d)].contiguous()
    out_p = ops.render_fwd(s["image_coord"][perm].to(d), ds.inv_K[perm.to(d)], parts_p, ds.cpose, tri_p, ops.triplane_pack(tri_p),
                           pack_p, Nc, Nf, bins=out.taps["bins"][perm.to(d)].contiguous(), mlp_mode="f16x3")
    assert torch.equal(out_p.mask, out.mask[perm.to(d)]) and torch.equal(out_p.color, out.color[perm.to(d)])


def test_c4_256_nc72_nf96_bf16_early_termination():
    """BASELINE C4: 256^2 rays, Nc 72 + Nf 96 (two samples per lane), bf16 MLP, early ray termination; 2 frames.
    Exact arithmetic (f32, eps 0) against the oracle on a slice of every image; the throughput configuration
    (bf16 + eps 1e-3) against the exact render on the whole frames."""
    S, B, Nc, Nf = 256, 2, 72, 96
    sc = Scene(S, B, "center_fixed", 256)
    ds = DeviceScene(sc)
    n = S * S
    exact = ds.render(sc.raw["image_coord"], Nc, Nf, None, seed=23, mlp_mode="f32", count=True, return_bins=True)
    _check_batch_properties(exact, B, n, Nf)
    ids, (rc, rm, rd), sub = _oracle_slices(sc, ds, exact.taps["bins"], Nc, Nf, per_image=24, seed=4, mlp_mode="f32")
    assert torch.equal(_cpu(sub.mask), torch.gather(_cpu(exact.mask), 1, ids))
    assert_close(_cpu(sub.color), rc, "C4 f32: colour vs oracle")
    assert_close(_cpu(sub.mask), rm, "C4 f32: mask vs oracle")
    assert_close(_cpu(sub.disparity), rd, "C4 f32: disparity vs oracle")
    fast = ds.render(sc.raw["image_coord"], Nc, Nf, _cpu(exact.taps["bins"]), mlp_mode="bf16", early_stop_eps=1e-3, count=True)
    _check_batch_properties(fast, B, n, Nf, exact=False)
    # one-term bf16 operands carry 2^-9 relative rounding into three 64-deep layers: a few per cent on the worst of
    # 131072 rays (measured 2.4e-2 mask / 8.0e-2 colour), 1e-2 at the 99.9th percentile, 1e-3 on average
    for a, b, what in ((fast.mask, exact.mask, "mask"), (fast.color, exact.color, "colour")):
        e = rel_err(_cpu(a), _cpu(b))
        assert e.max() < 0.2 and np.quantile(e, 0.999) < 3e-2, (what, e.max(), np.quantile(e, 0.999))
    assert float((fast.mask - exact.mask).abs().mean()) < 2e-3
    assert int(_cpu(fast.counters)[0]) <= int(_cpu(exact.counters)[0])
    # bf16 alone (no termination) on the same bins: the MLP rounding is the whole difference
    b16 = ds.render(sc.raw["image_coord"], Nc, Nf, _cpu(exact.taps["bins"]), mlp_mode="bf16")
    assert rel_err(_cpu(b16.mask), _cpu(exact.mask)).max() < 0.2


# ------------------------------------------------------------------------------------ range of the split-fp16 MLP arithmetic
def _to_f64(x):
    if torch.is_tensor(x):
        return x.double() if x.is_floating_point() else x
    if isinstance(x, dict):
        return {k: _to_f64(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_to_f64(v) for v in x)
    return x


@pytest.mark.parametrize("scale", [1e-4, 1.0, 1e4, 1e6])
def test_mlp_arithmetic_modes_over_feature_scales(scale):
    """`f16x3` (the default: 3-term split fp16 on MFMA) and `f32` with the FEATURE planes scaled by 1e-4 ... 1e+4 (the conv
    weights are row-normalised by the demodulation, custom_stylegan2/net.py:236-243, so their scale cancels; the feature
    magnitude is what reaches the MLP). The fp16 pair carries 22 bits while |operand| <= 65504; a tile in which a feature
    or a hidden activation exceeds that is evaluated again scaled by a power of two (exact: the network is positively
    homogeneous once the biases are scaled along, enarf_query.h mlp_tile), so the mode has fp32's range - x 1e4 and
    x 1e6 exercise that path (|feature| up to 1.4e5 / 1.4e7). The low half goes subnormal below |x| ~ 1e-4, where the
    absolute error is < 1e-7.

    Bound: 1e-4 against the fp32 oracle while fp32 itself is that well conditioned. At feature magnitudes of 1e2 and more
    it is not - the oracle in fp32 moves by 1e-3 ... 7e-3 against the same oracle in fp64 on a few tenths of a per cent of
    the colours (cancellation in the hidden layers) - so there both modes are held to the fp64 result with the error the
    reference's own fp32 arithmetic shows against it (x3, and twice its share of elements above 1e-4)."""
    sc = Scene(64, 1, "center_fixed", 20)
    sc.raw["tri_plane"] = sc.raw["tri_plane"].clone()
    sc.raw["tri_plane"][:, :96] *= scale
    ds = DeviceScene(sc)
    g = torch.Generator().manual_seed(3)
    centre = sc.pose_scaled[0, :, :3, 3]
    k = torch.randint(0, sc.P, (6000,), generator=g)
    pts = (centre[k] + torch.randn(6000, 3, generator=g) * 0.5).t()[None].contiguous()
    from oracle import enarf_oracle as O
    oden, ocol, ovalid, taps = O.query(pts, sc.pose_scaled, sc.scale, sc.cpose, sc.raw["tri_plane"], sc.weights(), return_taps=True)
    assert int(ovalid.any(dim=1).sum()) > 1000
    tden, tcol, tvalid = O.query(pts.double(), sc.pose_scaled.double(), sc.scale.double(), sc.cpose.double(),
                                 sc.raw["tri_plane"].double(), _to_f64(sc.weights()))
    assert torch.equal(tvalid, ovalid)
    ref_err = {"density": rel_err(oden, tden), "colour": rel_err(ocol, tcol)}
    feat_max = float(taps["feature"].abs().max())
    for mode in ("f32", "f16x3"):
        den, col = ds.query(pts, mlp_mode=mode)
        for name, ours, o32, t64 in (("density", den, oden, tden), ("colour", col, ocol, tcol)):
            what = f"{name}, features x {scale:g} (|feature| up to {feat_max:.3g}) [{mode}]"
            if ref_err[name].max() < 5e-5:                      # fp32 well conditioned: the parity bound proper
                assert_close(_cpu(ours), o32, what)
            else:
                e = rel_err(_cpu(ours), t64)
                assert e.max() <= 3 * ref_err[name].max(), (what, e.max(), ref_err[name].max())
                assert (e > 1e-4).mean() <= 2 * (ref_err[name] > 1e-4).mean() + 1e-4, what
    # and through the march: the bound on the image (fp32's own conditioning, measured above, where it is worse)
    tol = max(1e-4, 3 * float(ref_err["colour"].max()))
    coord = sc.raw["image_coord"][..., 64 * 28:64 * 28 + 128].contiguous()
    bins = torch.rand(1, 128, 32, generator=g).sort(-1).values
    rc, rm, rd = sc.oracle_render(coord, 48, 32, bins, taps=False)
    for mode in ("f32", "f16x3"):
        out = ds.render(coord, 48, 32, bins, mlp_mode=mode)
        assert_close(_cpu(out.mask), rm, f"mask, features x {scale:g} [{mode}]")
        assert_close(_cpu(out.color), rc, f"colour, features x {scale:g} [{mode}]", tol)


# ------------------------------------------------------------------------------------ batches marched in groups of frames
def test_grouped_batch_gives_the_same_bits_as_one_launch():
    """A batch with one tri-plane per frame is marched in groups of frames (enarf_render_args.group_frames, default 8: one
    frame per XCD). The near / far planes are reduced once over the WHOLE batch (rendering.py:15-17) and the in-kernel
    sampler counts its rays batch-wide, so every grouping gives the same bits as one launch: outputs, drawn bins and
    counters, through enarf_render_fwd and through the fused enarf_render_step_fwd; the backward's gradients agree to the
    rounding of their float atomics."""
    from enarf_gan_amd import ops
    S, B, Nc, Nf = 64, 12, 48, 32
    sc = Scene(S, B, "center_fixed", 256)
    ds = DeviceScene(sc)
    s = dict(sc.raw)
    s["pose_to_camera"] = s["pose_to_camera"].clone()
    s["pose_to_camera"][7, :, 2, 3] += 3.0               # one frame far behind the others: it alone sets the far plane
    d = ds.dev
    # re-prepare with the moved frame
    parts, pack = ops.prepare(s["pose_to_camera"].to(d), s["bone_length"].to(d), ds.sc.cbl.to(d), s["z_rend"].to(d), ds.mlp,
                              s["parents"], ds.sc.ol, ds.sc.cs)
    coord = s["image_coord"].to(d)

    def render(g):
        return ops.render_fwd(coord, ds.inv_K, parts, ds.cpose, ds.tri, ds.feat_cl, pack, Nc, Nf, seed=5, mlp_mode="f16x3",
                              count=True, return_bins=True, group_frames=g)

    one = render(B)
    assert float(one.mask.max()) > 0.5
    for g in (0, 5, 1):
        o = render(g)
        for name in ("color", "mask", "disparity", "fine_weights", "fine_depth"):
            assert torch.equal(getattr(o, name), getattr(one, name)), (g, name)
        assert torch.equal(o.taps["bins"], one.taps["bins"]) and torch.equal(o.counters, one.counters), g
    # each group reducing near / far over its own frames would NOT give these bits: the far frame moves the planes
    alone = ops.render_fwd(coord[:6], ds.inv_K[:6], parts[:6].contiguous(), ds.cpose, ds.tri[:6].contiguous(), ds.feat_cl[:6].contiguous(),
                           pack[:6].contiguous(), Nc, Nf, seed=5, mlp_mode="f16x3")
    assert not torch.equal(alone.fine_depth, one.fine_depth[:6])
    # the fused step (re-layout + prepare + set-up in the pre-march launch of every group)
    def step(g):
        feat = torch.empty_like(ds.feat_cl)
        st = ops.RenderStep(s["pose_to_camera"].to(d), s["bone_length"].to(d), ds.sc.cbl.to(d), s["z_rend"].to(d), ds.mlp,
                            s["parents"], ds.sc.ol, ds.sc.cs, coord, ds.inv_K, ds.cpose, ds.tri, feat, Nc, Nf, seed=5,
                            mlp_mode="f16x3", return_bins=True, group_frames=g)
        return st.run()
    s_one, s_grp = step(B), step(0)
    assert torch.equal(s_one.mask, one.mask) and torch.equal(s_grp.mask, one.mask) and torch.equal(s_grp.color, one.color)
    assert torch.equal(s_grp.taps["bins"], one.taps["bins"])
    # backward: grouped against one launch
    g = torch.Generator(device=d).manual_seed(1)
    n = S * S
    gc, gm = torch.randn(B, 3, n, device=d, generator=g), torch.randn(B, n, device=d, generator=g)
    outs = [ops.render_bwd(coord, ds.inv_K, parts, ds.cpose, ds.tri, ds.feat_cl, pack, Nf, one.taps["bins"], gc, gm, group_frames=gf)
            for gf in (B, 0)]
    assert_close(outs[1][0].cpu(), outs[0][0].cpu(), "grad_tri, grouped backward", 2e-5)
    for l in range(3):
        assert_close(outs[1][1][l].cpu(), outs[0][1][l].cpu(), f"dW{l}, grouped backward", 2e-5)


def test_first_launch_of_a_fresh_process_equals_the_second():
    """Regression test of commit 62f9e8b (render_kernel lost its barrier after staging the FIRST image context when the bin
    table moved up: the first rays of a launch could run on a half-staged MLP pack - visible only on the first launch of a
    process, with cold instruction caches and LDS, and not in a long-lived pytest process). A fresh interpreter renders the
    same frames twice with both march kernels, first launch on a fresh workspace included; every launch must give the same bits."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import hashlib, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from _helpers import DeviceScene, Scene
sc = Scene(64, 2, "center_fixed", 20)
ds = DeviceScene(sc)
for march in ("ray", "task"):
    sigs = []
    for rep in range(3):
        o = ds.render(sc.raw["image_coord"], 48, 32, None, seed=4, mlp_mode="f16x3", march=march, count=True, return_bins=True)
        torch.cuda.synchronize()
        h = hashlib.md5(b"".join(t.cpu().numpy().tobytes() for t in (o.color, o.mask, o.disparity, o.fine_depth, o.taps["bins"]))).hexdigest()
        sigs.append((h, tuple(o.counters.cpu().tolist())))
    print("SIG", march, len(set(sigs)), sigs[0][0], sigs[0][1][7])
''' % (root, os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    lines = [l.split() for l in r.stdout.splitlines() if l.startswith("SIG")]
    assert len(lines) == 2, r.stdout + r.stderr
    for _, march, distinct, h, watchdog in lines:
        assert distinct == "1" and watchdog == "0", (march, r.stdout)
    assert lines[0][3] == lines[1][3], "the two march kernels disagree on a first launch"
