"""GPU tests of the Python mirror of the reference interface (model / generator / operator entry points)."""
import numpy as np
import pytest
import torch

from _helpers import Scene, assert_close, load_golden, rel_err
from oracle import enarf_oracle as O
from test_host_cpu import Cfg, _nerf_cfg

pytestmark = pytest.mark.gpu


def _model(sc: Scene, Nc=48, Nf=64, style_dim=20, mlp_mode="f16x3"):
    from enarf_gan_amd.models.narf import TriPlaneNARF
    m = TriPlaneNARF(_nerf_cfg(origin_location=sc.ol, Nc=Nc, Nf=Nf, mlp_mode=mlp_mode), style_dim, 24,
                     parent=sc.raw["parents"], num_bone_param=23)
    m.register_canonical_pose(sc.raw["canonical_pose"])
    m.load_state_dict({f"mlp.{k}": v for k, v in sc.raw["mlp"].items()}, strict=False)
    with torch.no_grad():
        m.tri_plane.copy_(sc.raw["tri_plane"][:1])
    return m.cuda().eval()


@pytest.mark.parametrize("name", ["render_c1_128_b1_p23", "render_c1_128_b1_p24"])
def test_model_forward_matches_reference_golden(name):
    g = load_golden(name)
    sc = Scene(int(g["size"]), 1, str(g["origin_location"]), int(g["style_dim"]))
    m = _model(sc, style_dim=int(g["style_dim"]))
    idx = torch.from_numpy(g["ray_idx"].astype(np.int64))
    coord = torch.gather(sc.raw["image_coord"], 3, idx[:, None, None, :].expand(-1, 1, 3, -1)).contiguous().cuda()
    s = sc.raw
    with torch.no_grad():
        color, mask, disp = m(1, coord, s["pose_to_camera"].cuda(), s["inv_intrinsics"].cuda(), None, s["z_rend"].cuda(),
                              s["bone_length"].cuda(), Nc=48, Nf=64, return_disparity=True,
                              bins=torch.from_numpy(g["bins"]).cuda())
    assert_close(color.cpu(), g["color"], "colour vs reference", frac_ok=2e-3)
    assert_close(mask.cpu(), g["mask"], "mask vs reference", frac_ok=2e-3)
    assert_close(disp.cpu(), g["disparity"], "disparity vs reference", frac_ok=2e-3)
    assert m.buffers_tensors["fine_weights"].shape == (1, 1, coord.shape[-1], 63)
    assert m.buffers_tensors["fine_depth"].shape == (1, 1, coord.shape[-1], 64)
    assert m.buffers_tensors["tri_plane_feature"].shape[1] == 96 + 3 * sc.P


def test_model_query_entry_point_matches_reference_golden():
    g = load_golden("query_b1_p24")
    sc = Scene(64, 1, "center+head", 20)
    m = _model(sc)
    pts = torch.from_numpy(g["points"]).cuda()
    mi = {"z": None, "z_rend": sc.raw["z_rend"].cuda(), "bone_length": sc.bl_parts.cuda(), "truncation_psi": 1}
    with torch.no_grad():
        den, col = m.calc_density_and_color_from_camera_coord_v2(pts, sc.pose_scaled.cuda(), None, mi)
    same = m.temporal_state["valid_bits"].cpu().numpy().view(np.uint32) == g["valid"]
    assert (~same).sum() <= 2
    assert_close(den.cpu().numpy()[:, 0][same], g["density"][:, 0][same], "density vs reference")
    assert_close(col.cpu().numpy().transpose(0, 2, 1)[same], g["color"].transpose(0, 2, 1)[same], "colour vs reference")


def _dso_generator(sc, size, Nc=48, Nf=32, ray_batchsize=512):
    from enarf_gan_amd.models.generator import DSONARFGenerator
    gen = DSONARFGenerator(Cfg(use_triplane=True, ray_batchsize=ray_batchsize, nerf_params=_nerf_cfg(Nc=Nc, Nf=Nf)), size, 24,
                           sc.raw["parents"], 23)
    gen.register_canonical_pose(sc.raw["canonical_pose"])
    gen.nerf.load_state_dict({f"mlp.{k}": v for k, v in sc.raw["mlp"].items()}, strict=False)
    with torch.no_grad():
        gen.nerf.tri_plane.copy_(sc.raw["tri_plane"][:1])
    return gen.cuda().eval()


def _oracle_for(sc, coord, z_rend, Nc, Nf, bins, inv_K=None):
    s = sc.raw
    return O.render(coord, sc.pose_parts, sc.bl_parts, s["inv_intrinsics"] if inv_K is None else inv_K.cpu(), sc.cpose, sc.cbl,
                    s["tri_plane"], s["mlp"], z_rend.cpu(), 3.0, Nc, Nf, bins=bins.cpu())


@pytest.mark.parametrize("variant", ["plain", "bbox", "normalized"])
def test_dso_render_entire_img_matches_oracle(variant):
    """DSONARFGenerator.render_entire_img (models/generator.py:256-278 -> rendering.py:362-427), whole frame, against the
    oracle with the importance samples the call drew: plain, a bounding box (:383-385, pixel offsets) and normalised
    intrinsics (:390-394, coordinates / render_size with K scaled accordingly)."""
    S, Nc, Nf = 64, 48, 32
    sc = Scene(S, 1, "center_fixed", 20)
    gen = _dso_generator(sc, S, Nc, Nf)
    s = sc.raw
    ft = torch.tensor([0.37]).cuda()
    z1, _ = gen.get_latents(ft, s["pose_to_camera"].cuda())
    K_inv = s["inv_intrinsics"].cuda()
    kw, W, H, x0, y0 = {}, S, S, 0, 0
    if variant == "bbox":
        x0, y0, W, H = 8, 16, 48, 32
        kw["bbox"] = (x0, y0, x0 + W, y0 + H)
    if variant == "normalized":
        Kn = s["intrinsics"][0].clone()
        Kn[:2] /= S
        K_inv = torch.linalg.inv(Kn)[None].cuda()
        kw["use_normalized_intrinsics"] = True
    color, mask, disp = gen.render_entire_img(s["pose_to_camera"].cuda(), K_inv, ft, s["bone_length"].cuda(), None, S, no_grad=True, **kw)
    assert color.shape == (3, H, W) and mask.shape == (H, W) and disp.shape == (H, W)
    bins = gen.nerf.buffers_tensors["bins"]
    idx = torch.arange(W * H)
    coord = torch.stack([(idx % W).float() + 0.5 + x0, torch.div(idx, W, rounding_mode="floor").float() + 0.5 + y0,
                         torch.ones(W * H)], dim=0)
    if variant == "normalized":
        coord[:2] /= S                                            # rendering.py:391-392
    rc, rm, rd = _oracle_for(sc, coord[None, None], z1, Nc, Nf, bins, K_inv)
    assert float(rm.max()) > 0.2
    assert_close(color.cpu().reshape(1, 3, -1), rc, f"render_entire_img[{variant}] colour")
    assert_close(mask.cpu().reshape(1, -1), rm, f"render_entire_img[{variant}] mask")
    assert_close(disp.cpu().reshape(1, -1), rd, f"render_entire_img[{variant}] disparity")


def test_dso_generator_forward_matches_oracle_on_sampled_rays():
    """DSONARFGenerator.forward (models/generator.py:219-254): mask-based ray sampling on the device, the march on those
    rays, composition with the background - colour, mask and ray ids against the oracle on the same rays and samples."""
    S, Nc, Nf, nray = 64, 48, 32, 512
    sc = Scene(S, 1, "center_fixed", 20)
    gen = _dso_generator(sc, S, Nc, Nf, nray)
    s = sc.raw
    ft = torch.tensor([0.8]).cuda()
    with torch.no_grad():
        _, m0, _ = gen.render_entire_img(s["pose_to_camera"].cuda(), s["inv_intrinsics"].cuda(), ft, s["bone_length"].cuda(), None, S)
        fg = (m0 > 0.05).float()[None]
        col, msk, ray_idx = gen(s["pose_to_camera"].cuda(), None, fg, ft, s["bone_length"].cuda(), s["inv_intrinsics"].cuda(),
                                background=0.25)
    assert col.shape == (1, 3, nray) and msk.shape == (1, nray) and ray_idx.shape == (1, nray) and ray_idx.dtype == torch.int64
    ids = ray_idx[0].cpu()
    assert len(set(ids.tolist())) == nray and int(ids.min()) >= 0 and int(ids.max()) < S * S
    # every sampled ray lies in the 129 x 129 dilation of the foreground (here: the whole 64^2 frame) - checked for real
    # in test_mask_based_sampler_matches_torch; the renderer's part:
    z1, _ = gen.get_latents(ft, s["pose_to_camera"].cuda())
    coord = torch.stack([(ids % S).float() + 0.5, torch.div(ids, S, rounding_mode="floor").float() + 0.5, torch.ones(nray)])[None, None]
    rc, rm, _ = _oracle_for(sc, coord, z1, Nc, Nf, gen.nerf.buffers_tensors["bins"])
    assert_close(msk.cpu(), rm, "forward: mask")
    assert_close(col.cpu(), rc + 0.25 * (1 - rm[:, None]), "forward: colour over the background")


@pytest.mark.parametrize("B,h,w,k,r", [(2, 64, 64, 512, 64), (1, 200, 150, 4096, 64), (3, 40, 56, 100, 5), (1, 512, 512, 4096, 64)])
def test_mask_based_sampler_matches_torch(B, h, w, k, r):
    """enarf_mask_dilate_topk (separable window maximum + radix select) against the reference's formulation,
    F.max_pool2d(2r + 1, stride 1, padding r) + noise -> torch.topk (ray_sampler.py:23-30), with the same noise: the SAME
    set of pixels per image (the order of topk(sorted=False) is unspecified)."""
    import torch.nn.functional as F
    from enarf_gan_amd import ops
    from enarf_gan_amd.libraries.NeRF.ray_sampler import mask_based_sampler
    g = torch.Generator().manual_seed(h * w + k)
    mask = torch.zeros(B, h, w)
    for b in range(B):
        y0, x0 = int(torch.randint(0, h - 8, (1,), generator=g)), int(torch.randint(0, w - 8, (1,), generator=g))
        mask[b, y0:y0 + 8 + b, x0:x0 + 5] = 1.0
        mask[b, (y0 * 7) % h, (x0 * 3) % w] = 0.5            # a non-binary value
    noise = torch.rand(B, h * w, generator=g)
    ours = ops.mask_dilate_topk(mask.cuda(), noise.cuda(), k, r).cpu()
    dil = F.max_pool2d(mask.cuda()[:, None], 2 * r + 1, stride=1, padding=r)[:, 0].reshape(B, h * w).cpu()
    ref = torch.topk(dil + noise, k, dim=1, sorted=False)[1]
    for b in range(B):
        assert sorted(ours[b].tolist()) == sorted(ref[b].tolist()), f"image {b}"
    # ties: constant noise - the selection falls back to ascending pixel order inside the dilated region
    flat = torch.zeros(B, h * w)
    t = ops.mask_dilate_topk(mask.cuda(), flat.cuda(), k, r).cpu()
    for b in range(B):
        assert len(set(t[b].tolist())) == k
        thr = torch.topk(dil[b], k)[0][-1]
        assert bool((dil[b][t[b]] >= thr).all())
    # the entry point with the reference's signature (its window is fixed: 129 x 129)
    idx, homo = mask_based_sampler(mask.cuda(), k, noise=noise.cuda())
    if r == 64:
        assert sorted(idx[0].cpu().tolist()) == sorted(ref[0].tolist())
    assert idx.shape == (B, k) and idx.dtype == torch.int64 and homo.shape == (B, 1, 3, k)
    assert torch.equal(homo[0, 0, 0].cpu(), (idx[0].cpu() % w).float() + 0.5)
    assert torch.equal(homo[0, 0, 1].cpu(), torch.div(idx[0].cpu(), w, rounding_mode="floor").float() + 0.5)


def test_gan_generator_forward_matches_oracle():
    """TriNARFGenerator.forward (models/generator.py:56-118), batch 2 on a black background: image, mask, the side outputs
    fine_weights / fine_depth and the disparity variant against the oracle with the samples the call drew."""
    from enarf_gan_amd.models.generator import TriNARFGenerator
    S, B, Nc, Nf = 32, 2, 24, 32
    sc = Scene(S, B, "center_fixed", 256)
    gen = TriNARFGenerator(Cfg(z_dim=256, background_ratio=0.7, crop_background=True, pretrained_background=False,
                               nerf_params=_nerf_cfg(Nc=Nc, Nf=Nf, constant_triplane=False)), S, 24, sc.raw["parents"], 23,
                           black_background=True)
    gen.register_canonical_pose(sc.raw["canonical_pose"])
    gen.nerf.load_state_dict({f"mlp.{k}": v for k, v in sc.raw["mlp"].items()}, strict=False)
    gen = gen.cuda().eval()
    tri = sc.raw["tri_plane"].cuda()
    seen = {}

    def producer(z, enc, truncation_psi=1):       # stands in for the out-of-scope StyleGAN2-ADA synthesis network
        seen["z"], seen["enc"] = z, enc
        return tri
    gen.nerf.tri_plane_gen = producer
    s = sc.raw
    z = torch.cat([torch.randn(B, 512, generator=torch.Generator().manual_seed(0)), s["z_rend"]], dim=1).cuda()
    with torch.no_grad():
        img, mask, fw, fd = gen(s["pose_to_camera"].cuda(), None, s["bone_length"].cuda(), z, s["inv_intrinsics"].cuda())
    assert img.shape == (B, 3, S, S) and mask.shape == (B, S, S) and fw.shape == (B, 1, S * S, Nf - 1) and fd.shape == (B, 1, S * S, Nf)
    assert torch.equal(seen["z"], z[:, :512]) and seen["enc"].shape == (B, 23 * 8)          # z_nerf and the encoded bone lengths
    bins = gen.nerf.buffers_tensors["bins"].cpu()
    rc, rm, rd, taps = sc.oracle_render(s["image_coord"], Nc, Nf, bins)
    assert float(rm.max()) > 0.3
    assert_close(mask.cpu().reshape(B, -1), rm, "GAN forward: mask")
    assert_close(img.cpu().reshape(B, 3, -1), rc - (1 - rm[:, None]), "GAN forward: image on black (-1)")
    assert_close(fw.cpu()[:, 0], taps["fine_weights"], "GAN forward: fine_weights")
    assert_close(fd.cpu()[:, 0], taps["fine_depth"], "GAN forward: fine_depth", 1e-6)
    with torch.no_grad():
        img2, mask2, disp = gen(s["pose_to_camera"].cuda(), None, s["bone_length"].cuda(), z, s["inv_intrinsics"].cuda(),
                                return_disparity=True)
    rc2, rm2, rd2 = sc.oracle_render(s["image_coord"], Nc, Nf, gen.nerf.buffers_tensors["bins"].cpu(), taps=False)
    assert_close(disp.cpu(), rd2 * 3.0, "GAN forward: disparity x coordinate_scale")
    with torch.no_grad():
        fgc, fgm, bg = gen(s["pose_to_camera"].cuda(), None, s["bone_length"].cuda(), z, s["inv_intrinsics"].cuda(), return_bg=True)
    assert bg == -1 and fgc.shape == (B, 3, S, S)
    # render_mesh's density sweep through the generator (models/generator.py:120-129, base.py:65-77), one sample
    from enarf_gan_amd.libraries.NARF.mesh_rendering import density_volume
    one = lambda x: x[:1].cuda()
    gen.nerf.tri_plane_gen = lambda z_, enc, truncation_psi=1: tri[:1]
    vol = gen.density_volume(one(s["pose_to_camera"]), z[:1], one(s["bone_length"]), voxel_size=0.125)
    assert vol.shape == (17, 17, 17) and float(vol.max()) > 0
    center = one(s["pose_to_camera"])[:, 0, :3, 3:]
    mi = {"z": None, "z_rend": z[:1, 512:], "bone_length": sc.bl_parts[:1].cuda(), "truncation_psi": 0.4,
          "tri_plane_feature": tri[:1]}
    assert torch.equal(vol, density_volume(gen.nerf, sc.pose_parts[:1].cuda(), center, 0.125, mi))
    with pytest.raises(ImportError):
        gen.render_mesh(one(s["pose_to_camera"]), torch.eye(3).cuda(), z[:1], one(s["bone_length"]), voxel_size=0.125)
    with pytest.raises(AssertionError):          # one sample at a time (base.py:67-68)
        gen.density_volume(s["pose_to_camera"].cuda(), z, s["bone_length"].cuda(), voxel_size=0.125)


def test_sampling_api_mirror_matches_reference_golden():
    """libraries/triplane/sampling.py mirror (sample_feature: sum / prod / clamp_mask / batch_idx, sample_triplane_part_prob:
    prod / sum-softmax / uniform, sample_weighted_feature_v2 with gradients) against values and autograd gradients recorded
    from the reference's own functions (tests/golden/sampling_api.npz)."""
    import torch.nn.functional as F
    from enarf_gan_amd.libraries.triplane.sampling import sample_feature, sample_triplane_part_prob, sample_weighted_feature_v2
    g = load_golden("sampling_api")
    t = lambda k: torch.from_numpy(g[k]).cuda()
    assert_close(sample_feature(t("planes"), t("pos")).cpu(), g["out_sum"], "sample_feature sum, B = 2", 1e-5)
    valid = t("valid")
    assert_close(sample_triplane_part_prob(t("wplanes"), t("masked"), valid).cpu(), g["prob_prod"], "part prob prod", 1e-5)
    assert_close(sample_triplane_part_prob(t("wplanes"), t("masked"), valid, clamp_mask=True).cpu(), g["prob_clamp"], "prob clamp_mask", 1e-5)
    assert not np.allclose(g["prob_clamp"], g["prob_prod"])
    assert_close(sample_triplane_part_prob(t("wplanes"), t("masked"), valid, mode="sum").cpu(), g["prob_sum"], "prob sum + softmax", 1e-5)
    assert torch.equal(sample_triplane_part_prob(t("wplanes"), t("masked"), valid, mode="none").cpu(), torch.from_numpy(g["prob_uni"]))
    # batch_idx: the reference's side-by-side plane as input
    feat = t("feat")
    B, h = feat.shape[0], feat.shape[2]
    padded = F.pad(feat, (0, 1)).permute(1, 2, 0, 3).reshape(1, 96, h, (h + 1) * B)
    out = sample_feature(padded, t("pos1"), batch_idx=torch.from_numpy(g["bidx"]).cuda())
    assert_close(out.cpu(), g["out_bidx"], "sample_feature with batch_idx", 1e-5)
    # weighted feature, forward and the two gradients
    f, w = feat.clone().requires_grad_(True), t("weight").requires_grad_(True)
    wf = sample_weighted_feature_v2(32, f, t("masked"), w, valid)
    assert_close(wf.detach().cpu(), g["wf"], "sample_weighted_feature_v2", 1e-5)
    wf.backward(t("cot"))
    assert_close(f.grad.cpu(), g["g_feat"], "d / d tri-plane features", 1e-5)
    assert_close(w.grad.cpu(), g["g_weight"], "d / d weight", 1e-5)
    # dtypes: computed in fp32, returned in the input's dtype (TriplaneSampler.cpp:20)
    for dt in (torch.float16, torch.float64):
        o = sample_feature(t("planes")[:1].to(dt), t("pos")[:1].to(dt))
        assert o.dtype == dt and rel_err(o.float().cpu(), g["out_sum"][:1]).max() < (5e-3 if dt == torch.float16 else 1e-5)   # half: in/out rounding only
    assert sample_feature(t("wplanes").half(), t("masked").reshape(6, 3, -1).half(), reduction="prod").dtype == torch.float16


def test_sampler_autograd_function_gives_true_gradients():
    from enarf_gan_amd.cuda_extension.triplane_sampler import triplane_sampler
    g = torch.Generator().manual_seed(2)
    inp = torch.randn(1, 3 * 32, 24, 24, generator=g)
    grid = torch.rand(1, 50, 1, 3, generator=g) * 2 - 1
    go = torch.randn(1, 32, 50, 1, generator=g)
    a, b = inp.cuda().requires_grad_(True), grid.cuda().requires_grad_(True)
    out = triplane_sampler(a, b)
    out.backward(go.cuda())
    gi, gg = O.triplane_sampler_backward(go, inp, grid)
    assert_close(out.detach().cpu(), O.triplane_sampler_forward(inp, grid), "fwd", 1e-5)
    assert_close(a.grad.cpu(), gi, "grad_input", 1e-5)
    assert_close(b.grad.cpu(), gg, "grad_grid", 1e-5)
    # only the input needs grad
    a2 = inp.cuda().requires_grad_(True)
    triplane_sampler(a2, grid.cuda()).backward(go.cuda())
    assert_close(a2.grad.cpu(), gi, "grad_input only", 1e-5)


def test_density_volume_matches_oracle_grid_sweep():
    """create_mesh's density sweep (mesh_rendering.py:50-73): the grid generated chunk-wise on the device equals the
    reference's host-side meshgrid, and the densities equal the oracle's query on it."""
    from enarf_gan_amd.libraries.NARF.mesh_rendering import density_volume, create_mesh
    sc = Scene(32, 1, "center_fixed", 20)
    m = _model(sc)
    s = sc.raw
    voxel = 0.125                                   # 17^3 grid
    center = torch.tensor([0.02, -0.03, 1.0]).reshape(1, 3, 1)
    center[0, :, 0] += sc.pose_parts[0, :, :3, 3].mean(0) - torch.tensor([0.0, 0.0, 1.0])
    mi = {"z": None, "z_rend": s["z_rend"].cuda(), "bone_length": sc.bl_parts.cuda(), "truncation_psi": 1}
    pose = sc.pose_parts.cuda()
    keep = pose.clone()
    vol = density_volume(m, pose, center, voxel, mi, chunk=1000)       # several ragged chunks
    assert torch.equal(pose, keep), "the caller's pose must not be scaled in place"
    cube = int(1 / voxel)
    bins = torch.arange(-cube, cube + 1) / cube
    p = (torch.stack(torch.meshgrid(bins, bins, bins, indexing="ij")).reshape(1, 3, -1) + center) * 3.0
    den, _, valid = O.query(p, sc.pose_scaled, sc.scale, sc.cpose, s["tri_plane"], sc.weights())
    assert int(valid.any(dim=1).sum()) > 50
    assert_close(vol.cpu().reshape(-1), den.reshape(-1), "density volume")
    lattice = density_volume(m, pose, center, voxel, mi)                # one launch, points generated in the kernel
    assert torch.equal(lattice, vol), "lattice mode must place the points exactly where the tensor path does"
    with pytest.raises(ImportError):
        create_mesh(m, pose, center, voxel, 15, mi)


def test_forward_return_intermediate_gives_fine_points_and_density():
    """NARFBase.forward(return_intermediate=True) (libraries/NeRF/base.py:112-114, rendering.py:291): fine points and
    fine densities of every ray, against the oracle's taps (rays the reference drops hold zero density)."""
    sc = Scene(32, 1, "center_fixed", 20)
    m = _model(sc, Nc=24, Nf=32)
    s = sc.raw
    coord = s["image_coord"][..., 32 * 13:32 * 13 + 64].contiguous()
    g = torch.Generator().manual_seed(4)
    bins = torch.rand(1, 64, 32, generator=g).sort(-1).values
    with torch.no_grad():
        color, mask, (pts, den) = m(1, coord.cuda(), s["pose_to_camera"].cuda(), s["inv_intrinsics"].cuda(), None,
                                    s["z_rend"].cuda(), s["bone_length"].cuda(), Nc=24, Nf=32, return_intermediate=True,
                                    bins=bins.cuda())
    assert pts.shape == (1, 3, 64 * 32) and den.shape == (1, 1, 64 * 32)
    rc, rm, rd, taps = sc.oracle_render(coord, 24, 32, bins)
    live = taps["ray_validity"][0]
    assert_close(color.cpu(), rc, "colour")
    fd = den.cpu().reshape(1, 64, 32)
    assert_close(fd[0][live][:, :-1], taps["fine_density"][0][live][:, :-1], "fine density (the last sample is never queried)")
    assert float(fd[0][~live].abs().max()) == 0.0
    rd_ = taps["ray_dir"]
    start, end = taps["depth_min"][:, None] * rd_, taps["depth_max"][:, None] * rd_
    _, opts = O.fine_points(bins, taps["depth_min"], taps["depth_max"], start, end)
    assert_close(pts.cpu().reshape(1, 3, 64, 32)[0][:, live], opts[0][:, live], "fine points", 1e-5)
    with pytest.raises(NotImplementedError):
        m.train()
        m(1, coord.cuda(), s["pose_to_camera"].cuda(), s["inv_intrinsics"].cuda(), None, s["z_rend"].cuda(),
          s["bone_length"].cuda(), Nc=24, Nf=32, return_intermediate=True)


def test_sampler_image_ids_outside_the_batch_sample_nothing():
    """enarf_triplane_sample_ex_fwd / _bwd with a per-point image index (sample_feature's batch_idx, sampling.py:34-38): an
    id outside [0, n_images) - negative included - samples zeros and receives / sends no gradient, as a position in the
    reference's zero padding would; it must not become an out-of-bounds read or atomic. In-range points are
    unchanged by the presence of the others."""
    from enarf_gan_amd import ops
    g = torch.Generator().manual_seed(9)
    inp = torch.randn(3, 3 * 8, 16, 20, generator=g).cuda()
    n = 500
    grid = (torch.rand(1, n, 3, generator=g) * 2 - 1).cuda()
    ids = torch.randint(0, 3, (n,), generator=g, dtype=torch.int32)
    bad = ids.clone()
    bad[::7] = 3
    bad[3::11] = -1
    bad[5::13] = 1 << 30
    off = ((bad < 0) | (bad >= 3)).cuda()
    ref = ops.triplane_sample_ex_fwd(inp, grid, point_image=ids.cuda())
    out = ops.triplane_sample_ex_fwd(inp, grid, point_image=bad.cuda())
    assert float(out[..., off].abs().max()) == 0.0 and torch.equal(out[..., ~off], ref[..., ~off])
    sep = ops.triplane_sample_ex_fwd(inp, grid, separate=True, point_image=bad.cuda())
    assert float(sep[..., off].abs().max()) == 0.0
    go = torch.randn(1, 8, n, generator=g).cuda()
    gi, gg = ops.triplane_sample_ex_bwd(go, inp, grid, False, bad.cuda(), True, True)
    gi_ref, gg_ref = ops.triplane_sample_ex_bwd(go * (~off).float(), inp, grid, False, ids.cuda(), True, True)
    assert float(gg[0, off].abs().max()) == 0.0 and torch.equal(gg[0, ~off], gg_ref[0, ~off])
    assert_close(gi.cpu(), gi_ref.cpu(), "grad_input with out-of-batch ids", 1e-6)
