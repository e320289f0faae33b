#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python code on the CPU.

Runs only in the build container (needs /root/reference; the GPU box has neither the reference nor
this need - it reads the committed .npz files). Nothing from the reference is copied: the script
puts /root/reference on sys.path, imports its modules unmodified and records their outputs on the
seeded synthetic inputs of `enarf_gan_amd.synth`.

Harness-side accommodations, none of which touch arithmetic on the path:
  * modules that are not installed here and are never executed on this path (kornia, pytorch3d,
    dnnlib, the two un-vendored StyleGAN submodules, the compiled `triplane_sampler_cuda`) are
    pre-seeded in sys.modules as empty placeholders whose attributes raise if ever called;
  * the reference hard-codes device="cuda" in a few tensor factories (rendering.py:41,125,194;
    ray_sampler.py:37,56-57,65): torch.linspace / arange / ones are wrapped to drop that kwarg and
    torch.cuda.FloatTensor is aliased to torch.FloatTensor so the same code runs on the CPU;
  * torch.sort is wrapped to record the sorted importance-sampling `bins` (rendering.py:197), the
    only value needed to replay a render deterministically that the reference does not expose.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402


class _Absent:
    """Placeholder for a symbol of an uninstalled, never-executed dependency."""

    def __init__(self, name):
        self._name = name

    def __call__(self, *a, **k):
        raise RuntimeError(f"{self._name} is not available in this container and must not be reached")

    def __getattr__(self, item):
        return _Absent(f"{self._name}.{item}")

    def __mro_entries__(self, bases):      # allows `class X(Absent)` in never-instantiated code
        return (object,)


def _placeholder(name, attrs=()):
    m = types.ModuleType(name)
    m.__path__ = []
    for a in attrs:
        setattr(m, a, _Absent(f"{name}.{a}"))
    m.__getattr__ = lambda item, _n=name: _Absent(f"{_n}.{item}")
    sys.modules[name] = m
    return m


def install_placeholders():
    _placeholder("kornia")
    _placeholder("kornia.augmentation", ["RandomCrop"])
    _placeholder("libraries.stylegan2_pytorch")
    _placeholder("libraries.stylegan2_pytorch.op", ["FusedLeakyReLU", "fused_leaky_relu"])
    _placeholder("libraries.stylegan2_pytorch.model",
                 ["PixelNorm", "Upsample", "Blur", "ModulatedConv2d", "Generator"])
    _placeholder("pytorch3d")
    _placeholder("pytorch3d.renderer", ["FoVPerspectiveCameras", "PointLights", "RasterizationSettings",
                                        "MeshRenderer", "MeshRasterizer", "HardPhongShader", "Textures",
                                        "PerspectiveCameras", "look_at_view_transform"])
    _placeholder("pytorch3d.structures", ["Meshes"])
    _placeholder("dnnlib")
    _placeholder("triplane_sampler_cuda", ["triplane_sampler_forward", "triplane_sampler_backward"])


_SORT_LOG = []
# deterministic importance sampling for the full-frame cases (run_fullframe_case): while "on", torch.multinomial returns
# the stratified inverse-CDF bins of the weights it is given and the uniform jitter is 0.5 everywhere, so that the bins
# the reference marches are a function of small integers that a fixture can hold for EVERY ray of a frame
_DET = {"on": False, "idx": None}


class _HalfJitter:
    """stands in for torch.cuda.FloatTensor(*shape) while _DET is on: .uniform_() gives 0.5"""

    def __init__(self, *shape):
        self.shape = shape

    def uniform_(self):
        return torch.full(self.shape, 0.5)


def _float_tensor(*shape):
    return _HalfJitter(*shape) if _DET["on"] else torch.FloatTensor(*shape)


def redirect_cuda_factories():
    def strip(fn):
        def wrapped(*a, **k):
            if k.get("device") in ("cuda", torch.device("cuda")):
                k.pop("device")
            return fn(*a, **k)
        return wrapped
    torch.linspace = strip(torch.linspace)
    torch.arange = strip(torch.arange)
    torch.ones = strip(torch.ones)
    torch.cuda.FloatTensor = _float_tensor
    _multinomial = torch.multinomial

    def multinomial(w, num_samples, replacement=False, generator=None):
        if not _DET["on"]:
            return _multinomial(w, num_samples, replacement, generator=generator)
        cdf = torch.cumsum(w.double(), dim=-1)
        cdf = cdf / cdf[..., -1:]
        u = ((torch.arange(num_samples, dtype=torch.float64) + 0.5) / num_samples)[None].expand(w.shape[0], -1).contiguous()
        idx = torch.searchsorted(cdf, u, right=True).clamp(max=w.shape[-1] - 1)
        _DET["idx"] = idx.clone()
        return idx
    torch.multinomial = multinomial
    _sort = torch.sort

    def sort(*a, **k):
        out = _sort(*a, **k)
        _SORT_LOG.append(out[0].detach().clone())
        return out
    torch.sort = sort


class Cfg(dict):
    """attr-dict standing in for easydict.EasyDict (not installed)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def nerf_cfg(origin_location, Nc, Nf, **over):
    c = Cfg(hidden_size=32, Nc=Nc, Nf=Nf, origin_location=origin_location, coordinate_scale=3,
               render_bs=16384, no_ray_direction=True, multiply_density_with_triplane_wieght=False,
               clamp_mask=False, constant_triplane=True, constant_trimask=False,
               constant_trimask_lr_mul=1, deformation_field=False, selector_mlp=False,
               no_selector=False, time_conditional=True, pose_conditional=False, mask_input=False)
    c.update(over)
    return c


def build_reference_model(scene, Nc, Nf, style_dim, **cfg_over):
    from models.narf import TriPlaneNARF
    cfg = nerf_cfg(scene["origin_location"], Nc, Nf, **cfg_over)
    model = TriPlaneNARF(cfg, z_dim=style_dim, num_bone=24, bone_length=True,
                         parent=scene["parents"], num_bone_param=23, view_dependent=False)
    model.register_canonical_pose(scene["canonical_pose"])
    sd = {f"mlp.{k}": v for k, v in scene["mlp"].items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert set(missing) <= {"tri_plane", "canonical_joints", "canonical_parent_joints",
                            "canonical_bone_length", "canonical_pose"}, missing
    tri = scene["tri_plane"]
    # per-image tri-planes (GAN style): replace the producer, which is upstream of the path
    model.tri_plane_gen = lambda z, *a, **k: tri
    model.eval()
    return model


def bitmask(valid):
    """(B,P,...) bool -> (B,...) uint32 with bit k = part k."""
    v = valid.cpu().numpy().astype(np.uint32)
    P = v.shape[1]
    sh = (np.arange(P, dtype=np.uint32)).reshape((1, P) + (1,) * (v.ndim - 2))
    return (v << sh).sum(axis=1).astype(np.uint32)


def run_render_case(name, size, batch, Nc, Nf, origin_location, style_dim, n_keep, seed):
    from enarf_gan_amd import synth
    scene = synth.make_scene(size, batch, origin_location, style_dim)
    model = build_reference_model(scene, Nc, Nf, style_dim)
    B, n = batch, size * size
    _SORT_LOG.clear()
    torch.manual_seed(seed)
    with torch.no_grad():
        color, mask, disp = model.forward(B, scene["image_coord"], scene["pose_to_camera"],
                                          scene["inv_intrinsics"], None, scene["z_rend"],
                                          scene["bone_length"], Nc=Nc, Nf=Nf, return_disparity=True)
    ts = model.temporal_state
    # frustum taps via the reference's own function
    from libraries.NeRF.rendering import decide_frustrum_range
    from libraries.NARF.pose_utils import transform_pose
    pose_p, bl_p = transform_pose(scene["pose_to_camera"], scene["bone_length"], origin_location,
                                  scene["parents"])
    pose_s = pose_p.clone()
    pose_s[:, :, :3, 3] *= 3
    dmin, dmax, rdir, rval = decide_frustrum_range(scene["image_coord"], pose_s, scene["inv_intrinsics"],
                                                   0.3, 5, return_camera_coord=True)
    rval = rval.reshape(B, n)
    bins_c = _SORT_LOG[-1]                                     # (B,1,n',Nf)
    assert bins_c.shape[-1] == Nf

    def scatter(x_c, fill=0.0):
        """(B, n', ...) compacted (only when B == 1) -> (B, n, ...)."""
        if B != 1:
            return x_c
        full = torch.full((1, n) + tuple(x_c.shape[2:]), fill, dtype=x_c.dtype)
        full[0, rval[0]] = x_c[0]
        return full

    npr = bins_c.shape[2]
    bins = scatter(bins_c.reshape(B, npr, Nf), 0.5)
    cden = scatter(ts["coarse_density"].reshape(B, npr, Nc))
    fden = scatter(model.buffers_tensors["fine_density"].reshape(B, npr, Nf))
    fdepth = scatter(ts["fine_depth"].reshape(B, npr, Nf))
    fweights = scatter(model.buffers_tensors["fine_weights"].reshape(B, npr, Nf - 1))
    # validity bit masks of the fine samples, from the reference's own transform
    with torch.no_grad():
        fp = ts["fine_points"]                                 # (B,3,n'*Nf)
        local, canon = model.to_local_and_canonical(fp, pose_s, bl_p)
        from libraries.NeRF.utils import in_cube
        v = in_cube(local) * (canon.abs() < 1).all(dim=2)      # (B,P,n'*Nf)
    fvalid = scatter(torch.from_numpy(bitmask(v).astype(np.int64)).reshape(B, npr, Nf))

    # keep a deterministic subset of rays: mostly rays that hit, some that do not
    rs = np.random.RandomState(seed)
    keep = []
    for b in range(B):
        hit = np.where(rval[b].numpy())[0]
        miss = np.where(~rval[b].numpy())[0]
        nh = min(len(hit), int(n_keep * 0.85))
        nm = min(len(miss), n_keep - nh)
        sel = np.concatenate([rs.choice(hit, nh, replace=False), rs.choice(miss, nm, replace=False)])
        keep.append(np.sort(sel))
    m = min(len(k) for k in keep)
    keep = np.stack([k[:m] for k in keep])                     # (B,m)
    bi = np.arange(B)[:, None]

    out = dict(
        size=size, batch=batch, Nc=Nc, Nf=Nf, style_dim=style_dim, seed=seed,
        origin_location=origin_location, near=float(ts["near_plane"]), far=float(ts["far_plane"]),
        n_valid_rays=rval.sum(dim=1).numpy(),
        ray_idx=keep.astype(np.int32),
        ray_validity=rval.numpy()[bi, keep],
        depth_min=dmin.reshape(B, n).numpy()[bi, keep], depth_max=dmax.reshape(B, n).numpy()[bi, keep],
        bins=bins.numpy()[bi, keep], coarse_density=cden.numpy()[bi, keep],
        fine_density=fden.numpy()[bi, keep], fine_depth=fdepth.numpy()[bi, keep],
        fine_weights=fweights.numpy()[bi, keep], fine_valid=fvalid.numpy().astype(np.uint32)[bi, keep],
        color=color.numpy().transpose(0, 2, 1)[bi, keep].transpose(0, 2, 1),
        mask=mask.numpy()[bi, keep], disparity=disp.numpy()[bi, keep],
        full_mask_u8_sum=int((mask.numpy() * 255).astype(np.uint8).astype(np.int64).sum()),
    )
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: valid rays {rval.sum().item()}/{B * n}, kept {m}/image, "
          f"mask mean {mask.mean().item():.4f} -> {os.path.getsize(path) / 1024:.0f} KiB")


def run_fullframe_case(name, size, batch, Nc, Nf, origin_location, style_dim):
    """EVERY ray of a frame, for bit-exact checks of the integer outputs (the uint8 foreground mask of
    ENARF_GAN_demo.py:79, the ray-validity count): the reference renders with the deterministic sampler above and the
    fixture keeps, per marched ray, the Nf coarse-bin indices the sampler chose (uint8), from which a test rebuilds the
    reference's bins exactly: bins = idx / Nc + 0.5 / Nc in float32 (rendering.py:192-194)."""
    from enarf_gan_amd import synth
    scene = synth.make_scene(size, batch, origin_location, style_dim)
    model = build_reference_model(scene, Nc, Nf, style_dim)
    B, n = batch, size * size
    _SORT_LOG.clear()
    _DET["on"], _DET["idx"] = True, None
    try:
        with torch.no_grad():
            color, mask, disp = model.forward(B, scene["image_coord"], scene["pose_to_camera"], scene["inv_intrinsics"], None,
                                              scene["z_rend"], scene["bone_length"], Nc=Nc, Nf=Nf, return_disparity=True)
    finally:
        _DET["on"] = False
    from libraries.NeRF.rendering import decide_frustrum_range
    from libraries.NARF.pose_utils import transform_pose
    pose_p, _ = transform_pose(scene["pose_to_camera"], scene["bone_length"], origin_location, scene["parents"])
    pose_s = pose_p.clone()
    pose_s[:, :, :3, 3] *= 3
    _, _, _, rval = decide_frustrum_range(scene["image_coord"], pose_s, scene["inv_intrinsics"], 0.3, 5,
                                          return_camera_coord=True)
    rval = rval.reshape(B, n)
    idx = _DET["idx"]                                  # (rows, Nf): rows = marched rays (valid ones when B == 1)
    rows = int(rval.sum()) if B == 1 else B * n
    assert idx.shape == (rows, Nf) and int(idx.max()) < Nc <= 255
    bins_ref = _SORT_LOG[-1].reshape(rows, Nf)
    rebuilt = idx.float() / Nc + torch.full((rows, Nf), 0.5) / Nc
    assert torch.equal(bins_ref, rebuilt), "the fixture's bin indices must reproduce the reference's bins exactly"
    out = dict(size=size, batch=batch, Nc=Nc, Nf=Nf, style_dim=style_dim, origin_location=origin_location,
               ray_validity=np.packbits(rval.numpy(), axis=1), n_valid_rays=rval.sum(dim=1).numpy(),
               bin_idx=idx.numpy().astype(np.uint8),
               mask=mask.numpy(), mask_u8=(mask.numpy() * 255).astype(np.uint8),
               color_u8=np.clip(color.numpy() * 127.5 + 127.5, 0, 255).astype(np.uint8),
               disparity_sum=float(disp.double().sum()), color_abs_sum=float(color.double().abs().sum()))
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: valid rays {rval.sum().item()}/{B * n}, mask_u8 sum {int(out['mask_u8'].astype(np.int64).sum())} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def run_encoding_case(name, seed):
    """The host-side encodings of libraries/NeRF/utils.py:46-88 (the bone-length conditioning of the tri-plane producers,
    models/narf.py:286-288) and to_local / in_cube (:13-43) on small random inputs."""
    from libraries.NeRF.utils import in_cube, multi_part_positional_encoding, positional_encoding, to_local
    from enarf_gan_amd import synth
    g = torch.Generator().manual_seed(seed)
    scene = synth.make_scene(32, 3, "center+head", 20)
    from libraries.NARF.pose_utils import transform_pose
    pose_p, bl_p = transform_pose(scene["pose_to_camera"], scene["bone_length"], "center+head", scene["parents"])
    bl = bl_p.clone()
    bl[0, 3, 0] = 1.25                                          # a bone longer than 1: its channels are masked to zero
    x = torch.randn(2, 5, 7, generator=g)
    val = torch.rand(2, 24 * 3, 11, generator=g) * 2.4 - 1.2     # some parts leave [-1, 1]
    pts = torch.randn(3, 3, 50, generator=g) * 2 + torch.tensor([0.0, 0.0, 3.0])[None, :, None]
    out = dict(bone_length=bl.numpy(), enc_length=multi_part_positional_encoding(bl, 4, 24)[:, :, 0].numpy(),
               x=x.numpy(), pe_cos_first=positional_encoding(x, 6).numpy(),
               pe_sin_first_cat1=positional_encoding(x, 3, cos_first=False, cat_dim=1).numpy(),
               val=val.numpy(), mpe=multi_part_positional_encoding(val, 2, 24).numpy(),
               pts=pts.numpy(), local=to_local(pts, pose_p).numpy(), inside=in_cube(to_local(pts, pose_p)).numpy(),
               inside3=in_cube(pts[:, :, :] / 4).numpy())
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: masked channels {int((out['enc_length'] == 0).sum())} -> {os.path.getsize(path) / 1024:.0f} KiB")


def run_sampling_case(name, seed):
    """libraries/triplane/sampling.py:9-127 stand-alone: sample_feature (sum, prod with and without clamp_mask, the
    batch_idx side-by-side form), sample_triplane_part_prob (prod / sum / uniform) and sample_weighted_feature_v2, with the
    reference's own autograd gradients for the latter."""
    import torch.nn.functional as F
    from libraries.triplane.sampling import sample_feature, sample_triplane_part_prob, sample_weighted_feature_v2
    g = torch.Generator().manual_seed(seed)
    B, P, h, n = 2, 3, 16, 150
    planes = torch.randn(B, 3 * 4, h, h, generator=g)
    pos = torch.rand(B, 3, n, generator=g) * 2.4 - 1.2
    out_sum = sample_feature(planes, pos)                                   # B = 2: the F.grid_sample branch
    wplanes = torch.randn(B * P, 3, h, h, generator=g) * 3.0
    ppos = torch.rand(B, P, 3, n, generator=g) * 1.998 - 0.999      # valid pairs are inside the cube (models/narf.py:200-201)
    valid = torch.rand(B, P, n, generator=g) > 0.4
    masked = ppos * valid[:, :, None] + 2 * ~valid[:, :, None]              # models/narf.py:237
    prob_prod = sample_triplane_part_prob(wplanes, masked, valid)
    prob_clamp = sample_triplane_part_prob(wplanes, masked, valid, clamp_mask=True)
    prob_sum = sample_triplane_part_prob(wplanes, masked, valid, mode="sum")
    prob_uni = sample_triplane_part_prob(wplanes, masked, valid, mode="none")
    # batch_idx form on the side-by-side plane (sampling.py:96-97 builds it like this)
    feat = torch.randn(B, 96, h, h, generator=g).requires_grad_(True)
    padded = F.pad(feat, (0, 1)).permute(1, 2, 0, 3).reshape(1, 96, h, (h + 1) * B)
    bidx = torch.randint(0, B, (n,), generator=g)
    pos1 = (torch.rand(1, 3, n, generator=g) * 2.0 - 1.0)
    out_bidx = sample_feature(padded, pos1.clone(), batch_idx=bidx)
    weight = torch.rand(B, P, n, generator=g).requires_grad_(True)
    wf = sample_weighted_feature_v2(32, feat, masked, weight, valid)
    cot = torch.randn(wf.shape, generator=g)
    g_feat, g_weight = torch.autograd.grad(wf, (feat, weight), cot)
    out = dict(planes=planes.numpy(), pos=pos.numpy(), out_sum=out_sum.numpy(), wplanes=wplanes.numpy(), masked=masked.numpy(),
               valid=valid.numpy(), prob_prod=prob_prod.numpy(), prob_clamp=prob_clamp.numpy(), prob_sum=prob_sum.numpy(),
               prob_uni=prob_uni.numpy(), feat=feat.detach().numpy(), bidx=bidx.numpy().astype(np.int32),
               pos1=pos1.numpy(), out_bidx=out_bidx.detach().numpy(), weight=weight.detach().numpy(), wf=wf.detach().numpy(),
               cot=cot.numpy(), g_feat=g_feat.numpy(), g_weight=g_weight.numpy())
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: valid pairs {int(valid.sum())} -> {os.path.getsize(path) / 1024:.0f} KiB")


def run_query_case(name, batch, n_points, origin_location, style_dim, seed, mask_scale=1.0, **cfg_over):
    """calc_density_and_color_from_camera_coord_v2 (models/narf.py:176) on a point cloud. cfg_over: nerf_params switches
    (clamp_mask, no_selector, multiply_density_with_triplane_wieght); mask_scale stretches the part-probability planes so
    that clamp_mask's [-2, 5] range is exercised."""
    from enarf_gan_amd import synth
    from libraries.NARF.pose_utils import transform_pose
    from libraries.NeRF.utils import in_cube
    scene = synth.make_scene(64, batch, origin_location, style_dim)
    if mask_scale != 1.0:
        scene["tri_plane"] = scene["tri_plane"].clone()
        scene["tri_plane"][:, 96:] *= mask_scale
    model = build_reference_model(scene, 48, 64, style_dim, **cfg_over)
    pose_p, bl_p = transform_pose(scene["pose_to_camera"], scene["bone_length"], origin_location,
                                  scene["parents"])
    pose_s = pose_p.clone()
    pose_s[:, :, :3, 3] *= 3
    g = torch.Generator().manual_seed(seed)
    # points scattered around randomly chosen part centres (scaled space), so many are valid
    P = pose_s.shape[1]
    k = torch.randint(0, P, (batch, n_points), generator=g)
    centre = torch.gather(pose_s[:, :, :3, 3], 1, k[..., None].expand(-1, -1, 3))    # (B,N,3)
    pts = (centre + torch.randn(batch, n_points, 3, generator=g) * 0.7).permute(0, 2, 1).contiguous()
    model_input = {"z": None, "z_rend": scene["z_rend"], "bone_length": bl_p, "truncation_psi": 1}
    model.buffers_tensors = {}          # render() creates this attribute lazily (rendering.py:264-265)
    with torch.no_grad():
        den, col = model.calc_density_and_color_from_camera_coord_v2(pts, pose_s, None, model_input)
        local, canon = model.to_local_and_canonical(pts, pose_s, bl_p)
        v = in_cube(local) * (canon.abs() < 1).all(dim=2)
    w = model.temporal_state["weight"]
    out = dict(batch=batch, n_points=n_points, origin_location=origin_location, style_dim=style_dim,
               points=pts.numpy(), density=den.numpy(), color=col.numpy(), valid=bitmask(v),
               weight=w.numpy(), canonical=canon.numpy()[:, :, :, :256], mask_scale=mask_scale,
               clamp_mask=bool(cfg_over.get("clamp_mask", False)), no_selector=bool(cfg_over.get("no_selector", False)),
               mult_w=bool(cfg_over.get("multiply_density_with_triplane_wieght", False)))
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: valid pairs {int(v.sum())}, points with any valid {int(v.any(dim=1).sum())}/{batch * n_points}"
          f" -> {os.path.getsize(path) / 1024:.0f} KiB")


def run_grad_case(name, size, batch, Nc, Nf, origin_location, style_dim, n_rays, seed):
    """Gradients of the reference's own autograd through render (incl. MyReLU's custom backward) w.r.t. the tri-plane,
    the StyledMLP parameters and z_rend, for loss = sum(g_color * color) + sum(g_mask * mask) + sum(g_disp * disparity)."""
    import torch.nn.functional as F
    from enarf_gan_amd import synth
    scene = synth.make_scene(size, batch, origin_location, style_dim)
    model = build_reference_model(scene, Nc, Nf, style_dim)
    tri = scene["tri_plane"].clone().requires_grad_(True)
    model.tri_plane_gen = lambda z, *a, **k: tri
    z = scene["z_rend"].clone().requires_grad_(True)
    n = size * size
    start = n // 2 - n_rays // 2
    coord = scene["image_coord"][..., start:start + n_rays].contiguous()
    _SORT_LOG.clear()
    torch.manual_seed(seed)
    color, mask, disp = model.forward(batch, coord, scene["pose_to_camera"], scene["inv_intrinsics"], None, z,
                                      scene["bone_length"], Nc=Nc, Nf=Nf, return_disparity=True)
    from libraries.NeRF.rendering import decide_frustrum_range
    from libraries.NARF.pose_utils import transform_pose
    pose_p, _ = transform_pose(scene["pose_to_camera"], scene["bone_length"], origin_location, scene["parents"])
    pose_s = pose_p.clone()
    pose_s[:, :, :3, 3] *= 3
    _, _, _, rval = decide_frustrum_range(coord, pose_s, scene["inv_intrinsics"], 0.3, 5, return_camera_coord=True)
    rval = rval.reshape(batch, n_rays)
    bins_c = _SORT_LOG[-1]
    if batch == 1:
        bins = torch.full((1, n_rays, Nf), 0.5)
        bins[0, rval[0]] = bins_c.reshape(1, -1, Nf)[0]
    else:
        bins = bins_c.reshape(batch, n_rays, Nf)
    g = torch.Generator().manual_seed(seed + 1)
    gc, gm, gd = torch.randn(batch, 3, n_rays, generator=g), torch.randn(batch, n_rays, generator=g), torch.randn(batch, n_rays, generator=g)
    loss = (color * gc).sum() + (mask * gm).sum() + (disp * gd).sum()
    params = {k: v for k, v in model.mlp.named_parameters() if "noise" not in k}
    keys = sorted(params)
    grads = torch.autograd.grad(loss, [tri, z] + [params[k] for k in keys])
    out = dict(size=size, batch=batch, Nc=Nc, Nf=Nf, style_dim=style_dim, origin_location=origin_location, start=start,
               n_rays=n_rays, bins=bins.numpy(), g_color=gc.numpy(), g_mask=gm.numpy(), g_disp=gd.numpy(),
               color=color.detach().numpy(), mask=mask.detach().numpy(),
               # the tri-plane gradient (43 MB) is stored sum-pooled 16x16 per channel, plus its L1 norm
               grad_tri_pool16=(F.avg_pool2d(grads[0], 16) * 256).numpy(), grad_tri_abs_sum=float(grads[0].abs().sum()),
               grad_z=grads[1].numpy())
    for k, gk in zip(keys, grads[2:]):
        out["grad_" + k] = gk.numpy()
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss {float(loss):.4f}, |grad tri|_1 {out['grad_tri_abs_sum']:.4f} -> {os.path.getsize(path) / 1024:.0f} KiB")


def run_sampler_case(name, seed):
    """sample_feature's general branch = 3 x F.grid_sample + sum (sampling.py:27-44): the arithmetic
    the compiled triplane_sampler implements (kernel.cu:13-92)."""
    from libraries.triplane.sampling import sample_feature
    g = torch.Generator().manual_seed(seed)
    B, C, H, W, n = 2, 4, 16, 24, 300
    inp = torch.randn(B, 3 * C, H, W, generator=g)
    pos = (torch.rand(B, 3, n, generator=g) * 2.4 - 1.2)      # some taps out of bounds
    pos[:, :, :8] = torch.tensor([-1.0, 1.0, 0.0, -1.0 + 1 / W, 1.0 - 1 / H, 0.999, -0.999, 1.2])
    inp_r = inp.clone().requires_grad_(True)
    pos_r = pos.clone().requires_grad_(True)
    out = sample_feature(inp_r, pos_r, reduction="sum")        # (B,C,n)
    go = torch.randn(out.shape, generator=g)
    gi, gp = torch.autograd.grad(out, (inp_r, pos_r), go)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, input=inp.numpy(), position=pos.numpy(), output=out.detach().numpy(),
                        grad_output=go.numpy(), grad_input=gi.numpy(), grad_position=gp.numpy())
    print(f"{name}: -> {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    install_placeholders()
    redirect_cuda_factories()
    torch.set_num_threads(8)
    if "--only-c4" in sys.argv:     # BASELINE config C4's sample counts (Nc 72, Nf 96) at a small frame
        run_render_case("render_c4s_32_b2", size=32, batch=2, Nc=72, Nf=96, origin_location="center_fixed",
                        style_dim=256, n_keep=96, seed=25)
        return
    if "--only-modes" in sys.argv:
        run_query_case("query_b1_clamp_multw", batch=1, n_points=2048, origin_location="center_fixed", style_dim=20, seed=7,
                       mask_scale=3.0, clamp_mask=True, multiply_density_with_triplane_wieght=True)
        run_query_case("query_b1_noselector", batch=1, n_points=2048, origin_location="center_fixed", style_dim=20, seed=8,
                       no_selector=True, multiply_density_with_triplane_wieght=True)
        return
    if "--only-sampling" in sys.argv:
        run_sampling_case("sampling_api", seed=51)
        return
    if "--only-encoding" in sys.argv:
        run_encoding_case("encoding", seed=41)
        return
    if "--only-full" in sys.argv:   # every ray of a frame, deterministic sampler: bit-exact integer outputs
        run_fullframe_case("full_c1_128_b1_p23", size=128, batch=1, Nc=48, Nf=64, origin_location="center_fixed", style_dim=20)
        run_fullframe_case("full_gan_32_b2", size=32, batch=2, Nc=48, Nf=64, origin_location="center_fixed", style_dim=256)
        return
    if "--only-grad" in sys.argv:
        run_grad_case("grad_32_b1", size=32, batch=1, Nc=48, Nf=32, origin_location="center_fixed", style_dim=20,
                      n_rays=72, seed=31)
        run_grad_case("grad_32_b2", size=32, batch=2, Nc=48, Nf=32, origin_location="center_fixed", style_dim=256,
                      n_rays=48, seed=32)
        return
    run_sampler_case("sampler_b2", seed=3)
    run_query_case("query_b2_p23", batch=2, n_points=4096, origin_location="center_fixed", style_dim=256, seed=5)
    run_query_case("query_b1_p24", batch=1, n_points=4096, origin_location="center+head", style_dim=20, seed=6)
    run_render_case("render_c0_64_b1", size=64, batch=1, Nc=48, Nf=32, origin_location="center_fixed",
                    style_dim=20, n_keep=256, seed=21)
    run_render_case("render_c1_128_b1_p23", size=128, batch=1, Nc=48, Nf=64, origin_location="center_fixed",
                    style_dim=20, n_keep=160, seed=22)
    run_render_case("render_c1_128_b1_p24", size=128, batch=1, Nc=48, Nf=64, origin_location="center+head",
                    style_dim=20, n_keep=160, seed=23)
    run_render_case("render_gan_32_b2", size=32, batch=2, Nc=48, Nf=64, origin_location="center_fixed",
                    style_dim=256, n_keep=128, seed=24)
    run_grad_case("grad_32_b1", size=32, batch=1, Nc=48, Nf=32, origin_location="center_fixed", style_dim=20,
                  n_rays=72, seed=31)
    run_grad_case("grad_32_b2", size=32, batch=2, Nc=48, Nf=32, origin_location="center_fixed", style_dim=256,
                  n_rays=48, seed=32)
    run_render_case("render_c4s_32_b2", size=32, batch=2, Nc=72, Nf=96, origin_location="center_fixed",
                    style_dim=256, n_keep=96, seed=25)
    run_fullframe_case("full_c1_128_b1_p23", size=128, batch=1, Nc=48, Nf=64, origin_location="center_fixed", style_dim=20)
    run_fullframe_case("full_gan_32_b2", size=32, batch=2, Nc=48, Nf=64, origin_location="center_fixed", style_dim=256)
    run_encoding_case("encoding", seed=41)
    run_sampling_case("sampling_api", seed=51)
    run_query_case("query_b1_clamp_multw", batch=1, n_points=2048, origin_location="center_fixed", style_dim=20, seed=7,
                   mask_scale=3.0, clamp_mask=True, multiply_density_with_triplane_wieght=True)
    run_query_case("query_b1_noselector", batch=1, n_points=2048, origin_location="center_fixed", style_dim=20, seed=8,
                   no_selector=True, multiply_density_with_triplane_wieght=True)


if __name__ == "__main__":
    main()
