"""CPU-only checks (-m "not gpu"): the C-ABI library loads and exports every symbol include/enarf_hip.h declares,
argument validation works without touching a device, and the Python mirror keeps the reference's names."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "enarf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(enarf_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_all_exported_and_bound():
    from enarf_gan_amd import _lib
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 11
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in enarf_hip.h but not exported by libenarf_hip.so"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    assert set(_lib.SIGNATURES) == set(declared)
    assert lib.enarf_abi_version() == _lib.ABI_VERSION == 4
    assert lib.enarf_mlp_pack_bytes() % 16 == 0


def test_struct_layouts_match_the_header():
    """Compile a tiny C program against the header and compare sizeof / offsetof with the ctypes mirrors."""
    import subprocess
    import tempfile
    from enarf_gan_amd import _lib
    fields = {"enarf_render_args": ("RenderArgs", ["B", "render_scale", "image_coord", "feat_batch_stride", "seed",
                                                    "fine_depth", "dbg_bins", "counters", "workspace", "clamp_mask",
                                                    "march", "ws_epoch"]),
              "enarf_query_args": ("QueryArgs", ["N", "P", "points", "mask_batch_stride", "dbg_weight"]),
              "enarf_prepare_args": ("PrepareArgs", ["coordinate_scale", "parents", "pose_to_camera", "bias", "mlp_pack"]),
              "enarf_render_bwd_args": ("RenderBwdArgs", ["render_scale", "bins", "g_disparity", "grad_mask_batch_stride", "rows_x",
                                                          "rows_dz3", "rows_per_image", "row_blocks", "workspace", "counters",
                                                          "clamp_mask", "multiply_density_with_weight"]),
              "enarf_query_bwd_args": ("QueryBwdArgs", ["N", "points", "g_color", "rows_x", "rows_dz3", "rows_per_image",
                                                        "row_blocks", "multiply_density_with_weight"]),
              "enarf_weight_grad_args": ("WeightGradArgs", ["rows_x", "rows_dz3", "mlp_pack", "rows_per_image", "row_blocks",
                                                            "dW1", "db3", "workspace"]),
              "enarf_prepare_bwd_args": ("PrepareBwdArgs", ["style_dim", "z_rend", "dW", "d_mod_bias", "d_z_rend"])}
    prog = ['#include "enarf_hip.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void){"]
    for cs, (_, fl) in fields.items():
        prog.append(f'printf("%zu\\n", sizeof({cs}));')
        for f in fl:
            prog.append(f'printf("%zu\\n", offsetof({cs}, {f}));')
    prog.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write("\n".join(prog))
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")],
                       check=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    vals = iter(int(x) for x in out)
    for cs, (py, fl) in fields.items():
        st = getattr(_lib, py)
        assert C.sizeof(st) == next(vals), cs
        for f in fl:
            assert getattr(st, f).offset == next(vals), f"{cs}.{f}"


def test_argument_validation_needs_no_device():
    from enarf_gan_amd import _lib
    lib = _lib.load()
    assert lib.enarf_render_fwd(None, None) == -1 and b"null" in lib.enarf_last_error()
    a = _lib.RenderArgs()
    a.B, a.P, a.H, a.W = 1, 40, 256, 256
    assert lib.enarf_render_fwd(C.byref(a), None) == -1 and b"max 32 parts" in lib.enarf_last_error()
    q = _lib.QueryArgs()
    assert lib.enarf_query_fwd(C.byref(q), None) == -1
    p = _lib.PrepareArgs()
    p.B, p.num_joints, p.style_dim, p.origin_location = 1, 24, 20, 7
    assert lib.enarf_prepare(C.byref(p), None) == -1 and b"origin_location" in lib.enarf_last_error()
    assert lib.enarf_triplane_sample_fwd(None, None, None, 1, 1, 1, 1, 1, 0, 0, 0, None, None) == -1
    assert lib.enarf_device_status(None, 0) == -1 and b"flags is null" in lib.enarf_last_error()
    a.P, a.H, a.W = 23, 256, 1          # a one-column plane: the part-probability taps are fetched as pairs of adjacent floats
    assert lib.enarf_render_fwd(C.byref(a), None) == -2 and b"2x2" in lib.enarf_last_error()
    w = _lib.WeightGradArgs()
    w.B, w.rows_per_image = 1, 16
    assert lib.enarf_weight_grad(C.byref(w), None) == -1 and b"null pointer" in lib.enarf_last_error()
    assert lib.enarf_triplane_sample_workspace_bytes(2, 32, 256, 256) == 2 * 3 * 32 * 256 * 256 * 4
    assert lib.enarf_triplane_sample_workspace_bytes(2, 5, 256, 256) == 0


def test_tap_offsets_stay_inside_the_plane(tmp_path):
    """The bilinear tap index math of csrc/enarf_device.h, compiled for the HOST and swept over every float near the
    faces of the unit cube (tests/host/taps_check.hip): make_taps never leaves the plane for any input, and the light
    make_taps_valid used for valid pairs equals it for every |c| < 1 - including the last half texel before +1, where
    x1 == W / y1 == H (the address a round-1 experiment left unclamped)."""
    import subprocess
    exe = str(tmp_path / "taps_check")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "enarf-gan_amd", "csrc"), os.path.join(ROOT, "tests", "host", "taps_check.hip"),
                    "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
    assert int(r.stdout.split()[1]) > 100000


def test_build_tracks_every_included_header():
    """enarf_gan_amd.build rebuilds an object when any header its source includes (directly or through another header)
    changed: a header missing from the dependency list leaves the product library stale after a header-only edit."""
    import re
    from enarf_gan_amd import build as B
    tracked = {os.path.basename(h) for h in B.HEADERS}
    seen, todo = set(), list(B.SOURCES)
    while todo:
        f = todo.pop()
        path = os.path.join(B.CSRC, f) if os.path.exists(os.path.join(B.CSRC, f)) else os.path.join(ROOT, "include", f)
        for inc in re.findall(r'#include\s+"([^"]+)"', open(path).read()):
            if inc not in seen:
                seen.add(inc)
                todo.append(inc)
    assert seen and seen <= tracked, seen - tracked


def test_built_library_has_only_in_place_mfma_chains():
    """ISA lint of the built library (tools/check_mfma_chains.py): a chained MFMA whose vDst differs from its SrcC, or
    partially overlaps it, is issued by the compiler without a wait state and does not reliably see its predecessor's
    result on gfx950 - the cause of the round-1 renderer's run-to-run differences in the split-precision MLP modes."""
    import importlib.util
    from enarf_gan_amd import build
    spec = importlib.util.spec_from_file_location("check_mfma_chains", os.path.join(ROOT, "tools", "check_mfma_chains.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kernels, n_mfma, problems = mod.check(build.LIB)
    assert n_mfma > 1000 and kernels > 20
    assert not problems, problems[:5]


def test_product_reads_no_environment_switches():
    """A renderer whose output changes with an environment variable is a defect: the product sources
    read none, and the library path can only be changed by an explicit use_variant() call from a measurement tool."""
    import glob
    from enarf_gan_amd import _lib
    pkg = os.path.join(ROOT, "enarf-gan_amd")
    for f in glob.glob(os.path.join(pkg, "csrc", "*.h*")):
        assert "getenv" not in open(f).read(), f
    for f in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True):
        if os.path.basename(f) == "build.py":
            continue          # HIPCC (compiler path) only
        src = open(f).read()
        assert "os.environ" not in src and "getenv" not in src, f
    assert _lib.library_info() == {"path": os.path.join(pkg, "csrc", "libenarf_hip.so"), "variant": False}


def test_no_cpu_fallback():
    from enarf_gan_amd import _lib, ops
    with pytest.raises(_lib.EnarfHipError):
        ops.triplane_sample_fwd(torch.zeros(1, 3, 4, 4), torch.zeros(1, 2, 1, 3))
    with pytest.raises(_lib.EnarfHipError):
        ops.triplane_pack(torch.zeros(1, 165, 256, 256))


class Cfg(dict):
    __getattr__ = dict.__getitem__


def _nerf_cfg(**kw):
    c = Cfg(hidden_size=32, Nc=48, Nf=64, origin_location="center_fixed", coordinate_scale=3, render_bs=16384,
            no_ray_direction=True, multiply_density_with_triplane_wieght=False, clamp_mask=False, constant_triplane=True,
            constant_trimask=False, constant_trimask_lr_mul=1, deformation_field=False, selector_mlp=False,
            no_selector=False, time_conditional=True, pose_conditional=False)
    c.update(kw)
    return c


def test_model_mirror_keeps_reference_names_and_state_dict_keys():
    from enarf_gan_amd import synth
    from enarf_gan_amd.cuda_extension.triplane_sampler import (GRID_SAMPLE_INTERPOLATION_MODES,
                                                               GRID_SAMPLE_PADDING_MODES, TriplaneSamplerFunction,
                                                               triplane_sampler, triplane_sampler_cuda)
    from enarf_gan_amd.models.generator import DSONARFGenerator, TriNARFGenerator
    assert GRID_SAMPLE_INTERPOLATION_MODES == {"bilinear": 0, "nearest": 1}
    assert GRID_SAMPLE_PADDING_MODES == {"zeros": 0, "border": 1, "reflection": 2}
    assert callable(triplane_sampler) and hasattr(triplane_sampler_cuda, "triplane_sampler_forward")
    assert issubclass(TriplaneSamplerFunction, torch.autograd.Function)
    g = DSONARFGenerator(Cfg(use_triplane=True, ray_batchsize=4096, nerf_params=_nerf_cfg()), 128, 24,
                         synth.SMPL_PARENTS, 23)
    g.register_canonical_pose(synth.canonical_pose())
    keys = set(g.state_dict().keys())
    want = {"nerf.tri_plane", "nerf.canonical_pose", "nerf.canonical_bone_length", "nerf.canonical_joints",
            "nerf.canonical_parent_joints"}
    for i in range(3):
        for leaf in ("bias", "conv.weight", "conv.modulation.weight", "conv.modulation.bias", "noise.weight"):
            want.add(f"nerf.mlp.layers.{i}.{leaf}")
    assert keys == want
    assert g.nerf.tri_plane.shape == (1, 165, 256, 256) and g.nerf.num_bone == 23
    assert g.nerf.mlp.layers[0].conv.modulation.weight.shape == (32, 20)       # z = PE(frame_time, 10)
    # a reference-shaped snapshot loads strictly
    sd = {f"nerf.mlp.{k}": v for k, v in synth.make_mlp_params(20).items()}
    missing, unexpected = g.load_state_dict(sd, strict=False)
    assert not unexpected
    z1, z2 = g.get_latents(torch.tensor([0.5]), torch.zeros(1, 24, 4, 4))
    assert z1.shape == (1, 20)
    gan = TriNARFGenerator(Cfg(z_dim=256, background_ratio=0.7, crop_background=True, pretrained_background=False,
                               nerf_params=_nerf_cfg(origin_location="center+head")), 128, 24, synth.SMPL_PARENTS, 23,
                           black_background=True)
    assert gan.nerf.num_bone == 24 and gan.nerf.tri_plane.shape[1] == 32 * 3 + 24 * 3
    assert gan.nerf.mlp.layers[0].conv.modulation.weight.shape == (32, 256)
    assert gan.flops == 12800 and g.flops == 12800 and gan.memory_cost == 132          # per query point


def test_mesh_api_fails_like_the_reference_without_its_third_party_stages():
    """render_mesh / create_mesh (libraries/NARF/base.py:65-83, mesh_rendering.py:17-81) need PyMCubes and pytorch3d as
    the reference does: with neither installed they raise ImportError before any device work."""
    from enarf_gan_amd.libraries.NARF.mesh_rendering import create_mesh, render_mesh_
    from enarf_gan_amd.models.generator import TriNARFGenerator
    from enarf_gan_amd.models.narf import TriPlaneNARF
    for name in ("render_mesh", "density_volume"):
        assert callable(getattr(TriPlaneNARF, name)) and callable(getattr(TriNARFGenerator, name))
    assert callable(TriNARFGenerator.create_mesh)
    with pytest.raises(ImportError):
        create_mesh(None, torch.zeros(1, 24, 4, 4), torch.zeros(1, 3, 1))
    with pytest.raises(ImportError):
        render_mesh_((None, None, None), torch.eye(3), 128)


def test_ray_samplers_match_reference_formulas():
    from enarf_gan_amd import synth
    from enarf_gan_amd.libraries.NeRF.ray_sampler import mask_based_sampler, whole_image_grid_ray_sampler
    grid, homo = whole_image_grid_ray_sampler(16, 16, 2, device="cpu")
    assert torch.equal(homo, synth.pixel_centres(16, 2))
    assert grid.shape == (2, 16, 16, 2) and float(grid.min()) == -1 + 1 / 16
    # a patch coarser than the frame: centres scale with render_size / patch_size (ray_sampler.py:58)
    grid2, homo2 = whole_image_grid_ray_sampler(128, 32, 1, device="cpu")
    assert float(homo2[0, 0, 0, 0]) == 2.0 and float(homo2[0, 0, 1, 33]) == 6.0 and float(grid2[0, 1, 0, 1]) == 6.0 / 64 - 1
    mask = torch.zeros(1, 32, 32)
    with pytest.raises(Exception):          # device op: no CPU path
        mask_based_sampler(mask, 64)


def test_host_encodings_match_reference_golden(golden_dir):
    """libraries/NeRF/utils.py mirror (positional encodings, to_local, in_cube) against values recorded from the reference
    (tests/golden/encoding.npz) - bit-exact: the mirror rounds in the same order ((x 2^f) pi)."""
    from enarf_gan_amd import synth
    from enarf_gan_amd.libraries.NARF.pose_utils import transform_pose
    from enarf_gan_amd.libraries.NeRF.utils import in_cube, multi_part_positional_encoding, positional_encoding, to_local
    g = dict(np.load(os.path.join(golden_dir, "encoding.npz"), allow_pickle=False))
    t = lambda k: torch.from_numpy(g[k])
    assert torch.equal(multi_part_positional_encoding(t("bone_length"), 4, 24)[:, :, 0], t("enc_length"))
    assert torch.equal(positional_encoding(t("x"), 6), t("pe_cos_first"))
    assert torch.equal(positional_encoding(t("x"), 3, cos_first=False, cat_dim=1), t("pe_sin_first_cat1"))
    assert torch.equal(multi_part_positional_encoding(t("val"), 2, 24), t("mpe"))
    scene = synth.make_scene(32, 3, "center+head", 20)
    pose_p, _ = transform_pose(scene["pose_to_camera"], scene["bone_length"], "center+head", scene["parents"])
    loc = to_local(t("pts"), pose_p)
    assert loc.shape == t("local").shape and torch.allclose(loc, t("local"), rtol=0, atol=2e-6)   # torch.matmul order
    assert torch.equal(in_cube(t("local")), t("inside")) and torch.equal(in_cube(t("pts") / 4), t("inside3"))


def test_tri_plane_producers_get_the_encoded_bone_length():
    """models/narf.py:277-290: producers are conditioned on the bone-length positional encoding, not on the raw lengths;
    constant_trimask (narf.py:32-38) = generator feature planes + the learned part-probability planes x lr_mul."""
    from enarf_gan_amd import synth
    from enarf_gan_amd.libraries.NeRF.utils import multi_part_positional_encoding
    from enarf_gan_amd.models.narf import TriPlaneNARF
    bl = torch.rand(2, 23, 1) * 0.5 + 0.1
    seen = {}

    def producer(z, enc, truncation_psi=1):
        seen["enc"], seen["psi"] = enc, truncation_psi
        return torch.zeros(z.shape[0], 165, 4, 4)
    m = TriPlaneNARF(_nerf_cfg(constant_triplane=False), 20, 24, parent=synth.SMPL_PARENTS)
    m.tri_plane_gen = producer
    m.compute_tri_plane_feature(torch.zeros(2, 20), bl, truncation_psi=0.4)
    assert seen["enc"].shape == (2, 23 * 2 * 4) and seen["psi"] == 0.4
    assert torch.equal(seen["enc"], multi_part_positional_encoding(bl, 4, 23)[:, :, 0])
    # constant_trimask
    m = TriPlaneNARF(_nerf_cfg(constant_triplane=False, constant_trimask=True, constant_trimask_lr_mul=10), 20, 24,
                     parent=synth.SMPL_PARENTS)
    assert m.tri_plane.shape == (1, 69, 256, 256)
    from enarf_gan_amd import _lib as _l
    from enarf_gan_amd.libraries.stylegan2_ada.networks import Generator
    assert isinstance(m.generator, Generator) and m.generator.img_channels == 96      # the reference's own producer (narf.py:31) ...
    with pytest.raises(_l.EnarfHipError):                                              # ... which runs on the HIP ops: no CPU path
        m.compute_tri_plane_feature(torch.zeros(2, 20), bl)
    m.generator = lambda z, enc, truncation_psi=1: torch.ones(z.shape[0], 96, 256, 256) * enc[:, :1, None, None]
    with torch.no_grad():
        m.tri_plane.fill_(0.25)
    tri = m.compute_tri_plane_feature(torch.zeros(2, 20), bl)
    assert tri.shape == (2, 165, 256, 256) and float(tri[0, 96, 0, 0]) == 2.5 and tri.requires_grad
    assert torch.equal(tri[:, 0, 0, 0], multi_part_positional_encoding(bl, 4, 23)[:, 0, 0])
    # deformation field: the flow generator gets the encoding as well (checked on the CPU up to the HIP call)
    m = TriPlaneNARF(_nerf_cfg(constant_triplane=False, deformation_field=True), 20, 24, parent=synth.SMPL_PARENTS)

    def flow(z, enc, truncation_psi=1):
        seen["flow_enc"] = enc
        raise RuntimeError("stop before the device call")
    m.flow_generator = flow
    with pytest.raises(RuntimeError, match="stop before"):
        m.compute_tri_plane_feature(torch.zeros(2, 20), bl)
    assert seen["flow_enc"].shape == (2, 184)


def test_unsupported_configs_raise():
    from enarf_gan_amd import synth
    from enarf_gan_amd.models.narf import TriPlaneNARF
    with pytest.raises(NotImplementedError):
        TriPlaneNARF(_nerf_cfg(selector_mlp=True), 20, 24, parent=synth.SMPL_PARENTS)
    m = TriPlaneNARF(_nerf_cfg(no_selector=True, clamp_mask=True, multiply_density_with_triplane_wieght=True), 20, 24,
                     parent=synth.SMPL_PARENTS)
    assert m.kernel_flags() == dict(multiply_density_with_weight=True, clamp_mask=True, uniform_part_weight=True)
    with pytest.raises(NotImplementedError):
        TriPlaneNARF(_nerf_cfg(), 20, 24, parent=synth.SMPL_PARENTS, view_dependent=True)
    m = TriPlaneNARF(_nerf_cfg(constant_triplane=False), 20, 24, parent=synth.SMPL_PARENTS)
    from enarf_gan_amd import _lib as _l
    with pytest.raises(_l.EnarfHipError):          # the StyleGAN2-ADA producer is built (round 3) and, like every op here, device-only
        m.compute_tri_plane_feature(torch.zeros(1, 20), torch.ones(1, 23, 1))


# ---------------------------------------------------------------------------------------------- on-disk formats
def _dso_generator():
    from enarf_gan_amd import synth
    from enarf_gan_amd.models.generator import DSONARFGenerator
    g = DSONARFGenerator(Cfg(use_triplane=True, ray_batchsize=4096, nerf_params=_nerf_cfg()), 128, 24, synth.SMPL_PARENTS, 23)
    g.register_canonical_pose(synth.canonical_pose())
    return g


def test_reference_snapshot_schema_loads_into_the_mirror_generator(tmp_path):
    """A snapshot in the reference's schema (train_ENARF_GAN.py:278-294) - written here with synthetic weights, under
    DistributedDataParallel's "module." prefix, with keys of networks this repo does not build - loads by name into the
    mirror generator (DSO_demo.py:37-42: strict=False); what was not loaded is reported; round trip through save_snapshot."""
    from enarf_gan_amd import formats
    src = _dso_generator()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in src.parameters():
            p.copy_(torch.randn(p.shape, generator=g))
    gen_sd = {"module." + k: v.clone() for k, v in src.state_dict().items()}
    gen_sd["module.background_generator.conv1.weight"] = torch.randn(4, 4)          # un-vendored StyleGAN2 parts
    gen_sd["module.nerf.mlp.layers.0.conv.weight"] = gen_sd["module.nerf.mlp.layers.0.conv.weight"].clone()
    path = tmp_path / "snapshot_latest.pth"
    torch.save({"iteration": 1199, "start_time": 0.0, "gen": gen_sd, "dis": {"convs.0.weight": torch.zeros(2)},
                "gen_opt": {"state": {}, "param_groups": []}, "dis_opt": {}}, path)
    dst = _dso_generator()
    rep = formats.load_generator_snapshot(path, dst)
    assert rep.iteration == 1199 and rep.ignored == ["background_generator.conv1.weight"] and not rep.missing
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k
    # a snapshot with another style width: the modulation weights do not fit and are reported, strict=True refuses
    bad = {k: v for k, v in src.state_dict().items()}
    bad["nerf.mlp.layers.1.conv.modulation.weight"] = torch.zeros(64, 256)
    rep = formats.load_generator_snapshot({"gen": bad, "iteration": 3}, _dso_generator())
    assert rep.missing == ["nerf.mlp.layers.1.conv.modulation.weight"] and rep.ignored == ["nerf.mlp.layers.1.conv.modulation.weight"]
    with pytest.raises(RuntimeError):
        formats.load_generator_snapshot({"gen": bad}, _dso_generator(), strict=True)
    # our own snapshot in the same schema
    out = tmp_path / "snapshot_50000.pth"
    formats.save_snapshot(out, dst, 49999)
    snap = formats.read_snapshot(out)
    assert set(snap) == {"iteration", "start_time", "gen", "dis", "gen_opt", "dis_opt"} and snap["iteration"] == 49999
    assert torch.equal(snap["gen"]["nerf.tri_plane"], src.state_dict()["nerf.tri_plane"])


def test_cache_and_sample_data_readers(tmp_path):
    """cache.pickle (dataset/dataset.py:152-185) and sample_data.pickle (prepare_sample_data.py:59-66) written here with the
    reference's keys and shapes; the readers rebuild numpy arrays only and refuse a pickle that names any other global."""
    import pickle
    from enarf_gan_amd import formats, synth
    rs = np.random.RandomState(0)
    N = 5
    pose = synth.make_scene(32, N)["pose_to_camera"].numpy().astype(np.float64)
    K = np.broadcast_to(np.array([[150.0, 0, 64], [0, 150.0, 64], [0, 0, 1]]), (N, 3, 3)).copy()
    R = np.stack([np.linalg.qr(rs.randn(3, 3))[0] for _ in range(N)])
    T = rs.randn(N, 3, 1)
    cache = {"img": [bytes([i]) * 10 for i in range(N)], "camera_intrinsic": K, "smpl_pose": pose, "camera_rotation": R,
             "camera_translation": T, "frame_id": np.arange(N)}
    p = tmp_path / "cache.pickle"
    pickle.dump(cache, open(p, "wb"))
    c = formats.read_cache(p)
    ext = np.broadcast_to(np.eye(4), (N, 4, 4)).copy()
    ext[:, :3, :3], ext[:, :3, 3:] = R, T
    assert np.array_equal(c.pose_to_camera, np.matmul(ext[:, None], pose)) and np.array_equal(c.pose_to_world, pose)
    assert np.allclose(c.inv_intrinsics @ K, np.eye(3)) and len(c.img) == N and np.array_equal(c.frame_id, np.arange(N))
    with pytest.raises(ImportError):
        formats.unpack_image(c.img[0])                 # blosc is the reference's dependency, not installed here
    del cache["camera_rotation"], cache["camera_translation"]
    pickle.dump(cache, open(p, "wb"))
    assert np.array_equal(formats.read_cache(p).pose_to_camera, pose)
    rows = [{"pose_3d": pose[i], "intrinsics": K[i].astype(np.float32), "bone_length": rs.rand(23, 1)} for i in range(3)]
    q = tmp_path / "sample_data.pickle"
    pickle.dump(rows, open(q, "wb"))
    sd = formats.read_sample_data(q)
    assert sd.pose_3d.shape == (3, 24, 4, 4) and sd.intrinsics.dtype == np.float32 and sd.bone_length.shape == (3, 23, 1)
    # a pickle that would run code is refused before anything is called
    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > /dev/null",))
    pickle.dump({"img": [], "camera_intrinsic": Evil()}, open(p, "wb"))
    with pytest.raises(formats.UnsafePickleError):
        formats.read_cache(p)
