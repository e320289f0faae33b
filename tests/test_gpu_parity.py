"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the reference goldens.

Floating point: <= 1e-4 relative to the tensor's scale (the north-star bound) for mlp modes f32 and
bf16x3; validity bit masks, ray validity and depth ranges bit-exact against the oracle.
"""
import numpy as np
import pytest
import torch

from _helpers import DeviceScene, Scene, assert_close, bits_of, load_golden, rel_err
from oracle import enarf_oracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-4
# fp32 MFMA and the 3-term fp16 split meet the north-star bound; the 3-term bf16 split (8-bit halves) is ~3x looser
MODE_TOL = {"f32": RTOL, "f16x3": RTOL, "bf16x3": 4e-4}


@pytest.fixture(scope="module")
def ops():
    from enarf_gan_amd import ops as _ops
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return _ops


def _cpu(t):
    return t.detach().cpu()


# --------------------------------------------------------------------------------------------- prepare / pack
@pytest.mark.parametrize("ol,style_dim,B", [("center_fixed", 256, 2), ("center+head", 20, 1), ("center", 20, 3)])
def test_prepare_matches_oracle(ops, ol, style_dim, B):
    sc = Scene(32, B, ol, style_dim)
    ds = DeviceScene(sc)
    parts = _cpu(ds.parts)
    R = sc.pose_scaled[:, :, :3, :3].reshape(B, sc.P, 9)
    t = sc.pose_scaled[:, :, :3, 3]
    assert torch.equal(parts[:, :, :9], R), "part rotations must be copied exactly"
    assert torch.equal(parts[:, :, 9:12], t), "scaled part translations must be bit-exact"
    assert torch.equal(parts[:, :, 12], sc.scale), "canonical scale must be bit-exact"
    ref_w = sc.weights()
    for b in range(B):
        W1, W2, W3, b1, b2, b3 = [_cpu(x) for x in ops.mlp_unpack(ds.pack[b])]
        for ours, (w, bias) in zip(((W1, b1), (W2, b2), (W3, b3)), ref_w):
            assert_close(ours[0], w[b], "modulated weight", 2e-6)
            assert torch.equal(ours[1], bias)


def test_triplane_pack_is_a_pure_permutation(ops):
    sc = Scene(32, 2)
    tri = sc.raw["tri_plane"].cuda()
    cl = _cpu(ops.triplane_pack(tri))
    ref = sc.raw["tri_plane"][:, :96].reshape(2, 3, 32, 256, 256).permute(0, 1, 3, 4, 2)
    assert torch.equal(cl, ref)
    # ragged widths (4-byte path: W not a multiple of 4, and a last block narrower than 64) and an unaligned view
    g = torch.Generator().manual_seed(0)
    for H, W in ((5, 70), (3, 132), (2, 64)):
        t = torch.randn(1, 96 + 6, H, W, generator=g)
        want = t[:, :96].reshape(1, 3, 32, H, W).permute(0, 1, 3, 4, 2)
        assert torch.equal(_cpu(ops.triplane_pack(t.cuda())), want), (H, W)
    flat = torch.randn(1 + 102 * 4 * 64, generator=g).cuda()
    odd = flat[1:].reshape(1, 102, 4, 64)                     # data pointer 4 bytes off a 16-byte boundary
    assert odd.data_ptr() % 16 == 4
    assert torch.equal(_cpu(ops.triplane_pack(odd)), _cpu(odd)[:, :96].reshape(1, 3, 32, 4, 64).permute(0, 1, 3, 4, 2))


# --------------------------------------------------------------------------------------------- a1 operator
def test_sampler_golden_fwd_bwd(ops):
    g = load_golden("sampler_b2")
    inp = torch.from_numpy(g["input"]).cuda()
    pos = torch.from_numpy(g["position"]).cuda()
    grid = pos.permute(0, 2, 1)[:, :, None, :].contiguous()
    out = ops.triplane_sample_fwd(inp, grid)
    assert_close(_cpu(out)[..., 0], g["output"], "sampler fwd", 1e-5)
    go = torch.from_numpy(g["grad_output"]).cuda()[..., None].contiguous()
    gi, gg = ops.triplane_sample_bwd(go, inp, grid, 0, 0, False, True, True)
    assert_close(_cpu(gi), g["grad_input"], "sampler grad_input", 1e-5)
    assert_close(_cpu(gg)[:, :, 0].permute(0, 2, 1), g["grad_position"], "sampler grad_grid", 1e-5)


@pytest.mark.parametrize("C", [32, 8, 5])
def test_sampler_fast_path_vs_oracle(ops, C):
    g = torch.Generator().manual_seed(C)
    inp = torch.randn(2, 3 * C, 40, 56, generator=g)
    grid = torch.rand(2, 777, 1, 3, generator=g) * 2.2 - 1.1
    ref = O.triplane_sampler_forward(inp, grid)
    for ws in (True, False):
        out = ops.triplane_sample_fwd(inp.cuda(), grid.cuda(), use_workspace=ws)
        assert_close(_cpu(out), ref, f"sampler fwd C={C} ws={ws}", 1e-5)


@pytest.mark.parametrize("pad,align", [(1, False), (2, False), (0, True), (2, True)])
def test_sampler_padding_modes_vs_torch(ops, pad, align):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(9)
    C = 8
    inp = torch.randn(1, 3 * C, 20, 24, generator=g)
    grid = torch.rand(1, 300, 1, 3, generator=g) * 3.0 - 1.5
    mode = ["zeros", "border", "reflection"][pad]
    ref = 0
    for p in range(3):
        g2 = torch.stack([grid[..., p], grid[..., (p + 1) % 3]], dim=-1)
        ref = ref + F.grid_sample(inp[:, p * C:(p + 1) * C], g2, padding_mode=mode, align_corners=align)
    for ws in (True, False):
        out = ops.triplane_sample_fwd(inp.cuda(), grid.cuda(), 0, pad, align, use_workspace=ws)
        assert_close(_cpu(out), ref, f"pad={mode} align={align}", 1e-5)


@pytest.mark.parametrize("pad,align", [(0, False), (1, False), (2, True)])
def test_sampler_backward_fast_path_vs_direct_and_autograd(ops, pad, align):
    """C = 32 backward through the channel-last whole-line-atomics kernel (caller's workspace) against the direct
    NCHW kernel and against torch autograd through three F.grid_sample calls; ragged point count, batch 2, points
    outside [-1, 1]."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(21)
    C = 32
    inp = torch.randn(2, 3 * C, 24, 40, generator=g)
    grid = torch.rand(2, 1000 + 37, 1, 3, generator=g) * 2.4 - 1.2
    go = torch.randn(2, C, 1000 + 37, 1, generator=g)
    mode = ["zeros", "border", "reflection"][pad]
    x, q = inp.clone().requires_grad_(True), grid.clone().requires_grad_(True)
    ref = 0
    for p in range(3):
        g2 = torch.stack([q[..., p], q[..., (p + 1) % 3]], dim=-1)
        ref = ref + F.grid_sample(x[:, p * C:(p + 1) * C], g2, padding_mode=mode, align_corners=align)
    ref.backward(go)
    fast = ops.triplane_sample_bwd(go.cuda(), inp.cuda(), grid.cuda(), 0, pad, align, True, True, use_workspace=True)
    slow = ops.triplane_sample_bwd(go.cuda(), inp.cuda(), grid.cuda(), 0, pad, align, True, True, use_workspace=False)
    for name, (gi, gg) in (("fast", fast), ("direct", slow)):
        assert_close(_cpu(gi), x.grad, f"{name} grad_input pad={mode}", 1e-5)
        assert_close(_cpu(gg), q.grad, f"{name} grad_grid pad={mode}", 1e-5)
    only_grid = ops.triplane_sample_bwd(go.cuda(), inp.cuda(), grid.cuda(), 0, pad, align, False, True)
    assert only_grid[0] is None and torch.equal(only_grid[1], fast[1])
    only_in = ops.triplane_sample_bwd(go.cuda(), inp.cuda(), grid.cuda(), 0, pad, align, True, False)
    assert only_in[1] is None
    assert_close(_cpu(only_in[0]), x.grad, "grad_input alone", 1e-5)


def test_sampler_nearest_last_plane_wins(ops):
    g = torch.Generator().manual_seed(4)
    C = 4
    inp = torch.randn(1, 3 * C, 16, 16, generator=g)
    grid = torch.rand(1, 100, 1, 3, generator=g) * 1.9 - 0.95
    out = _cpu(ops.triplane_sample_fwd(inp.cuda(), grid.cuda(), 1, 0, False))
    ix = torch.round(((grid[0, :, 0, 2] + 1) * 16 - 1) / 2).long()
    iy = torch.round(((grid[0, :, 0, 0] + 1) * 16 - 1) / 2).long()
    ok = (ix >= 0) & (ix < 16) & (iy >= 0) & (iy < 16)
    ref = inp[0, 2 * C:, iy.clamp(0, 15), ix.clamp(0, 15)] * ok
    assert torch.allclose(out[0, :, :, 0], ref)


# --------------------------------------------------------------------------------------------- a9 query
@pytest.mark.parametrize("name", ["query_b2_p23", "query_b1_p24"])
@pytest.mark.parametrize("mode", ["f32", "f16x3", "bf16x3"])
def test_query_vs_oracle_and_golden(ops, name, mode):
    g = load_golden(name)
    sc = Scene(64, int(g["batch"]), str(g["origin_location"]), int(g["style_dim"]))
    ds = DeviceScene(sc)
    pts = torch.from_numpy(g["points"])
    den, col, vb, dc, dw = ds.query(pts, mlp_mode=mode, debug=True)
    oden, ocol, ovalid, taps = O.query(pts, sc.pose_scaled, sc.scale, sc.cpose, sc.raw["tri_plane"], sc.weights(),
                                       return_taps=True)
    obits = bits_of(ovalid)
    ours_bits = _cpu(vb).numpy().view(np.uint32)
    assert np.array_equal(ours_bits, obits), "validity bit masks must be bit-exact vs the oracle"
    assert torch.equal(_cpu(dc), taps["canonical"]), "canonical coordinates must be bit-exact vs the oracle"
    assert_close(_cpu(dw), taps["weight"], "part probability", 1e-5)
    tol = MODE_TOL[mode]
    assert_close(_cpu(den), oden, "density vs oracle", tol)
    assert_close(_cpu(col), ocol, "colour vs oracle", tol)
    # against the reference's own outputs (same points; masks equal except ulp-on-a-face pairs)
    same = ours_bits == g["valid"]
    assert (~same).sum() <= 2
    assert_close(_cpu(den).numpy()[:, 0][same], g["density"][:, 0][same], "density vs reference", tol)
    assert_close(_cpu(col).numpy().transpose(0, 2, 1)[same], g["color"].transpose(0, 2, 1)[same],
                 "colour vs reference", tol)


def _cube_face_points(sc, per=30000, seed=11):
    """(1, 3, N) camera-space points (fp32) aimed, in fp64, at the faces / edges / corners of every part's canonical cube
    (kept where the local cube test can pass: only some parts' canonical faces lie inside their local cube) and of every
    part's local cube; the snapped coordinates are 1, 1 - 2^-24, 1 - 2^-23, 1 - 2^-21 or 1 + 2^-23 in magnitude."""
    g = torch.Generator().manual_seed(seed)
    P = sc.P
    Rc, tc = sc.cpose[:, :3, :3].double(), sc.cpose[:, :3, 3].double()
    R, t = sc.pose_scaled[0, :, :3, :3].double(), sc.pose_scaled[0, :, :3, 3].double()
    scale = sc.scale[0].double()[:, None, None]

    def snapped(n):
        x = torch.rand(P, n, 3, generator=g, dtype=torch.float64) * 2 - 1
        eps = torch.tensor([0.0, 2.0 ** -24, 2.0 ** -23, 2.0 ** -21, -2.0 ** -23], dtype=torch.float64)[
            torch.randint(0, 5, (P, n, 3), generator=g)]
        snap = torch.rand(P, n, 3, generator=g) < 0.45
        snap[..., 0] |= ~snap.any(-1)                                        # at least one coordinate on a face
        sign = torch.where(torch.rand(P, n, 3, generator=g) < 0.5, -1.0, 1.0).double()
        return torch.where(snap, sign * (1.0 - eps), x)
    c = snapped(per)
    local_c = torch.einsum("pji,pnj->pni", Rc, c - tc[:, None]) / scale
    keep = (local_c.abs() <= 1).all(-1)
    local_l = snapped(per // 30)
    local = torch.cat([local_c[keep].reshape(-1, 3), local_l.reshape(-1, 3)])
    part = torch.cat([torch.arange(P)[:, None].expand(P, per)[keep], torch.arange(P)[:, None].expand(P, per // 30).reshape(-1)])
    cam = torch.einsum("nij,nj->ni", R[part], local) + t[part]
    return cam.t()[None].float().contiguous()


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_query_points_on_the_faces_of_the_canonical_cube(ops, mode):
    """Points whose canonical coordinates sit on, or a few ulp inside / outside, the faces, edges and corners of a part's
    cube - where the bilinear footprint's upper tap is texel W (its weight is 0; its address must still be legal,
    enarf_device.h make_taps) and where validity flips. Targets are built in canonical space and mapped back to camera
    space in fp64; validity and canonical coordinates must match the oracle bit for bit, values within the parity bound."""
    sc = Scene(32, 1, "center_fixed", 20)
    ds = DeviceScene(sc)
    pts = _cube_face_points(sc)
    den, col, vb, dc, dw = ds.query(pts, mlp_mode=mode, debug=True)
    oden, ocol, ovalid, taps = O.query(pts, sc.pose_scaled, sc.scale, sc.cpose, sc.raw["tri_plane"], sc.weights(), return_taps=True)
    near_face = (taps["canonical"].abs().amax(dim=2) > 1 - 1e-6) & ovalid                      # (B, P, N)
    near_local = (taps["local"].abs().amax(dim=2) > 1 - 1e-6) & ovalid
    just_out = ((taps["canonical"].abs().amax(dim=2) - 1).abs() < 1e-6) & ~ovalid
    assert int(near_face.sum()) > 300 and int(near_local.sum()) > 300 and int(just_out.sum()) > 300, \
        (int(near_face.sum()), int(near_local.sum()), int(just_out.sum()))
    assert np.array_equal(_cpu(vb).numpy().view(np.uint32), bits_of(ovalid))
    assert torch.equal(_cpu(dc), taps["canonical"])
    assert torch.isfinite(den).all() and torch.isfinite(col).all()
    assert_close(_cpu(den), oden, "density on the faces", MODE_TOL[mode])
    assert_close(_cpu(col), ocol, "colour on the faces", MODE_TOL[mode])


def test_query_bf16_mode_is_close(ops):
    g = load_golden("query_b2_p23")
    sc = Scene(64, 2, "center_fixed", 256)
    ds = DeviceScene(sc)
    pts = torch.from_numpy(g["points"])
    d32, c32 = ds.query(pts, mlp_mode="f32")
    d16, c16 = ds.query(pts, mlp_mode="bf16")
    assert rel_err(_cpu(d16), _cpu(d32)).max() < 8e-2
    assert rel_err(_cpu(c16), _cpu(c32)).max() < 8e-2


def test_query_ragged_and_empty(ops):
    sc = Scene(32, 1, "center_fixed", 20)
    ds = DeviceScene(sc)
    g = torch.Generator().manual_seed(0)
    for N in (1, 63, 65, 1025):
        pts = (sc.pose_scaled[0, 5, :3, 3][None, :, None] + torch.randn(1, 3, N, generator=g) * 0.5)
        den, col = ds.query(pts)
        oden, ocol, _ = O.query(pts, sc.pose_scaled, sc.scale, sc.cpose, sc.raw["tri_plane"], sc.weights())
        assert_close(_cpu(den), oden, f"density N={N}")
        assert_close(_cpu(col), ocol, f"colour N={N}")
    den, col = ds.query(torch.zeros(1, 3, 0))
    assert den.shape == (1, 1, 0)


def test_query_culling_does_not_depend_on_stale_lds(ops, tmp_path):
    """Regression test of ADVICE r02 (high): query_kernel computed the per-part cull radii from part frames in LDS right
    after stage_common(), which ends without a barrier - wave 0 could read frames waves 1-3 had not written yet and cull a
    part with a garbage radius. It passed as long as the LDS still held the same frames from an earlier workgroup. Here
    every CU's LDS is overwritten with NaN / huge bit patterns (tests/host/lds_poison.hip, built with hipcc) before each
    launch of enarf_query_fwd at B = 3 (other frames per image), and the culled production path must equal the debug path
    (which keeps every part as a candidate) bit for bit, launch after launch."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "liblds_poison.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC",
                    os.path.join(root, "tests", "host", "lds_poison.hip"), "-o", so], check=True, capture_output=True)
    lib = C.CDLL(so)
    lib.lds_poison.restype, lib.lds_poison.argtypes = C.c_int, [C.c_uint, C.c_void_p]
    sc = Scene(32, 3, "center+head", 20)
    ds = DeviceScene(sc)
    g = torch.Generator().manual_seed(4)
    jp = sc.pose_scaled[:, :, :3, 3]
    pick = torch.randint(0, jp.shape[1], (3, 20000), generator=g)
    pts = (torch.gather(jp, 1, pick[..., None].expand(-1, -1, 3)).permute(0, 2, 1) + 0.4 * torch.randn(3, 3, 20000, generator=g)).contiguous()
    ref = ds.query(pts, mlp_mode="f32", debug=True)            # density, colour, valid bits (+ taps): no culling
    assert int((ref[2] != 0).sum()) > 10000
    stream = torch.cuda.current_stream().cuda_stream
    for pattern in (0x7FC00000, 0x7F7FFFFF, 0xFFFFFFFF, 0x00000000):     # NaN, FLT_MAX, -NaN, zeros
        for _ in range(3):
            assert lib.lds_poison(pattern, stream) == 0
            den, col, vb = ds.query(pts, mlp_mode="f32", need_valid=True)
            assert torch.equal(vb, ref[2]), hex(pattern)
            assert torch.equal(den, ref[0]) and torch.equal(col, ref[1]), hex(pattern)


def test_query_multiply_density_with_weight(ops):
    sc = Scene(32, 1, "center_fixed", 20)
    ds = DeviceScene(sc)
    g = torch.Generator().manual_seed(1)
    pts = (sc.pose_scaled[0, 2, :3, 3][None, :, None] + torch.randn(1, 3, 2000, generator=g) * 0.6)
    den, _ = ds.query(pts, multiply_density_with_weight=True)
    oden, _, _ = O.query(pts, sc.pose_scaled, sc.scale, sc.cpose, sc.raw["tri_plane"], sc.weights(),
                         multiply_density_with_weight=True)
    assert_close(_cpu(den), oden, "density * max weight")


# --------------------------------------------------------------------------------------------- a13 render
RENDER_CASES = ["render_c0_64_b1", "render_c1_128_b1_p23", "render_c1_128_b1_p24", "render_gan_32_b2",
                "render_c4s_32_b2"]       # the last one: BASELINE config C4's sample counts, Nc 72 / Nf 96 (> 64)


def _render_case(name):
    g = load_golden(name)
    sc = Scene(int(g["size"]), int(g["batch"]), str(g["origin_location"]), int(g["style_dim"]))
    idx = torch.from_numpy(g["ray_idx"].astype(np.int64))
    coord = torch.gather(sc.raw["image_coord"], 3, idx[:, None, None, :].expand(-1, 1, 3, -1)).contiguous()
    return g, sc, coord, torch.from_numpy(g["bins"])


@pytest.mark.parametrize("name", RENDER_CASES)
@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_render_vs_oracle_and_golden(ops, name, mode):
    g, sc, coord, bins = _render_case(name)
    Nc, Nf, B = int(g["Nc"]), int(g["Nf"]), int(g["batch"])
    ds = DeviceScene(sc)
    out = ds.render(coord, Nc, Nf, bins, mlp_mode=mode, debug=True)
    rc, rm, rd, taps = sc.oracle_render(coord, Nc, Nf, bins)
    t = {k: _cpu(v) for k, v in out.taps.items()}
    # integer / boolean work: bit-exact against the oracle
    assert np.array_equal(t["ray_validity"].numpy().astype(bool), taps["ray_validity"].numpy())
    assert torch.equal(t["depth_min"], taps["depth_min"]) and torch.equal(t["depth_max"], taps["depth_max"])
    live = taps["ray_validity"].numpy() if B == 1 else np.ones((B, coord.shape[-1]), dtype=bool)
    ours_bits = t["fine_valid"].numpy().view(np.uint32)
    assert np.array_equal(ours_bits[live], bits_of(taps["fine_valid"])[live]), "fine-sample validity masks"
    assert_close(t["coarse_density"].numpy()[live], taps["coarse_density"].numpy()[live], "coarse density")
    assert_close(t["fine_density"].numpy()[live], taps["fine_density"].numpy()[live], "fine density")
    assert_close(_cpu(out.fine_depth)[:, 0].numpy()[live], taps["fine_depth"].numpy()[live], "fine depth", 1e-6)
    assert_close(_cpu(out.fine_weights)[:, 0].numpy()[live], taps["fine_weights"].numpy()[live], "fine weights")
    assert_close(_cpu(out.color), rc, "colour vs oracle")
    assert_close(_cpu(out.mask), rm, "mask vs oracle")
    assert_close(_cpu(out.disparity), rd, "disparity vs oracle")
    # the reference's own outputs
    assert np.array_equal(t["ray_validity"].numpy().astype(bool), g["ray_validity"])
    assert_close(_cpu(out.color), g["color"], "colour vs reference", frac_ok=2e-3)
    assert_close(_cpu(out.mask), g["mask"], "mask vs reference", frac_ok=2e-3)
    assert_close(_cpu(out.disparity), g["disparity"], "disparity vs reference", frac_ok=2e-3)
    if B == 1:   # dropped rays are exact zeros (rendering.py:337-350)
        dead = ~g["ray_validity"][0]
        assert float(_cpu(out.mask)[0][dead].abs().max()) == 0.0 and float(_cpu(out.color)[0][:, dead].abs().max()) == 0.0
    # integer foreground mask (ENARF_GAN_demo.py:79) away from quantisation steps
    frac = (g["mask"].astype(np.float64) * 255) % 1.0
    safe = (frac > 1e-2) & (frac < 1 - 1e-2)
    assert np.array_equal((_cpu(out.mask).numpy() * 255).astype(np.uint8)[safe], (g["mask"] * 255).astype(np.uint8)[safe])


def test_render_production_path_no_debug_matches_debug(ops):
    g, sc, coord, bins = _render_case("render_c1_128_b1_p23")
    ds = DeviceScene(sc)
    a = ds.render(coord, 48, 64, bins, debug=True)
    b = ds.render(coord, 48, 64, bins, debug=False, want_fine=False)
    assert torch.equal(a.color, b.color) and torch.equal(a.mask, b.mask) and torch.equal(a.disparity, b.disparity)


def test_render_in_kernel_sampling_replays_through_oracle(ops):
    """bins=None: the kernel draws its own importance samples; feeding the bins it reports to the oracle
    must reproduce its image, and the bins must be a sorted sample of [0, 1)."""
    g, sc, coord, _ = _render_case("render_c0_64_b1")
    ds = DeviceScene(sc)
    out = ds.render(coord, 48, 32, None, seed=99, debug=True, count=True)
    kb = _cpu(out.taps["bins"])
    live = g["ray_validity"][0]
    kbl = kb[0][torch.from_numpy(live)]
    assert float(kbl.min()) >= 0.0 and float(kbl.max()) < 1.0
    assert bool((kbl[:, 1:] >= kbl[:, :-1]).all())
    rc, rm, rd = sc.oracle_render(coord, 48, 32, kb, taps=False)
    assert_close(_cpu(out.color), rc, "colour, replayed bins")
    assert_close(_cpu(out.mask), rm, "mask, replayed bins")
    out2 = ds.render(coord, 48, 32, None, seed=99)
    assert torch.equal(out.color, out2.color), "same seed must reproduce"
    out3 = ds.render(coord, 48, 32, None, seed=100)
    assert not torch.equal(out.color, out3.color)
    # importance sampling concentrates bins where the coarse weights are: compare with a uniform draw
    cnt = _cpu(out.counters).numpy()
    assert cnt[2] == int(live.sum()) and cnt[0] > 0 and cnt[1] > 0


def test_render_counters_match_oracle_pair_count(ops):
    g, sc, coord, bins = _render_case("render_c0_64_b1")
    ds = DeviceScene(sc)
    out = ds.render(coord, 48, 32, bins, count=True)
    _, _, _, taps = sc.oracle_render(coord, 48, 32, bins)
    live = taps["ray_validity"][0]
    cv = taps["coarse_valid"][0][:, live]          # (P, m', Nc)
    fv = taps["fine_valid"][0][:, live][..., :-1]  # the last fine sample is never queried in production
    assert int(_cpu(out.counters)[0]) == int(cv.sum() + fv.sum())


def test_render_full_image_properties(ops):
    """Full 128x128 frame (BASELINE config C1): size-independent properties."""
    sc = Scene(128, 1, "center_fixed", 20)
    ds = DeviceScene(sc)
    coord = sc.raw["image_coord"]
    out = ds.render(coord, 48, 64, None, seed=5, debug=True)
    m, c = _cpu(out.mask)[0], _cpu(out.color)[0]
    rv = _cpu(out.taps["ray_validity"])[0].bool()
    assert torch.isfinite(m).all() and torch.isfinite(c).all() and torch.isfinite(_cpu(out.disparity)).all()
    # (a weight can be ~-1e-5: lerp(dmin, dmax, bin) is not monotonic to the last ulp for near-equal bins; same in the reference)
    assert float(m.min()) >= -1e-4 and float(m.max()) <= 1.0 + 1e-4
    assert float(c.abs().max()) <= 1.0 + 1e-5
    assert float(m[~rv].abs().max()) == 0.0
    assert 0.3 < float(rv.float().mean()) < 0.7
    assert_close(_cpu(out.fine_weights)[0, 0].sum(-1), m, "sum of fine weights == mask", 1e-5)
    # rays are independent: any subset rendered alone gives bit-identical pixels (given the same bins)
    kb = out.taps["bins"]
    idx = torch.arange(5000, 5000 + 777)
    sub = ds.render(coord[..., idx], 48, 64, _cpu(kb)[:, idx])
    assert torch.equal(_cpu(sub.mask)[0], m[idx])
    # oracle on a slice of the full frame
    sl = torch.arange(64 * 128 + 40, 64 * 128 + 40 + 48)
    rc, rm, rd = sc.oracle_render(coord[..., sl], 48, 64, _cpu(kb)[:, sl], taps=False)
    assert_close(m[sl][None], rm, "mask slice vs oracle")


def test_render_early_termination_is_bounded_and_opt_in(ops):
    """early_stop_eps > 0 skips fine tiles behind (coarse) transmittance < eps: the image moves by a few eps; eps = 0 is exact.
    An opaque slab is forced by scaling the density head's bias so that rays do terminate."""
    sc = Scene(64, 1, "center_fixed", 20)
    sc.raw["mlp"]["layers.2.bias"] = sc.raw["mlp"]["layers.2.bias"].clone()
    sc.raw["mlp"]["layers.2.bias"][0, 3, 0] = 3.0            # large positive sigma everywhere inside the cubes
    ds = DeviceScene(sc)
    coord = sc.raw["image_coord"]
    exact = ds.render(coord, 48, 64, None, seed=11, debug=True, count=True)
    bins = exact.taps["bins"]
    fast = ds.render(coord, 48, 64, _cpu(bins), early_stop_eps=1e-3, count=True)
    again = ds.render(coord, 48, 64, _cpu(bins), early_stop_eps=0.0)
    assert torch.equal(again.mask, exact.mask)
    skipped = int(_cpu(fast.counters)[4])
    assert skipped > 0 and int(_cpu(exact.counters)[4]) == 0
    # the coarse pass only estimates the fine transmittance (mid-point samples, other sample positions): the bound
    # is a small multiple of eps, not eps itself
    assert float((fast.mask - exact.mask).abs().max()) < 2e-2
    assert float((fast.color - exact.color).abs().max()) < 2e-2
    assert int(_cpu(fast.counters)[0]) < int(_cpu(exact.counters)[0])      # fewer (part, point) pairs gathered


def test_render_rejects_unsupported(ops):
    sc = Scene(32, 1, "center_fixed", 20)
    ds = DeviceScene(sc)
    with pytest.raises(NotImplementedError):
        ds.render(sc.raw["image_coord"], 130, 96, None)


def test_render_wide_sample_counts_in_kernel_sampling(ops):
    """Nc / Nf above 64 (two samples per lane): in-kernel importance sampling replayed through the oracle."""
    sc = Scene(32, 1, "center_fixed", 20)
    ds = DeviceScene(sc)
    coord = sc.raw["image_coord"][..., 32 * 12:32 * 12 + 96].contiguous()
    for (Nc, Nf) in ((72, 96), (128, 128), (48, 100), (100, 40)):
        out = ds.render(coord, Nc, Nf, None, seed=3, debug=True, mlp_mode="f32")
        kb = _cpu(out.taps["bins"])
        live = _cpu(out.taps["ray_validity"])[0].bool()
        kbl = kb[0][live]
        assert float(kbl.min()) >= 0.0 and float(kbl.max()) < 1.0 and bool((kbl[:, 1:] >= kbl[:, :-1]).all())
        rc, rm, rd = sc.oracle_render(coord, Nc, Nf, kb, taps=False)
        assert_close(_cpu(out.mask), rm, f"mask Nc={Nc} Nf={Nf}")
        assert_close(_cpu(out.color), rc, f"colour Nc={Nc} Nf={Nf}")
        assert_close(_cpu(out.disparity), rd, f"disparity Nc={Nc} Nf={Nf}")


@pytest.mark.parametrize("name,Nc,Nf", [("render_c1_128_b1_p23", 48, 64), ("render_c1_128_b1_p24", 48, 64),
                                         ("render_gan_32_b2", 48, 64)])
def test_fused_step_is_bit_identical_to_separate_calls(ops, name, Nc, Nf):
    """enarf_render_step_fwd (re-layout + prepare + ray set-up in one launch, then the march) against
    enarf_triplane_pack -> enarf_prepare -> enarf_render_fwd on the same inputs: every output bit for bit, including
    the part frames / MLP pack / channel-last planes the pre-march launch leaves behind."""
    g, sc, coord, bins = _render_case(name)
    ds = DeviceScene(sc)                               # separate calls
    ref = ds.render(coord, Nc, Nf, bins, debug=True)
    s, d = sc.raw, ds.dev
    feat = torch.full_like(ds.feat_cl, float("nan"))
    st = ops.RenderStep(s["pose_to_camera"].to(d), s["bone_length"].to(d), sc.cbl.to(d), s["z_rend"].to(d), ds.mlp,
                        s["parents"], sc.ol, sc.cs, coord.to(d), ds.inv_K, ds.cpose, ds.tri, feat, Nc, Nf,
                        bins=bins.to(d), debug=True)
    out = st.run()
    assert torch.equal(st.parts, ds.parts) and torch.equal(st.pack, ds.pack) and torch.equal(feat, ds.feat_cl)
    for k in ("color", "mask", "disparity", "fine_weights", "fine_depth"):
        assert torch.equal(getattr(out, k), getattr(ref, k)), k
    for k, v in ref.taps.items():
        assert torch.equal(out.taps[k], v), k
    # the two phases as two calls, and a cached re-layout (tri_nchw = NULL), give the same again
    st2 = ops.RenderStep(s["pose_to_camera"].to(d), s["bone_length"].to(d), sc.cbl.to(d), s["z_rend"].to(d), ds.mlp,
                         s["parents"], sc.ol, sc.cs, coord.to(d), ds.inv_K, ds.cpose, ds.tri, ds.feat_cl, Nc, Nf,
                         bins=bins.to(d), relayout=False)
    st2.run(ops.STEP_PRE)
    out2 = st2.run(ops.STEP_MARCH)
    assert torch.equal(out2.color, ref.color) and torch.equal(out2.mask, ref.mask)


def test_fused_step_rejects_mismatched_buffers(ops):
    g, sc, coord, bins = _render_case("render_c0_64_b1")
    ds = DeviceScene(sc)
    s, d = sc.raw, ds.dev
    st = ops.RenderStep(s["pose_to_camera"].to(d), s["bone_length"].to(d), sc.cbl.to(d), s["z_rend"].to(d), ds.mlp,
                        s["parents"], sc.ol, sc.cs, coord.to(d), ds.inv_K, ds.cpose, ds.tri, ds.feat_cl, 48, 32)
    st.ra.parts = ds.parts.data_ptr()                 # not the buffer the prepare stage writes
    with pytest.raises(RuntimeError, match="same buffers"):
        st.run()
    st.ra.parts = st.parts.data_ptr()
    with pytest.raises(RuntimeError, match="phases"):
        st.run(0)


def test_render_ragged_ray_count_batch3_every_ray_marched_once(ops):
    """n not a multiple of the 64-ray set-up blocks, B = 3: the banded, cost-classed ray lists must hand out every
    live ray exactly once (outputs for all rays, equal to the oracle; the live-ray counter equals B * n since batches
    > 1 drop nothing)."""
    sc = Scene(32, 3, "center+head", 20)
    ds = DeviceScene(sc)
    g = torch.Generator().manual_seed(3)
    idx = torch.randperm(32 * 32, generator=g)[:333].sort().values
    coord = sc.raw["image_coord"][..., idx].contiguous()
    bins = torch.rand(3, 333, 32, generator=g).sort(-1).values
    out = ds.render(coord, 24, 32, bins, count=True)
    rc, rm, rd = sc.oracle_render(coord, 24, 32, bins, taps=False)
    assert_close(_cpu(out.color), rc, "colour")
    assert_close(_cpu(out.mask), rm, "mask")
    assert_close(_cpu(out.disparity), rd, "disparity")
    assert int(_cpu(out.counters)[2]) == 3 * 333


def test_render_edge_cases_no_hit_single_ray_minimal_samples(ops):
    """Degenerate launches: a frame whose rays all miss (batch 1: everything dropped, nothing marched), a single ray,
    the smallest sample counts, and a batch in which one image has no hit at all - each against the oracle."""
    sc = Scene(32, 1, "center_fixed", 20)
    ds = DeviceScene(sc)
    full = sc.raw["image_coord"]
    # (a) rays pointing away from the body: flip the pixel coordinates far outside the frame
    away = full.clone()
    away[:, :, 0] += 1.0e4
    out = ds.render(away[..., :200].contiguous(), 48, 32, None, seed=1, count=True, debug=True)
    assert int(_cpu(out.counters)[2]) == 0 and int(_cpu(out.taps["ray_validity"]).sum()) == 0
    for t in (out.color, out.mask, out.disparity, out.fine_weights, out.fine_depth):
        assert float(t.abs().max()) == 0.0
    # (b) one ray (a hit: centre of the frame), (c) minimal sample counts
    mid = full[..., 16 * 32 + 16:16 * 32 + 17].contiguous()
    for Nc, Nf in ((48, 32), (2, 2), (16, 3)):
        g = torch.Generator().manual_seed(Nc)
        bins = torch.rand(1, 1, Nf, generator=g).sort(-1).values
        o1 = ds.render(mid, Nc, Nf, bins)
        rc, rm, rd = sc.oracle_render(mid, Nc, Nf, bins, taps=False)
        assert_close(_cpu(o1.color), rc, f"single ray colour Nc={Nc} Nf={Nf}")
        assert_close(_cpu(o1.mask), rm, f"single ray mask Nc={Nc} Nf={Nf}")
        assert_close(_cpu(o1.disparity), rd, f"single ray disparity Nc={Nc} Nf={Nf}")
    # (d) batch of 2 where image 1 sees nothing: batch > 1 drops no ray, the empty image's rays are marched over [near, far]
    sc2 = Scene(32, 2, "center_fixed", 20)
    ds2 = DeviceScene(sc2)
    coord = sc2.raw["image_coord"][..., 300:300 + 130].contiguous()
    coord[1, :, 0] += 1.0e4
    g = torch.Generator().manual_seed(7)
    bins = torch.rand(2, 130, 32, generator=g).sort(-1).values
    o2 = ds2.render(coord, 24, 32, bins, count=True)
    rc, rm, rd = sc2.oracle_render(coord, 24, 32, bins, taps=False)
    assert_close(_cpu(o2.color), rc, "half-empty batch colour")
    assert_close(_cpu(o2.mask), rm, "half-empty batch mask")
    assert int(_cpu(o2.counters)[2]) == 2 * 130 and float(_cpu(o2.mask)[1].abs().max()) == 0.0


@pytest.mark.parametrize("Nc,Nf", [(64, 64), (128, 128), (33, 17), (100, 70), (16, 128)])
def test_render_sample_count_variants_vs_oracle(ops, Nc, Nf):
    """Sample counts that change the tile / spare-wave layout of the march: four full coarse tiles (no spare wave),
    two samples per lane, counts that are not multiples of 16 - a band of consecutive rays against the oracle."""
    sc = Scene(32, 1, "center_fixed", 20)
    ds = DeviceScene(sc)
    coord = sc.raw["image_coord"][..., 32 * 12:32 * 12 + 96].contiguous()
    g = torch.Generator().manual_seed(Nc * 131 + Nf)
    bins = torch.rand(1, 96, Nf, generator=g).sort(-1).values
    out = ds.render(coord, Nc, Nf, bins)
    rc, rm, rd = sc.oracle_render(coord, Nc, Nf, bins, taps=False)
    assert float(rm.max()) > 0.1
    assert_close(_cpu(out.color), rc, f"colour Nc={Nc} Nf={Nf}")
    assert_close(_cpu(out.mask), rm, f"mask Nc={Nc} Nf={Nf}")
    assert_close(_cpu(out.disparity), rd, f"disparity Nc={Nc} Nf={Nf}")


@pytest.mark.parametrize("S,B,Nc,Nf,n0,nr", [(64, 1, 48, 64, 0, 4096), (32, 3, 24, 32, 5, 333), (32, 1, 72, 96, 300, 400),
                                            (32, 2, 100, 70, 0, 1024), (32, 1, 2, 2, 16 * 32 + 16, 1), (32, 1, 16, 128, 200, 65)])
def test_both_march_kernels_give_the_same_bits(ops, S, B, Nc, Nf, n0, nr):
    """ENARF_MARCH_RAY (a 4-wave workgroup per ray) and ENARF_MARCH_TASK (16-sample tiles as tasks, several rays in
    flight) share every stage; whichever `auto` picks, colour, mask, disparity, the fine outputs, the drawn bins and
    the work counters are equal bit for bit - with in-kernel importance sampling, early termination, ragged ray counts,
    batches (image switches drain the task pipeline), one ray, two samples per lane."""
    sc = Scene(S, B, "center_fixed", 20)
    ds = DeviceScene(sc)
    coord = sc.raw["image_coord"][..., n0:n0 + nr].contiguous()
    for kw in (dict(seed=5), dict(seed=5, early_stop_eps=1e-3), dict(seed=9, mlp_mode="f16x3")):
        a = ds.render(coord, Nc, Nf, None, count=True, return_bins=True, march="ray", **kw)
        b = ds.render(coord, Nc, Nf, None, count=True, return_bins=True, march="task", **kw)
        c = ds.render(coord, Nc, Nf, None, count=True, return_bins=True, **kw)
        for other in (b, c):
            for name in ("color", "mask", "disparity", "fine_weights", "fine_depth"):
                assert torch.equal(getattr(a, name), getattr(other, name)), (name, kw)
            assert torch.equal(a.taps["bins"], other.taps["bins"]), kw
            assert torch.equal(a.counters[:5], other.counters[:5]), (kw, a.counters, other.counters)
        assert int(_cpu(a.counters)[7]) == 0 and int(_cpu(b.counters)[7]) == 0        # no watchdog exit
        if B > 1:        # rays that miss every cube take a short cut in production runs and the general stages in debug runs
            dbg = ds.render(coord, Nc, Nf, None, count=True, debug=True, march="ray", **kw)
            assert int((_cpu(dbg.taps["ray_validity"]) == 0).sum()) > 20
            for name in ("color", "mask", "disparity", "fine_weights", "fine_depth"):
                assert torch.equal(getattr(a, name), getattr(dbg, name)), (name, kw, "debug vs production")
            assert torch.equal(a.taps["bins"], dbg.taps["bins"]), (kw, "bins: debug vs production")
    assert nr < 100 or float(a.mask.max()) > 0.05
    with pytest.raises(KeyError):
        ds.render(coord, Nc, Nf, None, march="fastest")


def test_steps_on_two_streams_do_not_interfere(ops):
    """Steps issued on two HIP streams with private intermediates (bench.py --streams 2) give the same bits as serial
    steps: the workspace is per stream, everything else is read-only."""
    g, sc, coord, bins = _render_case("render_c1_128_b1_p23")
    ds = DeviceScene(sc)
    s, d = sc.raw, ds.dev
    full = s["image_coord"].to(d)
    ref = ds.render(full, 48, 64, None, seed=11)
    streams = [torch.cuda.Stream(d), torch.cuda.Stream(d)]
    torch.cuda.synchronize()
    outs = []
    for i in range(6):
        with torch.cuda.stream(streams[i % 2]):
            feat = torch.empty_like(ds.feat_cl)
            st = ops.RenderStep(s["pose_to_camera"].to(d), s["bone_length"].to(d), sc.cbl.to(d), s["z_rend"].to(d), ds.mlp,
                                s["parents"], sc.ol, sc.cs, full, ds.inv_K, ds.cpose, ds.tri, feat, 48, 64, seed=11)
            outs.append(st.run())
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o.color, ref.color) and torch.equal(o.mask, ref.mask)


def test_workspace_epochs_alternate_headers_without_a_fill(ops):
    """ws_epoch protocol (include/enarf_hip.h): consecutive forward calls alternate between the two queue headers, each
    clearing the other for its successor; a backward in between falls back to epoch 0. Every call gives the same bits."""
    from enarf_gan_amd import ops as O_
    sc = Scene(32, 1, "center_fixed", 20)
    ds = DeviceScene(sc)
    coord = sc.raw["image_coord"]
    key = O_._ws_key(ds.dev)
    O_._render_ws.pop(key, None)                  # earlier tests may have grown the cached workspace: start from none
    O_._render_epoch.pop(key, None)
    ref = ds.render(coord, 24, 32, None, seed=3, count=True)
    seen = []
    for i in range(5):
        seen.append(O_._render_epoch[key])
        out = ds.render(coord, 24, 32, None, seed=3, count=True)
        assert torch.equal(out.color, ref.color) and torch.equal(out.counters, ref.counters), i
    assert seen == list(range(seen[0], seen[0] + 5)) and seen[0] >= 1
    bins = ds.render(coord, 24, 32, None, seed=3, return_bins=True).taps["bins"]
    O_.render_bwd(coord.to(ds.dev), ds.inv_K, ds.parts, ds.cpose, ds.tri, ds.feat_cl, ds.pack, 32, bins,
                  torch.ones(1, 3, 1024, device=ds.dev), torch.ones(1, 1024, device=ds.dev))
    assert O_._render_epoch[key] == 1                      # the backward used header 0 with its own fill
    for i in range(3):
        out = ds.render(coord, 24, 32, None, seed=3, count=True)
        assert torch.equal(out.color, ref.color) and torch.equal(out.counters, ref.counters)
    # a launch that has to re-allocate the workspace starts counting again
    before = O_._render_ws[key].numel()
    big = Scene(160, 1, "center_fixed", 20)
    dsb = DeviceScene(big)
    dsb.render(big.raw["image_coord"], 24, 32, None, seed=3)
    assert O_._render_ws[key].numel() > before and O_._render_epoch[key] == 1


def test_in_kernel_importance_sampler_follows_the_reference_law(ops):
    """The kernel draws its importance samples as sorted uniforms pushed through the inverse CDF of the smoothed
    coarse weights; the reference draws multinomial bins + uniform jitter and sorts (rendering.py:192-197). Same law:
    over many seeds the kernel's bins must fall into coarse bin j with probability ws_j / sum(ws) (z-scores ~ N(0,1))
    and be uniform inside a bin - checked against the oracle's weights AND against the oracle's own sampler."""
    sc = Scene(32, 1, "center_fixed", 20)
    ds = DeviceScene(sc)
    coord = sc.raw["image_coord"][..., 32 * 10:32 * 10 + 256].contiguous()
    Nc, Nf, S = 48, 64, 40
    first = ds.render(coord, Nc, Nf, None, seed=1000, debug=True)
    live = _cpu(first.taps["ray_validity"])[0].bool()
    assert int(live.sum()) > 60
    edges = O.linspace_sym(0.0, 1.0, Nc + 1)
    dmin, dmax = _cpu(first.taps["depth_min"])[0][live], _cpu(first.taps["depth_max"])[0][live]
    cdepth = dmin[:, None] * (1 - edges) + dmax[:, None] * edges
    _, cw = O.ray_weights(_cpu(first.taps["coarse_density"])[0][live], cdepth)
    ws = O.smooth_weights(cw)
    p = (ws / ws.sum(-1, keepdim=True)).double()                              # (m, Nc)
    hist = torch.zeros_like(p)
    frac_sum, frac_n = 0.0, 0
    for s in range(S):
        kb = _cpu(ds.render(coord, Nc, Nf, None, seed=1000 + s, return_bins=True).taps["bins"])[0][live].double()
        j = torch.clamp((kb * Nc).floor().long(), 0, Nc - 1)
        hist.scatter_add_(1, j, torch.ones_like(kb))
        frac_sum += float((kb * Nc - j).sum()); frac_n += kb.numel()
    n_draw = S * Nf
    exp = n_draw * p
    ok = exp > 20
    z2 = ((hist - exp) ** 2 / (n_draw * p * (1 - p)))[ok]
    assert 0.85 < float(z2.mean()) < 1.15, float(z2.mean())
    assert abs(frac_sum / frac_n - 0.5) < 0.005
    # the oracle's own sampler on the same weights has the same statistics
    g = torch.Generator().manual_seed(0)
    hist_o = torch.zeros_like(p)
    for s in range(S):
        ob = O.draw_bins(ws, Nf, Nc, g).double()
        hist_o.scatter_add_(1, torch.clamp((ob * Nc).floor().long(), 0, Nc - 1), torch.ones_like(ob))
    z2o = ((hist_o - exp) ** 2 / (n_draw * p * (1 - p)))[ok]
    assert abs(float(z2.mean()) - float(z2o.mean())) < 0.15


@pytest.mark.parametrize("B,H,W", [(2, 32, 40), (1, 256, 256)])
def test_deformation_field_producer_vs_grid_sample(ops, B, H, W):
    """enarf_triplane_warp_fwd / _bwd (models/narf.py:40-58): the constant feature planes warped by a flow, written
    channel-last, against F.grid_sample (bilinear, zeros, align_corners False) and its autograd; flows of a few texels,
    some pointing out of the plane."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(H)
    planes = torch.randn(1, 96, H, W, generator=g)
    flow = 3.0 * torch.randn(B, 6, H, W, generator=g)
    flow[:, :, :2] -= 6.0                                  # rows that sample above the plane
    gout = torch.randn(B, 3, H, W, 32, generator=g)

    x, f = planes.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    gx = (torch.arange(W) + 0.5 + f[:, 0::2]) / (W / 2) - 1
    gy = (torch.arange(H)[:, None] + 0.5 + f[:, 1::2]) / (H / 2) - 1
    grid = torch.stack([gx, gy], dim=-1).reshape(B * 3, H, W, 2)
    src = x.reshape(1, 3, 32, H, W).expand(B, -1, -1, -1, -1).reshape(B * 3, 32, H, W)
    ref = F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    ref_cl = ref.reshape(B, 3, 32, H, W).permute(0, 1, 3, 4, 2)
    (ref_cl * gout).sum().backward()

    src_cl = ops.triplane_pack(planes.cuda())                         # (1, 3, H, W, 32)
    out = ops.triplane_warp_fwd(src_cl, flow.cuda())
    assert_close(_cpu(out), ref_cl.detach(), "warped planes", 1e-5)
    gs, gf = ops.triplane_warp_bwd(gout.cuda(), src_cl, flow.cuda())
    assert_close(_cpu(gs).permute(0, 3, 1, 2).reshape(1, 96, H, W), x.grad, "d planes", 1e-5)
    assert_close(_cpu(gf), f.grad, "d flow", 1e-5)
    only_flow = ops.triplane_warp_bwd(gout.cuda(), src_cl, flow.cuda(), need_src=False)
    assert only_flow[0] is None and torch.equal(only_flow[1], gf)
