"""CPU side of SURVEY.md 8(f) rank 4 (discriminator + background generator): the oracle's restatement of the two
un-vendored ops against independent definitions, the host logic of the HIP ops (sizes, adjoint pads), and the mirror
networks' state-dict layout against the imported reference's (tests/golden/gan2d_*.npz). No GPU compute here."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from enarf_gan_amd import _lib  # noqa: E402
from enarf_gan_amd.libraries.custom_stylegan2 import net, op  # noqa: E402
from enarf_gan_amd.libraries.gan import loss as gan_loss  # noqa: E402
from oracle import gan_ops_oracle as third  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
# (up, down, pad) as the networks use them, plus crops and a lopsided pad
UPFIRDN_CASES = [(1, 1, (2, 1)), (1, 1, (1, 1)), (2, 1, (2, 1)), (1, 2, (2, 2)), (1, 2, (1, 1)), (1, 1, (-1, 2)), (1, 1, (0, 0, 3, -1)),
                 (2, 1, (0, 0)), (1, 2, (0, 3, 1, 0))]


def _scipy_upfirdn2d(x, k1y, k1x, up, down, pads):
    """separable reference from scipy.signal.upfirdn (1-D upfirdn along each axis), then the pad / crop of the 2-D op:
    scipy's output is the FULL convolution of the zero-stuffed signal, i.e. padding (taps - 1) on both sides; positions that
    the 2-D op's own pads do not reach are cut, positions beyond are zeros"""
    from scipy.signal import upfirdn
    px0, px1, py0, py1 = pads

    def axis(a, taps, ax, p0, p1):
        n = a.shape[ax]
        full = upfirdn(taps, a, up=up, down=1, axis=ax)                  # length (n - 1) * up + len(taps)
        # the 2-D op sees n * up stuffed samples (zeros after the last one too), padded by p0 / p1, 'valid' correlation with the
        # flipped taps = full convolution cropped: output index j <-> full index j + (len(taps) - 1) - p0
        length = n * up + p0 + p1 - len(taps) + 1
        idx = np.arange(length) + (len(taps) - 1) - p0
        ok = (idx >= 0) & (idx < full.shape[ax])
        out = np.zeros([length if i == ax else s for i, s in enumerate(full.shape)], dtype=full.dtype)
        sl_out = [slice(None)] * full.ndim
        sl_in = [slice(None)] * full.ndim
        sl_out[ax] = np.nonzero(ok)[0]
        sl_in[ax] = idx[ok]
        out[tuple(sl_out)] = full[tuple(sl_in)]
        sl = [slice(None)] * full.ndim
        sl[ax] = slice(None, None, down)
        return out[tuple(sl)]
    return axis(axis(x, k1y, 2, py0, py1), k1x, 3, px0, px1)


@pytest.mark.parametrize("up,down,pad", UPFIRDN_CASES)
def test_oracle_upfirdn2d_matches_scipy(up, down, pad):
    rng = np.random.RandomState(0)
    x = rng.randn(2, 3, 9, 13)
    ky, kx = np.array([1.0, 3.0, 3.0, 1.0]), np.array([1.0, 2.0, -1.0])          # asymmetric: a flip error would show
    pads = pad if len(pad) == 4 else (pad[0], pad[1], pad[0], pad[1])
    want = _scipy_upfirdn2d(x, ky, kx, up, down, pads)
    got = third.upfirdn2d(torch.from_numpy(x), torch.from_numpy(np.outer(ky, kx)), up=up, down=down, pad=pad).numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("up,down,pad", UPFIRDN_CASES)
def test_out_size_and_adjoint_pads(up, down, pad):
    """the library's size rule equals the restatement's, and the adjoint pads give the transpose of the op (checked on the
    restatement, whose autograd is the truth): <A x, g> = <x, A^T g>"""
    lib = _lib.load()
    pads = pad if len(pad) == 4 else (pad[0], pad[1], pad[0], pad[1])
    H, W, kh, kw = 11, 14, 4, 3
    k = torch.randn(kh, kw, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    x = torch.randn(1, 2, H, W, dtype=torch.float64, generator=torch.Generator().manual_seed(2), requires_grad=True)
    y = third.upfirdn2d(x, k, up=up, down=down, pad=pad)
    OH, OW = y.shape[2:]
    assert OH == lib.enarf_upfirdn2d_out_size(H, kh, up, down, pads[2], pads[3])
    assert OW == lib.enarf_upfirdn2d_out_size(W, kw, up, down, pads[0], pads[1])
    g = torch.randn(y.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    (want,) = torch.autograd.grad(y, x, g)
    ap = op.adjoint_pads(H, W, OH, OW, kh, kw, up, down, pads)
    got = third.upfirdn2d(g, torch.flip(k, [0, 1]), up=down, down=up, pad=ap)
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=1e-12, atol=1e-12)
    # and the adjoint's adjoint is the op again (what the second derivative of R1 runs)
    ap2 = op.adjoint_pads(OH, OW, H, W, kh, kw, down, up, ap)
    again = third.upfirdn2d(x.detach(), k, up=up, down=down, pad=ap2)
    torch.testing.assert_close(again, y.detach(), rtol=1e-12, atol=1e-12)


def test_out_size_edge_cases():
    lib = _lib.load()
    assert lib.enarf_upfirdn2d_out_size(4, 4, 1, 1, 0, 0) == 1
    assert lib.enarf_upfirdn2d_out_size(3, 4, 1, 1, 0, 0) == 0          # filter longer than the padded input: empty
    assert lib.enarf_upfirdn2d_out_size(0, 4, 1, 1, 2, 2) == 0
    assert lib.enarf_upfirdn2d_out_size(128, 4, 1, 2, 2, 2) == 65
    assert lib.enarf_upfirdn2d_out_size(64, 4, 2, 1, 2, 1) == 128


def test_oracle_fused_leaky_relu():
    x = torch.randn(3, 5, 4, 4)
    b = torch.randn(5)
    want = F.leaky_relu(x + b.view(1, 5, 1, 1), 0.2) * 2 ** 0.5
    torch.testing.assert_close(third.fused_leaky_relu(x, b), want)
    torch.testing.assert_close(third.fused_leaky_relu(x[:, :, 0, 0], b), F.leaky_relu(x[:, :, 0, 0] + b, 0.2) * 2 ** 0.5)


def test_ops_refuse_host_tensors():
    with pytest.raises(_lib.EnarfHipError):
        op.fused_leaky_relu(torch.zeros(2, 3, 4, 4), torch.zeros(3))
    with pytest.raises(_lib.EnarfHipError):
        op.upfirdn2d(torch.zeros(1, 1, 8, 8), op.make_kernel([1, 3, 3, 1]), pad=(2, 1))


@pytest.mark.parametrize("name", ["gan2d_dis_32_std", "gan2d_dis_16_nostd"])
def test_discriminator_state_dict_is_the_reference_layout(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    dis = net.Discriminator(SimpleNamespace(minibatch_std=bool(g["minibatch_std"])), size=int(g["size"]))
    sd = dis.state_dict()
    assert sorted(sd.keys()) == list(g["keys"])
    assert [",".join(map(str, sd[k].shape)) for k in sorted(sd.keys())] == list(g["shapes"])


@pytest.mark.parametrize("name", ["gan2d_gen_32_crop", "gan2d_gen_16"])
def test_background_generator_state_dict_is_the_reference_layout(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    gen = net.Generator(size=int(g["size"]), style_dim=int(g["style_dim"]), n_mlp=4, last_channel=3,
                        crop_background=bool(g["crop_background"]))
    sd = gen.state_dict()
    assert sorted(sd.keys()) == list(g["keys"])
    assert [",".join(map(str, sd[k].shape)) for k in sorted(sd.keys())] == list(g["shapes"])
    assert gen.n_latent == int(g["n_latent"])
    # the blur filters are the reference's constants
    torch.testing.assert_close(sd["convs.0.conv.blur.kernel"], third.make_kernel([1, 3, 3, 1]) * 4)
    torch.testing.assert_close(sd["to_rgbs.0.upsample.kernel"], third.make_kernel([1, 3, 3, 1]) * 4)


def test_snapshot_round_trip_of_both_networks(tmp_path):
    """the reference's snapshot schema (train_ENARF_GAN.py:278-294) holds `gen` and `dis` state dicts: written by
    formats.save_snapshot from the mirror networks, they load back by name"""
    from enarf_gan_amd import formats
    dis = net.Discriminator(SimpleNamespace(minibatch_std=False), size=16)
    gen = net.Generator(16, 8, 4)
    path = tmp_path / "snapshot_latest.pth"
    formats.save_snapshot(path, gen, 7, discriminator=dis)
    snap = formats.read_snapshot(path)
    dis2 = net.Discriminator(SimpleNamespace(minibatch_std=False), size=16)
    dis2.load_state_dict(snap["dis"])
    for k, v in dis.state_dict().items():
        assert torch.equal(v, dis2.state_dict()[k])
    rep = formats.load_generator_snapshot(snap, net.Generator(16, 8, 4), strict=True)
    assert not rep.missing and not rep.ignored and rep.iteration == 7


def test_minibatch_stddev_feature():
    feat = torch.randn(8, 6, 4, 4)
    out = net.minibatch_stddev(feat, 4)
    assert out.shape == (8, 7, 4, 4)
    # samples i and i + 2 share a group (view(group, -1, ...)): one statistic per group, constant over the map
    y = feat.view(4, 2, 6, 4, 4)
    want = torch.sqrt(y.var(0, unbiased=False) + 1e-8).mean([1, 2, 3])
    for i in range(8):
        torch.testing.assert_close(out[i, 6], want[i % 2].expand(4, 4))
    torch.testing.assert_close(out[:, :6], feat)


def test_losses():
    real, fake = torch.tensor([0.5, -2.0, 3.0]), torch.tensor([-0.5, 2.0])
    assert float(gan_loss.adv_loss_dis(real, fake, "hinge")) == pytest.approx((0.5 + 3.0 + 0.0) / 3 + (0.5 + 3.0) / 2)
    assert float(gan_loss.adv_loss_gen(fake, "hinge")) == pytest.approx(-0.75)
    assert float(gan_loss.adv_loss_dis(real, fake, "ce", 2.0)) == pytest.approx(
        float(F.softplus(-2 * real).mean() + F.softplus(2 * fake).mean()))
    assert float(gan_loss.adv_loss_gen(fake, "ce")) == pytest.approx(float(F.softplus(-fake).mean()))
    with pytest.raises(AssertionError):
        gan_loss.adv_loss_gen(fake, "wgan")
    x = torch.randn(4, 3, 5, 5, requires_grad=True)
    w = torch.randn(3, 5, 5)
    pred = (x * x * w).sum([1, 2, 3])                      # d pred / d x = 2 x w
    want = (2 * x * w).pow(2).flatten(1).sum(1).mean()
    torch.testing.assert_close(gan_loss.d_r1_loss(pred, x), want)


def test_random_window_is_an_integer_crop():
    img = torch.arange(2 * 1 * 3 * 12, dtype=torch.float32).view(2, 1, 3, 12)
    torch.manual_seed(0)
    out = net._random_window(img, 6)
    assert out.shape == (2, 1, 3, 6)
    for b in range(2):
        x0 = int(out[b, 0, 0, 0] - img[b, 0, 0, 0])
        assert 0 <= x0 <= 6 and torch.equal(out[b], img[b, :, :, x0:x0 + 6])


# ------------------------------------------------------------------------------------------ tri-plane producer (8f rank 2)
def test_triplane_generator_structure():
    """prepare_triplane_generator (libraries/triplane/triplane_nerf.py:17-29): 256 x 256 planes, widths min(32768 / res, 512),
    8 mapping layers with the bone-length embedding, 14 style vectors, the published module names"""
    from enarf_gan_amd.libraries.stylegan2_ada.networks import prepare_triplane_generator
    g = prepare_triplane_generator(512, 512, (32 + 23) * 3, c_dim=23 * 8)
    assert g.num_ws == 14 and g.synthesis.block_resolutions == [4, 8, 16, 32, 64, 128, 256]
    sd = g.state_dict()
    assert sd["synthesis.b4.const"].shape == (512, 4, 4)
    assert sd["synthesis.b64.conv1.weight"].shape == (512, 512, 3, 3)
    assert sd["synthesis.b128.conv0.weight"].shape == (256, 512, 3, 3) and sd["synthesis.b256.conv0.weight"].shape == (128, 256, 3, 3)
    assert sd["synthesis.b256.torgb.weight"].shape == (165, 128, 1, 1) and sd["synthesis.b256.torgb.affine.weight"].shape == (128, 512)
    assert sd["mapping.embed.weight"].shape == (512, 184) and sd["mapping.fc0.weight"].shape == (512, 1024)
    assert sd["mapping.fc7.weight"].shape == (512, 512) and sd["mapping.w_avg"].shape == (512,)
    assert not any("noise" in k for k in sd)                                     # use_noise=False
    assert "synthesis.b4.conv0.weight" not in sd and "synthesis.b8.conv0.resample_filter" in sd
    torch.testing.assert_close(sd["synthesis.b8.resample_filter"], third.make_kernel([1, 3, 3, 1]))
    # affine biases start at 1, mapping weights are stored divided by the learning-rate multiplier 0.01
    assert float(sd["synthesis.b8.conv0.affine.bias"].min()) == 1.0 and 50 < float(sd["mapping.fc3.weight"].std()) < 200


def test_model_builds_the_reference_producers_and_accepts_callables():
    from test_host_cpu import _nerf_cfg
    from enarf_gan_amd import synth
    from enarf_gan_amd.libraries.stylegan2_ada.networks import Generator
    from enarf_gan_amd.models.narf import TriPlaneNARF
    parents = synth.SMPL_PARENTS if hasattr(synth, "SMPL_PARENTS") else synth.make_scene(8, 1, "center_fixed", 4)["parents"]
    m = TriPlaneNARF(_nerf_cfg(constant_triplane=False), [64, 32], 24, parent=parents, num_bone_param=23)
    assert isinstance(m.tri_plane_gen, Generator) and m.tri_plane_gen.img_channels == (32 + 23) * 3 and m.tri_plane_gen.c_dim == 23 * 8
    assert "tri_plane_gen.synthesis.b4.const" in m.state_dict() and m.generator is None and m.flow_generator is None
    m.tri_plane_gen = lambda z, enc, truncation_psi=1: torch.zeros(z.shape[0], 165, 256, 256)       # any callable takes the slot
    assert not any(k.startswith("tri_plane_gen.") for k in m.state_dict())
    assert m.compute_tri_plane_feature(torch.zeros(2, 64), torch.ones(2, 23, 1)).shape == (2, 165, 256, 256)
    m.tri_plane_gen = Generator(64, 184, 512, 256, 165)                                             # ... and a network again
    assert "tri_plane_gen.mapping.w_avg" in m.state_dict()
    t = TriPlaneNARF(_nerf_cfg(constant_triplane=False, constant_trimask=True), [64, 32], 24, parent=parents, num_bone_param=23)
    assert isinstance(t.generator, Generator) and t.generator.img_channels == 96 and "generator.synthesis.b256.torgb.bias" in t.state_dict()
    f = TriPlaneNARF(_nerf_cfg(constant_triplane=False, deformation_field=True), [64, 32], 24, parent=parents, num_bone_param=23)
    assert isinstance(f.flow_generator, Generator) and f.flow_generator.img_channels == 6
    c = TriPlaneNARF(_nerf_cfg(), [64, 32], 24, parent=parents, num_bone_param=23)
    assert not any(k.startswith(("tri_plane_gen.", "generator.", "flow_generator.")) for k in c.state_dict())
