"""GPU tests of SURVEY.md 8(f) rank 4: the HIP ops under the 2-D GAN networks against a plain PyTorch fp32 / the oracle's
restatement (forward, backward, second derivative), the mirror discriminator and background generator against the
imported reference's outputs (tests/golden/gan2d_*.npz), and one whole GAN step through renderer + discriminator."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _helpers import Scene
from oracle import gan_ops_oracle as third
from test_host_cpu import Cfg, _nerf_cfg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# ------------------------------------------------------------------------------------------------ bias + leaky ReLU
@pytest.mark.parametrize("shape", [(4, 16, 32, 32), (3, 7, 5, 5), (2, 5, 9), (6, 33), (1, 512, 4, 4), (2, 3, 1, 1030)])
def test_fused_leaky_relu_forward_backward_and_second_derivative(shape):
    from enarf_gan_amd.libraries.custom_stylegan2 import op
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g)
    b = torch.randn(shape[1], generator=g)
    gy = torch.randn(shape, generator=g)
    ggx = torch.randn(shape, generator=g)

    def run(fn, dev):
        xx, bb = x.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        y = fn(xx, bb, 0.2, 2 ** 0.5)
        gx, gb = torch.autograd.grad(y, [xx, bb], gy.to(dev), create_graph=True)
        # second derivative: d <gx, ggx> / d (grad_output) is what R1's backward needs; take it w.r.t. an upstream scale
        s = torch.ones((), device=dev, requires_grad=True)
        y2 = fn(xx * s, bb, 0.2, 2 ** 0.5)
        (g1,) = torch.autograd.grad(y2, xx, gy.to(dev), create_graph=True)
        (g2,) = torch.autograd.grad((g1 * ggx.to(dev)).sum(), s)
        return y.detach().cpu(), gx.detach().cpu(), gb.detach().cpu(), g2.cpu()

    want = run(third.fused_leaky_relu, "cpu")
    got = run(op.fused_leaky_relu, "cuda")
    assert torch.equal(got[0], want[0].float()) or _rel(got[0], want[0]) < 1e-6
    assert _rel(got[1], want[1]) < 1e-6
    assert _rel(got[2], want[2]) < 1e-5          # a sum over outer x inner terms in another order
    assert _rel(got[3], want[3]) < 1e-5


def test_fused_leaky_relu_module_and_no_bias():
    from enarf_gan_amd.libraries.custom_stylegan2 import op
    m = op.FusedLeakyReLU(8).cuda()
    with torch.no_grad():
        m.bias.copy_(torch.linspace(-1, 1, 8))
    x = torch.randn(2, 8, 6, 6, device="cuda")
    torch.testing.assert_close(m(x), F.leaky_relu(x + m.bias.view(1, 8, 1, 1), 0.2) * 2 ** 0.5)
    torch.testing.assert_close(op.fused_leaky_relu(x), F.leaky_relu(x, 0.2) * 2 ** 0.5)
    # a non-contiguous input is taken as it is meant (the wrapper makes it dense)
    xt = x.permute(0, 1, 3, 2)
    torch.testing.assert_close(op.fused_leaky_relu(xt, m.bias), F.leaky_relu(xt + m.bias.view(1, 8, 1, 1), 0.2) * 2 ** 0.5)


# ------------------------------------------------------------------------------------------------------ upfirdn2d
UPFIRDN_CASES = [(1, 1, (2, 1)), (1, 1, (1, 1)), (2, 1, (2, 1)), (1, 2, (2, 2)), (1, 2, (1, 1)), (1, 1, (-1, 2)), (1, 1, (0, 0, 3, -1)),
                 (2, 1, (0, 0)), (1, 2, (0, 3, 1, 0)), (2, 1, (3, 3)), (1, 2, (5, 4)), (1, 1, (2, 2))]          # the last: 129 out of 128


@pytest.mark.parametrize("up,down,pad", UPFIRDN_CASES)
@pytest.mark.parametrize("hw", [(9, 13), (64, 64), (33, 130), (128, 128)])
@pytest.mark.parametrize("taps_x", [(1.0, 2.0, -1.0), (1.0, 2.0, -1.0, 0.5)])          # 4 x 3: the generic loop; 4 x 4: the unrolled one
def test_upfirdn2d_matches_restatement(up, down, pad, hw, taps_x):
    from enarf_gan_amd.libraries.custom_stylegan2 import op
    H, W = hw
    g = torch.Generator().manual_seed(H * 1000 + W + up * 7 + down)
    x = torch.randn(2, 3, H, W, generator=g)
    k = torch.outer(torch.tensor([1.0, 3.0, 3.0, 1.0]), torch.tensor(taps_x)) / 7.0          # asymmetric: a flip error would show
    want = third.upfirdn2d(x.double(), k.double(), up=up, down=down, pad=pad)
    got = op.upfirdn2d(x.cuda(), k.cuda(), up=up, down=down, pad=pad)
    assert got.shape == want.shape
    assert _rel(got, want) < 2e-6


@pytest.mark.parametrize("up,down,pad", [(1, 1, (2, 1)), (2, 1, (2, 1)), (1, 2, (2, 2)), (1, 1, (-1, 2)), (1, 2, (1, 1))])
def test_upfirdn2d_backward_and_second_derivative(up, down, pad):
    from enarf_gan_amd.libraries.custom_stylegan2 import op
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 2, 21, 18, generator=g)
    k = op.make_kernel([1, 3, 3, 1]) * (up ** 2)

    def run(fn, dev, dt):
        xx = x.to(dev, dt).requires_grad_(True)
        y = fn(xx, k.to(dev, dt), up=up, down=down, pad=pad)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(dev, dt)
        # a non-linear function of the op's output, so that the second derivative is not zero
        (gx,) = torch.autograd.grad((y * y * gy).sum(), xx, create_graph=True)
        (ggx,) = torch.autograd.grad(gx.pow(2).sum(), xx)
        return y.detach().cpu(), gx.detach().cpu(), ggx.cpu()

    want = run(third.upfirdn2d, "cpu", torch.float64)
    got = run(op.upfirdn2d, "cuda", torch.float32)
    for a, b, tol in zip(got, want, (2e-6, 1e-5, 1e-4)):
        assert a.shape == b.shape and _rel(a, b) < tol


def test_upfirdn2d_many_planes_and_modules():
    """more planes than one grid axis holds (the launch strides over them), and the Blur / Upsample modules"""
    from enarf_gan_amd.libraries.custom_stylegan2 import op
    x = torch.randn(70000, 1, 8, 8, generator=torch.Generator().manual_seed(1))
    k = op.make_kernel([1, 3, 3, 1])
    got = op.upfirdn2d(x.cuda(), k.cuda(), pad=(2, 1))
    want = third.upfirdn2d(x, k, pad=(2, 1))
    assert _rel(got, want) < 2e-6
    y = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(2))
    assert _rel(op.Blur([1, 3, 3, 1], pad=(2, 2)).cuda()(y.cuda()), third.Blur([1, 3, 3, 1], pad=(2, 2))(y)) < 2e-6
    up = op.Upsample([1, 3, 3, 1]).cuda()(y.cuda())
    assert up.shape == (2, 4, 32, 32) and _rel(up, third.Upsample([1, 3, 3, 1])(y)) < 2e-6
    # a constant image stays constant under the up-sampler (unit DC gain after the factor^2 scaling), away from the border
    c = op.Upsample([1, 3, 3, 1]).cuda()(torch.full((1, 1, 8, 8), 3.0, device="cuda"))
    torch.testing.assert_close(c[:, :, 2:-2, 2:-2], torch.full((1, 1, 12, 12), 3.0, device="cuda"))
    with pytest.raises(Exception):
        op.upfirdn2d(y.cuda(), k.cuda(), up=2, down=2)


# ---------------------------------------------------------------------------------------- networks vs the reference
@pytest.mark.parametrize("name", ["gan2d_dis_32_std", "gan2d_dis_16_nostd"])
def test_discriminator_matches_reference_golden(name):
    from enarf_gan_amd.libraries.custom_stylegan2 import net
    from enarf_gan_amd.libraries.gan.loss import d_r1_loss
    g = np.load(os.path.join(GOLD, name + ".npz"))
    dis = net.Discriminator(SimpleNamespace(minibatch_std=bool(g["minibatch_std"])), size=int(g["size"]))
    third.fill_by_name(dis)                         # the weights of the fixture: a function of the state-dict keys
    dis = dis.cuda()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    out = dis(x)
    assert out.shape == g["out"].shape
    assert _rel(out.detach(), g["out"]) < 2e-4
    r1 = d_r1_loss(out, x)
    (gx,) = torch.autograd.grad(out.sum(), x, retain_graph=True)
    assert _rel(gx, g["grad_x"]) < 2e-4
    assert abs(float(r1) - float(g["r1"])) < 2e-4 * abs(float(g["r1"]))
    g_lin, g_first = torch.autograd.grad(r1, [dis.final_linear[1].weight, dis.convs[0][0].weight])
    assert _rel(g_lin, g["r1_grad_final_linear_1_weight"]) < 5e-4          # second derivatives through every HIP op
    assert _rel(g_first, g["r1_grad_convs_0_0_weight"]) < 5e-4


@pytest.mark.parametrize("name", ["gan2d_gen_32_crop", "gan2d_gen_16"])
def test_background_generator_matches_reference_golden(name):
    from enarf_gan_amd.libraries.custom_stylegan2 import net
    g = np.load(os.path.join(GOLD, name + ".npz"))
    gen = net.Generator(size=int(g["size"]), style_dim=int(g["style_dim"]), n_mlp=4, last_channel=3,
                        crop_background=bool(g["crop_background"]))
    third.fill_by_name(gen)
    gen = gen.cuda().eval()
    noise = [torch.from_numpy(g[f"noise_{i}"]).cuda() for i in range(gen.num_layers)]
    z_bg, z_r = torch.from_numpy(g["z_bg"]).cuda(), torch.from_numpy(g["z_render"]).cuda()
    with torch.no_grad():
        img, none = gen([z_bg, z_r], inject_index=gen.n_latent - 4, noise=noise)
        one, lat = gen([z_bg], return_latents=True, noise=noise)
    assert none is None and img.shape == g["image"].shape
    assert _rel(img, g["image"]) < 2e-4
    assert _rel(one, g["image_one_style"]) < 2e-4
    assert _rel(lat, g["latent_one_style"]) < 1e-5
    # training mode of the cropped variant: a random window of the same wide image
    if bool(g["crop_background"]):
        gen.train()
        with torch.no_grad():
            win, _ = gen([z_bg, z_r], inject_index=gen.n_latent - 4, noise=noise)
        assert win.shape == img.shape and bool(torch.isfinite(win).all())


# ----------------------------------------------------------------------------------------------- the GAN loop closed
def test_one_gan_step_through_renderer_background_and_discriminator():
    """train_ENARF_GAN.py:102-170 in small: generator forward (HIP renderer + background generator), generator loss through
    the discriminator, backward to the renderer's parameters and the background generator's; discriminator loss and the R1
    step. Checks the composite (generator.py:107) and that every parameter group receives a finite, non-zero gradient."""
    from enarf_gan_amd.libraries.custom_stylegan2 import net
    from enarf_gan_amd.libraries.gan.loss import adv_loss_dis, adv_loss_gen, d_r1_loss
    from enarf_gan_amd.models.generator import TriNARFGenerator
    S, B, Nc, Nf, zd = 32, 2, 24, 32, 32
    sc = Scene(S, B, "center_fixed", zd)
    torch.manual_seed(0)
    gen = TriNARFGenerator(Cfg(z_dim=zd, background_ratio=0.7, crop_background=True, pretrained_background=False,
                               nerf_params=_nerf_cfg(Nc=Nc, Nf=Nf, constant_triplane=False)), S, 24, sc.raw["parents"], 23)
    gen.register_canonical_pose(sc.raw["canonical_pose"])
    assert isinstance(gen.background_generator, net.Generator) and gen.background_generator.crop_background
    gen = gen.cuda().train()
    tri = sc.raw["tri_plane"].cuda().requires_grad_(True)            # stands in for the un-vendored tri-plane synthesis net
    gen.nerf.tri_plane_gen = lambda z, enc, truncation_psi=1: tri
    dis = net.Discriminator(SimpleNamespace(minibatch_std=True), size=S).cuda()
    s = sc.raw
    z = torch.randn(B, 4 * zd, device="cuda")
    # generator step
    dis.requires_grad_(False)
    gen.eval()                 # the centre window of the background and, with the same seed, the same importance samples
    with torch.no_grad():
        torch.manual_seed(1)
        fg, fg_mask, bg = gen(s["pose_to_camera"].cuda(), None, s["bone_length"].cuda(), z, s["inv_intrinsics"].cuda(), return_bg=True)
        torch.manual_seed(1)
        whole, mask_e, _, _ = gen(s["pose_to_camera"].cuda(), None, s["bone_length"].cuda(), z, s["inv_intrinsics"].cuda())
    assert bg.shape == (B, 3, S, S) and torch.equal(mask_e, fg_mask)
    torch.testing.assert_close(whole, fg + (1 - fg_mask[:, None]) * bg)              # generator.py:107
    gen.train()
    fake, mask, _, _ = gen(s["pose_to_camera"].cuda(), None, s["bone_length"].cuda(), z, s["inv_intrinsics"].cuda())
    assert fake.shape == (B, 3, S, S) and float(mask.max()) > 0.2
    loss_g = adv_loss_gen(dis(fake), "ce")
    loss_g.backward()
    groups = {"tri-plane": tri.grad, "StyledMLP": gen.nerf.mlp.layers[0].conv.weight.grad,
              "background": gen.background_generator.convs[0].conv.weight.grad,
              "background style": gen.background_generator.style[1].weight.grad}
    for name, gr in groups.items():
        assert gr is not None and bool(torch.isfinite(gr).all()) and float(gr.abs().max()) > 0, name
    assert all(p.grad is None for p in dis.parameters())
    # discriminator step + R1
    dis.requires_grad_(True)
    real = torch.randn(B, 3, S, S, device="cuda").requires_grad_(True)
    loss_d = adv_loss_dis(dis(real), dis(fake.detach()), "ce")
    loss_d.backward()
    r1 = d_r1_loss(dis(real), real)
    (0.5 * r1 * 16 * 10).backward()
    for n_, p in dis.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n_
    assert float(dis.convs[1].conv2[1].weight.grad.abs().max()) > 0


def test_gan_iteration_tool_runs_and_reports():
    """tools/bench_gan_step.py (BASELINE configs[2] in small): two timed iterations of generator + discriminator + R1 with two
    micro-batches, in a process of its own; the line it prints carries the phases and a finite image statistic"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_gan_step.py"), "--size", "32", "--batch", "4", "--accum", "2",
                        "--nc", "16", "--nf", "16", "--steps", "2", "--warmup", "1", "--producer", "planes"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["unit"] == "it/s" and line["value"] > 0 and line["n_gpus"] == 1
    assert set(line["phases_ms_mean_rank0"]) == {"generator forward + backward (+ exchange)", "discriminator step", "R1 step"}
    assert 0 < line["fake_image_abs_mean"] < 10


def test_mask_regularisers_match_their_formulas():
    from enarf_gan_amd.models.loss import nerf_bone_loss, nerf_patch_loss, push_to_background
    g = torch.Generator().manual_seed(0)
    mask = torch.rand(2, 16, 16, generator=g).cuda()
    bone = (torch.rand(2, 32, 32, generator=g) > 0.9).float().cuda()
    k = int(mask.numel() * 0.3)
    want_bg = mask.flatten().sort()[0][:k].square().mean()
    torch.testing.assert_close(push_to_background(mask, 0.3), want_bg)
    assert push_to_background(mask, 0.0) == 0
    pooled = F.max_pool2d(bone[:, None], 2, 2)[:, 0] > 0.5
    want_bone = ((1 - mask).square() * pooled).sum() / pooled.sum()
    torch.testing.assert_close(nerf_bone_loss(mask, bone), want_bone)
    torch.testing.assert_close(nerf_patch_loss(mask, bone, 0.3, coef=10), (want_bg + want_bone) * 10)


def test_ops_edge_cases():
    """empty batches, maps smaller than the filter, a single pixel, and arguments the library refuses"""
    from enarf_gan_amd.libraries.custom_stylegan2 import op
    k = op.make_kernel([1, 3, 3, 1]).cuda()
    # a 1 x 1 and a 2 x 3 map under the blur (padding makes room for the 4 x 4 filter)
    for h, w in ((1, 1), (2, 3), (5, 1)):
        x = torch.randn(3, 2, h, w, generator=torch.Generator().manual_seed(h * 10 + w))
        for kw in (dict(pad=(2, 1)), dict(up=2, pad=(2, 1)), dict(down=2, pad=(2, 2))):
            want = third.upfirdn2d(x, k.cpu() * (4 if "up" in kw else 1), **kw)
            got = op.upfirdn2d(x.cuda(), k * (4 if "up" in kw else 1), **kw)
            assert got.shape == want.shape and _rel(got, want) < 2e-6, (h, w, kw)
    # no room for the filter: an error, as an empty convolution output would be
    with pytest.raises(ValueError):
        op.upfirdn2d(torch.zeros(1, 1, 2, 2, device="cuda"), k, pad=(0, 0))
    # empty batch: nothing launched, shapes kept
    e = op.upfirdn2d(torch.zeros(0, 4, 8, 8, device="cuda"), k, pad=(2, 1))
    assert e.shape == (0, 4, 8, 8)
    assert op.fused_leaky_relu(torch.zeros(0, 4, 8, 8, device="cuda"), torch.zeros(4, device="cuda")).shape == (0, 4, 8, 8)
    # a 9-tap filter and a bias of the wrong length are refused, with a message
    with pytest.raises(NotImplementedError, match="filter"):          # ENARF_ERR_UNSUPPORTED
        op.upfirdn2d(torch.zeros(1, 1, 16, 16, device="cuda"), torch.ones(9, 9, device="cuda"), pad=(4, 4))
    with pytest.raises(ValueError, match="bias"):
        op.fused_leaky_relu(torch.zeros(2, 4, 3, 3, device="cuda"), torch.zeros(5, device="cuda"))
    # a 1-D filter is the separable outer product, as make_kernel builds it
    y = torch.randn(1, 2, 12, 12, device="cuda")
    torch.testing.assert_close(op.upfirdn2d(y, torch.tensor([1.0, 3.0, 3.0, 1.0], device="cuda") / 8, pad=(2, 1)),
                               op.upfirdn2d(y, k, pad=(2, 1)))


# ------------------------------------------------------------------------------------ tri-plane producer (8f rank 2)
def _published_generator_forward(g, z, c, truncation_psi=1.0):
    """The same network written the other published way, in plain PyTorch on the CPU: per-sample weights
    w'_{b,o,i,k} = w_{o,i,k} s_{b,i} [normalised per (b, o)] in ONE grouped convolution (StyleGAN2 sec. 2.2), bias and leaky
    ReLU as separate ops, up-sampling by explicit zero-stuffing + padding + correlation with the flipped filter
    (oracle/gan_ops_oracle.upfirdn2d). Shares nothing with libraries/stylegan2_ada/networks.py but the state dict."""
    sd = {k: v.detach().double().cpu() for k, v in g.state_dict().items()}
    z, c = z.double().cpu(), c.double().cpu()

    def fc(x, name, act, lr, in_f):
        y = x @ (sd[name + ".weight"] * (lr / in_f ** 0.5)).t() + sd[name + ".bias"] * lr
        return F.leaky_relu(y, 0.2) * 2 ** 0.5 if act else y

    def nrm(x):
        return x * (x.square().mean(1, keepdim=True) + 1e-8).rsqrt()
    x = torch.cat([nrm(z), nrm(fc(c, "mapping.embed", False, 1.0, c.shape[1]))], 1)
    for i in range(g.mapping.num_layers):
        x = fc(x, f"mapping.fc{i}", True, 0.01, x.shape[1])
    w = sd["mapping.w_avg"].lerp(x, truncation_psi) if truncation_psi != 1 else x
    filt = sd["synthesis.b4.resample_filter"]

    def modconv(x, name, wlat, up, demod, gain_styles=1.0):
        wt = sd[name + ".weight"]
        s = (wlat @ (sd[name + ".affine.weight"] / wlat.shape[1] ** 0.5).t() + sd[name + ".affine.bias"]) * gain_styles
        ws = wt[None] * s[:, None, :, None, None]                                   # (b, out, in, k, k)
        if demod:
            ws = ws * (ws.square().sum([2, 3, 4], keepdim=True) + 1e-8).rsqrt()
        b, cin, h, wd = x.shape
        k = wt.shape[-1]
        if up == 1:
            y = F.conv2d(x.reshape(1, b * cin, h, wd), ws.reshape(-1, cin, k, k), padding=k // 2, groups=b)
        else:
            wtr = ws.transpose(1, 2).reshape(b * cin, -1, k, k)
            y = F.conv_transpose2d(x.reshape(1, b * cin, h, wd), wtr, stride=2, groups=b)
            y = y.reshape(b, -1, y.shape[2], y.shape[3])
            y = third.upfirdn2d(y, filt * 4, pad=(1, 1))
        return y.reshape(b, -1, y.shape[2], y.shape[3])

    def layer(x, name, wlat, up):
        y = modconv(x, name, wlat, up, True) + sd[name + ".bias"].view(1, -1, 1, 1)
        return F.leaky_relu(y, 0.2) * 2 ** 0.5
    img, x, i = None, None, 0
    for res in g.synthesis.block_resolutions:
        p = f"synthesis.b{res}"
        if res == 4:
            x = sd[p + ".const"][None].expand(z.shape[0], -1, -1, -1)
        else:
            x = layer(x, p + ".conv0", w, 2)
        x = layer(x, p + ".conv1", w, 1)
        if img is not None:
            img = third.upfirdn2d(img, filt * 4, up=2, pad=(2, 1))
        tw = sd[p + ".torgb.weight"]
        y = modconv(x, p + ".torgb", w, 1, False, gain_styles=1 / (tw.shape[1] * tw.shape[2] ** 2) ** 0.5) + sd[p + ".torgb.bias"].view(1, -1, 1, 1)
        img = y if img is None else img + y
    return img


def test_triplane_generator_matches_the_published_formulation():
    from enarf_gan_amd.libraries.stylegan2_ada.networks import Generator
    torch.manual_seed(3)
    g = Generator(16, 6, 32, 32, 9, mapping_kwargs=dict(num_layers=3), synthesis_kwargs=dict(channel_base=512, channel_max=24)).eval()
    with torch.no_grad():                      # trained-looking values: non-zero biases and a non-zero average latent
        for n_, p_ in g.named_parameters():
            if n_.endswith("bias") and "affine" not in n_:
                p_.normal_(0, 0.3)
        g.mapping.w_avg.normal_()
    z, c = torch.randn(3, 16), torch.randn(3, 6)
    gc = g.cuda()
    for psi in (1.0, 0.6):
        with torch.no_grad():
            got = gc(z.cuda(), c.cuda(), truncation_psi=psi)
        want = _published_generator_forward(g, z, c, psi)
        assert got.shape == want.shape == (3, 9, 32, 32)
        assert _rel(got, want) < 5e-5, psi
    # truncation to zero: every sample is the average latent's image
    with torch.no_grad():
        flat = gc(z.cuda(), c.cuda(), truncation_psi=0.0)
    torch.testing.assert_close(flat[0], flat[2])
    # training mode moves the average latent (beta 0.995), evaluation does not
    before = gc.mapping.w_avg.clone()
    gc.train()
    gc(z.cuda(), c.cuda()).sum().backward()
    assert not torch.equal(gc.mapping.w_avg, before) and float(gc.synthesis.b4.const.grad.abs().max()) > 0
    gc.eval()
    mid = gc.mapping.w_avg.clone()
    with torch.no_grad():
        gc(z.cuda(), c.cuda())
    assert torch.equal(gc.mapping.w_avg, mid)


def test_gan_generator_end_to_end_with_its_own_producer():
    """TriNARFGenerator as the reference builds it (models/generator.py:14-38): z -> StyleGAN2-ADA tri-planes conditioned on
    the bone lengths -> HIP renderer -> composite over the StyleGAN2 background; one backward through all of it"""
    from enarf_gan_amd.libraries.stylegan2_ada.networks import Generator
    from enarf_gan_amd.models.generator import TriNARFGenerator
    S, B, zd = 32, 2, 32
    sc = Scene(S, B, "center_fixed", zd)
    torch.manual_seed(0)
    gen = TriNARFGenerator(Cfg(z_dim=zd, background_ratio=0.7, crop_background=True, pretrained_background=False,
                               nerf_params=_nerf_cfg(Nc=16, Nf=16, constant_triplane=False)), S, 24, sc.raw["parents"], 23)
    gen.register_canonical_pose(sc.raw["canonical_pose"])
    assert isinstance(gen.nerf.tri_plane_gen, Generator)
    gen = gen.cuda().train()
    s = sc.raw
    z = torch.randn(B, 4 * zd, device="cuda")
    img, mask, fw, fd = gen(s["pose_to_camera"].cuda(), None, s["bone_length"].cuda(), z, s["inv_intrinsics"].cuda())
    tri = gen.nerf.buffers_tensors["tri_plane_feature"]
    assert tri.shape == (B, (32 + 23) * 3, 256, 256) and img.shape == (B, 3, S, S) and bool(torch.isfinite(img).all())
    (img.square().mean() + mask.mean()).backward()
    for name in ("nerf.tri_plane_gen.synthesis.b256.torgb.weight", "nerf.tri_plane_gen.mapping.fc0.weight", "nerf.tri_plane_gen.mapping.embed.weight",
                 "nerf.mlp.layers.0.conv.weight", "background_generator.convs.0.conv.weight"):
        gr = dict(gen.named_parameters())[name].grad
        assert gr is not None and bool(torch.isfinite(gr).all()), name
    assert float(dict(gen.named_parameters())["nerf.tri_plane_gen.synthesis.b256.torgb.weight"].grad.abs().max()) > 0


def test_discriminator_runs_under_bf16_autocast():
    """an opt-in the reference does not have: torch.autocast around the 2-D networks. The library convolutions run in bf16, the HIP
    ops take whatever arrives and compute in fp32; outputs and gradients stay close to the fp32 run"""
    from enarf_gan_amd.libraries.custom_stylegan2 import net
    dis = net.Discriminator(SimpleNamespace(minibatch_std=True), size=32)
    third.fill_by_name(dis)
    dis = dis.cuda()
    x = torch.randn(4, 3, 32, 32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)).requires_grad_(True)
    ref = dis(x)
    (g_ref,) = torch.autograd.grad(ref.sum(), x)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = dis(x)
    (g_amp,) = torch.autograd.grad(out.float().sum(), x)
    assert bool(torch.isfinite(out).all()) and _rel(out.float(), ref) < 0.1 and _rel(g_amp, g_ref) < 0.3          # bf16: 8 bits
