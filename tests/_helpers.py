"""Shared set-up of the parity tests: one synthetic scene -> oracle inputs (CPU) and device inputs (GPU)."""
import os

import numpy as np
import torch

from enarf_gan_amd import synth
from oracle import enarf_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def fullframe_case(name):
    """A full-frame fixture of make_golden.run_fullframe_case -> (g, ray_validity (B, n) bool, bins (B, n, Nf) float32).

    bins are rebuilt exactly as the reference formed them under the fixture's deterministic sampler: idx / Nc + 0.5 / Nc
    in float32 (rendering.py:192-194); rays the reference dropped (B == 1, no cube hit) get 0.5 - they are never marched."""
    g = load_golden(name)
    B, n, Nc, Nf = int(g["batch"]), int(g["size"]) ** 2, int(g["Nc"]), int(g["Nf"])
    rv = np.unpackbits(g["ray_validity"], axis=1)[:, :n].astype(bool)
    idx = torch.from_numpy(g["bin_idx"].astype(np.int64))
    rows = idx.float() / Nc + torch.full(idx.shape, 0.5) / Nc
    if B == 1:
        bins = torch.full((1, n, Nf), 0.5)
        bins[0, torch.from_numpy(rv[0])] = rows
    else:
        bins = rows.reshape(B, n, Nf)
    return g, rv, bins


def u8_mismatches(ours_f32, ref_f32):
    """Pixels whose uint8 quantisation (x * 255).astype(uint8) (ENARF_GAN_demo.py:79) differs: [(flat index, ours, ref)]."""
    a, b = np.asarray(ours_f32, dtype=np.float32).reshape(-1), np.asarray(ref_f32, dtype=np.float32).reshape(-1)
    qa, qb = (a * 255).astype(np.uint8), (b * 255).astype(np.uint8)
    bad = np.nonzero(qa != qb)[0]
    return [(int(i), float(a[i]), float(b[i])) for i in bad]


def assert_u8_mask_matches(ours_f32, ref_f32, what, max_float_gap=1e-5, max_frac=2e-3):
    """The integer foreground mask, (mask * 255).astype(uint8) (ENARF_GAN_demo.py:79), against the reference's: equal on
    EVERY pixel except true straddlers of a quantisation step - returned as [(index, ours, reference)] - i.e. pixels whose
    two float masks differ by less than `max_float_gap` (10x tighter than the 1e-4 parity bound) and truncate to ADJACENT
    integers. Two kinds:
      * the saturation step 254 | 255: an opaque pixel's mask is 1 - T_end with T_end ~ 1e-8, i.e. 1.0 to the last bit or
        two, and the truncating quantiser steps exactly at 1.0; which side a pixel lands on is decided by the rounding of
        a 63-term float sum (the reference against itself on another BLAS or thread count flips them as well). Any
        number of these is accepted, each within 3 ulp of 1.0 on both sides;
      * everywhere else at most `max_frac` of the pixels (with errors ~1e-6 and 255 steps that is the expected count)."""
    bad = u8_mismatches(ours_f32, ref_f32)
    away = []
    for i, a, b in bad:
        qa, qb = int(np.float32(a) * np.float32(255)), int(np.float32(b) * np.float32(255))
        assert abs(a - b) <= max_float_gap and abs(qa - qb) == 1, \
            f"{what}: pixel {i} differs in the integer mask and is not a straddler: ours {a!r} reference {b!r}"
        if {qa, qb} == {254, 255}:
            assert abs(a - 1.0) <= 4e-7 and abs(b - 1.0) <= 4e-7, f"{what}: pixel {i}: {a!r} vs {b!r} at the saturation step"
        else:
            away.append((i, a, b))
    n = np.asarray(ref_f32).size
    assert len(away) <= max(3, max_frac * n), f"{what}: {len(away)} of {n} pixels straddle a step below saturation: {away[:20]}"
    return bad


class Scene:
    """CPU-side inputs of the oracle for one synthetic scene."""

    def __init__(self, size, batch, origin_location="center_fixed", style_dim=256, **kw):
        self.raw = synth.make_scene(size, batch, origin_location, style_dim, **kw)
        s = self.raw
        self.B, self.ol, self.cs = batch, origin_location, 3.0
        self.pose_parts, self.bl_parts = O.transform_pose(s["pose_to_camera"], s["bone_length"], origin_location,
                                                          s["parents"])
        self.cpose, self.cbl = O.register_canonical_pose(s["canonical_pose"], s["parents"], origin_location)
        self.pose_scaled = O.scale_pose_translation(self.pose_parts, self.cs)
        self.scale = O.canonical_scale(self.cbl, self.bl_parts, self.cs)
        self.P = self.pose_parts.shape[1]

    def weights(self):
        return O.modulated_weights(self.raw["mlp"], self.raw["z_rend"])

    def oracle_render(self, coord, Nc, Nf, bins, dtype=torch.float32, taps=True, images=None):
        """`images`: restate only these images of the batch (coord / bins already hold just them); their near / far planes are
        the whole batch's, as in a launch over the batch (rendering.py:15-17) - the oracle's cost grows with the batch"""
        s = self.raw
        if images is None:
            return O.render(coord.to(dtype), self.pose_parts, self.bl_parts, s["inv_intrinsics"], self.cpose, self.cbl,
                            s["tri_plane"], s["mlp"], s["z_rend"], self.cs, Nc, Nf, bins=bins, return_taps=taps)
        planes = O.near_far(self.pose_scaled)
        return O.render(coord.to(dtype), self.pose_parts[images], self.bl_parts[images], s["inv_intrinsics"][images], self.cpose,
                        self.cbl, s["tri_plane"][images], s["mlp"], s["z_rend"][images], self.cs, Nc, Nf, bins=bins,
                        return_taps=taps, near_far_planes=planes)


class DeviceScene:
    """The same scene prepared for the HIP path (enarf_prepare + enarf_triplane_pack)."""

    def __init__(self, sc: Scene, device="cuda:0"):
        from enarf_gan_amd import ops
        s = sc.raw
        d = torch.device(device)
        self.sc, self.dev = sc, d
        self.tri = s["tri_plane"].to(d).contiguous()
        self.feat_cl = ops.triplane_pack(self.tri)
        self.mlp = {k: v.to(d) for k, v in s["mlp"].items()}
        self.cpose = sc.cpose.to(d)
        self.parts, self.pack = ops.prepare(s["pose_to_camera"].to(d), s["bone_length"].to(d), sc.cbl.to(d),
                                            s["z_rend"].to(d), self.mlp, s["parents"], sc.ol, sc.cs)
        self.inv_K = s["inv_intrinsics"].to(d)

    def render(self, coord, Nc, Nf, bins=None, **kw):
        from enarf_gan_amd import ops
        b = None if bins is None else bins.to(self.dev)
        return ops.render_fwd(coord.to(self.dev), self.inv_K, self.parts, self.cpose, self.tri, self.feat_cl,
                              self.pack, Nc, Nf, bins=b, **kw)

    def query(self, pts, **kw):
        from enarf_gan_amd import ops
        return ops.query_fwd(pts.to(self.dev), self.parts, self.cpose, self.tri, self.feat_cl, self.pack, **kw)


def bits_of(valid_bool):
    """(B,P,...) bool tensor -> (B,...) uint32 numpy bit masks."""
    v = valid_bool.numpy().astype(np.uint32)
    P = v.shape[1]
    sh = np.arange(P, dtype=np.uint32).reshape((1, P) + (1,) * (v.ndim - 2))
    return (v << sh).sum(axis=1).astype(np.uint32)


def rel_err(ours, ref):
    ours = np.asarray(ours, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return np.abs(ours - ref) / max(np.abs(ref).max(), 1e-6)


def assert_close(ours, ref, what, rtol=1e-4, frac_ok=0.0):
    e = rel_err(ours, ref)
    bad = float((e > rtol).mean()) if e.size else 0.0
    assert bad <= frac_ok, f"{what}: max rel err {e.max():.3e}, {bad * 100:.4f}% of elements above {rtol}"
    return float(e.max()) if e.size else 0.0
