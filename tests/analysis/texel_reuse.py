"""CPU analysis (oracle; not a test, not product): how many distinct 128-B texel lines the 16 samples of a tile touch per
(part, plane), against the texel loads the gather rounds issue - the reuse an LDS-staged, de-duplicated gather could
exploit. Behind DESIGN.md 3.1's "texel de-duplication" row. Run: python tests/analysis/texel_reuse.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from _helpers import Scene
from oracle import enarf_oracle as O
torch.set_num_threads(8)
S=128
sc = Scene(S, 1, "center_fixed", 256)
coord = sc.raw["image_coord"]
Nc, Nf = 48, 64
c = coord[..., 6144:8192].contiguous()
rc, rm, rd, taps = sc.oracle_render(c, Nc, Nf, None)
print({k: tuple(v.shape) for k, v in taps.items() if torch.is_tensor(v)})
rd_ = taps["ray_dir"][0]            # (3, n)
fd = taps["fine_depth"][0]          # (n, Nf)
n = fd.shape[0]
def analyse(depth, valid, name):
    N = depth.shape[1]
    pts = (rd_[:, :, None] * depth[None]).reshape(1, 3, n * N)
    local, can = O.to_local_and_canonical(pts, sc.pose_scaled, sc.scale, sc.cpose)
    can = can[0].reshape(23, 3, n, N).numpy()
    v = valid[0].numpy().astype(bool)          # (P, n, N)
    W = 256
    tot_lines = 0; uniq_tile = 0; uniq_ray = 0; pairs = 0; uniq_fp_tile = 0
    planes = [(0, 1), (1, 2), (2, 0)]
    for (a, b) in planes:
        ix = ((can[:, a] + 1) * W - 1) / 2; iy = ((can[:, b] + 1) * W - 1) / 2
        x0 = np.floor(ix).astype(np.int64); y0 = np.floor(iy).astype(np.int64)
        fp = (y0 * 1024 + x0)                    # footprint id (P, n, N)
        for r in range(n):
            for k in range(23):
                vk = v[k, r]
                if not vk.any(): continue
                f = fp[k, r]
                for t0 in range(0, N, 16):
                    sel = vk[t0:t0 + 16]
                    if not sel.any(): continue
                    ff = f[t0:t0 + 16][sel]
                    lines = np.unique(np.concatenate([ff, ff + 1, ff + 1024, ff + 1025]))
                    uniq_tile += lines.size
                    uniq_fp_tile += np.unique(ff).size
                    tot_lines += 4 * ff.size
                ffr = f[vk]
                uniq_ray += np.unique(np.concatenate([ffr, ffr + 1, ffr + 1024, ffr + 1025])).size
    pairs = int(v.sum())
    print(f"{name}: pairs {pairs}  texel loads {tot_lines}  unique per (tile, part, plane) {uniq_tile} ({tot_lines / uniq_tile:.2f}x reuse)  "
          f"unique footprints per tile {uniq_fp_tile} ({tot_lines / 4 / uniq_fp_tile:.2f}x)  unique per (ray, part, plane) {uniq_ray} ({tot_lines / uniq_ray:.2f}x)")
cd = taps["coarse_depth"][0]
cmid = 0.5 * (cd[:, 1:] + cd[:, :-1])
analyse(fd, taps["fine_valid"], "fine")
analyse(cmid, taps["coarse_valid"], "coarse")
