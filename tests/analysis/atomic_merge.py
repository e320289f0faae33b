"""CPU analysis (oracle; not a test, not product): 128-B atomic line-adds the backward's feature-plane scatter sends for the
FINE samples of a band of rays, under different on-chip merging schemes. Behind DESIGN.md 3.4. Counts per (part, plane):
  loads       4 taps x valid samples (no merging)
  slot runs   per tap slot, runs of consecutive samples of the tile that hit the same texel (the round-2 kernel)
  parity runs the two taps of a texel ROW dealt to two walkers by the parity of the texel's offset (x0 and x0 + 1 always
              differ in it), each walker merging runs of equal texels: catches a sample's x0 + 1 meeting the next sample's x0
  tile uniq   distinct texels over the 4 taps x 16 samples of a tile (an LDS table per tile and part)
  ray uniq    distinct texels over the whole ray (an LDS table per ray and part)
Run: python tests/analysis/atomic_merge.py [first_ray n_rays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from _helpers import Scene
from oracle import enarf_oracle as O
torch.set_num_threads(8)
S = 128
r0 = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sc = Scene(S, 1, "center_fixed", 256)
c = sc.raw["image_coord"][..., r0:r0 + nr].contiguous()
Nc, Nf = 48, 64
rc, rm, rd, taps = sc.oracle_render(c, Nc, Nf, None)
rd_ = taps["ray_dir"][0]
fd = taps["fine_depth"][0]
n = fd.shape[0]
N = Nf
pts = (rd_[:, :, None] * fd[None]).reshape(1, 3, n * N)
local, can = O.to_local_and_canonical(pts, sc.pose_scaled, sc.scale, sc.cpose)
can = can[0].reshape(23, 3, n, N).numpy()
v = taps["fine_valid"][0].numpy().astype(bool)
v[:, :, N - 1] = False                      # the last fine sample carries no weight (rendering.py:307-321)
W = 256
loads = slot_runs = tile_uniq = ray_uniq = parity_runs = 0
mask_loads = mask_runs = 0
for (a, b) in [(0, 1), (1, 2), (2, 0)]:
    ix = ((can[:, a] + 1) * W - 1) / 2; iy = ((can[:, b] + 1) * W - 1) / 2
    x0 = np.floor(ix).astype(np.int64); y0 = np.floor(iy).astype(np.int64)
    fp = y0 * 1024 + x0
    for r in range(n):
        for k in range(23):
            vk = v[k, r]
            if not vk.any(): continue
            f = fp[k, r]
            for t0 in range(0, N, 16):
                sel = vk[t0:t0 + 16]
                if not sel.any(): continue
                ff = f[t0:t0 + 16][sel]
                loads += 4 * ff.size
                runs = 1 + int((np.diff(ff) != 0).sum())
                slot_runs += 4 * runs
                for row in (0, 1024):                       # texel rows y0 and y0 + 1, each walked by two parity walkers
                    for par in (0, 1):
                        seq = np.where(((ff + row) & 1) == par, ff + row, ff + row + 1)      # the row's tap of that parity
                        parity_runs += 1 + int((np.diff(seq) != 0).sum())
                tile_uniq += np.unique(np.concatenate([ff, ff + 1, ff + 1024, ff + 1025])).size
            ffr = f[vk]
            ray_uniq += np.unique(np.concatenate([ffr, ffr + 1, ffr + 1024, ffr + 1025])).size
pairs = int(v.sum())
print(f"rays {n} (from {r0}), valid fine pairs {pairs}")
print(f"parity runs {parity_runs} ({loads / parity_runs:.2f}x, {parity_runs * 128 / pairs:.0f} B per pair)")
print(f"feature line-adds: loads {loads} | slot runs {slot_runs} ({loads / slot_runs:.2f}x) | tile uniq {tile_uniq} ({loads / tile_uniq:.2f}x) | "
      f"ray uniq {ray_uniq} ({loads / ray_uniq:.2f}x)")
print(f"bytes per pair: loads {loads * 128 / pairs:.0f} | slot runs {slot_runs * 128 / pairs:.0f} | tile uniq {tile_uniq * 128 / pairs:.0f} | ray uniq {ray_uniq * 128 / pairs:.0f}")
