"""CPU analysis (oracle; not a test, not product): per-sample valid-part counts of the C1 frame and what they allow the
gather rounds - rounds as built (max count per 16-sample tile), with samples sorted by count inside a pass, with quads
j / j + 8 sharing work, and perfectly dense tiles. Behind DESIGN.md 3.1's "dense gather rounds" row.
Run: python tests/analysis/lane_utilisation.py   (a few minutes on 8 cores)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from _helpers import Scene
torch.set_num_threads(8)
S=128
sc = Scene(S, 1, "center_fixed", 256)
coord = sc.raw["image_coord"]
n = coord.shape[-1]
Nc, Nf = 48, 64
g = torch.Generator().manual_seed(1)
res = {"c": [], "f": []}
t0=time.time()
CH=2048
for s in range(0, n, CH):
    c = coord[..., s:s+CH].contiguous()
    bins = None
    rc, rm, rd, taps = sc.oracle_render(c, Nc, Nf, bins)
    rv = taps["ray_validity"].reshape(-1).bool().numpy()
    cv = taps["coarse_valid"][0].numpy()   # (P, n, Nc)
    fv = taps["fine_valid"][0].numpy()
    # taps are for the kept rays only? check shape
    res["c"].append(cv.sum(0)); res["f"].append(fv.sum(0))
    print(s, cv.shape, fv.shape, rv.sum(), time.time()-t0, flush=True)
pc = np.concatenate(res["c"], 0); pf = np.concatenate(res["f"], 0)
np.savez(os.path.join(os.environ.get('TMPDIR', '/tmp'), 'popc.npz'), pc=pc, pf=pf)

live = (pc.sum(1) + pf.sum(1)) > 0
print("rays with any pair", int(live.sum()), "pairs", int(pc.sum() + pf.sum()))


def stats(p, name):
    n_, N = p.shape
    T = N // 16
    pt = p.reshape(n_, T, 16)
    cur = pt.max(2).sum()
    pairs = p.sum()
    ps = -np.sort(-p, axis=1).reshape(n_, T, 16)
    srt = ps.max(2).sum()
    sh = np.ceil((pt[:, :, :8] + pt[:, :, 8:]) / 2).max(2).sum()
    dense = np.ceil(pt.sum(2) / 16).sum()
    print(f"{name}: pairs {pairs}  rounds now {cur} (lane utilisation {pairs / 16 / cur:.3f})  sorted by count {srt} ({pairs / 16 / srt:.3f})  "
          f"quads j / j+8 share {sh} ({pairs / 16 / sh:.3f})  dense tiles {dense} ({pairs / 16 / dense:.3f})")
    return cur, srt, sh, dense


a, b = stats(pc, "coarse"), stats(pf, "fine")
print("total rounds now", a[0] + b[0], "sorted", a[1] + b[1], "share", a[2] + b[2], "dense", a[3] + b[3])
