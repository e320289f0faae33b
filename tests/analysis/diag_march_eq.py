"""Diagnosis: the two march kernels (and repeated runs of each) on one case; prints where outputs differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _helpers import DeviceScene, Scene
if os.environ.get("ENARF_VARIANT"):
    from enarf_gan_amd import _lib
    _lib.use_variant(os.path.join(ROOT, "variants", f"libenarf_{os.environ['ENARF_VARIANT']}.so"))
    print("variant", _lib.library_info()["path"])

S, B, Nc, Nf, n0, nr = [int(x) for x in sys.argv[1:7]] if len(sys.argv) >= 7 else (64, 1, 48, 64, 0, 4096)
sc = Scene(S, B, "center_fixed", 20)
ds = DeviceScene(sc)
coord = sc.raw["image_coord"][..., n0:n0 + nr].contiguous()
runs = {}
for rep in range(3):
    for m in ("ray", "task"):
        poison = [torch.full((B * 3 * nr * 4 + (i + 1) * 4096,), float("nan"), device="cuda") for i in range(6)]   # freed: the outputs land on NaNs
        del poison
        o = ds.render(coord, Nc, Nf, None, count=True, return_bins=True, march=m, seed=5, debug=(os.environ.get("DEBUG", "1") == "1"))
        print("   nan in outputs:", [int(torch.isnan(getattr(o, nm)).sum()) for nm in ("color", "mask", "disparity", "fine_weights", "fine_depth")],
              "rays with all-zero fine_depth:", int((o.fine_depth.abs().sum(-1) == 0).sum()))
        if rep == 0 and m == "ray":
            first_taps = {k: v.clone() for k, v in o.taps.items() if torch.is_tensor(v)}
        runs[(m, rep)] = o
        print(m, rep, "counters", o.counters.tolist(), flush=True)
ref = runs[("ray", 0)]
for k, o in runs.items():
    for name in ("color", "mask", "disparity", "fine_weights", "fine_depth"):
        a, b = getattr(ref, name), getattr(o, name)
        if not torch.equal(a, b):
            d = (a != b)
            rays = d.reshape(-1, d.shape[-1] if name in ("fine_weights", "fine_depth") else 1).any(-1) if name in ("fine_weights", "fine_depth") else d
            idx = torch.nonzero(d.reshape(-1))[:8].reshape(-1).tolist()
            print(k, name, "differs in", int(d.sum()), "elements; first flat indices", idx, "max abs diff", float((a - b).abs().max()))
    if not torch.equal(ref.taps["bins"], o.taps["bins"]):
        print(k, "bins differ", int((ref.taps["bins"] != o.taps["bins"]).sum()))
if "ray_validity" in ref.taps:
    print("valid rays", int(ref.taps["ray_validity"].sum()))

last = runs[("ray", 2)].taps
for k, v in (first_taps.items() if os.environ.get("DEBUG", "1") == "1" else []):
    if k in last and torch.is_tensor(last[k]) and not torch.equal(v, last[k]):
        d = (v != last[k])
        print("tap", k, "first launch differs from a later one in", int(d.sum()), "elements, first indices", torch.nonzero(d.reshape(-1))[:6].reshape(-1).tolist())
