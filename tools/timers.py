"""Phase breakdown of the march from a -DENARF_TIMERS=1 build (tools/build_variant.sh timers -DENARF_TIMERS=1; ENARF_VARIANT=timers): per-wave
cycle sums per phase, printed as fractions of the summed wave time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from enarf_gan_amd import ops, synth

def _maybe_variant():
    """tools only: ENARF_VARIANT=<name> loads variants/libenarf_<name>.so (tools/build_variant.sh) instead of the in-tree build"""
    import os
    v = os.environ.get("ENARF_VARIANT")
    if v:
        from enarf_gan_amd import _lib
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        _lib.use_variant(v if os.path.sep in v else os.path.join(root, "variants", f"libenarf_{v}.so"))
        print("variant library:", _lib.library_info()["path"], flush=True)


_maybe_variant()

S, Nc, Nf = 128, 48, 64
sc = synth.make_scene(S, 1, "center_fixed", 20, pose_seed=1234, shared_triplane=True)
dev = torch.device("cuda:0")
cpose, cbl = synth.canonical_buffers(sc, "center_fixed")
d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
tri = sc["tri_plane"][:1].contiguous().to(dev)
mlp = {k: v.to(dev) for k, v in sc["mlp"].items()}
feat = ops.triplane_pack(tri)
names2 = ["0 round set-up + issue p0 p1", "1 mask wait", "2 sigmoid", "3 plane-0 wait", "4 reduce p0 p1 + issue p2", "5 plane-2 wait",
          "6 reduce p2 + finish", "7 everything outside the rounds"]
names = ["0 ray header", "1 pass A (candidate tests)", "2 round set-up + mask taps", "3 feature gathers + FMA",
         "4 transpose + MLP", "5 barrier wait", "6 S2 weights/sampling", "7 S4 composite"]
for rep in range(12):
    st = ops.RenderStep(d["pose_to_camera"], d["bone_length"], cbl.to(dev), d["z_rend"], mlp, sc["parents"], "center_fixed",
                        3.0, d["image_coord"].reshape(1, 3, S * S), d["inv_intrinsics"], cpose.to(dev), tri, feat, Nc, Nf,
                        seed=99, mlp_mode=os.environ.get("MODE", "f16x3"), count=True)
    if os.environ.get("TIMERS") == "3":
        st.out.counters[0] = 2 ** 62
        st.out.counters[5] = 2 ** 62
    out = st.run()
    torch.cuda.synchronize()
if os.environ.get("TIMERS") == "2":
    names = names2
if os.environ.get("TIMERS") == "4":
    names = ["0 ray header (queue, record, direction)", "1 S2 weights (density, scan, smoothing)", "2 S2 sampling (philox, cdf, search)",
             "3 S4 heads (tanh, density)", "4 S4 scan + weights", "5 S4 sums + stores", "6 barrier waits", "7 S1 + S3 (queries)"]
if os.environ.get("TIMERS") == "5":
    names = ["0 query tiles (pass A, rounds, MLP)", "1 S2 weights + importance samples", "2 S4 compositing + outputs",
             "3 refill after S4 (+ start-up)", "4 idle (no tile to claim)", "5 scan + claim", "6 pop-ahead of the next ray", "-"]
if os.environ.get("TIMERS") == "3":
    c = out.counters.cpu().tolist()
    span = c[1] - c[0]
    print(f"workgroups {c[3]}  launch span {span / 100:.1f} us  mean workgroup busy {c[2] / c[3] / 100:.1f} us "
          f"({c[2] / c[3] / span * 100:.1f} % of the span)  last start +{(c[4] - c[0]) / 100:.1f} us  first end +{(c[5] - c[0]) / 100:.1f} us  "
          f"mean last-ray {c[6] / c[3] / 100:.1f} us  longest ray {c[7] / 100:.1f} us")
    sys.exit(0)
c = out.counters.cpu().double()
tot = float(c.sum())
for k in range(8):
    print(f"{names[k]:32s} {float(c[k]) / tot * 100:6.2f} %   {float(c[k]):.3e} cycles")
print("total wave-cycles", tot)
