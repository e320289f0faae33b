"""GPU diagnostic: histogram of popcount(cand) over live rays (set-up pass records) on the bench frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from enarf_gan_amd import ops, synth
S, Nc, Nf = 128, 48, 64
sc = synth.make_scene(S, 1, "center_fixed", 20, pose_seed=1234, shared_triplane=True)
dev = torch.device("cuda:0")
cpose, cbl = synth.canonical_buffers(sc, "center_fixed")
d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
tri = sc["tri_plane"][:1].contiguous().to(dev)
mlp = {k: v.to(dev) for k, v in sc["mlp"].items()}
feat = ops.triplane_pack(tri)
st = ops.RenderStep(d["pose_to_camera"], d["bone_length"], cbl.to(dev), d["z_rend"], mlp, sc["parents"], "center_fixed",
                    3.0, d["image_coord"].reshape(1, 3, S * S), d["inv_intrinsics"], cpose.to(dev), tri, feat, Nc, Nf,
                    seed=99, mlp_mode="f16x3", debug=True)
out = st.run()
torch.cuda.synchronize()
ws = list(ops._render_ws.values())[0].cpu().numpy().view(np.uint32)
hdr_ints = 2 * (16 + 32 + 32 * 16)          # two headers of kWsHeaderBytes / 4 ints (enarf_march.h)
hdr = ws[:hdr_ints // 2]
recs = ws[hdr_ints:hdr_ints + 8 * S * S].reshape(-1, 8)
cand, valid = recs[:, 2], recs[:, 3]
pc = np.array([bin(int(c)).count("1") for c in cand])
live = valid == 1
print("live", live.sum())      # (the header holding this launch's counts is already cleared or reused by later launches)
print("popcount(cand) histogram over live rays:", np.bincount(pc[live], minlength=24).tolist())
fv = out.taps["fine_valid"][0].cpu().numpy()        # (n, Nf) bit masks
cost = np.array([[bin(int(x)).count("1") for x in row] for row in fv[live][:, :]]).max(1)
print("corr(popcount(cand), max fine valid count per ray) =", np.corrcoef(pc[live], cost)[0, 1])
for lo, hi in ((0, 3), (3, 6), (6, 10), (10, 24)):
    m = (pc[live] >= lo) & (pc[live] < hi)
    print(f"cand in [{lo},{hi}): rays {m.sum()}  mean max-valid {cost[m].mean() if m.any() else 0:.2f}")
