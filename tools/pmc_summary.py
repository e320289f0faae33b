"""Summarise gpurun_out/pmc_final (tools/gpu_pmc.sh) into profiles/<tag>_pmc_summary.txt, profiles/<tag>_kernel_stats.csv
and profiles/traffic.json. Usage: python tools/pmc_summary.py r01_final"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "pmc_final")
tag = sys.argv[1]
names = {"render_kernel": "march", "pre_march_kernel": "pre", "ray_setup_kernel": "setup"}
lines = ["# rocprofv3 --pmc (one group per run, with --kernel-trace only) on: python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-p24 --spinup-ms 0",
         "# workload C1: 128x128 rays, B=1, Nc 48 + Nf 64, P=23, f16x3; mean over the launches of each kernel",
         "# march = enarf::render_kernel<3,1>, pre = enarf::pre_march_kernel (re-layout + prepare + ray set-up); SQ_* cycle counters are quad-cycles (x4 = cycles)",
         "# FETCH_SIZE / WRITE_SIZE in KB; gfx950 tallies wide reads at half size (MI355X_MICROARCH.md): traffic.json doubles FETCH_SIZE"]
vals = {}
for grp in ("sq1", "sq2", "tcp", "tcc", "fetch", "write", "grbm"):
    acc = defaultdict(list)
    files = sorted(glob.glob(os.path.join(src, grp, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:      # gpurun_out accumulates: only the newest run of each group
        for r in csv.DictReader(open(f)):
            for key, short in names.items():
                if key in r["Kernel_Name"]:
                    acc[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (short, c), v in sorted(acc.items()):
        m = sum(v) / len(v)
        vals[(short, c)] = m
        lines.append(f"{short:6s} {c:34s} {m:.6g}   (n={len(v)})")
open(os.path.join(root, "profiles", f"{tag}_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
st = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
if st:
    shutil.copy(st[-1], os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
if ("march", "FETCH_SIZE") in vals:
    fetch, write = vals[("march", "FETCH_SIZE")], vals.get(("march", "WRITE_SIZE"), 0.0)
    json.dump({"render_kernel_hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "fetch_size_kb_raw": fetch,
               "write_size_kb": write,
               "note": f"rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (profiles/{tag}_pmc_summary.txt); FETCH_SIZE doubled "
                       "per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); L2->fabric requests, Infinity-Cache hits included"},
              open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print("\n".join(lines[4:]))
