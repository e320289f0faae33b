"""Summarise gpurun_out/pmc_final (tools/gpu_pmc.sh) into profiles/<tag>_pmc_summary.txt, profiles/<tag>_kernel_stats.csv
and profiles/traffic.json. Usage: python tools/pmc_summary.py r01_final [gpurun_out subdirectory, default pmc_final; WORKLOAD_KEY=... for a non-default workload]"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "pmc_final"
src = os.path.join(root, "gpurun_out", sub)
default_workload = sub == "pmc_final"           # only the default C1 profile feeds bench.py (traffic.json, r03_roofline.json)
names = {"enarf::march_kernel<": "march", "enarf::render_kernel<": "march", "enarf::pre_march_kernel": "pre", "ray_setup_kernel": "setup"}
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
if ("march", "FETCH_SIZE") in vals and default_workload:
    fetch, write = vals[("march", "FETCH_SIZE")], vals.get(("march", "WRITE_SIZE"), 0.0)
    json.dump({"render_kernel_hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "fetch_size_kb_raw": fetch,
               "write_size_kb": write,
               "note": f"rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (profiles/{tag}_pmc_summary.txt); FETCH_SIZE doubled "
                       "per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); L2->fabric requests, Infinity-Cache hits included"},
              open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
# ---- counter-backed utilisation of the march (read by bench.py as roofline.counters when the workload matches) ----------
def kernel_avg_ns(stats_csv, key):
    for r in csv.DictReader(open(stats_csv)):
        if key in r.get("Name", ""):
            return float(r["AverageNs"])
    return None


if st and ("march", "SQ_WAVE_CYCLES") in vals:
    t_ns = kernel_avg_ns(st[-1], "enarf::march_kernel<") or kernel_avg_ns(st[-1], "enarf::render_kernel<")
    t_src = f"profiles/{tag}_kernel_stats.csv (rocprofv3 --stats pass)"
    if not default_workload:
        # a non-default workload's stats pass also times the f32 / P = 24 extras of bench.py under the same kernel name for
        # some shapes, and a batch marched in groups launches the march once per group: take the mean duration of the
        # launches the counters themselves were collected on (the FETCH_SIZE pass's own kernel trace)
        tr = sorted(glob.glob(os.path.join(root, "gpurun_out", sub, "fetch", "*", "*kernel_trace.csv")), key=os.path.getmtime)
        if tr:
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[-1]))
                 if any(k in r["Kernel_Name"] for k in ("enarf::march_kernel<", "enarf::render_kernel<"))]
            if d:
                t_ns, t_src = sum(d) / len(d), f"mean of the {len(d)} march launches of the FETCH_SIZE pass (gpurun_out/{sub}/fetch)"
    t = t_ns * 1e-9
    g = lambda c: vals.get(("march", c), 0.0)
    # persistent waves live for the whole launch: their mean lifetime in shader cycles is the launch's cycle count
    cycles = 4.0 * g("SQ_WAVE_CYCLES") / max(g("SQ_WAVES"), 1.0)
    hbm_bytes = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024
    src = f"profiles/{tag}_pmc_summary.txt"
    fr = {
        "ta_busy": {"value": g("TA_TA_BUSY_sum") / (256 * cycles), "formula": "TA_TA_BUSY_sum / (256 CUs x cycles)", "source": src},
        "valu_busy": {"value": 4 * g("SQ_ACTIVE_INST_VALU") / (1024 * cycles), "formula": "4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles)", "source": src},
        "mfma_busy": {"value": g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cycles), "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)", "source": src},
        "lds_busy": {"value": 4 * g("SQ_ACTIVE_INST_LDS") / (1024 * cycles), "formula": "4 x SQ_ACTIVE_INST_LDS / (1024 SIMDs x cycles)", "source": src},
        "l2_read": {"value": g("TCP_TCC_READ_REQ_sum") * 64 / t / 34.5e12, "formula": "TCP_TCC_READ_REQ_sum x 64 B / t / 34.5 TB/s", "source": src},
        "hbm": {"value": hbm_bytes / t / 8e12, "formula": "(2 x FETCH_SIZE + WRITE_SIZE) KB / t / 8 TB/s (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md)", "source": src},
        "wave_parked": {"value": g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0), "formula": "SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt, barriers, sleep)", "source": src},
        "wave_issue_stalled": {"value": g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0), "formula": "SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES", "source": src},
        "l1_hit": {"value": 1.0 - g("TCP_TCC_READ_REQ_sum") / max(g("TCP_TOTAL_CACHE_ACCESSES_sum"), 1.0), "formula": "1 - TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum", "source": src},
        "l2_hit": {"value": g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0), "formula": "TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)", "source": src},
    }
    limiter = max(("ta_busy", "valu_busy", "mfma_busy", "l2_read", "hbm"), key=lambda k: fr[k]["value"])
    json.dump({"workload_key": os.environ.get("WORKLOAD_KEY", "C1:128:1:48:64:23:f16x3:0:0.0"), "kernel_ms": t_ns * 1e-6, "kernel_ms_source": t_src,
               "cycles_per_launch": cycles,
               "clock_ghz_under_profiler": cycles / t / 1e9, "hbm_bytes_per_launch": int(hbm_bytes), "limiter": limiter, "fractions": fr,
               "kernel_stats": f"profiles/{tag}_kernel_stats.csv"},
              open(os.path.join(root, "profiles", "r03_roofline.json" if default_workload else f"{tag}_roofline.json"), "w"), indent=1)
print("\n".join(lines[4:]))
