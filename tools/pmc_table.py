"""Summarise a tools/gpu_bwd_pmc.sh run (gpurun_out/<sub>) for profiles/: <tag>_pmc_summary.txt (mean counter value per launch
for every kernel named below), <tag>_kernel_stats.csv (rocprofv3 --stats) and <tag>_roofline.json (derived fractions of the
backward kernel). Usage: python tools/pmc_table.py <tag> [sub, default pmc_bwd]"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "pmc_bwd"
src = os.path.join(root, "gpurun_out", sub)
names = {"enarf::render_bwd_kernel<": "bwd", "enarf::weight_grad_partial_kernel": "wgrad", "enarf::weight_grad_reduce_kernel": "wred",
         "enarf::unpack_add_kernel": "unpack", "enarf::prepare_bwd_kernel": "prepb", "ray_setup_kernel": "setup"}
lines = [f"# rocprofv3 --pmc (one group per run, with --kernel-trace only) on: python3 tools/bench_bwd.py (ITERS=3); source gpurun_out/{sub}",
         "# mean over the launches of each kernel; SQ_* cycle counters are quad-cycles (x4 = cycles)",
         "# FETCH_SIZE / WRITE_SIZE in KB; gfx950 tallies wide reads at half size (MI355X_MICROARCH.md): derived traffic doubles FETCH_SIZE"]
try:
    lines.append("# workload: " + [l for l in open(os.path.join(src, "stats.log")) if l.startswith("{")][-1].strip())
except Exception:
    pass
vals = {}
for grp in ("sq1", "sq2", "tcp", "tcc", "atom", "fetch", "write"):
    acc = defaultdict(list)
    files = sorted(glob.glob(os.path.join(src, grp, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:
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


def kernel_avg_ns(stats_csv, key):
    for r in csv.DictReader(open(stats_csv)):
        if key in r.get("Name", ""):
            return float(r["AverageNs"])
    return None


if st and ("bwd", "SQ_WAVE_CYCLES") in vals:
    t_ns = kernel_avg_ns(st[-1], "enarf::render_bwd_kernel<")
    t = t_ns * 1e-9
    g = lambda c: vals.get(("bwd", c), 0.0)
    cycles = 4.0 * g("SQ_WAVE_CYCLES") / max(g("SQ_WAVES"), 1.0)       # persistent waves live for the whole launch
    hbm_bytes = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024
    s = f"profiles/{tag}_pmc_summary.txt"
    fr = {
        "ta_busy": {"value": g("TA_TA_BUSY_sum") / (256 * cycles), "formula": "TA_TA_BUSY_sum / (256 CUs x cycles)", "source": s},
        "valu_busy": {"value": 4 * g("SQ_ACTIVE_INST_VALU") / (1024 * cycles), "formula": "4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles)", "source": s},
        "mfma_busy": {"value": g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cycles), "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)", "source": s},
        "lds_busy": {"value": 4 * g("SQ_ACTIVE_INST_LDS") / (1024 * cycles), "formula": "4 x SQ_ACTIVE_INST_LDS / (1024 SIMDs x cycles)", "source": s},
        "hbm": {"value": hbm_bytes / t / 8e12, "formula": "(2 x FETCH_SIZE + WRITE_SIZE) KB / t / 8 TB/s", "source": s},
        "atomic_rate": {"value": g("TCC_EA0_ATOMIC_sum") * 64 / t / 1.3e12,
                        "formula": "TCC_EA0_ATOMIC_sum x 64 B / t / 1.3 TB/s (MI355X_MICROARCH.md, Global float atomics; cdna_hip_programming.md rule on TCC_EA0_ATOMIC_sum)", "source": s},
        "wave_parked": {"value": g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0), "formula": "SQ_WAIT_ANY / SQ_WAVE_CYCLES", "source": s},
        "wave_issue_stalled": {"value": g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0), "formula": "SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES", "source": s},
        "l1_hit": {"value": 1.0 - g("TCP_TCC_READ_REQ_sum") / max(g("TCP_TOTAL_CACHE_ACCESSES_sum"), 1.0), "formula": "1 - TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum", "source": s},
        "l2_hit": {"value": g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0), "formula": "TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)", "source": s},
    }
    json.dump({"kernel": "enarf::render_bwd_kernel", "kernel_ms": t_ns * 1e-6, "cycles_per_launch": cycles, "waves_per_launch": g("SQ_WAVES"),
               "clock_ghz_under_profiler": cycles / t / 1e9, "hbm_bytes_per_launch": int(hbm_bytes),
               "atomic_bytes_per_launch_from_counters": int(g("TCC_EA0_ATOMIC_sum") * 64), "fractions": fr,
               "kernel_stats": f"profiles/{tag}_kernel_stats.csv"},
              open(os.path.join(root, "profiles", f"{tag}_roofline.json"), "w"), indent=1)
print("\n".join(lines[3:]))
