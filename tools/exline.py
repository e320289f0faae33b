#!/usr/bin/env python3
"""Condense bench.py's JSON line (stdin) to one short line: tools/exline.py <label>."""
import json
import sys

for line in sys.stdin:
    if line.startswith("{"):
        j = json.loads(line)
        r = j["roofline"]
        print(f"{sys.argv[1]:>14s}  rays/s {j['value']:.4g}  kernel_ms {r['kernel_ms']:.4f}  ms/step {j['ms_per_step']:.4f}  "
              f"alg GB/s {r['achieved']:.0f}  frac {r['frac']:.3f}", flush=True)
