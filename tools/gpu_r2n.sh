#!/bin/bash
# GPU call: counters of the batch workload (16 frames, per-frame tri-planes) to see what its +22 % per frame is; default line with counters attached
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2n; mkdir -p $O
python bench.py --steps 200 2>/dev/null | tee $O/bench_default.json | python tools/exline.py default
bash tools/gpu_pmc.sh pmc_b16d --batch 16 --distinct-triplanes --no-f32
bash tools/gpu_pmc.sh pmc_b8 --batch 8 --no-f32
