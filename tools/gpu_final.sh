#!/bin/bash
# GPU call at the end of a round: smoke, default bench line, kernel stats + PMC passes of the default workload, 2-rank rehearsal
set -eo pipefail
ulimit -c 0
O=gpurun_out/final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/gpu_pmc.sh
bash tools/gpu_multi_rehearsal.sh
python bench.py --steps 200 2>/dev/null | tee $O/bench_default.json | python tools/exline.py default
python bench.py --steps 20 --warmup 5 2>/dev/null | tee $O/bench_driver_like.json | python tools/exline.py driver-like-20-steps
