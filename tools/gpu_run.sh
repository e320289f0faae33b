#!/bin/bash
# tests + smoke + bench + kernel-trace profile on the GPU box
set -eo pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -15
python __graft_entry__.py --smoke 2>&1 | tail -2
python bench.py 2>/dev/null | tee gpurun_out/bench.json | python tools/exline.py default
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-two-streams > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/prof/*/*_kernel_stats.csv | cut -c1-160 | head -6
