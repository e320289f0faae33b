#!/bin/bash
# tests + smoke + bench + kernel-trace profile on the GPU box
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -25
python __graft_entry__.py --smoke 2>&1 | tail -3
python bench.py --steps 100 --warmup 10 2>&1 | tee gpurun_out/bench.log | tail -3
for m in f32 bf16x3 bf16; do python bench.py --steps 100 --warmup 10 --mlp-mode $m --no-cpu-baseline 2>&1 | tee gpurun_out/bench_$m.log | tail -1; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof -name "*kernel_stats*" | head -3
