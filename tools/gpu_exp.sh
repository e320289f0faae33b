#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tee gpurun_out/pytest_gpu.log | tail -8
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline"
E="python tools/exline.py"
{
$B 2>/dev/null | $E trims
$B --batch 8 2>/dev/null | $E trims_batch8
$B --mlp-mode f32 2>/dev/null | $E trims_f32
} | tee gpurun_out/exp12.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof12 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/prof12/*/*_kernel_stats.csv | cut -c1-140 | head -6
