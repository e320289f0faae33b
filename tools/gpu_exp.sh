#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -6
timeout -k 10 120 python tools/bench_bwd.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_bwd.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bwd -- python3 $GRAFT_REPO_ROOT/tools/bench_bwd.py > /dev/null 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/prof_bwd/*/*_kernel_stats.csv | cut -c1-150 | head -8
