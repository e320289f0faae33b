#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -x -k "render" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof15 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/prof15/*/*_kernel_stats.csv | cut -c1-130 | head -6
