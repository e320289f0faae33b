#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_backward.py -q -x 2>&1 | tee gpurun_out/pytest_bwd.log | tail -5
timeout -k 10 120 python tools/bench_bwd.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_bwd.log
