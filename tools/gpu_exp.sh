#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -x -k "early or full_image or in_kernel or counters" 2>&1 | tail -5
B="timeout -k 10 100 python bench.py --steps 60 --warmup 5 --no-cpu-baseline"
E="python tools/exline.py"
$B 2>/dev/null | $E exact
$B --early-stop-eps 1e-3 2>/dev/null | $E eps1e-3
$B --early-stop-eps 1e-2 2>/dev/null | $E eps1e-2
