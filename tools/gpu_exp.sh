#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -12
B="timeout -k 10 100 python bench.py --steps 60 --warmup 5 --no-cpu-baseline"
E="python tools/exline.py"
$B 2>/dev/null | $E spl_default
$B --nc 72 --nf 96 2>/dev/null | $E nc72_nf96
