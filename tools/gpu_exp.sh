#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
B="timeout -k 10 100 python bench.py --steps 60 --warmup 5 --no-cpu-baseline"
E="python tools/exline.py"
for i in 1 2; do $B 2>/dev/null | $E saddr_$i; done
$B --batch 8 2>/dev/null | $E saddr_b8
