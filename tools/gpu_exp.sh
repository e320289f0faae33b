#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tee gpurun_out/pytest_gpu.log | tail -5
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline"
E="python tools/exline.py"
{
$B 2>/dev/null | $E xcdq
for w in 2 4; do ENARF_WGS_PER_CU=$w $B 2>/dev/null | $E xcdq_wgs$w; done
$B --batch 8 2>/dev/null | $E xcdq_batch8
} | tee gpurun_out/exp11.log
