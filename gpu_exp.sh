#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tee gpurun_out/pytest_gpu.log | tail -6
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline"
E="python tools/exline.py"
{
$B 2>/dev/null | $E queue_3percu
for w in 2 4 6; do ENARF_WGS_PER_CU=$w $B 2>/dev/null | $E queue_wgs$w; done
for a in 1 3 7; do ENARF_ABLATE=$a $B 2>/dev/null | $E queue_ablate$a; done
$B --batch 8 2>/dev/null | $E queue_batch8
$B --batch 32 --steps 20 2>/dev/null | $E queue_batch32
} | tee gpurun_out/exp5.log
