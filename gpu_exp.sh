#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tee gpurun_out/pytest_gpu.log | tail -12
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline"
E="python tools/exline.py"
{
$B 2>/dev/null | $E prepass
for w in 2 4; do ENARF_WGS_PER_CU=$w $B 2>/dev/null | $E prepass_wgs$w; done
for a in 7 63; do ENARF_ABLATE=$a $B 2>/dev/null | $E prepass_abl$a; done
$B --batch 8 2>/dev/null | $E prepass_batch8
} | tee gpurun_out/exp10.log
