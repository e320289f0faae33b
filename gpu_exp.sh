#!/bin/bash
mkdir -p gpurun_out
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline"
E="python tools/exline.py"
{
$B 2>/dev/null | $E strided_rpw8
ENARF_NO_STRIDE=1 $B 2>/dev/null | $E nostride_rpw8
for r in 4 16 32 64; do ENARF_RAYS_PER_WG=$r $B 2>/dev/null | $E strided_rpw$r; done
$B --batch 8 2>/dev/null | $E batch8
} | tee gpurun_out/exp2.log
