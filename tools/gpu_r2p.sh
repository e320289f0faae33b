#!/bin/bash
# GPU call: cost classes for batches with several frames per band (A/B), the re-layout alone
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2p; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
var() { echo "--allow-variant --variant variants/libenarf_$1.so"; }
{
run B16d --steps 30 --batch 16 --distinct-triplanes
run B16d-bc1 --steps 30 --batch 16 --distinct-triplanes $(var bc1)
run B16d-bc2 --steps 30 --batch 16 --distinct-triplanes $(var bc2)
run B32d --steps 15 --batch 32 --distinct-triplanes
run B32d-bc1 --steps 15 --batch 32 --distinct-triplanes $(var bc1)
run B32d-bc2 --steps 15 --batch 32 --distinct-triplanes $(var bc2)
run B16 --steps 30 --batch 16
run B16-bc1 --steps 30 --batch 16 $(var bc1)
run B64d --steps 8 --batch 64 --distinct-triplanes
run B64d-bc1 --steps 8 --batch 64 --distinct-triplanes $(var bc1)
run B8d --steps 60 --batch 8 --distinct-triplanes
} | tee $O/bench.log
