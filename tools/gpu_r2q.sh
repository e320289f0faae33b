#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2q; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
{
run B8 --steps 60 --batch 8
run B8-drop --steps 60 --batch 8 --drop-missed-rays
run B8d --steps 60 --batch 8 --distinct-triplanes
run B8d-drop --steps 60 --batch 8 --distinct-triplanes --drop-missed-rays
run B2 --steps 100 --batch 2
run B2-drop --steps 100 --batch 2 --drop-missed-rays
} | tee $O/bench.log
