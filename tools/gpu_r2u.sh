#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2u; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
var() { echo "--allow-variant --variant variants/libenarf_$1.so"; }
{
run task-s6 --steps 300 --march task
for k in 4 5 7 8 10; do run task-s$k --steps 300 --march task $(var s$k); done
run ray --steps 300
run task-7296-s6 --steps 100 --nc 72 --nf 96
for k in 5 8 10; do run task-7296-s$k --steps 100 --nc 72 --nf 96 $(var s$k); done
} | tee $O/bench.log
