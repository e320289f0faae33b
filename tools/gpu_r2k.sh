#!/bin/bash
# GPU call: full suite, TA microbenchmark (5 patterns), software-pipelined rounds A/B on both marches
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2k; mkdir -p $O
rc=0; python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -15 | tee $O/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 120 variants/ub_ta 2>&1 | tee $O/ub_ta.txt
B="python bench.py --no-cpu-baseline --no-p24 --no-f32"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
var() { echo "--allow-variant --variant variants/libenarf_$1.so"; }
{
run task-C1 --steps 300
run v2-C1 --steps 300 $(var v2march)
run task-noswp-C1 --steps 300 $(var noswp)
run v2-noswp-C1 --steps 300 $(var v2noswp)
run task-vtaps-C1 --steps 300 $(var vtaps)
run v2-vtaps-C1 --steps 300 $(var v2vtaps)
run task-C1-again --steps 300
run v2-C1-again --steps 300 $(var v2march)
run task-B8 --steps 60 --batch 8
run v2-B8 --steps 60 --batch 8 $(var v2march)
run v2-noswp-B8 --steps 60 --batch 8 $(var v2noswp)
run task-7296 --steps 100 --nc 72 --nf 96
run v2-7296 --steps 100 --nc 72 --nf 96 $(var v2march)
} | tee $O/bench.log
