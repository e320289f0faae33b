#!/bin/bash
# GPU call: full suite; row-pair mask loads A/B on both march kernels; batches; C4-like
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2l; mkdir -p $O
rc=0; python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -15 | tee $O/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
var() { echo "--allow-variant --variant variants/libenarf_$1.so"; }
{
run ray-C1 --steps 300 --march ray
run task-C1 --steps 300 --march task
run ray-nopairs-C1 --steps 300 --march ray $(var nopairs)
run task-nopairs-C1 --steps 300 --march task $(var nopairs)
run ray-vtaps-C1 --steps 300 --march ray $(var vtaps)
run auto-C1 --steps 300
run ray-B8 --steps 60 --batch 8 --march ray
run ray-nopairs-B8 --steps 60 --batch 8 --march ray $(var nopairs)
run ray-B16d --steps 30 --batch 16 --distinct-triplanes
run task-7296 --steps 100 --nc 72 --nf 96 --march task
run ray-7296 --steps 100 --nc 72 --nf 96 --march ray
run task-nopairs-7296 --steps 100 --nc 72 --nf 96 --march task $(var nopairs)
run task-7296-B8 --steps 30 --nc 72 --nf 96 --batch 8 --march task
run ray-7296-B8 --steps 30 --nc 72 --nf 96 --batch 8 --march ray
run ray-C0 --steps 100 --size 64 --nf 32 --march ray
run task-C0 --steps 100 --size 64 --nf 32 --march task
} | tee $O/bench.log
