#!/bin/bash
# GPU call: quick parity subset, mul24 A/B, kernel stats + PMC passes for profiles/r02_*, N = 2 rehearsal
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2m; mkdir -p $O
rc=0; python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -8 | tee $O/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
var() { echo "--allow-variant --variant variants/libenarf_$1.so"; }
{
run ray-C1 --steps 300
run ray-nomul24-C1 --steps 300 $(var nomul24)
run task-C1 --steps 300 --march task
run task-nomul24-C1 --steps 300 --march task $(var nomul24)
run ray-C1-again --steps 300
} | tee $O/bench.log
bash tools/gpu_pmc.sh
bash tools/gpu_multi_rehearsal.sh
python bench.py --steps 200 2>/dev/null | tee $O/bench_default.json | python tools/exline.py default
