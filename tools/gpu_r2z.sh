#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2z; mkdir -p $O
rc=0; python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -4
if [ $rc -gt 1 ]; then exit $rc; fi
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
run C1 --steps 400
run C1b --steps 400
run B8d --steps 60 --batch 8 --distinct-triplanes
