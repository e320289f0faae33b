#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2s; mkdir -p $O
DEBUG=0 python tools/diag_march_eq.py 32 3 24 32 5 333 2>&1 | grep -v amdgpu.ids | grep "counters\|differs" | tee $O/diag_b3.log
rc=0; python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_configs.py -m gpu -q > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -6 | tee $O/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
B="python bench.py --no-cpu-baseline --no-p24 --no-f32"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
run B8 --steps 60 --batch 8
run B8d --steps 60 --batch 8 --distinct-triplanes
run B64d --steps 8 --batch 64 --distinct-triplanes
