#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2x; mkdir -p $O
rc=0; python -m pytest tests/test_gpu_parity.py -m gpu -q -k "march_kernels or ragged or edge_cases or replays" > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -4 | tee $O/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
var() { echo "--allow-variant --variant variants/libenarf_$1.so"; }
{
run B8-chunk8 --steps 60 --batch 8
run B8-chunk1 --steps 60 --batch 8 $(var mc1)
run B8-chunk32 --steps 60 --batch 8 $(var mc32)
run B8-drop --steps 60 --batch 8 --drop-missed-rays
run B8d-chunk8 --steps 60 --batch 8 --distinct-triplanes
run B64d-chunk8 --steps 8 --batch 64 --distinct-triplanes
run B64d-chunk32 --steps 8 --batch 64 --distinct-triplanes $(var mc32)
} | tee $O/bench.log
