#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2f; mkdir -p $O
{ echo "=== task march, 700 rays"; NRAYS=700 REPS=2 MODES=f32,f16x3 timeout -k 10 120 python tools/tapcheck.py 2>&1 | grep -v amdgpu.ids
  echo "=== task march, full frame"; REPS=3 timeout -k 10 200 python tools/tapcheck.py 2>&1 | grep -v amdgpu.ids; } | tee $O/first.log
if grep -q "NOT deterministic\|Error\|error" $O/first.log; then echo "PROBLEM - stopping"; exit 0; fi
python bench.py --no-cpu-baseline --steps 300 2>&1 | grep -v amdgpu.ids | python tools/exline.py task-march | tee $O/bench.log
timeout -k 10 120 python bench.py --no-cpu-baseline --steps 300 --allow-variant --variant variants/libenarf_v2march.so 2>&1 | grep -v amdgpu.ids | python tools/exline.py v2march | tee -a $O/bench.log
python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -30 | tee $O/pytest.log
