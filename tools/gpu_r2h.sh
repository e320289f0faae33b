#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2h; mkdir -p $O
{ echo "=== determinism"; REPS=2 MODES=f32,f16x3 timeout -k 10 200 python tools/tapcheck.py 2>&1 | grep -v amdgpu.ids; } | tee $O/first.log
if grep -q "NOT deterministic\|Error\|error" $O/first.log; then echo "PROBLEM - stopping"; exit 0; fi
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --steps 300"
$B 2>&1 | grep -v amdgpu.ids | python tools/exline.py task-w12s8 | tee $O/bench.log
for v in s4 s5 s6 w16s6 w16s8 v2march; do
  timeout -k 10 120 $B --allow-variant --variant variants/libenarf_$v.so 2>&1 | grep -v amdgpu.ids | python tools/exline.py $v | tee -a $O/bench.log
done
TIMERS=5 ENARF_VARIANT=timers5 timeout -k 10 120 python tools/timers.py 2>&1 | grep -v amdgpu.ids | tee $O/timers5.log
TIMERS=5 ENARF_VARIANT=t5s5 timeout -k 10 120 python tools/timers.py 2>&1 | grep -v amdgpu.ids | tee $O/timers5_s5.log
python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -30 | tee $O/pytest.log
