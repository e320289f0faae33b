#!/bin/bash
# GPU call: missed-ray pass (batches): first-launch determinism, full suite, then batches
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2r; mkdir -p $O
python tests/analysis/diag_march_eq.py 2>&1 | grep -v amdgpu.ids | grep "counters\|differs" | tee $O/diag.log
DEBUG=0 python tests/analysis/diag_march_eq.py 32 3 24 32 5 333 2>&1 | grep -v amdgpu.ids | grep "counters\|differs" | tee $O/diag_b3.log
rc=0; python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -12 | tee $O/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
{
run C1 --steps 300
run C1-task --steps 300 --march task
run B2 --steps 100 --batch 2
run B8 --steps 60 --batch 8
run B8d --steps 60 --batch 8 --distinct-triplanes
run B16d --steps 30 --batch 16 --distinct-triplanes
run B64d --steps 8 --batch 64 --distinct-triplanes
run B8-7296 --steps 30 --batch 8 --nc 72 --nf 96
run 7296 --steps 100 --nc 72 --nf 96
} | tee $O/bench.log
