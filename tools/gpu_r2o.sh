#!/bin/bash
# GPU call: row bands across the batch vs image bands (A/B), vectorised re-layout, smoke, C1 counters without the f32 pass
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2o; mkdir -p $O
rc=0; python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_backward.py -m gpu -q -x > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -8 | tee $O/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
var() { echo "--allow-variant --variant variants/libenarf_$1.so"; }
{
run C1 --steps 300
run C1-imgbands --steps 300 $(var imgbands)
run B8 --steps 60 --batch 8
run B8-imgbands --steps 60 --batch 8 $(var imgbands)
run B8d --steps 60 --batch 8 --distinct-triplanes
run B8d-imgbands --steps 60 --batch 8 --distinct-triplanes $(var imgbands)
run B16d --steps 30 --batch 16 --distinct-triplanes
run B16d-imgbands --steps 30 --batch 16 --distinct-triplanes $(var imgbands)
run B32d --steps 15 --batch 32 --distinct-triplanes
run B32d-imgbands --steps 15 --batch 32 --distinct-triplanes $(var imgbands)
run B2 --steps 100 --batch 2
run B2-imgbands --steps 100 --batch 2 $(var imgbands)
} | tee $O/bench.log
bash tools/gpu_pmc.sh
