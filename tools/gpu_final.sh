#!/bin/bash
# GPU call at the end of a round: full GPU suite, smoke, kernel stats + PMC passes of the default workload, 2-rank rehearsal, default bench lines
set -eo pipefail
ulimit -c 0
O=gpurun_out/final; mkdir -p $O
rc=0; python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -6 | tee $O/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/gpu_pmc.sh
bash tools/gpu_multi_rehearsal.sh
python bench.py --steps 200 2>/dev/null | tee $O/bench_default.json | python tools/exline.py default
python bench.py --steps 20 --warmup 5 2>/dev/null | tee $O/bench_driver_like.json | python tools/exline.py driver-like-20-steps
