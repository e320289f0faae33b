#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2t; mkdir -p $O
rc=0; python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -6 | tee $O/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
BATCH=8 python tools/bench_bwd.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/bwd.log
