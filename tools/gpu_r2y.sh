#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2y; mkdir -p $O
rc=0; python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -k "sampler or sampling" > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -4
if [ $rc -gt 1 ]; then exit $rc; fi
python tools/bench_sampler.py 2>&1 | grep -v amdgpu.ids | tail -4
