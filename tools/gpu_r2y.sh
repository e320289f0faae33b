#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2y; mkdir -p $O
python tools/bench_bwd.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/bwd.log
BATCH=8 python tools/bench_bwd.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/bwd.log
python tools/cand_hist.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/cand.log
python tools/bench_mesh.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/mesh.log
