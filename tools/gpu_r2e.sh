#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2e; mkdir -p $O
python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -40 | tee $O/pytest.log
python bench.py --no-cpu-baseline --steps 300 2>&1 | grep -v amdgpu.ids | python tools/exline.py base | tee $O/bench.log
python bench.py --no-cpu-baseline --steps 300 --mlp-mode f32 --no-p24 2>&1 | grep -v amdgpu.ids | python tools/exline.py base-f32 | tee -a $O/bench.log
timeout -k 10 120 python bench.py --no-cpu-baseline --steps 300 --allow-variant --variant variants/libenarf_vtaps.so 2>&1 | grep -v amdgpu.ids | python tools/exline.py vtaps | tee -a $O/bench.log
