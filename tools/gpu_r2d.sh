#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2d; mkdir -p $O
run() { echo "=== $*"; env "$@" timeout -k 10 200 python tools/tapcheck.py 2>&1 | grep -v amdgpu.ids; }
{ run ENARF_VARIANT=tapcheck2 REPS=4; run ENARF_VARIANT=tapcheck2 REPS=3 NRAYS=3000 MODES=f16x3,bf16x3; } | tee $O/tapcheck2.log
if grep -q "NOT deterministic\|VIOLATIONS [1-9]" $O/tapcheck2.log; then echo "STILL BROKEN - stopping"; exit 0; fi
echo "=== clean: in-tree bench (clamped taps), then the valid-taps variant" | tee -a $O/tapcheck2.log
python bench.py --no-cpu-baseline --steps 300 2>&1 | grep -v amdgpu.ids | python tools/exline.py base | tee $O/bench.log
python bench.py --no-cpu-baseline --steps 300 --mlp-mode f32 --no-p24 2>&1 | grep -v amdgpu.ids | python tools/exline.py base-f32 | tee -a $O/bench.log
timeout -k 10 120 python bench.py --no-cpu-baseline --steps 300 --allow-variant --variant variants/libenarf_vtaps2.so 2>&1 | grep -v amdgpu.ids | python tools/exline.py vtaps2 | tee -a $O/bench.log
