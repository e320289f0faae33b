#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2s; mkdir -p $O
python tools/diag_march_eq.py 2>&1 | grep -v amdgpu.ids | grep -v "differs in\|bins differ" | tee $O/diag.log
ENARF_VARIANT=imgbands python tools/diag_march_eq.py 2>&1 | grep -v amdgpu.ids | grep -v "differs in\|bins differ" | tee $O/diag_prev.log
ENARF_VARIANT=imgbands python tools/diag_march_eq.py 2>&1 | grep -v amdgpu.ids | grep "tap\|differs" | head -5
