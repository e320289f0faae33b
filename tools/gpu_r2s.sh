#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2s; mkdir -p $O
ENARF_VARIANT=csplit DEBUG=0 python tests/analysis/diag_march_eq.py 32 3 24 32 5 333 2>&1 | grep -v amdgpu.ids | grep "counters" | tee $O/diag_b3.log
