#!/bin/bash
# round 2, first GPU call: suite + A/B of the round-1-equivalent build vs the trimmed round + phase timers
set -o pipefail
ulimit -c 0
O=gpurun_out/r2a; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.log
bash tools/gpu_ab.sh r1eq gtaps 2>&1 | tee $O/ab.log
ENARF_VARIANT=timers1 python tools/timers.py 2>&1 | tee $O/timers1.log
TIMERS=4 ENARF_VARIANT=timers4 python tools/timers.py 2>&1 | tee $O/timers4.log
TIMERS=3 ENARF_VARIANT=timers3 python tools/timers.py 2>&1 | tee $O/timers3.log
