#!/bin/bash
# kernel durations of the 2-D GAN ops (tools/run_gan2d_ops.py) -> gpurun_out/gan2d/prof/*_kernel_stats.csv
set -eo pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/gan2d/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/run_gan2d_ops.py > $out/run.log 2>&1 || true
f=$(ls $out/*/*kernel_stats.csv | head -1)
cut -d, -f1-7 $f | head -12
