#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/bwd_stats -- python3 $GRAFT_REPO_ROOT/tools/bench_bwd.py > $GRAFT_REPO_ROOT/$O/bwd_stats.log 2>&1
find $GRAFT_REPO_ROOT/$O/bwd_stats -name "*kernel_stats.csv" | while read f; do cut -d, -f1-4 $f | head -12; done
