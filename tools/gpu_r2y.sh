#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2y; mkdir -p $O
python tools/bench_mesh.py 2>&1 | grep -v amdgpu.ids | tail -2
OFFSET=100 python tools/bench_mesh.py 2>&1 | grep -v amdgpu.ids | tail -2
OFFSET=1.5 python tools/bench_mesh.py 2>&1 | grep -v amdgpu.ids | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/mesh_stats -- python3 $GRAFT_REPO_ROOT/tools/bench_mesh.py > $GRAFT_REPO_ROOT/$O/mesh_stats.log 2>&1
find $GRAFT_REPO_ROOT/$O/mesh_stats -name "*kernel_stats.csv" | while read f; do cut -d, -f1-7 $f | head -5; done
