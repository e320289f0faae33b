#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2v; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
{
run default --steps 400
run cache-triplane --steps 400 --cache-triplane
run default2 --steps 400
run cache-triplane2 --steps 400 --cache-triplane
run unfused --steps 400 --unfused
run streams2 --steps 400 --streams 2
} | tee $O/bench.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_unfused -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-p24 --no-f32 --unfused > $GRAFT_REPO_ROOT/$O/stats_unfused.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_cache -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-p24 --no-f32 --cache-triplane > $GRAFT_REPO_ROOT/$O/stats_cache.log 2>&1
find $GRAFT_REPO_ROOT/$O -name "*kernel_stats.csv" | while read f; do echo $f; cut -d, -f1-4 $f | head -6; done
