#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2w; mkdir -p $O
rc=0; python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -8 | tee $O/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams"
run() { name=$1; shift; $B "$@" 2>&1 | grep -v amdgpu.ids | python tools/exline.py $name; }
var() { echo "--allow-variant --variant variants/libenarf_$1.so"; }
{
run lanes8 --steps 400
run lanes4 --steps 400 $(var setup4)
run lanes8b --steps 400
run lanes4b --steps 400 $(var setup4)
run B8d-lanes8 --steps 60 --batch 8 --distinct-triplanes
run B8d-lanes4 --steps 60 --batch 8 --distinct-triplanes $(var setup4)
} | tee $O/bench.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-p24 --no-f32 --no-two-streams > $GRAFT_REPO_ROOT/$O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_unfused -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-p24 --no-f32 --no-two-streams --unfused > $GRAFT_REPO_ROOT/$O/stats_unfused.log 2>&1
find $GRAFT_REPO_ROOT/$O -name "*kernel_stats.csv" | while read f; do cut -d, -f1-4 $f | head -5; done
