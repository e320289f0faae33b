#!/bin/bash
# GPU call: full suite (test failures do not stop the measurements, a crashed process does), TA microbenchmark,
# task march vs v2 march vs valid-taps on several workloads, default bench line
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2j; mkdir -p $O
rc=0; python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -40 | tee $O/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
timeout -k 10 120 variants/ub_ta 2>&1 | tee $O/ub_ta.txt
B="python bench.py --no-cpu-baseline --no-p24 --no-f32"
V="--allow-variant --variant variants/libenarf_v2march.so"
T="--allow-variant --variant variants/libenarf_vtaps.so"
{
$B --steps 300 2>&1 | grep -v amdgpu.ids | python tools/exline.py task-C1
$B --steps 300 $V 2>&1 | grep -v amdgpu.ids | python tools/exline.py v2-C1
$B --steps 300 $T 2>&1 | grep -v amdgpu.ids | python tools/exline.py vtaps-C1
$B --steps 60 --batch 8 2>&1 | grep -v amdgpu.ids | python tools/exline.py task-B8
$B --steps 60 --batch 8 $V 2>&1 | grep -v amdgpu.ids | python tools/exline.py v2-B8
$B --steps 30 --batch 16 --distinct-triplanes 2>&1 | grep -v amdgpu.ids | python tools/exline.py task-B16d
$B --steps 30 --batch 16 --distinct-triplanes $V 2>&1 | grep -v amdgpu.ids | python tools/exline.py v2-B16d
$B --steps 100 --nc 72 --nf 96 2>&1 | grep -v amdgpu.ids | python tools/exline.py task-7296
$B --steps 100 --nc 72 --nf 96 $V 2>&1 | grep -v amdgpu.ids | python tools/exline.py v2-7296
$B --steps 100 --size 64 --nf 32 2>&1 | grep -v amdgpu.ids | python tools/exline.py task-C0
$B --steps 100 --size 64 --nf 32 $V 2>&1 | grep -v amdgpu.ids | python tools/exline.py v2-C0
} | tee $O/bench.log
python bench.py --steps 200 2>/dev/null | tee $O/bench_default.json | python tools/exline.py default
