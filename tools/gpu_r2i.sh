#!/bin/bash
set -eo pipefail
ulimit -c 0
O=gpurun_out/r2i; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -30 | tee $O/pytest.log
B="python bench.py --no-cpu-baseline --no-p24 --no-f32"
V="--allow-variant --variant variants/libenarf_v2march.so"
{
$B --steps 300 2>&1 | grep -v amdgpu.ids | python tools/exline.py task-C1
$B --steps 300 $V 2>&1 | grep -v amdgpu.ids | python tools/exline.py v2-C1
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
