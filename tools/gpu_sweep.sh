#!/bin/bash
# the numbers quoted in DESIGN.md / BASELINE.md: one line per configuration
set -eo pipefail
ulimit -c 0
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams"
$B --steps 300 | python tools/exline.py C1
$B --steps 300 --unfused | python tools/exline.py C1-unfused
$B --steps 300 --cache-triplane | python tools/exline.py C1-cached-planes
$B --steps 300 --origin center+head | python tools/exline.py C1-P24
$B --steps 50 --batch 8 | python tools/exline.py C1x8
$B --steps 50 --batch 8 --distinct-triplanes | python tools/exline.py C3-share-B8d
$B --steps 30 --batch 16 --distinct-triplanes | python tools/exline.py C2-like-B16
$B --steps 8 --batch 64 --distinct-triplanes | python tools/exline.py C3-whole-B64d
$B --steps 100 --nc 72 --nf 96 | python tools/exline.py Nc72-Nf96
$B --steps 20 --size 256 --batch 8 --nc 72 --nf 96 --mlp-mode bf16 --early-stop-eps 1e-3 | python tools/exline.py C4-like
$B --steps 100 --size 64 --nf 32 | python tools/exline.py C0
python tools/bench_bwd.py 2>&1 | grep -v amdgpu.ids | tail -1
BATCH=8 python tools/bench_bwd.py 2>&1 | grep -v amdgpu.ids | tail -1
python tools/bench_sampler.py 2>&1 | grep -v amdgpu.ids | tail -4
python tools/bench_mesh.py 2>&1 | grep -v amdgpu.ids | tail -1
python tools/bench_warp.py 2>&1 | grep -v amdgpu.ids | tail -2
