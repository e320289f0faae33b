#!/bin/bash
# backward A/B on the GPU box: the in-tree library, then every variant named (tools/build_variant.sh NAME ...)
# usage: tools/gpu_bwd_ab.sh [NAME ...]   env BATCH / DISTINCT / SIZE / NF as tools/bench_bwd.py
set -eo pipefail
ulimit -c 0
echo "== in-tree"; timeout -k 10 300 python tools/bench_bwd.py 2>&1 | grep -v amdgpu.ids | tail -3
for n in "$@"; do echo "== $n"; ENARF_VARIANT=$n timeout -k 10 300 python tools/bench_bwd.py 2>&1 | grep -v amdgpu.ids | tail -3; done
