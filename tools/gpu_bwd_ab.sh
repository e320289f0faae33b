#!/bin/bash
set -eo pipefail
ulimit -c 0
python tools/bench_bwd.py 2>&1 | grep backward | tail -1
for n in "$@"; do echo "== $n"; ENARF_VARIANT=$n timeout -k 10 120 python tools/bench_bwd.py 2>&1 | grep backward | tail -1; done
