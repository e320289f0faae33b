#!/bin/bash
# A/B: bench.py with every variants/libenarf_<name>.so named on the command line (and the in-tree build first)
# BENCH_ARGS: extra bench flags
ulimit -c 0
python bench.py --no-cpu-baseline --steps 300 $BENCH_ARGS | python tools/exline.py base
for n in "$@"; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 300 --allow-variant --variant variants/libenarf_$n.so $BENCH_ARGS | python tools/exline.py $n || exit 1
done
