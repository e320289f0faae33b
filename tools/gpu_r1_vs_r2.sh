#!/bin/bash
# same-box comparison of the round-1 final tree (variants/r1tree: `git archive df19949`, built there) with the current one
set -eo pipefail
ulimit -c 0
O=gpurun_out/r1r2; mkdir -p $O
for i in 1 2; do
  (cd variants/r1tree && python bench.py --steps 300 --no-cpu-baseline --no-p24 2>/dev/null) | tee $O/r1_$i.json | python tools/exline.py round1-C1
  python bench.py --steps 300 --no-cpu-baseline --no-p24 --no-f32 --no-two-streams 2>/dev/null | tee $O/r2_$i.json | python tools/exline.py round2-C1
done
(cd variants/r1tree && python bench.py --steps 60 --batch 8 --no-cpu-baseline --no-p24 2>/dev/null) | python tools/exline.py round1-B8
python bench.py --steps 60 --batch 8 --no-cpu-baseline --no-p24 --no-f32 --no-two-streams 2>/dev/null | python tools/exline.py round2-B8
(cd variants/r1tree && python bench.py --steps 30 --batch 16 --distinct-triplanes --no-cpu-baseline --no-p24 2>/dev/null) | python tools/exline.py round1-B16d
python bench.py --steps 30 --batch 16 --distinct-triplanes --no-cpu-baseline --no-p24 --no-f32 --no-two-streams 2>/dev/null | python tools/exline.py round2-B16d
(cd variants/r1tree && python bench.py --steps 100 --nc 72 --nf 96 --no-cpu-baseline --no-p24 2>/dev/null) | python tools/exline.py round1-7296
python bench.py --steps 100 --nc 72 --nf 96 --no-cpu-baseline --no-p24 --no-f32 --no-two-streams 2>/dev/null | python tools/exline.py round2-7296
