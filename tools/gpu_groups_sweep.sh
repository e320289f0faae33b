#!/bin/bash
# batches with one tri-plane per frame: one launch against groups of frames (enarf_render_args.group_frames) -> profiles/r03_groups_sweep.log
set -o pipefail
B="python bench.py --no-cpu-baseline --no-p24 --no-f32 --no-two-streams --distinct-triplanes"
$B --steps 40 --batch 8 | python tools/exline.py B8d
$B --steps 30 --batch 16 --group-frames 64 | python tools/exline.py B16d-1launch
$B --steps 30 --batch 16 | python tools/exline.py B16d-g8
$B --steps 8 --batch 64 --group-frames 64 | python tools/exline.py B64d-1launch
$B --steps 8 --batch 64 | python tools/exline.py B64d-g8
$B --steps 8 --batch 64 --group-frames 4 | python tools/exline.py B64d-g4
