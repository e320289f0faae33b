#!/bin/bash
# GPU call at the end of a round, part 2: kernel stats + PMC passes of the default workload (march), of 16 frames with
# per-frame tri-planes marched in groups, and of the backward; then the sweeps quoted in DESIGN.md / BASELINE.md
ulimit -c 0
bash tools/gpu_pmc.sh pmc_final > gpurun_out/pmc_final.log 2>&1
bash tools/gpu_pmc.sh pmc_b16d --batch 16 --distinct-triplanes > gpurun_out/pmc_b16d.log 2>&1
bash tools/gpu_bwd_pmc.sh pmc_bwd_final > gpurun_out/pmc_bwd_final.log 2>&1
bash tools/gpu_sweep.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sweep.log
BATCH=8 DISTINCT=1 python tools/bench_bwd.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/sweep.log
SIZE=256 BATCH=2 DISTINCT=1 NC=72 NF=96 python tools/bench_bwd.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/sweep.log
