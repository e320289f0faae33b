#!/bin/bash
# GPU call at the end of a round, part 1: full GPU suite, smoke, determinism re-take, 2-rank rehearsal, default bench lines
# (part 2 = tools/gpu_final_profiles.sh: the PMC passes and the sweeps)
set -eo pipefail
ulimit -c 0
O=gpurun_out/final; mkdir -p $O
rc=0; python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1 || rc=$?
grep -v amdgpu.ids $O/pytest_full.log | tail -6 | tee $O/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
# determinism of the product library on the ray subset of profiles/r02_mfma_hazard_fixed.log's last two runs
NRAYS=3000 REPS=4 MODES=f16x3,bf16x3 python tools/tapcheck.py 2>&1 | grep -v amdgpu.ids | tee $O/determinism_nrays3000.log
bash tools/gpu_multi_rehearsal.sh
python bench.py --steps 200 2>/dev/null | tee $O/bench_default.json | python tools/exline.py default
python bench.py --steps 20 --warmup 5 2>/dev/null | tee $O/bench_driver_like.json | python tools/exline.py driver-like-20-steps
python bench.py --train-step --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tee $O/bench_train_c1.json | python tools/exline.py train-C1
python bench.py --train-step --batch 8 --distinct-triplanes --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tee $O/bench_train_b8d.json | python tools/exline.py train-B8d
