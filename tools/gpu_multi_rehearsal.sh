#!/bin/bash
# N = 2 rehearsal of bench.py on the 1-GPU box: two ranks share the card, torch.distributed over gloo (the real N > 1 runs
# are the driver's, on RCCL). Forward (C3 shape, 2 x 4 frames here, with the one-GPU reference of the same job), --train-step
# (two micro-batches, asynchronous bucketed gradient all-reduce) and --shard-frame.
set -eo pipefail
O=gpurun_out/multi; mkdir -p $O
export ENARF_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 --batch 4 2>/dev/null | grep '^{' | tee $O/n2_forward.json | python tools/exline.py n2-fwd
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --warmup 1 --batch 2 --train-step 2>/dev/null | grep '^{' | tee $O/n2_train_step.json | python tools/exline.py n2-train
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 5 --warmup 2 --batch 1 --shard-frame 2>/dev/null | grep '^{' | tee $O/n2_shard_frame.json | python tools/exline.py n2-shard
