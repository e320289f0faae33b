#!/bin/bash
# first GPU contact: parity tests
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -40
