#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $1"; env $1 timeout 900 python -m pytest $2 -q 2>&1 | tail -12; }
run "OCC4D_GRID_GAP_FILTER=0" "tests/test_gpu_sampler.py"
run "OCC4D_PAIR_MLP=0 OCC4D_TRAIN_ROWLIN_HALF_CU=0" "tests/test_gpu_training.py"
run "OCC4D_DETERMINISTIC=1" "tests/test_gpu_training.py tests/test_gpu_kernels_random.py"
