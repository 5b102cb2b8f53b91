#!/bin/bash
# The A/B switches of INTEGRATION.md section G, each flipped once over the tests that exercise its path (GPU box).
cd "$(dirname "$0")/.."
run() { echo "== $1"; env $1 timeout 900 python -m pytest $2 -x -q 2>&1 | tail -2; }
run "OCC4D_NESTED_FPS=0 OCC4D_POOL_FROM_SELF_KNN=0" "tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fps_pruned.py"
run "OCC4D_SORTED_SCATTER=0 OCC4D_SOFTMAX_BWD4=0 OCC4D_ROW_BALANCE=0" "tests/test_gpu_training.py tests/test_gpu_kernels_random.py tests/test_gpu_contracts.py"
run "OCC4D_SAMPLER_FAST=0" "tests/test_gpu_sampler.py"
run "OCC4D_GRID_GAP_FILTER=0" "tests/test_gpu_sampler.py"
run "OCC4D_TRUNK4=1" "tests/test_gpu_parity.py tests/test_gpu_trunk.py tests/test_gpu_fullsize.py"
run "OCC4D_PAIR_MLP=0 OCC4D_TRAIN_ROWLIN_HALF_CU=0" "tests/test_gpu_training.py"
run "OCC4D_DETERMINISTIC=1" "tests/test_gpu_training.py"
run "OCC4D_FPS_PRUNE=0" "tests/test_gpu_parity.py tests/test_gpu_fps_pruned.py"
run "OCC4D_DECODE_STREAMS=1 OCC4D_PINNED_HOST_IO=0" "tests/test_gpu_parity.py tests/test_gpu_postops.py"
run "OCC4D_LOGIT_PRECISION=bf16x6 OCC4D_TRUNK_PRECISION=bf16x6" "tests/test_gpu_fullsize.py"
run "OCC4D_KNN_GRID=0" "tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_contracts.py"
run "OCC4D_KNN_GRID_MIN_PAIRS=1 OCC4D_KNN_GRID_MIN_DATA=1" "tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_training.py tests/test_gpu_sampler.py"
run "OCC4D_KNN_GRID_MIN_PAIRS=1 OCC4D_KNN_GRID_MIN_DATA=1 OCC4D_KNN_GRID_ONE_THREAD_FROM=1" "tests/test_gpu_parity.py tests/test_gpu_fullsize.py"
run "OCC4D_GRADIENT_OVERLAP=0" "tests/test_gpu_training.py tests/test_gpu_contracts.py"
# (split-precision training: the strict tests against the ORACLE; the fp32-twin comparison of the fused / unfused pair chain is
#  not in this list -- two fp32-class paths put a hidden unit of +-2.4e-7 on different sides of its ReLU there, the audited
#  kink effect of DESIGN.md 7b, one entry of one gradient)
echo "== OCC4D_TRAIN_PRECISION=bf16x6 OCC4D_LOGIT_PRECISION=bf16x6"; OCC4D_TRAIN_PRECISION=bf16x6 OCC4D_LOGIT_PRECISION=bf16x6 timeout 900 python -m pytest tests/test_gpu_training.py -x -q -k "split_precision or decoder_gradients or end_to_end or checkpointed or encoder_gradients or reduces" 2>&1 | tail -2
