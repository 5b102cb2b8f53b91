#!/bin/bash
# one training-path iteration on the GPU box: tests of the touched kernels, bench_train eager + replayed, kernel stats
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04b; mkdir -p $OUT
python -m pytest tests/test_gpu_kernels_random.py tests/test_gpu_training.py tests/test_gpu_contracts.py -q -x 2>&1 | tail -3
python bench_train.py --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_train.json
python bench_train.py --steps 5 --warmup 2 --graph 2>/dev/null | tail -1 >> $OUT/bench_train.json
cut -c1-330 $OUT/bench_train.json
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof3 -- python bench_train.py --steps 4 --warmup 2 > /dev/null 2>&1
cp "$(find $OUT/prof3 -name '*kernel_stats.csv' | head -1)" $OUT/train_kernel_stats.csv
rm -rf $OUT/prof3
