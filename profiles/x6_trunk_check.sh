#!/bin/bash
# bf16x6 trunk + attention: parity gates, step time, per-kernel time (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_regimes.py -q -k "split_precision_rowlin or entirely or bf16x6" 2>&1 | tail -2
OCC4D_LOGIT_PRECISION=bf16x6 OCC4D_TRUNK_PRECISION=bf16x6 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('x6 all: ms/step', round(l['ms_per_step'],2), 'q/s', round(l['value']))"
OCC4D_LOGIT_PRECISION=bf16x6 OCC4D_TRUNK_PRECISION=bf16x6 OCC4D_DECODE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x6prof4 -- python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > /dev/null 2>&1
head -5 $(find gpurun_out/x6prof4 -name "*kernel_stats.csv" | head -1) | cut -c1-150
rm -rf gpurun_out/x6prof4
