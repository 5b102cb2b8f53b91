#!/bin/bash
# bf16x6 attention: parity gates + step time + kernel time (GPU box): bash profiles/x6_check.sh [pmc]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_regimes.py -q -k "bf16x6 or three_way" 2>&1 | tail -3
OCC4D_LOGIT_PRECISION=bf16x6 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('bf16x6 ms/step', round(l['ms_per_step'],2), 'q/s', round(l['value']), 'launch ms', round(l['roofline']['avg_launch_ms'],4))"
if [ "${1:-}" = pmc ]; then
  OCC4D_LOGIT_PRECISION=bf16x6 bash profiles/run_pmc.sh gpurun_out/pmc_x6 32256 greater > /dev/null 2>&1
  python profiles/summarize_pmc.py gpurun_out/pmc_x6 | grep -A22 "cross_attn_bf16x6" | grep -E "INSTS_VALU|INSTS_MFMA|WAIT|->"
fi
