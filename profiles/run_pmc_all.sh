#!/bin/bash
# All PMC evidence of a round in one gpurun call: the fused attention kernel of the fp32 default and of both split
# schemes, GREATER and CARLA decode chunks (profiles/run_pmc.sh = separate rocprofv3 --pmc passes, never with a trace
# domain), summarised by profiles/summarize_pmc.py.   bash profiles/run_pmc_all.sh <out_dir under gpurun_out/>
set -u
OUT=${1:?out dir}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
for kind in greater carla; do
  for prec in f32 bf16x6 f16x3; do
    OCC4D_LOGIT_PRECISION=$prec OCC4D_TRUNK_PRECISION=$prec bash "$ROOT/profiles/run_pmc.sh" "$OUT/${kind}_$prec" 32256 $kind
    key=$kind; [ "$prec" != f32 ] && key=${kind}_$prec
    python "$ROOT/profiles/summarize_pmc.py" "$OUT/${kind}_$prec" --json "$OUT/pmc_traffic.json" --kind $key > "$OUT/pmc_summary_${kind}_$prec.txt" 2>&1 || echo "summary ${kind}_$prec failed"
    find "$OUT/${kind}_$prec" -name "*.csv" -size +4M -delete
  done
done
ls -la "$OUT"
