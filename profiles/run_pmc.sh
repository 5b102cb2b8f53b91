#!/bin/bash
# PMC passes of one decoder mini-batch (rocprofv3 --pmc, one pass per counter group; never combined with a trace
# domain).  Usage (GPU box, from the repo root):  bash profiles/run_pmc.sh <out_dir> [n_queries] [kind]
# Writes <out_dir>/<group>/..._counter_collection.csv; summarise with profiles/summarize_pmc.py <out_dir>.
set -u
OUT=${1:?out dir}
NQ=${2:-32256}
KIND=${3:-greater}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -o pmc -- \
      python "$ROOT/profiles/probe.py" decode "$NQ" 2 "$KIND" > "$OUT/$name.log" 2>&1 || echo "pass $name failed (see $OUT/$name.log)"
done
