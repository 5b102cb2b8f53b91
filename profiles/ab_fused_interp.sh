#!/bin/bash
# A/B of the lin_z table term: stand-alone interp_add4 pass (default) vs fused into the residual block's epilogue
# (OCC4D_FUSED_INTERP=1 -> OCC4D_PATH_FUSED_INTERP).  Usage (GPU box): bash profiles/ab_fused_interp.sh <out_dir>
set -euo pipefail
OUT=${1:?out dir}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
for mode in 0 1; do
  export OCC4D_FUSED_INTERP=$mode
  python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_fused$mode.json"
  OCC4D_DECODE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof$mode" -- \
      python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > /dev/null 2>&1
  cp "$(find "$OUT/prof$mode" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_fused$mode.csv"
  rm -rf "$OUT/prof$mode"
  ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE \
      --output-format csv -d "$OUT/pmc$mode" -o pmc -- python "$ROOT/profiles/probe.py" decode 32256 2 greater > "$OUT/pmc$mode.log" 2>&1 ) || echo "pmc pass $mode failed"
done
unset OCC4D_FUSED_INTERP
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
for mode in (0, 1):
    b = json.load(open('%s/bench_fused%d.json' % (out, mode)))
    print('mode %d: %.2f ms/step  %.0f q/s' % (mode, b['ms_per_step'], b['value']))
    for r in csv.DictReader(open('%s/kernel_stats_fused%d.csv' % (out, mode))):
        if any(k in r['Name'] for k in ('resblock_kernel', 'interp_add4', 'rowlin_kernel')):
            print('   %-60s calls %5s avg %9.1f us total %8.2f ms' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob('%s/pmc%d/**/*counter_collection.csv' % (out, mode), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            if 'resblock_kernel' in name or 'interp_add4' in name:
                key = 'resblock' if 'resblock' in name else 'interp_add4'
                acc[key][r['Counter_Name']] += float(r['Counter_Value'])
                acc[key]['_launches_' + r['Counter_Name']] += 1
    for key, c in acc.items():
        n = c.get('_launches_SQ_WAVE_CYCLES', 1)
        print('   PMC %-12s per launch:' % key, {k: round(v / n) for k, v in c.items() if not k.startswith('_')})
PY
