#!/bin/bash
# Regenerates the round's evidence files on the GPU box (one gpurun call, ~12 min):
#   bash profiles/collect_evidence.sh <out_dir under gpurun_out/>
# then copy <out_dir>/* into profiles/ under the round's prefix.  PMC passes are separate: profiles/run_pmc.sh.
# Every step runs under `set -euo pipefail` in its own subshell: a step that fails is reported in <out_dir>/FAILED.txt, its
# (partial or empty) output file is removed -- no 0-byte evidence -- and the script ends with a non-zero status.
set -uo pipefail
OUT=${1:?out dir}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
: > "$OUT/FAILED.txt"
step() {   # step <output file (relative to OUT) or -> <command string>
  local out=$1; shift
  if ! ( set -euo pipefail; eval "$*" ); then
    echo "FAILED: $*" | tee -a "$OUT/FAILED.txt" >&2
    [ "$out" != "-" ] && rm -f "$OUT/$out"
  elif [ "$out" != "-" ] && [ ! -s "$OUT/$out" ]; then
    echo "EMPTY OUTPUT: $out <- $*" | tee -a "$OUT/FAILED.txt" >&2
    rm -f "$OUT/$out"
  fi
}
step bench.json 'python bench.py --steps 20 --warmup 5 2> "$OUT/bench.err" | tail -1 > "$OUT/bench.json"'
step bench_carla.json 'python bench.py --kind carla --steps 10 --warmup 3 --no-secondary 2>> "$OUT/bench.err" | tail -1 > "$OUT/bench_carla.json"'
step bench_kernel_stats_streams1.csv 'OCC4D_DECODE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof1" -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > "$OUT/bench_streams1_under_rocprof.json"; cp "$(find "$OUT/prof1" -name "*kernel_stats.csv" | head -1)" "$OUT/bench_kernel_stats_streams1.csv"'
step bench_kernel_stats.csv 'rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof2" -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > /dev/null 2>&1; cp "$(find "$OUT/prof2" -name "*kernel_stats.csv" | head -1)" "$OUT/bench_kernel_stats.csv"'
step bench_kernel_stats_f16x3_streams1.csv 'OCC4D_LOGIT_PRECISION=f16x3 OCC4D_TRUNK_PRECISION=f16x3 OCC4D_DECODE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof7" -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > "$OUT/bench_f16x3_streams1_under_rocprof.json"; cp "$(find "$OUT/prof7" -name "*kernel_stats.csv" | head -1)" "$OUT/bench_kernel_stats_f16x3_streams1.csv"'
step bench_kernel_stats_bf16x6_streams1.csv 'OCC4D_LOGIT_PRECISION=bf16x6 OCC4D_TRUNK_PRECISION=bf16x6 OCC4D_DECODE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof6" -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > "$OUT/bench_bf16x6_streams1_under_rocprof.json"; cp "$(find "$OUT/prof6" -name "*kernel_stats.csv" | head -1)" "$OUT/bench_kernel_stats_bf16x6_streams1.csv"'
: > "$OUT/bench_train.jsonl"
for flags in "" "--no-checkpoint" "--precision bf16x6" "--precision bf16x6 --no-checkpoint"; do
  step - "python bench_train.py --steps 20 --warmup 3 $flags 2>/dev/null | tail -1 >> \"$OUT/bench_train.jsonl\""
done
: > "$OUT/bench_train_sampler.jsonl"
for flags in "" "--sampler" "--sampler --sampler-serial"; do
  step - "python bench_train.py --steps 20 --warmup 3 $flags 2>/dev/null | tail -1 >> \"$OUT/bench_train_sampler.jsonl\""
done
step time_pair_mlp.txt 'python profiles/time_pair_mlp.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/time_pair_mlp.txt"'
step time_wgrad.txt 'python profiles/time_wgrad.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/time_wgrad.txt"'
step profile_sampler.txt 'python profiles/profile_sampler.py 2>/dev/null | head -3 > "$OUT/profile_sampler.txt"'
step train_kernel_stats.csv 'rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof3" -- python bench_train.py --steps 4 --warmup 2 > /dev/null 2>&1; cp "$(find "$OUT/prof3" -name "*kernel_stats.csv" | head -1)" "$OUT/train_kernel_stats.csv"; python profiles/train_timeline.py "$(find "$OUT/prof3" -name "*kernel_trace.csv" | head -1)" 30 - 1 > "$OUT/train_timeline.txt"'
step train_phases.txt 'python profiles/train_phases.py 10 2>/dev/null | grep -v amdgpu.ids > "$OUT/train_phases.txt"'
step train_shapes.txt 'python profiles/train_shapes.py 2>/dev/null > "$OUT/train_shapes.txt"'
step time_rowlin_tail.txt 'python profiles/time_rowlin_tail.py 2>/dev/null > "$OUT/time_rowlin_tail.txt"'
step time_fps.txt 'python profiles/time_fps.py 2>/dev/null > "$OUT/time_fps.txt"; OCC4D_FPS_PRUNE=0 python profiles/time_fps.py 2>/dev/null >> "$OUT/time_fps.txt"'
step fps_stamps.txt 'python profiles/stamp_fps.py 14336 4779 2>/dev/null > "$OUT/fps_stamps.txt"; python profiles/stamp_fps.py 4779 1593 2>/dev/null >> "$OUT/fps_stamps.txt"'
step x6_stamps.txt '(cd profiles && python stamp_x6.py 2>/dev/null | tail -2) > "$OUT/x6_stamps.txt"'
step encode_timeline.txt 'rocprofv3 --kernel-trace --output-format csv -d "$OUT/enc" -- python profiles/probe.py encode 4 > /dev/null 2>&1; python profiles/encode_timeline.py "$(find "$OUT/enc" -name "*kernel_trace.csv" | head -1)" 6 > "$OUT/encode_timeline.txt"'
step host_boundary_probe.txt 'python profiles/host_boundary_probe.py 6 2>/dev/null | grep -v amdgpu.ids > "$OUT/host_boundary_probe.txt"'
step attn_split_ablations.txt 'python profiles/time_attn_split.py f16x3 bf16x6 -DOCC4D_XA_ABL_NOSPLIT -DOCC4D_XA_ABL_NOGEMM1 -DOCC4D_XA_ABL_NOINIT -DOCC4D_XA_ABL_NODMA -DOCC4D_XA_ABL_NOLDS -DOCC4D_XA_ABL_NODEP -DOCC4D_XA_ABL_NOLDS,-DOCC4D_XA_ABL_NODEP,-DOCC4D_XA_ABL_NOSPLIT,-DOCC4D_XA_ABL_NOINIT,-DOCC4D_XA_ABL_NODMA 2>/dev/null | grep -v amdgpu.ids > "$OUT/attn_split_ablations.txt"'
step forced_dist_rccl.log '( OCC4D_FORCE_DIST=1 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline; OCC4D_FORCE_DIST=1 NCCL_DEBUG=INFO python bench_train.py --steps 3 --warmup 1 ) > "$OUT/forced_dist_rccl.log" 2>&1'
rm -rf "$OUT/prof1" "$OUT/prof2" "$OUT/prof3" "$OUT/prof6" "$OUT/prof7" "$OUT/enc"
ls -la "$OUT"
if [ -s "$OUT/FAILED.txt" ]; then echo "some steps FAILED:"; cat "$OUT/FAILED.txt"; exit 1; fi
rm -f "$OUT/FAILED.txt"
