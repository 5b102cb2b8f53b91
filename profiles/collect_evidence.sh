#!/bin/bash
# Regenerates the round's evidence files on the GPU box (one gpurun call, ~12 min):
#   bash profiles/collect_evidence.sh <out_dir under gpurun_out/>
# then copy <out_dir>/* into profiles/ under the round's prefix.  PMC passes are separate: profiles/run_pmc.sh.
set -u
OUT=${1:?out dir}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
python bench.py --steps 20 --warmup 5 2> "$OUT/bench.err" | tail -1 > "$OUT/bench.json"
python bench.py --kind carla --steps 10 --warmup 3 2>> "$OUT/bench.err" | tail -1 > "$OUT/bench_carla.json"
OCC4D_DECODE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof1" -- \
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > "$OUT/bench_streams1_under_rocprof.json"
cp "$(find "$OUT/prof1" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_kernel_stats_streams1.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof2" -- \
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > /dev/null 2>&1
cp "$(find "$OUT/prof2" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_kernel_stats.csv"
: > "$OUT/bench_train.jsonl"
for flags in "" "--graph" "--no-checkpoint" "--graph --no-checkpoint"; do
  python bench_train.py --steps 20 --warmup 3 $flags 2>/dev/null | tail -1 >> "$OUT/bench_train.jsonl"
done
OCC4D_PAIR_MLP=0 OCC4D_TRAIN_ROWLIN_HALF_CU=0 python bench_train.py --steps 20 --warmup 3 --graph 2>/dev/null | tail -1 >> "$OUT/bench_train.jsonl"
: > "$OUT/bench_train_sampler.jsonl"
for flags in "--graph" "--sampler --graph" "--sampler" "--sampler --sampler-serial"; do
  python bench_train.py --steps 20 --warmup 3 $flags 2>/dev/null | tail -1 >> "$OUT/bench_train_sampler.jsonl"
done
python profiles/time_pair_mlp.py > "$OUT/time_pair_mlp.txt" 2>/dev/null
python profiles/time_wgrad.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/time_wgrad.txt"
( echo "16-byte-lane kernel (default):"; python profiles/time_softmax_bwd.py 2>/dev/null | grep -v amdgpu.ids;
  echo "scalar kernel (OCC4D_SOFTMAX_BWD4=0):"; OCC4D_SOFTMAX_BWD4=0 python profiles/time_softmax_bwd.py 2>/dev/null | grep -v amdgpu.ids ) > "$OUT/time_softmax_bwd.txt"
python profiles/profile_sampler.py 2>/dev/null | head -3 > "$OUT/profile_sampler.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof3" -- python bench_train.py --steps 4 --warmup 2 > /dev/null 2>&1
cp "$(find "$OUT/prof3" -name '*kernel_stats.csv' | head -1)" "$OUT/train_kernel_stats.csv"
python profiles/train_shapes.py > "$OUT/train_shapes.txt" 2>/dev/null
python profiles/time_rowlin_tail.py > "$OUT/time_rowlin_tail.txt" 2>/dev/null
python profiles/time_fps.py > "$OUT/time_fps.txt" 2>/dev/null
OCC4D_FPS_PRUNE=0 python profiles/time_fps.py >> "$OUT/time_fps.txt" 2>/dev/null
python profiles/stamp_fps.py 14336 4779 > "$OUT/fps_stamps.txt" 2>/dev/null
python profiles/stamp_fps.py 4779 1593 >> "$OUT/fps_stamps.txt" 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d "$OUT/enc" -- python profiles/probe.py encode 4 > /dev/null 2>&1
python profiles/encode_timeline.py "$(find "$OUT/enc" -name '*kernel_trace.csv' | head -1)" 6 > "$OUT/encode_timeline.txt"
( OCC4D_FORCE_DIST=1 NCCL_DEBUG=INFO python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline;
  OCC4D_FORCE_DIST=1 NCCL_DEBUG=INFO python bench_train.py --steps 3 --warmup 1 ) > "$OUT/forced_dist_rccl.log" 2>&1
rm -rf "$OUT/prof1" "$OUT/prof2" "$OUT/prof3" "$OUT/enc"
ls -la "$OUT"
