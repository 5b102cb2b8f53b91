#!/usr/bin/env python
"""Benchmark of the occlusions-4d hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Either launched by `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...` (RANK / WORLD_SIZE in the environment), or from a bare shell: without WORLD_SIZE the script
re-launches itself under torch.distributed.run on 127.0.0.1 (and exits with a clear message when fewer than N GPUs
are visible).

One "step" = one perform_inference unit of work: ONE encode of the
(1, 14336, 8) point-cloud video + decode of every grid query point of one output frame,
inputs (point cloud, query grid, weights) already resident in HBM, outputs left in HBM.

Workload (BASELINE.json configs[1], "GREATER inference 1xMI355X"): n_points 14336,
video_len 12, num_sample 524288 -> 534 528 grid queries, implicit_batch_size 32768,
fp32, synthetic data + seeded random-init weights of the published architecture.
N > 1 GPUs run BASELINE.json configs[3] ("GREATER dense grid 2 M queries sharded across
8 x MI355X"): num_sample 2097152 -> 2 125 568 grid queries, rank 0 encodes and broadcasts the
abstract cloud, every rank decodes a contiguous 1/N slice (265 696 queries per GPU at N = 8):
the total work is the same for N = 2, 4, 8 -> "scaling": "strong".  value = total queries /
max-over-ranks time.  Each line also carries the other grid as a secondary leg, so that both
strong-scaling curves have all their points: `config4_single_gpu` on the N = 1 line (the 2 M
grid on one GPU) and `strong_config2` on the N > 1 lines (the 534 528-query grid split N ways,
where the serial 7.6 ms encode is the Amdahl term).

Extra objects on the JSON line:
  roofline      the dominant kernel = cross_attn16p_kernel (fused vector attention, 14
                neighbours, D = 416), timed with HIP events on its launch stream.  achieved / frac
                = the FLOP the kernel EXECUTES (after the exact-in-R refactoring of DESIGN.md 4,
                counted once: 2 * 14 * (32*832 + 832*416 + 32*416) per query) / time, against the
                157.3 TFLOP/s fp32 MFMA peak: <= 1 by construction.  achieved_as_written /
                frac_as_written = the FLOP of the reference ops it replaces (per
                pair 3*32 + 32*H + 2H*H + 2H*H MAC) / the same time: may exceed 1, because the
                refactoring removes 45 % of that work.  traffic = HBM bytes per launch from the
                rocprofv3 PMC passes committed under profiles/ (file named in traffic_source).
  cpu_baseline  the CPU oracle (oracle/path.py = the reference's PyTorch-CPU op sequence)
                timed on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import contextlib
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import occlusions4d_amd as pk  # noqa: E402

FP32_MFMA_PEAK = 157.3e12
N_POINTS, VIDEO_LEN, BATCH = 14336, 12, int(os.environ.get('OCC4D_BENCH_BATCH', '32768'))   # BASELINE: 32768
NUM_SAMPLE = 524288            # configs[1] / configs[2]: -> 534 528 (GREATER) / 541 314 (CARLA) grid queries
NUM_SAMPLE_DENSE = 2097152     # configs[3]: -> 2 125 568 grid queries, sharded over the GPUs of the node
SEED = 1830


def as_written_flops(n_queries, n_calls, m_abstract, g_out):
    """Matmul FLOPs of the reference's op sequence (SURVEY.md §8(d) closed form)."""
    H, E, P, B, L, K, KL = 416, 288, 68, 6, 2, 14, 8
    per_query = 2 * (P * H + B * 3 * H * H + L * (3 * H * H + K * (3 * 32 + 32 * H + 2 * H * H + 2 * H * H) + K * H)
                     + H * g_out + KL * E)
    per_call = L * 2 * (2 * E * H) * m_abstract
    return per_query * n_queries + per_call * n_calls + 18.35e9


def executed_flops(n_queries, m_abstract, g_out):
    """Matmul FLOPs the build executes for the same results: lin_z through the per-scene table
    (8 x H interpolation instead of H x H), attn_mlp[0] merged into the query / key projections (per pair
    32 x 2H instead of H x 2H), to_k / to_v / W1 k per scene instead of per call; each counted once."""
    H, E, P, B, L, K, KL = 416, 288, 68, 6, 2, 14, 8
    per_query = 2 * (P * H + B * (2 * H * H + KL * H) + L * (H * 2 * H + K * (3 * 32 + 32 * 2 * H + 2 * H * H + 32 * H + H)
                                                             + H * H) + H * g_out)
    per_scene = 2 * m_abstract * E * (L * (2 * H + H) + B * H)
    return per_query * n_queries + per_scene + 18.35e9


def cpu_baseline_worker(kind):
    """Child process: the oracle timed on the host.  Prints one JSON object.  Decode first (cheap,
    exactly linear in N_q), then the encode (dominated by the oracle's Python FPS stand-in and the
    reference's N x N argsort kNN)."""
    from oracle import path as op
    # many-core hosts thrash on the oracle's small ops (measured on the GPU box: 256 threads are
    # ~100x slower than 16); the baseline uses at most 16 threads and reports that as `cores`
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16)))
    pa, ia, inf = pk.configs.model_args(kind, N_POINTS)
    esd, dsd = pk.configs.synthetic_weights(pa, ia, SEED)
    pcl = pk.configs.synthetic_pcl(kind, N_POINTS, VIDEO_LEN, SEED)
    q = pk.geometry.sample_implicit_points_blind_numpy(NUM_SAMPLE, inf['min_z'], inf['cube_bounds'], 3,
                                                       inf['data_kind'], inf['cube_mode'], 'grid')
    m = pk.distributed.abstract_shape(type('E', (), pa), N_POINTS)[0]
    rng = np.random.default_rng(0)
    ab = torch.from_numpy(np.concatenate([pcl[0, :m, :3].numpy(), 0.5 * rng.normal(size=(m, 288))], 1).astype(np.float32))
    fg = torch.from_numpy((0.3 * rng.normal(size=(128,))).astype(np.float32))
    with torch.no_grad():
        op.decoder_forward(dsd, ia, torch.from_numpy(q[:256]), ab, fg)            # warm-up
        # the small sample first, so that a slow host still reports a figure; then two FULL implicit_batch_size
        # mini-batches, decoded one after the other as eval/inference.py:204-246 does
        for sample in (2048, 2 * BATCH):
            t0 = time.time()
            for lo in range(0, sample, BATCH):
                op.decoder_forward(dsd, ia, torch.from_numpy(q[lo:min(sample, lo + BATCH)]), ab, fg)
            t_dec = time.time() - t0
            print(json.dumps(dict(stage='decode', t_dec=t_dec, sample=sample, cores=torch.get_num_threads(),
                                  n_total=int(q.shape[0]))), flush=True)
            if t_dec > 4.0:
                break
        t0 = time.time()
        op.encoder_forward(esd, pa, pcl)
        t_enc = time.time() - t0
    print(json.dumps(dict(stage='encode', t_enc=t_enc)), flush=True)
    # the same decode with EVERY host core (SURVEY 8(d)'s wording), on a 256-query sample (a many-core host thrashes on
    # the oracle's small ops: ~100 x slower than 16 threads on the 256-core GPU box, so the sample must be small): reported
    # beside the 16-thread figure, which stays the baseline
    with torch.no_grad():
        torch.set_num_threads(os.cpu_count() or 1)
        t0 = time.time()
        op.decoder_forward(dsd, ia, torch.from_numpy(q[:256]), ab, fg)
        print(json.dumps(dict(stage='all_cores', t_all=time.time() - t0, all_sample=256, all_cores=torch.get_num_threads())),
              flush=True)


def cpu_baseline_start(kind):
    """Starts cpu_baseline_worker in a child (16 host threads of the box's 256; the GPU legs that run meanwhile need one)."""
    import subprocess
    return subprocess.Popen([sys.executable, os.path.abspath(__file__), '--_cpu-worker', '--kind', kind],
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)


def secondary_leg(script, argv, pick, budget_s=150.0):
    """One more BASELINE configuration on the driver-timed line: runs `script argv` in a child process on this GPU
    (the parent is idle meanwhile), parses its JSON line and keeps the keys in `pick`.  A failure is recorded on the
    line, it never fails the headline."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, script)] + argv
    t0 = time.time()
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=budget_s)
    except subprocess.TimeoutExpired:
        return dict(error='%s %s: no result within %.0f s' % (script, ' '.join(argv), budget_s))
    rec = None
    for ln in res.stdout.splitlines():
        try:
            rec = json.loads(ln)
        except ValueError:
            pass
    if res.returncode != 0 or not isinstance(rec, dict):
        return dict(error='%s %s: exit code %d: %s' % (script, ' '.join(argv), res.returncode, res.stderr.strip()[-300:]))
    out = {k: rec.get(k) for k in pick}
    out['command'] = 'python %s %s' % (script, ' '.join(argv))
    out['wall_s'] = round(time.time() - t0, 1)
    return out


def cpu_baseline(kind, budget_s=240.0, child=None):
    """Collects cpu_baseline_worker (started here unless the caller did) within a wall-clock budget (the child is killed
    by PID if the encode overruns; the decode-only figure is then reported and said so)."""
    import subprocess
    child = child or cpu_baseline_start(kind)
    try:
        out, _ = child.communicate(timeout=budget_s)
    except subprocess.TimeoutExpired:
        child.kill()
        out, _ = child.communicate()
    rec = {}
    for ln in out.splitlines():
        try:
            rec.update(json.loads(ln))
        except ValueError:
            pass
    if 't_dec' not in rec:
        return dict(value=None, unit='query-points/s', cores=None, kind='port', sample='cpu baseline failed')
    n, sample = rec['n_total'], rec['sample']
    t_dec_full = rec['t_dec'] * n / sample
    if 't_enc' in rec:
        total = rec['t_enc'] + t_dec_full
        note = ('oracle/path.py on host CPU: %d-query decode %.2f s (extrapolated linearly to %d queries = %.0f s) + '
                '1 encode (n_points=%d, FPS by the oracle\'s torch_cluster stand-in) %.1f s'
                % (sample, rec['t_dec'], n, t_dec_full, N_POINTS, rec['t_enc']))
    else:
        total = t_dec_full
        note = ('oracle/path.py on host CPU: %d-query decode %.2f s extrapolated linearly to %d queries; the encode '
                'did not finish within the %.0f s budget and is NOT included (decode-only upper bound)'
                % (sample, rec['t_dec'], n, budget_s))
    all_cores = None
    if 't_all' in rec:       # (did not finish within the budget on a thrashing host: then absent)
        all_cores = dict(cores=rec['all_cores'], decode_value=rec['all_sample'] / rec['t_all'], unit='query-points/s (decode only)',
                         sample='%d-query decode with torch.set_num_threads(%d): %.1f s' % (rec['all_sample'], rec['all_cores'],
                                                                                           rec['t_all']))
    return dict(value=n / total, unit='query-points/s', cores=rec['cores'], host_cores=os.cpu_count(), kind='port',
                all_cores=all_cores,
                sample=note + ' (torch threads capped at %d of the host\'s %d logical cores: the oracle\'s small ops thrash '
                'beyond that)' % (rec['cores'], os.cpu_count() or 0))


def self_launch(n_gpus):
    """`python bench.py --gpus N` from a bare shell: re-run this command line as N ranks under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1).  Returns the launcher's exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n_gpus and os.environ.get('OCC4D_BENCH_SHARE_GPU') != '1':
        print('bench.py: --gpus %d needs %d visible GPUs, this machine has %d' % (n_gpus, n_gpus, have), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % n_gpus,
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this image's host driver only supports dmabuf IPC; without it RCCL's device-memory
    # sharing across processes fails with `hipIpcGetMemHandle: invalid argument` (the build environment's own note; the
    # driver's launcher exports it as well -- kept here so that a bare `python bench.py --gpus N` behaves the same)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(cmd, env=env)


def rccl_log_setup(rank):
    """N > 1 runs record RCCL's own initialisation log (NCCL_DEBUG=INFO into a per-process file) so that the bench line can
    carry which transport / ring RCCL really built: the first 8-GPU run is then diagnosable from its JSON tail."""
    fd, path = tempfile.mkstemp(prefix='occ4d_rccl_rank%d_' % rank, suffix='.log')      # (removed in main() once read)
    os.close(fd)
    os.environ.setdefault('NCCL_DEBUG', 'INFO')
    os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH')
    os.environ['NCCL_DEBUG_FILE'] = path
    return path


def rccl_log_excerpt(path, limit=14):
    """The lines of RCCL's log that name the version, the topology search result, rings / trees and transports."""
    keep = []
    try:
        with open(path, errors='replace') as f:
            for ln in f:
                if 'NCCL INFO' not in ln:
                    continue
                msg = ln.split('NCCL INFO', 1)[1].strip()
                if any(k in msg for k in ('version', 'RCCL', 'Ring ', 'Trees', 'Channel 00', 'via ', 'Connected all', 'comm 0x',
                                          'nChannels', 'Pattern', 'XGMI', 'P2P')):
                    keep.append(msg[:160])
    except OSError as e:
        return ['(no RCCL log: %s)' % e]
    seen, out = set(), []
    for m in keep:
        key = m[:40]
        if key not in seen:
            seen.add(key)
            out.append(m)
    return out[:limit]


def pmc_traffic(kind):
    """HBM bytes per full launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, written by profiles/summarize_pmc.py from separate --pmc runs of this command)."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            rec = json.load(f)[kind]
        return rec['hbm_bytes_per_launch'], 'profiles/pmc_traffic.json: ' + rec['source']
    except (OSError, KeyError, ValueError):
        return None, 'no PMC pass committed for this workload'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--kind', default='greater', choices=['greater', 'carla'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the opt-in split-bf16 logits measurement')
    ap.add_argument('--no-extra', action='store_true', help='skip every informational leg (secondary grid, pipelined, '
                    'host boundary)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary BASELINE configurations '
                    '(carla_config3 = configs[2], train_config5 = configs[4]) that the N = 1 line carries')
    ap.add_argument('--_cpu-worker', dest='cpu_worker', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_baseline_worker(args.kind)
        return 0

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        print('bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d)' % (args.gpus, world, args.gpus),
              file=sys.stderr)
        return 2
    # Harness self-test only (tests/test_gpu_bench_multirank.py): OCC4D_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and
    # OCC4D_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device), so that the complete N > 1 code
    # path -- sharding, broadcast, per-rank timing exchange, both grids -- runs on a 1-GPU box.  Timings of such a
    # run mean nothing and the line says so.
    share_gpu = os.environ.get('OCC4D_BENCH_SHARE_GPU') == '1'
    backend = os.environ.get('OCC4D_BENCH_BACKEND', 'nccl')
    dev_index = 0 if share_gpu else local_rank
    if torch.cuda.device_count() <= dev_index:
        print('bench.py: rank %d needs GPU %d, only %d visible' % (rank, dev_index, torch.cuda.device_count()),
              file=sys.stderr)
        return 2
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    use_dist = world > 1 or os.environ.get('OCC4D_FORCE_DIST') == '1'   # (1-rank RCCL: smoke test of the N > 1 path)
    rccl_ranks = 1
    rccl_log = None
    if use_dist:
        if backend == 'nccl':
            rccl_log = rccl_log_setup(rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
        ones = torch.ones((), device=device)
        dist.all_reduce(ones)                   # the ranks RCCL really connected
        rccl_ranks = int(ones.item())

    pa, ia, inf = pk.configs.model_args(args.kind, N_POINTS)
    esd, dsd = pk.configs.synthetic_weights(pa, ia, SEED)
    pcl_cpu = pk.configs.synthetic_pcl(args.kind, N_POINTS, VIDEO_LEN, SEED)
    enc = pk.model.PointCompletionNetV3(**pa).to(device).eval()
    dec = pk.implicit.LocalPclResnetFC(**ia).to(device).eval()
    enc.load_state_dict(esd)
    dec.load_state_dict(dsd)
    pcl = pcl_cpu.to(device)

    def grid(num_sample):
        return pk.geometry.sample_implicit_points_blind_device(
            num_sample, inf['min_z'], inf['cube_bounds'], 3, inf['data_kind'], inf['cube_mode'], 'grid', device)

    # N = 1: configs[1] (the configuration the metric is quoted on).  N > 1: configs[3], the dense grid, sharded.
    num_sample = NUM_SAMPLE if world == 1 else NUM_SAMPLE_DENSE
    queries = grid(num_sample)
    n_total = queries.shape[0]
    lo, hi = pk.distributed.shard_bounds(n_total, rank, world)

    def step(q=None):
        return pk.distributed.sharded_inference(pcl, queries if q is None else q, enc, dec, BATCH, inf['color_mode'],
                                                inf['predict_segmentation'], 'none', 13)

    def fence():
        torch.cuda.synchronize()
        pk.ops.check_pending()     # cooperative-FPS status words: a timed-out launch voids the run
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup, finish=None):
        """`warmup` untimed + exactly `steps` timed calls of fn between fences; returns (max-over-ranks seconds,
        per-rank seconds, last result).  `finish`: called after the last timed step, before the closing fence."""
        res = None
        for _ in range(warmup):
            res = fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = fn()
        if finish is not None:
            finish()
        fence()
        mine = time.perf_counter() - t0
        per_rank = [mine]
        if use_dist:
            tt = torch.tensor([mine], dtype=torch.float64, device=device)
            allt = [torch.empty_like(tt) for _ in range(world)]
            dist.all_gather(allt, tt)
            per_rank = [float(x.item()) for x in allt]
        return max(per_rank), per_rank, res

    # Schedule of the timed steps.  N = 1: sequential (encode, then decode: the single-GPU figure of rounds 1-5).  N > 1:
    # PIPELINED (distributed.ClipPipeline, round 6): rank 0's encode + the one packed broadcast of clip i + 1 are issued
    # on a side stream while every rank decodes clip i, so the serial 3.7 ms encode leaves the strong-scaling Amdahl
    # term.  Every step still encodes and decodes a clip in full: K timed steps = K encodes + K broadcasts + K decodes
    # (the last encode is waited for inside the timed region).  OCC4D_BENCH_SCHEDULE=sequential restores the old N > 1.
    pipe = pk.distributed.ClipPipeline(enc, dec, BATCH, inf['color_mode'], inf['predict_segmentation'], 'none', 13)

    def pipe_step(q=None):
        if pipe.pending is None:
            pipe.submit(pcl)
        taken = pipe.take()
        pipe.submit(pcl)
        return pipe.decode(taken, queries if q is None else q)

    def pipe_finish():
        pipe.take()                                         # the last encode issued inside the timed region
    schedule = os.environ.get('OCC4D_BENCH_SCHEDULE', 'pipelined' if world > 1 else 'sequential')
    assert schedule in ('pipelined', 'sequential'), schedule
    with torch.no_grad():
        if schedule == 'pipelined':
            elapsed, per_rank_s, (out, _) = timed(pipe_step, args.steps, max(1, args.warmup), finish=pipe_finish)
        else:
            elapsed, per_rank_s, (out, _) = timed(step, args.steps, args.warmup)
        # Roofline leg: HIP events around every launch of the dominant kernel on its launch stream.
        # The timed steps above interleave mini-batches on two streams, where an event bracket also
        # covers the other stream's kernels; this extra (untimed) step runs the same launches on ONE
        # stream so that the brackets are exclusive kernel durations.
        timer = pk.ops.KernelTimer(lambda name, **sh: name == 'cross_attn' and sh.get('d') == 416)
        pk.ops.set_kernel_timer(timer)
        with pk.kernels(decode_streams=1):
            step()
        pk.ops.set_kernel_timer(None)
        psum = timer.summary().get('cross_attn', dict(launches=0, total_ms=0.0, total_flops=0.0))
        extra = not args.no_extra
        # Secondary grid (informational): the other strong-scaling curve's point for this N.
        other = None
        if extra:
            q2 = grid(NUM_SAMPLE_DENSE if world == 1 else NUM_SAMPLE)
            k2 = max(2, (args.steps + 3) // 4) if world == 1 else args.steps
            if schedule == 'pipelined':
                e2, _, _ = timed(lambda: pipe_step(q2), k2, 1, finish=pipe_finish)
            else:
                e2, _, _ = timed(lambda: step(q2), k2, 1)
            lo2, hi2 = pk.distributed.shard_bounds(q2.shape[0], rank, world)
            other = dict(workload='%s grid of %d queries%s' % (args.kind.upper(), q2.shape[0],
                                                               ' (%d per GPU)' % (hi2 - lo2) if world > 1 else ''),
                         steps=k2, ms_per_step=1e3 * e2 / k2, value=q2.shape[0] * k2 / e2, n_gpus=world)
            del q2
        # Opt-in split-precision mode, reported next to the official fp32 number (never as `value`): bf16x6 (round 5) --
        # EVERY GEMM of the two cross-attention layers (csrc/crossattn_bf16x6.hip) and the trunk's 416-input Linear
        # layers (csrc/trunk_bf16x6.hip: residual blocks, merged query projection, layer3) on v_mfma_f32_16x16x32_bf16
        # with both operands split three ways into bf16 pieces (exact), six partial products, fp32 accumulate:
        # fp32-class -- held by tests/test_gpu_regimes.py to the fp32 paths' own bound in every saturated regime.  Its
        # own roofline object: executed bf16 MFMA FLOP of the attention kernel / its HIP-event time / dense bf16 peak.
        # (The round-1 bf16x3 logit mode -- two pieces, NOT fp32-class -- is slower than this and no longer reported.)
        alt = None
        alt_f16 = None
        if not args.no_alt and extra and world == 1:
            def split_leg(scheme):
                """The whole decoder on one split scheme, selected for THIS thread's calls (`with pk.kernels(...)`: no
                module global is assigned; the decoder keeps one prepared weight set per selection)."""
                k_attn = max(2, args.steps // 2)
                with pk.kernels(logit_precision=scheme):
                    attn_elapsed, _, _ = timed(step, k_attn, 1)
                with pk.kernels(precision=scheme):
                    alt_elapsed, _, (out_alt, _) = timed(step, args.steps, 1)
                    timer6 = pk.ops.KernelTimer(lambda name, **sh: name == 'cross_attn' and sh.get('d') == 416)
                    pk.ops.set_kernel_timer(timer6)
                    with pk.kernels(decode_streams=1):
                        step()
                    pk.ops.set_kernel_timer(None)
                ps6 = timer6.summary().get('cross_attn', dict(launches=0, total_ms=0.0, total_flops=0.0))
                # MFMAs the attention kernel executes per launch: workgroups (18 queries x 2 channel halves) x 8 waves x
                # (26 stages x (2 x 2 GEMM1 + 13 x 2 GEMM2 tiles) + 14 epilogue tile pairs x 2) x products per tile
                # (6 | 3) v_mfma_f32_16x16x32_{bf16 | f16} of 2 * 16 * 16 * 32 FLOP
                prod = 6 if scheme == 'bf16x6' else 3
                t6 = ps6['total_ms'] * 1e-3
                bs6 = pk.inference.decode_chunk(BATCH)
                batches6 = [min(bs6, hi - b) for b in range(lo, hi, bs6)]
                wgs6 = sum(2 * (-(-c // 18)) for b in batches6 for c in pk.ops.path_row_chunks(b)) * ia['cross_attn_layers']
                mfma6 = wgs6 * 8 * (26 * 30 + 14 * 2) * prod * (2.0 * 16 * 16 * 32)
                traffic6, traffic6_source = pmc_traffic(args.kind + '_' + scheme)
                what = ('bf16 x 3 pieces per operand, 6 of 9 products, f32 accumulate (fp32-class)' if scheme == 'bf16x6' else
                        'fp16 x 2 pieces per operand (round to nearest), 3 of 4 products, f32 accumulate; weights packed * 2^8; '
                        'inference forwards only, |w| < 255, |activation| < 65504')
                return dict(
                    mode='%s: the GEMMs of the cross-attention layers and of the trunk (residual blocks, query projection, '
                         'layer3: 99 %% of the decode FLOP) on split-precision 16x16x32 MFMAs (csrc/crossattn_bf16x6.hip, '
                         'csrc/trunk_bf16x6.hip, templates over the scheme); encoder, lin_in / lin_out, tables fp32' % scheme,
                    dtype=what, ms_per_step=1e3 * alt_elapsed / args.steps, value=n_total * args.steps / alt_elapsed,
                    attention_only=dict(ms_per_step=1e3 * attn_elapsed / k_attn, value=n_total * k_attn / attn_elapsed),
                    max_abs_diff_vs_f32=float((out_alt - out).abs().max()),
                    regime_bound='every gate of the fp32 path: golden vectors at 1e-4 and, with weights x 4 / x 8, equal / '
                                 'dominant logits, far queries, |hip - ref64| <= max(1e-4, 2 |ref32 - ref64|) '
                                 '(tests/test_gpu_regimes.py, variants "%s", "%s_trunk", "%s_all")' % (scheme, scheme, scheme),
                    roofline=dict(bound='mfma', unit='TFLOP/s', peak=2500.0,
                                  achieved=mfma6 / t6 / 1e12 if t6 > 0 else None,
                                  frac=mfma6 / t6 / 2.5e15 if t6 > 0 else None,
                                  fp32_equivalent_tflops=ps6['total_flops'] / t6 / 1e12 if t6 > 0 else None,
                                  launches=ps6['launches'], avg_launch_ms=ps6['total_ms'] / max(1, ps6['launches']),
                                  traffic=traffic6, traffic_source=traffic6_source,
                                  kernel='cross_attn_split_kernel<%s>; achieved = executed v_mfma_f32_16x16x32 FLOP (%d '
                                         'products, duplicated GEMM1 and dead rows included) / HIP-event time; fp32_equivalent '
                                         '= the fp32 kernel\'s executed FLOP count / the same time'
                                         % ('SplitBf16x6' if scheme == 'bf16x6' else 'SplitF16x3', prod)))
            alt = split_leg('bf16x6')
            alt_f16 = split_leg('f16x3')
        # Throughput mode (informational, never `value`): clips pipelined across steps -- the encode of step i + 1 is
        # issued on a side stream while step i decodes (distributed.ClipPipeline).  Every step still encodes and
        # decodes in full; K steps contain K encode launches and K decodes.
        pipelined = None             # the OTHER schedule, informational: 'pipelined' on the N = 1 line, 'sequential' on N > 1
        if extra:
            if schedule == 'pipelined':
                o_elapsed, _, (out_other, _) = timed(step, args.steps, 1)
                mode = 'sequential: every rank waits for rank 0\'s encode + broadcast before it decodes (the N > 1 schedule of rounds 1-5)'
            else:
                o_elapsed, _, (out_other, _) = timed(pipe_step, args.steps, max(1, args.warmup), finish=pipe_finish)
                mode = 'encode of step i+1 issued on a side stream while step i decodes (K encodes + K decodes)'
            pipelined = dict(mode=mode, ms_per_step=1e3 * o_elapsed / args.steps, value=n_total * args.steps / o_elapsed,
                             max_abs_diff_vs_timed_schedule=float((out_other - out).abs().max()))
        # Host-boundary figure (informational, never `value`): the full perform_inference call as the reference's
        # eval loop makes it -- host point cloud in (H2D), grid generated on the device, encode + decode, split /
        # compress_air on the device, every result array copied back to host numpy (D2H over PCIe).
        host_boundary = None
        if extra and world == 1:
            # As the reference's eval loop calls it (eval/test.py:75) and keeps each result while the next call runs:
            # two untimed calls first (the pinned host buffers of TWO result sets and the side streams' device pools
            # exist after them; round 5 timed 3 calls from cold, two of which were allocating: 144 ms), then 5 timed.
            def host_call():
                return pk.inference.perform_inference(
                    pcl_cpu.clone(), None, None, [enc, dec], device, 'if', inf['min_z'], inf['cube_bounds'],
                    inf['color_mode'], 3, None, sample_implicit=True, num_sample=NUM_SAMPLE, point_sample_mode='grid',
                    batch_size=BATCH, predict_segmentation=inf['predict_segmentation'], track_mode='none',
                    semantic_classes=13, density_threshold=0.5, data_kind=inf['data_kind'], cube_mode=inf['cube_mode'],
                    compress_air=True)
            reps = 5
            res = host_call()
            res = host_call()
            torch.cuda.synchronize()
            each = []
            for _ in range(reps):
                th = time.perf_counter()
                res = host_call()
                each.append(time.perf_counter() - th)
            t_host = sum(each) / reps
            host_boundary = dict(ms_per_call=1e3 * t_host, value=res['points_query'].shape[0] / t_host,
                                 calls=reps, warmup_calls=2, ms_each=[round(1e3 * t, 2) for t in each],
                                 what='perform_inference with host numpy inputs and outputs (PCIe inclusive), the '
                                      'previous result held while the next call runs')
            del res
        # the exchange step, measured on one more (untimed) step: HIP events around the encoder (rank 0) and around
        # the two broadcasts, per rank
        timing = {}
        pk.distributed.sharded_inference(pcl, queries, enc, dec, BATCH, inf['color_mode'], inf['predict_segmentation'],
                                         'none', 13, timing=timing)
        torch.cuda.synchronize()
        ex = [timing['encode'][0].elapsed_time(timing['encode'][1]), timing['broadcast'][0].elapsed_time(timing['broadcast'][1])]
        exchange = [ex]
        if use_dist:
            tt = torch.tensor(ex, dtype=torch.float64, device=device)
            allt = [torch.empty_like(tt) for _ in range(world)]
            dist.all_gather(allt, tt)
            exchange = [[float(v) for v in x.tolist()] for x in allt]
        # encode share, measured separately (informational; rank 0 is the rank that encodes)
        t_encode = None
        if rank == 0:
            torch.cuda.synchronize()
            te = time.perf_counter()
            enc(pcl, False)
            torch.cuda.synchronize()
            t_encode = time.perf_counter() - te

    ms_per_step = 1e3 * elapsed / args.steps
    value = n_total * args.steps / elapsed

    if rank == 0:
        m_abs = pk.distributed.abstract_shape(enc, N_POINTS)[0]
        chunk = pk.inference.decode_chunk(BATCH)
        calls = -(-(hi - lo) // chunk)
        fl = as_written_flops(hi - lo, calls, m_abs, ia['d_out'])            # rank 0's share (it also encodes)
        fl_exec = executed_flops(hi - lo, m_abs, ia['d_out'])
        t_kernel = psum['total_ms'] * 1e-3
        executed = psum['total_flops'] / t_kernel if t_kernel > 0 else 0.0
        H = 416
        as_written_pair = 2.0 * (3 * 32 + 32 * H + 2 * H * H + 2 * H * H)       # FLOP per (query, neighbour)
        executed_pair = 2.0 * (32 * 2 * H + 2 * H * H + 32 * H)
        as_written = executed * as_written_pair / executed_pair
        traffic, traffic_source = pmc_traffic(args.kind)
        line = {
            'metric': '4D query-points/sec (encode+decode) at n_points=14336',
            'value': value, 'unit': 'query-points/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s inference (BASELINE configs[%d]): n_points=%d video_len=%d num_sample=%d (-> %d grid '
                                   'queries%s) implicit_batch_size=%d (decoded in mini-batches of %d = 3584 workgroups of 9 '
                                   'queries; every query is decoded, results do not depend on the split), seeded '
                                   'random-init weights'
                                   % (args.kind.upper(), (1 if world == 1 else 3) if args.kind == 'greater' else 2, N_POINTS, VIDEO_LEN, num_sample, n_total,
                                      ', %d per GPU' % (hi - lo) if world > 1 else '', BATCH, chunk),
                       'abstract_points': m_abs, 'outputs_per_query': ia['d_out'],
                       'parallelism': 'query-sharded x%d, rank 0 encodes, abstract cloud + global embedding in one packed '
                                      'broadcast (RCCL)' % world if world > 1 else 'single GPU',
                       'schedule': schedule + (' (encode + broadcast of clip i + 1 beside the decode of clip i; K timed steps = '
                                               'K encodes + K decodes)' if schedule == 'pipelined' else ''),
                       'scaling_note': 'strong scaling: the total work is fixed as N grows.  N = 1 runs configs[1] (534 528 '
                                       'queries, the configuration the metric is quoted on); N = 2, 4, 8 run configs[3] '
                                       '(2 125 568 queries, the same total for every N).  Each line carries the other grid as a '
                                       'secondary leg (config4_single_gpu / strong_config2), so both strong curves have their '
                                       'N = 1 point.',
                       'encode_ms_per_rank': [e[0] for e in exchange], 'broadcast_ms_per_rank': [e[1] for e in exchange],
                       'rccl_ranks': rccl_ranks, 'backend': ('rccl' if backend == 'nccl' else backend) if use_dist else None,
                       'rccl_info': rccl_log_excerpt(rccl_log) if rccl_log else None,
                       'ranks_share_one_gpu': share_gpu or None,
                       'per_rank_ms_per_step': [1e3 * t / args.steps for t in per_rank_s],
                       'decode_streams': pk.kernels.defaults().decode_streams, 'decode_chunk': chunk},
            'roofline': {
                'bound': 'mfma', 'achieved': executed / 1e12, 'peak': FP32_MFMA_PEAK / 1e12, 'unit': 'TFLOP/s',
                'frac': executed / FP32_MFMA_PEAK,
                'achieved_as_written': as_written / 1e12, 'frac_as_written': as_written / FP32_MFMA_PEAK,
                'traffic': traffic, 'traffic_source': traffic_source,
                'traffic_over_algorithmic': (traffic / (chunk * (2 * H + H + 14) * 4.0)) if traffic else None,
                'kernel': ('cross_attn16p_kernel (csrc/crossattn16p.hip: fused vector attention = pos-MLP + attn-MLP + softmax '
                           '+ aggregate, 14 neighbours, D=416, v_mfma_f32_16x16x4_f32, two 4-wave workgroups per CU); HIP '
                           'events recorded by the library around each launch on its launch stream '
                           '(occ4d_launch_events of occ4d_decoder_query_fwd_f32)'),
                'launches': psum['launches'], 'avg_launch_ms': psum['total_ms'] / max(1, psum['launches']),
                'flop_per_launch_executed': psum['total_flops'] / max(1, psum['launches']),
                'flop_per_launch_as_written': psum['total_flops'] / max(1, psum['launches'])
                * as_written_pair / executed_pair},
            'pipeline': {
                'executed_tflop_per_step_per_gpu': fl_exec / 1e12,
                'executed_frac': fl_exec / (ms_per_step * 1e-3) / FP32_MFMA_PEAK,
                'as_written_tflop_per_step_per_gpu': fl / 1e12,
                'as_written_fp32_mfma_frac': fl / (ms_per_step * 1e-3) / FP32_MFMA_PEAK,
                'encode_ms': 1e3 * t_encode},
        }
        d0 = pk.kernels.defaults()
        if (d0.logit_precision, d0.trunk_precision) != ('f32', 'f32'):
            # a profiling run with OCC4D_LOGIT_PRECISION / OCC4D_TRUNK_PRECISION as the PROCESS default: say so on the line
            line['dtype'] = 'NOT the fp32 figure: process default precision logits %s / trunk %s (OCC4D_*_PRECISION)' % (
                d0.logit_precision, d0.trunk_precision)
            line['roofline']['note'] = ('priced against the fp32 MFMA peak although a split-precision kernel ran: not a figure of '
                                        'merit; the split modes\' rooflines are the alt_precision* objects of a default run')
        if other is not None:
            line['config4_single_gpu' if world == 1 else 'strong_config2'] = other
        if alt is not None:
            line['alt_precision'] = alt
        if alt_f16 is not None:
            line['alt_precision_f16x3'] = alt_f16
        if pipelined is not None:
            line['sequential' if schedule == 'pipelined' else 'pipelined'] = pipelined
        if host_boundary is not None:
            line['host_boundary'] = host_boundary
        # The headline is complete here: a copy goes to STDERR now, so that a harness time limit that cuts the secondary
        # legs or the CPU baseline below short still leaves the measured figure in the log (stdout keeps its ONE line)
        print('bench.py headline (secondary legs follow): ' + json.dumps({k: line[k] for k in (
            'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step')} | {'roofline_frac': line['roofline']['frac']}),
            file=sys.stderr, flush=True)
        # BASELINE configs[2] and configs[4] on the same driver-timed line: each leg is its own
        # process on this GPU, K timed steps between fences exactly as above, its own `roofline` object.  (The CPU
        # baseline starts after them: beside them its 16 threads cost the host-bound eager training step 1.5 ms.)
        if world == 1 and extra and not args.no_secondary and args.kind == 'greater':
            del out
            torch.cuda.empty_cache()
            line['carla_config3'] = secondary_leg(
                'bench.py', ['--kind', 'carla', '--steps', '10', '--warmup', '3', '--no-extra', '--no-cpu-baseline'],
                ['metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'dtype', 'config', 'roofline', 'pipeline'])
            line['train_config5'] = secondary_leg(
                'bench_train.py', ['--steps', '10', '--warmup', '3'],
                ['metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'dtype', 'scaling', 'graph',
                 'attention_backward', 'geometry_prefetch', 'gradient_overlap', 'peak_mem_gb', 'roofline', 'config',
                 'losses'])
            if 'error' not in line['train_config5']:
                # the same step with its 416-input Linear layers, their data gradients and the forward attention kernel on
                # the split-precision kernels (fp32-class, opt-in): informational, `train_config5.value` stays fp32
                line['train_config5']['alt_precision'] = secondary_leg(
                    'bench_train.py', ['--steps', '10', '--warmup', '3', '--precision', 'bf16x6'],
                    ['value', 'ms_per_step', 'dtype', 'precision', 'peak_mem_gb', 'losses'])
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.kind)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()
    if rccl_log:
        with contextlib.suppress(OSError):
            os.remove(rccl_log)
    return 0


if __name__ == '__main__':
    sys.exit(main())
