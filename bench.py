#!/usr/bin/env python
"""Benchmark of the occlusions-4d hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" = one perform_inference unit of work: ONE encode of the
(1, 14336, 8) point-cloud video + decode of every grid query point of one output frame,
inputs (point cloud, query grid, weights) already resident in HBM, outputs left in HBM.

Workload (BASELINE.json configs[1], "GREATER inference 1xMI355X"): n_points 14336,
video_len 12, num_sample 524288 -> 534 528 grid queries, implicit_batch_size 32768,
fp32, synthetic data + seeded random-init weights of the published architecture.
At N GPUs the grid is num_sample = 524288 * N (N = 4 is configs[3]'s 2 M-query grid),
rank 0 encodes and broadcasts the abstract cloud, every rank decodes a contiguous 1/N
slice -> per-GPU work is fixed: "scaling": "weak".  value = total queries of all ranks /
max-over-ranks time.

Extra objects on the JSON line:
  roofline      the dominant kernel = cross_attn_kernel<13> (fused vector attention, 14
                neighbours, D = 416).  achieved = ALGORITHMIC FLOP of the reference ops it
                replaces (per pair 3*32 + 32*H + 2H*H + 2H*H MAC, as written)
                / HIP-event time of those launches inside the timed region; the FLOP the
                kernel really executes after the exact-in-R refactoring are reported next to
                it (achieved_executed).  peak = 157.3 TFLOP/s fp32 MFMA.
  cpu_baseline  the CPU oracle (oracle/path.py = the reference's PyTorch-CPU op sequence)
                timed on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import occlusions4d_amd as pk  # noqa: E402

FP32_MFMA_PEAK = 157.3e12
N_POINTS, VIDEO_LEN, NUM_SAMPLE, BATCH = 14336, 12, 524288, int(os.environ.get('OCC4D_BENCH_BATCH', '32768'))   # BASELINE: 32768
SEED = 1830


def as_written_flops(n_queries, n_calls, m_abstract, g_out):
    """Matmul FLOPs of the reference's op sequence (SURVEY.md §8(d) closed form)."""
    H, E, P, B, L, K, KL = 416, 288, 68, 6, 2, 14, 8
    per_query = 2 * (P * H + B * 3 * H * H + L * (3 * H * H + K * (3 * 32 + 32 * H + 2 * H * H + 2 * H * H) + K * H)
                     + H * g_out + KL * E)
    per_call = L * 2 * (2 * E * H) * m_abstract
    return per_query * n_queries + per_call * n_calls + 18.35e9


def cpu_baseline_worker(kind, world):
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
    q = pk.geometry.sample_implicit_points_blind_numpy(NUM_SAMPLE * world, inf['min_z'], inf['cube_bounds'], 3,
                                                       inf['data_kind'], inf['cube_mode'], 'grid')
    m = pk.distributed.abstract_shape(type('E', (), pa), N_POINTS)[0]
    rng = np.random.default_rng(0)
    ab = torch.from_numpy(np.concatenate([pcl[0, :m, :3].numpy(), 0.5 * rng.normal(size=(m, 288))], 1).astype(np.float32))
    fg = torch.from_numpy((0.3 * rng.normal(size=(128,))).astype(np.float32))
    with torch.no_grad():
        op.decoder_forward(dsd, ia, torch.from_numpy(q[:256]), ab, fg)            # warm-up
        for sample in (2048, 16384):      # the small sample first, so that a slow host still reports a figure
            t0 = time.time()
            op.decoder_forward(dsd, ia, torch.from_numpy(q[:sample]), ab, fg)
            t_dec = time.time() - t0
            print(json.dumps(dict(stage='decode', t_dec=t_dec, sample=sample, cores=torch.get_num_threads(),
                                  n_total=int(q.shape[0]))), flush=True)
            if t_dec > 4.0:
                break
        t0 = time.time()
        op.encoder_forward(esd, pa, pcl)
        t_enc = time.time() - t0
    print(json.dumps(dict(stage='encode', t_enc=t_enc)), flush=True)


def cpu_baseline(kind, world, budget_s=240.0):
    """Runs cpu_baseline_worker in a child with a wall-clock budget (the child is killed by PID if the
    encode overruns; the decode-only figure is then reported and said so)."""
    import subprocess
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--_cpu-worker', '--kind', kind,
                              '--gpus', str(world)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
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
                '1 encode (n_points=%d) %.1f s' % (sample, rec['t_dec'], n, t_dec_full, N_POINTS, rec['t_enc']))
    else:
        total = t_dec_full
        note = ('oracle/path.py on host CPU: %d-query decode %.2f s extrapolated linearly to %d queries; the encode '
                'did not finish within the %.0f s budget and is NOT included (decode-only upper bound)'
                % (sample, rec['t_dec'], n, budget_s))
    return dict(value=n / total, unit='query-points/s', cores=rec['cores'], kind='port', sample=note)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--kind', default='greater', choices=['greater', 'carla'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the opt-in split-bf16 logits measurement')
    ap.add_argument('--_cpu-worker', dest='cpu_worker', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_baseline_worker(args.kind, args.gpus)
        return

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    use_dist = world > 1 or os.environ.get('OCC4D_FORCE_DIST') == '1'   # (1-rank RCCL: smoke test of the N > 1 path)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        dist.init_process_group('nccl', device_id=device)

    pa, ia, inf = pk.configs.model_args(args.kind, N_POINTS)
    esd, dsd = pk.configs.synthetic_weights(pa, ia, SEED)
    pcl_cpu = pk.configs.synthetic_pcl(args.kind, N_POINTS, VIDEO_LEN, SEED)
    enc = pk.model.PointCompletionNetV3(**pa).to(device).eval()
    dec = pk.implicit.LocalPclResnetFC(**ia).to(device).eval()
    enc.load_state_dict(esd)
    dec.load_state_dict(dsd)
    queries = pk.geometry.sample_implicit_points_blind_device(
        NUM_SAMPLE * world, inf['min_z'], inf['cube_bounds'], 3, inf['data_kind'], inf['cube_mode'], 'grid', device)
    pcl = pcl_cpu.to(device)
    n_total = queries.shape[0]
    lo, hi = pk.distributed.shard_bounds(n_total, rank, world)

    def step():
        return pk.distributed.sharded_inference(pcl, queries, enc, dec, BATCH, inf['color_mode'],
                                                inf['predict_segmentation'], 'none', 13)

    def fence():
        torch.cuda.synchronize()
        pk.ops.check_pending()     # cooperative-FPS status words: a timed-out launch voids the run
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out, _ = step()
        fence()
        elapsed = time.perf_counter() - t0
        # Roofline leg: HIP events around every launch of the dominant kernel on its launch stream.
        # The timed steps above interleave mini-batches on two streams, where an event bracket also
        # covers the other stream's kernels; this extra (untimed) step runs the same launches on ONE
        # stream so that the brackets are exclusive kernel durations.
        timer = pk.ops.KernelTimer(lambda name, **sh: name == 'cross_attn' and sh.get('d') == 416)
        streams_saved = pk.inference.DECODE_STREAMS
        pk.inference.DECODE_STREAMS = 1
        pk.ops.set_kernel_timer(timer)
        step()
        pk.ops.set_kernel_timer(None)
        pk.inference.DECODE_STREAMS = streams_saved
        psum = timer.summary().get('cross_attn', dict(launches=0, total_ms=0.0, total_flops=0.0))
        # Opt-in mode, reported next to the official fp32 number (never as `value`): the attention-logit
        # GEMM on split-bf16 MFMAs (three bf16 products, fp32 accumulate); outputs stay within 1e-6 of
        # the fp32 path (tests/test_gpu_parity.py::test_decoder_with_split_bf16_logits).
        alt = None
        if not args.no_alt and world == 1:
            pk.point_transformer_layer.LOGIT_PRECISION = 'bf16x3'
            step()
            fence()
            ta = time.perf_counter()
            for _ in range(args.steps):
                out_alt, _ = step()
            fence()
            alt_elapsed = time.perf_counter() - ta
            pk.point_transformer_layer.LOGIT_PRECISION = 'f32'
            alt = dict(mode='attention-logit GEMM on bf16x3 split MFMA (fp32 accumulate), all else fp32',
                       ms_per_step=1e3 * alt_elapsed / args.steps, value=n_total * args.steps / alt_elapsed,
                       max_abs_diff_vs_f32=float((out_alt - out).abs().max()))
        # Throughput mode (informational, never `value`): clips pipelined across steps -- the encode of step i + 1 is
        # issued on a side stream while step i decodes (distributed.ClipPipeline).  Every step still encodes and
        # decodes in full; K steps contain K encode launches and K decodes.
        pipelined = None
        if world == 1:          # extra legs only on the single-GPU run: nothing optional may endanger an N > 1 line
            pipe = pk.distributed.ClipPipeline(enc, dec, BATCH, inf['color_mode'], inf['predict_segmentation'], 'none', 13)
            pipe.submit(pcl)
            for _ in range(max(1, args.warmup)):
                taken = pipe.take()
                pipe.submit(pcl)
                out_pipe, _ = pipe.decode(taken, queries)
            fence()
            tp = time.perf_counter()
            for _ in range(args.steps):
                taken = pipe.take()
                pipe.submit(pcl)
                out_pipe, _ = pipe.decode(taken, queries)
            pipe.take()                                         # the last encode issued inside the timed region
            fence()
            pipe_elapsed = time.perf_counter() - tp
            pipelined = dict(mode='encode of step i+1 issued on a side stream while step i decodes (K encodes + K decodes)',
                             ms_per_step=1e3 * pipe_elapsed / args.steps, value=n_total * args.steps / pipe_elapsed,
                             max_abs_diff_vs_sequential=float((out_pipe - out).abs().max()))
        # Host-boundary figure (informational, never `value`): the full perform_inference call as the reference's
        # eval loop makes it -- host point cloud in (H2D), grid generated on the device, encode + decode, split /
        # compress_air on the device, every result array copied back to host numpy (D2H over PCIe).
        host_boundary = None
        if world == 1:
            reps = 3
            torch.cuda.synchronize()
            th = time.perf_counter()
            for _ in range(reps):
                res = pk.inference.perform_inference(
                    pcl_cpu.clone(), None, None, [enc, dec], device, 'if', inf['min_z'], inf['cube_bounds'],
                    inf['color_mode'], 3, None, sample_implicit=True, num_sample=NUM_SAMPLE, point_sample_mode='grid',
                    batch_size=BATCH, predict_segmentation=inf['predict_segmentation'], track_mode='none',
                    semantic_classes=13, density_threshold=0.5, data_kind=inf['data_kind'], cube_mode=inf['cube_mode'],
                    compress_air=True)
            t_host = (time.perf_counter() - th) / reps
            host_boundary = dict(ms_per_call=1e3 * t_host, value=res['points_query'].shape[0] / t_host,
                                 what='perform_inference with host numpy inputs and outputs (PCIe inclusive)')
        # encode share, measured separately (informational)
        torch.cuda.synchronize()
        te = time.perf_counter()
        enc(pcl, False)
        torch.cuda.synchronize()
        t_encode = time.perf_counter() - te

    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = n_total * args.steps / elapsed

    if rank == 0:
        m_abs = pk.distributed.abstract_shape(enc, N_POINTS)[0]
        calls = -(-(hi - lo) // pk.inference.decode_chunk(BATCH))
        fl = as_written_flops(hi - lo, calls, m_abs, ia['d_out'])
        executed = psum['total_flops'] / (psum['total_ms'] * 1e-3) if psum['total_ms'] > 0 else 0.0
        H, K = 416, 14
        as_written_pair = 2.0 * (3 * 32 + 32 * H + 2 * H * H + 2 * H * H)       # FLOP per (query, neighbour)
        executed_pair = 2.0 * (32 * 2 * H + 2 * H * H + 32 * H)
        achieved = executed * as_written_pair / executed_pair
        line = {
            'metric': '4D query-points/sec (encode+decode) at n_points=14336',
            'value': value, 'unit': 'query-points/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s inference: n_points=%d video_len=%d num_sample=%d (-> %d grid queries'
                                   '%s) implicit_batch_size=%d, seeded random-init weights'
                                   % (args.kind.upper(), N_POINTS, VIDEO_LEN, NUM_SAMPLE * world, n_total,
                                      ', %d per GPU' % (hi - lo) if world > 1 else '', BATCH),
                       'abstract_points': m_abs, 'outputs_per_query': ia['d_out'],
                       'parallelism': 'query-sharded x%d, abstract cloud broadcast' % world,
                       'decode_streams': pk.inference.DECODE_STREAMS,
                       'decode_chunk': pk.inference.decode_chunk(BATCH)},
            'roofline': {
                'bound': 'mfma', 'achieved': achieved / 1e12, 'peak': FP32_MFMA_PEAK / 1e12, 'unit': 'TFLOP/s',
                'frac': achieved / FP32_MFMA_PEAK,
                # HBM bytes per full 32256-query launch from the PMC passes committed as profiles/r01_pmc_final.txt:
                # (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE correction); not re-measured live
                'traffic': 236.5e6 if args.kind == 'greater' else None,
                'traffic_source': 'profiles/r01_pmc_final.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)',
                'kernel': 'cross_attn_kernel<13> (fused vector attention: pos-MLP + attn-MLP + softmax + '
                          'aggregate, 14 neighbours, D=416)',
                'launches': psum['launches'], 'avg_launch_ms': psum['total_ms'] / max(1, psum['launches']),
                'flop_per_launch_as_written': psum['total_flops'] / max(1, psum['launches'])
                * as_written_pair / executed_pair,
                'achieved_executed': executed / 1e12, 'frac_executed': executed / FP32_MFMA_PEAK},
            'pipeline': {
                'as_written_tflop_per_step_per_gpu': fl / 1e12,
                'as_written_fp32_mfma_frac': fl / (ms_per_step * 1e-3) / FP32_MFMA_PEAK,
                'encode_ms': 1e3 * t_encode},
        }
        if alt is not None:
            line['alt_precision'] = alt
        if pipelined is not None:
            line['pipelined'] = pipelined
        if host_boundary is not None:
            line['host_boundary'] = host_boundary
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.kind, world)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
