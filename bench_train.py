#!/usr/bin/env python
"""Training-step benchmark for BASELINE config 5 (CARLA-4D DDP training step): batch 1 per GPU,
n_points = 28672 (levels 9558 / 3186 / 1062, M = 4248), 4 target frames x 17203 supervision
queries (num_cr_solid 7168 + air 10035, args.py:254,257), density + segmentation losses, AdamW,
gradient all-reduce on RCCL.  Separate from bench.py (whose single JSON line is the headline
inference metric): prints one JSON line with the step time.

    python bench_train.py [--steps K --warmup W]            (1 GPU)
    python bench_train.py --gpus N                           (re-launches itself as N ranks on 127.0.0.1)
    python -m torch.distributed.run --nproc-per-node N bench_train.py --gpus N

The line carries, next to the step time: `roofline` (executed FLOP of the step -- summed over every GEMM / fused
kernel launch of one step -- and the as-written FLOP of forward + backward, over the step time and the fp32 MFMA
peak; the kernel with the largest total time and its average launch duration from HIP events on the launch stream),
`rccl_ranks`, the per-rank step times and the time of the gradient all-reduce alone.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import occlusions4d_amd as pk  # noqa: E402

N_POINTS, FRAMES, QUERIES, SEED = 28672, 4, 17203, 1830
PEAK_F32_MFMA = 157.3      # TFLOP/s, MI355X fp32 matrix peak (MI355X_MICROARCH.md)


def self_launch(n_gpus):
    """`python bench_train.py --gpus N` from a bare shell: re-run this command line as N ranks under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1).  Returns the launcher's exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n_gpus and os.environ.get('OCC4D_BENCH_SHARE_GPU') != '1':
        print('bench_train.py: --gpus %d needs %d visible GPUs, this machine has %d' % (n_gpus, n_gpus, have),
              file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % n_gpus,
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # (HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver only supports dmabuf IPC; see bench.py:self_launch)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(cmd, env=env)


def as_written_flops(pa, ia, n_points, n_queries, n_abstract):
    """Matmul FLOP (2 MAC) of ONE forward pass as the reference writes it (SURVEY.md 8(d) closed forms: per query
    F_query, per decoder call F_call = L 2 (2 E H) M, encoder proportional to n_points), and 3 x that for
    forward + backward (data and weight gradients)."""
    H, E, P, B, L, K, Kloc = 416, 288, 68, 6, 2, 14, 8
    G = ia['d_out']
    f_query = 2 * (P * H + B * 3 * H * H + L * (3 * H * H + K * (3 * 32 + 32 * H + 2 * H * H + 2 * H * H) + K * H)
                   + H * G + Kloc * E)
    f_call = L * 2 * (2 * E * H) * n_abstract
    f_enc = 18.48e9 * n_points / 14336.0          # (abstract_levels = 2; [probe] of SURVEY.md 8(d))
    fwd = f_query * n_queries + f_call + f_enc
    return 3.0 * fwd


class StepCounter:
    """ops.set_kernel_timer hook: HIP events around every reported launch (GEMMs and fused kernels carry their FLOP)
    of one step, on the launch stream."""

    def __init__(self):
        self.events = []

    def want(self, name, **shape):
        return True

    def launch(self, name, flops, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn()
        e1.record()
        self.events.append((name, e0, e1, float(flops)))
        return rc

    def summary(self):
        """Per name: launches, total time, FLOP.  A bracket can start early: the first launch an autograd worker
        thread issues after a cross-stream dependency has its start event recorded BEFORE the stream's wait on the
        forward pass, and then reads tens of ms for a 2.6 ms kernel (rocprofv3 kernel trace of the same step: the
        launches of a shape are equal to 2 %).  Brackets above 3 x the median of their (name, FLOP) group are replaced
        by that median."""
        torch.cuda.synchronize()
        groups = {}
        for name, a, b, f in self.events:
            groups.setdefault((name, f), []).append(a.elapsed_time(b))
        out = {}
        for (name, f), ts in groups.items():
            med = sorted(ts)[len(ts) // 2]
            d = out.setdefault(name, dict(launches=0, total_ms=0.0, total_flops=0.0))
            d['launches'] += len(ts)
            d['total_ms'] += sum(t if t <= 3.0 * med else med for t in ts)
            d['total_flops'] += f * len(ts)
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--no-checkpoint', action='store_true', help='store the attention pair tensors (merged form; OCC4D_STORED_ATTENTION_FORM=as_written for the round-1 path) instead of recomputing them in backward')
    ap.add_argument('--per-frame', action='store_true', help='decode the target frames one after the other (the reference\'s loop) instead of in one batched decoder call')
    ap.add_argument('--no-prefetch', action='store_true', help='do not prefetch the next step\'s FPS chain / kNNs under this step\'s backward')
    ap.add_argument('--precision', choices=['f32', 'bf16x6'], default='f32', help='bf16x6 (opt-in, fp32-class): the 416-input Linear layers of the step (forward and data gradients) the forward attention kernel and the pair-tensor recompute on three-way split bf16 MFMAs, 6 partial products, fp32 accumulate; weight gradients stay fp32')
    ap.add_argument('--sampler', action='store_true',
                    help='draw the supervision points of every step with GuidedImplicitPointSampler (57344-point target '
                         'frames, bias low_moving_vehped_sembal) instead of fixed synthetic queries: the NEXT step\'s points '
                         'are drawn on a side stream while this step runs (they depend on the data and the random '
                         'generators only, like a dataloader); --sampler-serial draws them in front of the step')
    ap.add_argument('--sampler-serial', action='store_true', help='with --sampler: sample, synchronise, then step')
    args = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        print('bench_train.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d)'
              % (args.gpus, world, args.gpus), file=sys.stderr)
        return 2
    # Harness self-test only (tests/test_gpu_bench_multirank.py): OCC4D_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and
    # OCC4D_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device); timings of such a run mean nothing.
    share_gpu = os.environ.get('OCC4D_BENCH_SHARE_GPU') == '1'
    backend = os.environ.get('OCC4D_BENCH_BACKEND', 'nccl')
    dev_index = 0 if share_gpu else local_rank
    if torch.cuda.device_count() <= dev_index:
        print('bench_train.py: rank %d needs GPU %d, only %d visible' % (rank, dev_index, torch.cuda.device_count()),
              file=sys.stderr)
        return 2
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    use_dist = world > 1 or os.environ.get('OCC4D_FORCE_DIST') == '1'   # (1-rank RCCL: smoke test of the N > 1 path)
    rccl_ranks = 1
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29534')
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
        ones = torch.ones((), device=device)
        dist.all_reduce(ones)                   # the ranks RCCL really connected
        rccl_ranks = int(ones.item())
    selection = dict(checkpoint_attention=not args.no_checkpoint)     # kernels.Selection fields of this step (no globals)
    if args.precision == 'bf16x6':
        selection.update(train_precision='bf16x6', logit_precision='bf16x6')   # (logit: the recompute path's fused forward kernel)
    pa, ia, inf = pk.configs.model_args('carla', N_POINTS)
    esd, dsd = pk.configs.synthetic_weights(pa, ia, SEED)
    enc = pk.model.PointCompletionNetV3(**pa).to(device).train()
    dec = pk.implicit.LocalPclResnetFC(**ia).to(device).train()
    enc.load_state_dict(esd)
    dec.load_state_dict(dsd)
    pcl = pk.configs.synthetic_pcl('carla', N_POINTS, 12, SEED + rank).to(device)
    rng = np.random.default_rng(SEED + 100 + rank)
    (x0, x1), (y0, y1), (z0, z1) = (0.0, 40.0), (-16.0, 16.0), (-1.0, 6.4)
    q = np.concatenate([rng.uniform([x0, y0, z0], [x1, y1, z1], size=(FRAMES, QUERIES, 3)),
                        np.broadcast_to(np.arange(FRAMES, dtype=np.float64)[:, None, None], (FRAMES, QUERIES, 1))], -1)
    target = np.concatenate([rng.integers(0, 2, size=(FRAMES, QUERIES, 1)), rng.uniform(size=(FRAMES, QUERIES, 3)),
                             np.zeros((FRAMES, QUERIES, 1)), rng.integers(-1, 13, size=(FRAMES, QUERIES, 1))], -1)
    q = torch.from_numpy(q.astype(np.float32)).to(device)
    target = torch.from_numpy(target.astype(np.float32)).to(device)
    # static_shapes: the loss's masked means by weighting instead of boolean indexing (training.implicit_loss) -- the
    # form without data-dependent shapes; it removes the 12 device->host reads (one per boolean index) that
    # stall the host between forward and backward.  OCC4D_BENCH_INDEXED_LOSS=1: the reference's indexing form.
    lkw = dict(density_lw=1.0, segmentation_lw=0.6, static_shapes=os.environ.get('OCC4D_BENCH_INDEXED_LOSS') != '1')
    step = pk.training.TrainStep(enc, dec, lr=1e-3, grad_clip=0.2, loss_kwargs=lkw, kernel_selection=selection)

    def fence():
        torch.cuda.synchronize()
        pk.ops.check_pending()     # cooperative-FPS status words: a timed-out launch voids the run
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    step.batch_frames = not args.per_frame
    # the next batch's geometry is prefetched under this step's backward / replay; the synthetic bench feeds the same
    # resident cloud every step
    nxt = {} if args.no_prefetch else {'next_pcl_input': pcl}
    sampler_ms = None
    if args.sampler:
        frames, sizes, valo, num_valo = pk.configs.synthetic_target_frames('carla', 57344, FRAMES, SEED + 200 + rank)
        frames = [f.to(device) for f in frames]
        sizes = [z.to(device) for z in sizes]
        valo, num_valo = valo.to(device), num_valo.to(device)
        sampler = pk.geometry.GuidedImplicitPointSampler(
            None, min_z=-1.0, cube_bounds=16.0, point_occupancy_radius=0.2, num_solid=7168, num_air=QUERIES - 7168,
            predict_segmentation=True, semantic_classes=13, data_kind='carla',
            point_sample_bias='low_moving_vehped_sembal', cube_mode=4)
        np.random.seed(SEED + rank)
        torch.manual_seed(SEED + rank)
        sampler_time = [0.0]

        side = pk.training.SideStreamSampler(sampler, FRAMES)
        uploaded = torch.cuda.Event()
        uploaded.record()        # the target frames are complete here; a real loop records one such event per upload

        def draw():
            ts = time.perf_counter()
            side.draw(frames, sizes, valo, num_valo, ready=uploaded)
            sampler_time[0] += time.perf_counter() - ts

        primed = [False]

        def run_step():
            if args.sampler_serial:
                draw()
                qq, tt = side.take()
                torch.cuda.synchronize()
                return step(pcl, qq, tt, **nxt)
            if not primed[0]:
                draw()
                primed[0] = True
            qq, tt = side.take()
            loss = step(pcl, qq, tt, **nxt)        # queued, not waited for
            draw()                                 # the next step's points, beside it
            return loss
    else:
        def run_step():
            return step(pcl, q, target, **nxt)

    # The loss of a step is a device scalar: it is read after the timed region (a loop that logs every step's loss with
    # .item() stalls the host once per step; the usual loop logs every N-th).
    read = lambda t: t                                                           # noqa: E731
    losses = []
    for _ in range(args.warmup):
        losses.append(read(run_step()))
    fence()
    if args.sampler:
        sampler_time[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(read(run_step()))
    fence()
    elapsed = time.perf_counter() - t0
    losses = [float(t) for t in losses]
    if args.sampler:
        sampler_ms = 1e3 * sampler_time[0] / args.steps
    per_rank = [elapsed]
    if use_dist:
        tm = torch.tensor([elapsed], dtype=torch.float64, device=device)
        allt = [torch.empty_like(tm) for _ in range(world)]
        dist.all_gather(allt, tm)
        per_rank = [float(x.item()) for x in allt]
        elapsed = max(per_rank)
    # the gradient all-reduce alone (one flat 28.8 MB bucket): the gradients of the last step are still there
    allreduce_ms = None
    if use_dist:
        fence()
        t1 = time.perf_counter()
        for _ in range(3):
            pk.training.allreduce_gradients(step.params)
        fence()
        allreduce_ms = 1e3 * (time.perf_counter() - t1) / 3
    # roofline leg: one more (untimed) step with HIP events around every reported launch
    roof = None
    if True:
        counter = StepCounter()
        pk.ops.set_kernel_timer(counter)
        overlap, pk.autograd.GRADIENT_OVERLAP = pk.autograd.GRADIENT_OVERLAP, False   # (one stream: a launch's events
        run_step()                                                                     #  bracket that kernel alone)
        pk.autograd.GRADIENT_OVERLAP = overlap
        pk.ops.set_kernel_timer(None)
        summ = counter.summary()
        fence()
        executed = sum(v['total_flops'] for v in summ.values())
        top = max(summ.items(), key=lambda kv: kv[1]['total_ms']) if summ else None
        m_abs = pk.distributed.abstract_shape(enc, N_POINTS)[0]
        written = as_written_flops(pa, ia, N_POINTS, FRAMES * QUERIES, m_abs)
        sec = elapsed / args.steps
        roof = dict(bound='mfma', peak=PEAK_F32_MFMA, unit='TFLOP/s',
                    achieved=executed / sec / 1e12, frac=executed / sec / 1e12 / PEAK_F32_MFMA,
                    achieved_as_written=written / sec / 1e12, frac_as_written=written / sec / 1e12 / PEAK_F32_MFMA,
                    executed_tflop_per_step=executed / 1e12, as_written_tflop_per_step=written / 1e12,
                    note='executed = sum of the FLOP of every GEMM / fused-kernel launch of one step (forward, data and '
                         'weight gradients, recompute); as written = 3 x the reference forward count (SURVEY.md 8(d)); the per-kernel '
                         'table is taken on ONE extra step with the parameter-gradient stream switched off, so that a '
                         'launch\'s HIP events bracket that kernel alone',
                    kernels={k: dict(launches=v['launches'], total_ms=v['total_ms'],
                                     avg_launch_ms=v['total_ms'] / max(1, v['launches']),
                                     tflops=v['total_flops'] / max(v['total_ms'], 1e-9) / 1e9)
                             for k, v in sorted(summ.items(), key=lambda kv: -kv[1]['total_ms'])},
                    dominant=None if top is None else top[0])
        if args.precision != 'f32':
            roof['mixed_precision_note'] = ('part of the executed FLOP ran on bf16 MFMAs (six partial products each, not counted '
                                            'six times): frac against the fp32 MFMA peak is informational for this mode')
    if rank == 0:
        print(json.dumps({
            'metric': 'training step (BASELINE config 5: CARLA-4D, batch 1/GPU, n_points=28672, 4 x 17203 queries)',
            'value': world * args.steps / elapsed, 'unit': 'examples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'dtype': 'f32' if args.precision == 'f32' else 'f32 weight gradients; forward Linears, their data gradients, the forward attention kernel (and any pair-tensor recompute) on bf16 x 3 pieces, 6 products, f32 accumulate (fp32-class)', 'precision': args.precision, 'data': 'synthetic', 'losses': losses, 'sampler_ms_per_step': sampler_ms, 'sampler': (None if not args.sampler else 'in front of the step (serial)' if args.sampler_serial else "next step's points drawn on a side stream beside this step (host time per step in sampler_ms_per_step)"), 'graph': False, 'loss_read': 'after the timed region', 'geometry_prefetch': bool(nxt), 'gradient_overlap': (('parameter gradients on a second stream beside the data-gradient chain, at most %.0f GB of operands held for it' % (pk.autograd.GRADIENT_OVERLAP_BYTES / 2 ** 30)) if pk.autograd.GRADIENT_OVERLAP else False), 'frames_batched': bool(step.batch_frames), 'attention_backward': ('stored pair tensors (%s form)' % pk.kernels.defaults().stored_attention_form) if args.no_checkpoint else {'all': 'fused forward kernel keeps a, logits and pe; backward walks them in equal chunks of at most %d queries, nothing recomputed', 'logits': 'fused forward kernel keeps its logits; a and pe recomputed in backward in equal chunks of at most %d queries', 'none': 'recompute in backward (equal chunks of at most %d queries)'}[pk.kernels.defaults().store_pairs] % pk.kernels.defaults().checkpoint_chunk,
            'peak_mem_gb': torch.cuda.max_memory_allocated() / 2 ** 30,
            'roofline': roof,
            'config': {'workload': 'CARLA-4D training step (BASELINE configs[4]): batch 1 per GPU, n_points=%d, %d x %d '
                                   'supervision queries, density + segmentation losses, AdamW, clip 0.2' % (N_POINTS, FRAMES, QUERIES),
                       'parallelism': 'single GPU' if world == 1 else 'dp%d (one process per GPU, one flat gradient all-reduce)' % world,
                       'rccl_ranks': rccl_ranks, 'backend': backend if use_dist else None,
                       'ranks_share_one_gpu': share_gpu if use_dist else None,
                       'per_rank_ms_per_step': [1e3 * t / args.steps for t in per_rank],
                       'allreduce_ms': allreduce_ms}}), flush=True)
    if use_dist:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
