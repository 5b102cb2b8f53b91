#!/usr/bin/env python
"""Training-step benchmark for BASELINE config 5 (CARLA-4D DDP training step): batch 1 per GPU,
n_points = 28672 (levels 9558 / 3186 / 1062, M = 4248), 4 target frames x 17203 supervision
queries (num_cr_solid 7168 + air 10035, args.py:254,257), density + segmentation losses, AdamW,
gradient all-reduce on RCCL.  Separate from bench.py (whose single JSON line is the headline
inference metric): prints one JSON line with the step time.

    python bench_train.py [--steps K --warmup W]            (1 GPU)
    python -m torch.distributed.run --nproc-per-node N bench_train.py --gpus N
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--no-checkpoint', action='store_true', help='store the attention pair tensors (merged form; OCC4D_STORED_ATTENTION_FORM=as_written for the round-1 path) instead of recomputing them in backward')
    ap.add_argument('--per-frame', action='store_true', help='decode the target frames one after the other (the reference\'s loop) instead of in one batched decoder call')
    ap.add_argument('--no-prefetch', action='store_true', help='do not prefetch the next step\'s FPS chain / kNNs under this step\'s backward')
    ap.add_argument('--graph', action='store_true', help='replay the step as one captured hipGraph (GraphedTrainStep)')
    ap.add_argument('--sampler', action='store_true',
                    help='draw the supervision points inside the step with GuidedImplicitPointSampler '
                         '(57344-point target frames, bias low_moving_vehped_sembal) instead of fixed synthetic queries')
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=device)
    pk.point_transformer_layer.CHECKPOINT_ATTENTION = not args.no_checkpoint
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
    lkw = dict(density_lw=1.0, segmentation_lw=0.6)
    if args.graph:
        assert not args.sampler, 'the guided sampler draws on the host: it cannot be part of a captured step'
        step = pk.training.GraphedTrainStep(enc, dec, lr=1e-3, grad_clip=0.2, loss_kwargs=lkw,
                                            external_geometry=not args.no_prefetch)
        step.batch_frames = not args.per_frame
        step.capture(pcl, q, target)
    else:
        step = pk.training.TrainStep(enc, dec, lr=1e-3, grad_clip=0.2, loss_kwargs=lkw)

    def fence():
        torch.cuda.synchronize()
        pk.ops.check_pending()     # cooperative-FPS status words: a timed-out launch voids the run
        if world > 1:
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

        def run_step():
            ts = time.perf_counter()
            qs, ts_ = [], []
            for t in range(FRAMES):
                (si, ai, st, at, _, _) = sampler(frames, sizes, valo, num_valo, t)
                qs.append(torch.cat([si, ai], dim=1)[0])
                ts_.append(torch.cat([st, at], dim=1)[0])
            torch.cuda.synchronize()
            sampler_time[0] += time.perf_counter() - ts
            return step(pcl, torch.stack(qs), torch.stack(ts_), **nxt)
    else:
        def run_step():
            return step(pcl, q, target, **nxt)

    losses = []
    for _ in range(args.warmup):
        losses.append(float(run_step()))
    fence()
    if args.sampler:
        sampler_time[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(float(run_step()))
    fence()
    elapsed = time.perf_counter() - t0
    if args.sampler:
        sampler_ms = 1e3 * sampler_time[0] / args.steps
    if world > 1:
        tm = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        elapsed = float(tm.item())
    if rank == 0:
        print(json.dumps({
            'metric': 'training step (BASELINE config 5: CARLA-4D, batch 1/GPU, n_points=28672, 4 x 17203 queries)',
            'value': world * args.steps / elapsed, 'unit': 'examples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'dtype': 'f32', 'data': 'synthetic', 'losses': losses, 'sampler_ms_per_step': sampler_ms, 'graph': bool(args.graph), 'geometry_prefetch': bool(nxt), 'frames_batched': bool(step.batch_frames), 'attention_backward': ('stored pair tensors (%s form)' % pk.point_transformer_layer.STORED_ATTENTION_FORM) if args.no_checkpoint else 'recompute in backward (chunks of %d queries)' % pk.point_transformer_layer._CHECKPOINT_CHUNK,
            'peak_mem_gb': torch.cuda.max_memory_allocated() / 2 ** 30}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
