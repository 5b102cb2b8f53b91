"""Phases of the eager config-5 training step WITHOUT a profiler attached: device time between events recorded at the
phase boundaries (encoder / decoder / losses / backward / clip + AdamW) and the HOST time spent issuing each phase.  A
phase whose host time is close to its device time is issue-bound: the device waits for launches there.
Usage: python profiles/train_phases.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402
from occlusions4d_amd import training as tr  # noqa: E402

N_POINTS, FRAMES, QUERIES, SEED = 28672, 4, 17203, 1830
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda:0')
pa, ia, inf = pk.configs.model_args('carla', N_POINTS)
esd, dsd = pk.configs.synthetic_weights(pa, ia, SEED)
enc = pk.model.PointCompletionNetV3(**pa).to(dev).train()
dec = pk.implicit.LocalPclResnetFC(**ia).to(dev).train()
enc.load_state_dict(esd)
dec.load_state_dict(dsd)
pcl = pk.configs.synthetic_pcl('carla', N_POINTS, 12, SEED).to(dev)
rng = np.random.default_rng(SEED + 100)
q = np.concatenate([rng.uniform([0, -16, -1], [40, 16, 6.4], size=(FRAMES, QUERIES, 3)),
                    np.broadcast_to(np.arange(FRAMES, dtype=np.float64)[:, None, None], (FRAMES, QUERIES, 1))], -1)
target = np.concatenate([rng.integers(0, 2, size=(FRAMES, QUERIES, 1)), rng.uniform(size=(FRAMES, QUERIES, 3)),
                         np.zeros((FRAMES, QUERIES, 1)), rng.integers(-1, 13, size=(FRAMES, QUERIES, 1))], -1)
q = torch.from_numpy(q.astype(np.float32)).to(dev)
target = torch.from_numpy(target.astype(np.float32)).to(dev)
step = tr.TrainStep(enc, dec, lr=1e-3, grad_clip=0.2,
                    loss_kwargs=dict(density_lw=1.0, segmentation_lw=0.6, static_shapes=True),
                    fused_optimizer=os.environ.get('OCC4D_FUSED_OPTIMIZER', '1') == '1')
NAMES = ['encoder', 'decoder', 'losses', 'prefetch issue', 'backward', 'clip + AdamW']


def one(record):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(NAMES) + 1)]
    host = [time.perf_counter()]

    def mark(i):
        ev[i].record()
        host.append(time.perf_counter())

    pk.ops.check_pending(wait=False)
    step.optimizer.zero_grad(set_to_none=True)
    ev[0].record()
    (pcl_abstract, features_global, _) = enc(pcl, False)
    mark(1)
    out = dec(q.reshape(FRAMES * QUERIES, 4), pcl_abstract[0], features_global[0], None)[0].reshape(FRAMES, QUERIES, -1)
    mark(2)
    loss = tr.implicit_loss(out, target, **step.loss_kwargs)
    mark(3)
    enc.prefetch_geometry(pcl)
    mark(4)
    with pk.autograd.gradient_overlap():
        loss.backward()
    mark(5)
    tr.allreduce_gradients(step.params, participation=step.participation)
    if step.fused:
        step.optimizer.step(max_norm=step.grad_clip)        # one library call: clip + AdamW over the flat buffers (round 6)
    else:
        torch.nn.utils.clip_grad_norm_(step.params, step.grad_clip)
        step.optimizer.step()
    tr.invalidate_weight_caches()
    mark(6)
    if record is not None:
        record.append((ev, host))


for _ in range(3):
    one(None)
torch.cuda.synchronize()
rec = []
t0 = time.perf_counter()
for _ in range(steps):
    one(rec)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
devt = np.zeros(len(NAMES))
hostt = np.zeros(len(NAMES))
for ev, host in rec:
    for i in range(len(NAMES)):
        devt[i] += ev[i].elapsed_time(ev[i + 1])
        hostt[i] += (host[i + 1] - host[i]) * 1e3
print('step %.2f ms (wall over %d back-to-back steps)' % (wall, steps))
print('%-16s %12s %12s' % ('phase', 'device ms', 'host issue ms'))
for n, d, h in zip(NAMES, devt / steps, hostt / steps):
    print('%-16s %12.2f %12.2f' % (n, d, h))
print('%-16s %12.2f %12.2f' % ('sum', devt.sum() / steps, hostt.sum() / steps))
