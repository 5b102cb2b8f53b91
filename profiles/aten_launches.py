"""Which ATen operators of one eager training step (BASELINE config 5) launch kernels, how often and on which shapes
(torch.profiler; the library's own launches go through ctypes and do not appear here)."""
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402

N_POINTS, FRAMES, QUERIES, SEED = 28672, 4, 17203, 2030
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
step = pk.training.TrainStep(enc, dec, lr=1e-3, grad_clip=0.2, loss_kwargs=dict(density_lw=1.0, segmentation_lw=0.6, static_shapes=True))
step.batch_frames = True
for _ in range(2):
    step(pcl, q, target, next_pcl_input=pcl)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(pcl, q, target, next_pcl_input=pcl)
    torch.cuda.synchronize()
import collections  # noqa: E402
by_site = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith('aten::') or not e.kernels:
        continue
    site = '?'
    for fr in (e.stack or []):
        if ('occlusions' in fr or 'bench_train' in fr or 'torch/optim' in fr or 'clip_grad' in fr) and 'profiles/' not in fr:
            site = fr.strip()[-90:]
            break
    else:
        site = 'autograd engine (gradient accumulation / tape glue)' if not e.stack else (e.stack[0].strip()[-90:])
    o = by_site[(e.name, site)]
    o[0] += len(e.kernels)
    o[1] += sum(k.duration for k in e.kernels)
print('kernels launched by ATen operators, by operator and the nearest frame of this package:')
print('%6s %9s  %-24s %s' % ('n', 'us', 'op', 'site'))
for (name, site), (n, us) in sorted(by_site.items(), key=lambda kv: -kv[1][0]):
    print('%6d %9.1f  %-24s %s' % (n, us, name, site))
print('total: %d kernels, %.1f us' % (sum(v[0] for v in by_site.values()), sum(v[1] for v in by_site.values())))
print()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = getattr(e, 'self_cuda_time_total', 0)
    if dt > 0:
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print('self device time (us), count, op, input shapes')
for dt, cnt, key, shp in rows[:70]:
    print('%9.0f %5d  %-28s %s' % (dt, cnt, key, shp))
print('total', sum(r[0] for r in rows))
print('reductions (aten::sum / mean / norm) by input shape:')
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ('aten::sum', 'aten::mean', 'aten::linalg_vector_norm', 'aten::norm', 'aten::_foreach_norm', 'aten::amax', 'aten::max', 'aten::any', 'aten::all', 'aten::cumsum', 'aten::sort', 'aten::nonzero', 'aten::index_add_', 'aten::bincount', 'aten::unique'):
        print('   %-26s x%-4d %s' % (e.key, e.count, str(e.input_shapes)[:120]))
print('memset activities (each would be a MEMSET NODE of a captured step):')
for dt, cnt, key, shp in rows:
    if 'emset' in key:
        print('%9.0f %5d  %s' % (dt, cnt, key))
