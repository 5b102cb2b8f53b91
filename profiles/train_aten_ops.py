"""Where do the ATen launches of a config-5 training step come from?  One step under torch.profiler with python stacks;
every device kernel that is not one of the library's is listed by ATen operator and by the innermost frame inside
occlusions-4d_amd/ that issued it (autograd-engine launches have no python frame: listed as <autograd engine>).
Usage: python profiles/train_aten_ops.py [precision]"""
import collections
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402
from occlusions4d_amd import training as tr  # noqa: E402

N_POINTS, FRAMES, QUERIES, SEED = 28672, 4, 17203, 1830
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
sel = dict(train_precision=sys.argv[1]) if len(sys.argv) > 1 else None
step = tr.TrainStep(enc, dec, lr=1e-3, grad_clip=0.2,
                    loss_kwargs=dict(density_lw=1.0, segmentation_lw=0.6, static_shapes=True), kernel_selection=sel)
for _ in range(3):
    step(pcl, q, target, pcl)
torch.cuda.synchronize()
import traceback  # noqa: E402

from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

PKG = 'occlusions-4d_amd'
LAUNCHING = ('copy_', 'fill_', 'zero_', 'add_', 'add', 'mul', 'mul_', 'div', 'div_', 'sub', 'sub_', 'ge', 'sum', 'cat', 'mean',
             'clamp', 'sigmoid', 'clone', 'contiguous', 'zeros', 'zeros_like', 'ones', 'ones_like', 'full', 'where', 'neg',
             'index_select', 'gather', 'scatter_add_', 'masked_fill', 'masked_fill_', 'relu', 'threshold_backward', 'exp', 'log',
             '_to_copy', 'new_zeros', 'stack', 'index', 'index_put_', 'sqrt', 'rsqrt', 'pow', 'abs', 'max', 'min', 'eq', 'ne',
             'gt', 'lt', 'le', 'logical_and', 'logical_not', 'bitwise_and', 'cumsum', 'any', 'all', 'nonzero', 'arange')
sites = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split('.')[0]
        if name in LAUNCHING:
            site = '<no package frame>'
            for fr in reversed(traceback.extract_stack()):
                if PKG in fr.filename or fr.filename.endswith('bench_train.py'):
                    site = '%s:%d %s' % (fr.filename.split(PKG + '/')[-1], fr.lineno, fr.name)
                    break
            sites[(site, name)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    step(pcl, q, target, pcl)
    torch.cuda.synchronize()
print('python-visible dispatches of launching ATen operators in one step, by innermost package frame')
for (site, op), v in sites.most_common(80):
    print('  %5d  %-14s %s' % (v, op, site))
print()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(pcl, q, target, pcl)
    torch.cuda.synchronize()

events = prof.events()
PKG = 'occlusions-4d_amd'
by_op = collections.Counter()
by_site = collections.Counter()
n_kernels = 0
n_lib = 0
for e in events:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    if not e.name.startswith('aten::'):
        continue
    # count only leaf operators (an outer aten op owns the same kernels as its inner ones: take the innermost = those
    # whose cpu children have no kernels)
    if any(c.kernels for c in e.cpu_children):
        continue
    site = '<autograd engine>'
    for fr in e.stack or []:
        if PKG in fr or 'bench_train' in fr or 'train_aten_ops' in fr:
            site = fr.split(PKG + '/')[-1] if PKG in fr else fr
            break
    by_op[e.name] += len(e.kernels)
    by_site[(site, e.name)] += len(e.kernels)
    n_kernels += len(e.kernels)
total = sum(1 for e in events if e.device_type == torch.autograd.DeviceType.CUDA)
print('device activities in the step: %d; launched by leaf ATen operators: %d' % (total, n_kernels))
print('\nby operator')
for k, v in by_op.most_common():
    print('  %5d  %s' % (v, k))
print('\nby call site (innermost frame in the package)')
for (site, op), v in by_site.most_common(60):
    print('  %5d  %-28s %s' % (v, op, site))
