"""Brute-force kNN kernel against the exact grid search (same lists) at the shapes of the encoder's self-kNNs and of a
training step's decoder lists; HIP events, 10 repeats."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402

ops = pk.ops
rng = np.random.default_rng(0)


def scene(n):
    return torch.from_numpy(rng.uniform([0, -16, -1], [40, 16, 6.4], size=(n, 3)).astype(np.float32)).cuda()


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / 10


for name, nq, nd, k, metric in [('self-kNN training level 0', 28672, 28672, 16, 0), ('self-kNN training level 1', 9558, 9558, 16, 0),
                                ('self-kNN inference level 0', 14336, 14336, 16, 0), ('self-kNN inference level 1', 4779, 4779, 16, 0),
                                ('self-kNN level 2', 1593, 1593, 16, 0), ('decoder lists, training', 68812, 4248, 14, 0),
                                ('decoder interpolation lists, training', 68812, 4248, 8, 1), ('decoder lists, inference chunk', 32256, 531, 14, 0)]:
    d = scene(nd)
    q = d if nq == nd else scene(nq)
    res = {}
    for grid in (False, True):
        ops.KNN_GRID, ops.KNN_GRID_MIN_DATA, ops.KNN_GRID_MIN_PAIRS = grid, 1, 1
        res[grid] = timed(lambda: ops.knn(q, d, k, metric=metric))
    print('%-40s %6d x %6d k %2d  brute %8.1f us   grid %8.1f us' % (name, nq, nd, k, res[False], res[True]), flush=True)
