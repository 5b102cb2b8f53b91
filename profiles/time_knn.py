"""kNN kernel timing vs threads-per-query (OCC4D_KNN_TPQ) for the shapes on the path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402

g = torch.Generator(device='cuda').manual_seed(0)
R = lambda n: torch.rand((n, 3), device='cuda', generator=g) * 10 - 5
shapes = [(14336, 14336, 16, 0), (4779, 14336, 12, 0), (1593, 4779, 12, 0), (32256, 531, 14, 0), (32256, 531, 8, 1),
          (32256, 2124, 14, 0), (4600, 57344, 1, 1), (20000, 57344, 1, 1), (11469, 11469, 1, 1), (534528, 400, 1, 1)]
for nq, nd, k, metric in shapes:
    q, d = R(nq), R(nd)
    line = 'nq=%6d nd=%6d k=%2d m=%d:' % (nq, nd, k, metric)
    for tpq in (1, 4, 16):
        os.environ['OCC4D_KNN_TPQ'] = str(tpq)
        pk.ops.knn(q, d, k, metric=metric, return_dist=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            pk.ops.knn(q, d, k, metric=metric, return_dist=True)
        e1.record()
        torch.cuda.synchronize()
        line += '  tpq%-2d %8.1f us' % (tpq, 1e3 * e0.elapsed_time(e1) / 5)
    print(line, flush=True)
