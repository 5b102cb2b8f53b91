"""interp_add timing at the decode shape (32256 x 416, 8 neighbours of 531 table rows)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk
n, m, d, k = 32256, 531, 416, 8
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn((n, d), device='cuda', generator=g)
table = torch.randn((m, 6 * d), device='cuda', generator=g)
cv = torch.randn((d,), device='cuda', generator=g)
q = torch.rand((n, 3), device='cuda', generator=g)
a = torch.rand((m, 3), device='cuda', generator=g)
idx, dist = pk.ops.knn(q, a, k, metric=1, return_dist=True)
w = pk.ops.interp_weights(dist)
for _ in range(3):
    pk.ops.interp_add(x, cv, table[:, d:2 * d], idx, w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    pk.ops.interp_add(x, cv, table[:, d:2 * d], idx, w)
e1.record()
torch.cuda.synchronize()
print('interp_add %.1f us' % (1e3 * e0.elapsed_time(e1) / 50))
