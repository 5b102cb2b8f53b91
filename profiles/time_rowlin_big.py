import sys, os
sys.path.insert(0, '/root/repo')
import torch, numpy as np
import occlusions4d_amd as pk
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (M, N) in [(32256, 832), (68812, 416), (458752, 832), (458752, 416), (229376, 832)]:
    x = torch.randn(M, 416, device='cuda'); w = torch.randn(N, 416, device='cuda') * 0.05; b = torch.zeros(N, device='cuda')
    p = pk.ops.pack_trunk_rows(w)
    out = torch.empty(M, N, device='cuda')
    t = timeit(lambda: pk.ops.rowlin(x, p, b, N, out=out)); t2 = timeit(lambda: pk.ops.linear(x, w, b, out=out))
    fl = 2.0 * M * N * 416
    print('M=%7d N=%4d rowlin %8.3f ms %6.1f TF | linear %8.3f ms %6.1f TF' % (M, N, t, fl / t / 1e9, t2, fl / t2 / 1e9))
