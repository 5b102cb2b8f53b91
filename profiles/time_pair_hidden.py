import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import occlusions4d_amd as pk
from occlusions4d_amd import ops
n, m, k, d = 22976, 2124, 14, 416
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).cuda()
aq, kt, r = rnd(n, 2 * d), rnd(m, 2 * d), torch.relu(rnd(n * k, 32))
wp, W2, b2, P2, c2 = rnd(2 * d, 32), rnd(d, 2 * d), rnd(d), rnd(d, 32), rnd(d)
idx = torch.randint(0, m, (n, k), generator=g).int().cuda()
stream = ops.pack_attn16p_stream(W2, b2, wp, P2, c2)
a, lg, pe = ops.pt_pair_mlp(aq, kt, r, idx, c2, stream)
def t(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print('full  %.3f ms' % t(lambda: ops.pt_pair_mlp(aq, kt, r, idx, c2, stream)))
print('short %.3f ms (%.2f GB stored)' % (t(lambda: ops.pt_pair_mlp(aq, kt, r, idx, c2, stream, logits=lg)), n * k * 3 * d * 4 / 1e9))
x = torch.empty(n * k * 3 * d, device='cuda')
print('fill of the same bytes %.3f ms' % t(lambda: x.fill_(1.0)))
# the reductions that read the (n k, 832) pair gradient in backward
da = torch.randn(n * k, 2 * d, device='cuda')
print('segment_sum of (n k, 832) over k: %.3f ms (%.2f GB read)' % (t(lambda: ops.segment_sum(da, k)), n * k * 2 * d * 4 / 1e9))
print('scatter_add_rows of the same into %d rows: %.3f ms' % (m, t(lambda: ops.scatter_add_rows(da, idx, m, scale=-1.0))))
