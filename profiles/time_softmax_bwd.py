"""softmax-aggregate backward (csrc/backward.hip) at the training step's cross-attention chunk (32768 queries x 14
neighbours x 416 channels, 4736 abstract points): with the value-gradient atomics, without (dv = null), HBM bytes / time."""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402

ops, _lib = pk.ops, pk._lib
n, k, d, m = 32768, 14, 416, 4736
g = torch.Generator(device='cuda').manual_seed(0)
logits = torch.randn((n * k, d), device='cuda', generator=g)
pe = torch.randn((n * k, d), device='cuda', generator=g)
v = torch.randn((m, d), device='cuda', generator=g)
dagg = torch.randn((n, d), device='cuda', generator=g)
idx = torch.randint(0, m, (n, k), device='cuda', generator=g, dtype=torch.int32)
# (neighbour lists of a real scene are spatially coherent: consecutive queries share neighbours)
idx_local = ((torch.arange(n, device='cuda')[:, None] // 8 + torch.arange(k, device='cuda')[None]) % m).to(torch.int32)
dl, dpe = torch.empty_like(logits), torch.empty_like(logits)
dv = torch.zeros((m, d), device='cuda')
div = float(torch.tensor(math.sqrt(d), dtype=torch.float32))
P = ops._ptr
st = ops._stream()


def run(ix, with_dv, with_dpe=True):
    _lib.check(_lib.lib().occ4d_pt_softmax_agg_bwd_f32(P(logits), P(v), d, P(pe), P(ix), n, k, d, div, P(dagg), d, P(dl),
                                                       P(dpe) if with_dpe else None, P(dv) if with_dv else None, d, st))


for name, ix in (('random lists', idx), ('coherent lists', idx_local)):
    for with_dv in (True, False):
        for _ in range(2):
            run(ix, with_dv)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(ix, with_dv)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gb = 4 * n * k * d * 4 / 1e9
        print('%-15s dv atomics %-5s %7.1f us  %5.2f TB/s (4 pair tensors = %.2f GB)' % (name, with_dv, 1e3 * ms, gb / ms, gb), flush=True)
dv.zero_()
run(idx, True)
torch.cuda.synchronize()
print('checksums: |dlogits| %.6e  |dpe| %.6e  |dv| %.6e' % (float(dl.double().abs().sum()), float(dpe.double().abs().sum()),
                                                          float(dv.double().abs().sum())))
