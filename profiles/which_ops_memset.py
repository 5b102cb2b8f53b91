"""Which torch operators of the training step issue a device memset (a MEMSET NODE once captured)?"""
import torch
import torch.nn.functional as F
from torch.profiler import ProfilerActivity, profile

dev = 'cuda'
x = torch.randn(17203, device=dev, requires_grad=True)
y = (torch.rand(17203, device=dev) > 0.5).float()
m = (torch.rand(17203, device=dev) > 0.3)
logits = torch.randn(17203, 13, device=dev, requires_grad=True)
lab = torch.randint(0, 13, (17203,), device=dev)
big = torch.randn(68812, 128, device=dev)
params = [torch.randn(416, 416, device=dev, requires_grad=True) for _ in range(8)]
for p in params:
    p.grad = torch.randn_like(p)
opt = torch.optim.AdamW(params, lr=1e-3, capturable=True)
cases = {
    'x.sum() [17203]': lambda: x.sum(),
    'bce_with_logits mean [17203]': lambda: F.binary_cross_entropy_with_logits(x, y),
    'bce backward': lambda: F.binary_cross_entropy_with_logits(x, y).backward(),
    '(v*m).sum()/m.sum()': lambda: (x * m.float()).sum() / m.float().sum().clamp(min=1.0),
    'cross_entropy none [17203,13]': lambda: F.cross_entropy(logits, lab, reduction='none'),
    'cross_entropy none backward': lambda: F.cross_entropy(logits, lab, reduction='none').sum().backward(),
    'big.sum(0) [68812,128]': lambda: big.sum(0),
    'clip_grad_norm_': lambda: torch.nn.utils.clip_grad_norm_(params, 0.2),
    'AdamW(capturable).step': lambda: opt.step(),
    'torch.cat': lambda: torch.cat([big, big], dim=1),
    'zeros': lambda: torch.zeros(68812, 416, device=dev),
}
for name, fn in cases.items():
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    n = sum(e.count for e in prof.key_averages() if 'emset' in e.key)
    print('%-34s memsets: %d' % (name, n))
