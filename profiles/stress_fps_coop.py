"""Race / residency screen of the cooperative FPS kernel: its indices must not change while other streams keep every
CU busy with long register- and LDS-heavy kernels (workgroups then become resident one by one), and no bounded spin
may time out."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402

torch.manual_seed(0)
g = torch.Generator(device='cuda').manual_seed(1)
clouds = [(torch.rand((n, 3), device='cuda', generator=g) * 10 - 5, m) for n, m in
          [(28672, 9558), (172032, 14336), (14336, 4779), (60000, 3000)]]
ref = [pk.ops.fps_coop(p, m, start=5, return_order=True)[1].clone() for p, m in clouds]
torch.cuda.synchronize()

# background load: big fp32 GEMMs through the library's own linear kernel on two other streams
x = torch.randn((65536, 416), device='cuda')
w = torch.randn((416, 416), device='cuda')
side = [torch.cuda.Stream(), torch.cuda.Stream()]
bad = 0
for rep in range(6):
    for s in side:
        with torch.cuda.stream(s):
            for _ in range(500):
                pk.ops.linear(x, w)
    for (p, m), r in zip(clouds, ref):
        o = pk.ops.fps_coop(p, m, start=5, return_order=True, check=True)[1]
        bad += int(not torch.equal(o, r))
    torch.cuda.synchronize()
    print('rep', rep, 'mismatches so far', bad, flush=True)
print('OK' if bad == 0 else 'FAILED')
