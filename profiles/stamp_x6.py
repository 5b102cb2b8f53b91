"""Phase breakdown of cross_attn_bf16x6_kernel (GPU box): OCC4D_X6_STAMPS=1 OCC4D_LOGIT_PRECISION=bf16x6 python profiles/stamp_x6.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('OCC4D_X6_STAMPS', '1')
os.environ.setdefault('OCC4D_LOGIT_PRECISION', 'bf16x6')
import occlusions4d_amd as pk  # noqa: E402
from probe import nets  # noqa: E402

with torch.no_grad():
    enc, dec, pcl, inf = nets('greater')
    ab, fg, _ = enc(pcl, False)
    q = torch.from_numpy(pk.geometry.sample_implicit_points_blind_numpy(524288, inf['min_z'], inf['cube_bounds'], 3, 'greater',
                                                                        4, 'grid')[:32256]).cuda()
    for _ in range(2):
        dec(q, ab[0], fg[0], None)
    torch.cuda.synchronize()
buf = np.zeros(1024 * 2 * 6, dtype=np.uint64)
pk._lib.check(pk._lib.lib().occ4d_debug_x6_stamps(buf.ctypes.data_as(C.c_void_p), buf.size))
t = buf.reshape(1024, 2, 6).astype(np.int64)
t = t[t[:, 0, 0] > 0]
for w in (0, 1):
    d = np.diff(t[:, w, :5], axis=1)
    print('wave %d (group B share %.2f): prologue %6.0f  wait0 %6.0f  loop %7.0f  epilogue %6.0f  total %7.0f cycles (median of %d workgroups)'
          % (4 * w, t[:, w, 5].mean(), *np.median(d, axis=0), np.median(t[:, w, 4] - t[:, w, 0]), t.shape[0]))
