"""HIP-event timings of the fused trunk kernels against the generic chain they replace (one decode chunk of rows).
Usage: python profiles/time_trunk.py [rows]"""
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402

H = 416


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3     # us


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32256
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.normal(size=(n, H)).astype(np.float32)).cuda()
    ws = [torch.from_numpy((0.05 * rng.normal(size=(H, H))).astype(np.float32)).cuda() for _ in range(2)]
    bs = [torch.from_numpy((0.1 * rng.normal(size=(H,))).astype(np.float32)).cuda() for _ in range(2)]
    wq = torch.from_numpy((0.05 * rng.normal(size=(2 * H, H))).astype(np.float32)).cuda()
    bq = torch.from_numpy((0.1 * rng.normal(size=(2 * H,))).astype(np.float32)).cuda()
    ztab = torch.from_numpy(rng.normal(size=(531, 6 * H)).astype(np.float32)).cuda()
    zc = torch.from_numpy(rng.normal(size=(6 * H,)).astype(np.float32)).cuda()
    idx = torch.from_numpy(rng.integers(0, 531, size=(n, 8)).astype(np.int32)).cuda()
    w8 = torch.from_numpy(rng.uniform(size=(n, 8)).astype(np.float32)).cuda()
    p0, p1, pq = pk.ops.pack_trunk_rows(ws[0]), pk.ops.pack_trunk_cols(ws[1]), pk.ops.pack_trunk_rows(wq)
    p3 = pk.ops.pack_trunk_rows(ws[1])
    h0, h1, hq, h3 = (pk.ops.pack_trunk4_rows(ws[0]), pk.ops.pack_trunk4_cols(ws[1]), pk.ops.pack_trunk4_rows(wq),
                      pk.ops.pack_trunk4_rows(ws[1]))
    interp = (zc[H:2 * H], ztab[:, H:2 * H], idx, w8)
    y = torch.empty_like(x)
    aq = torch.empty((n, 2 * H), device='cuda')
    h = torch.empty_like(x)
    flop = 4.0 * n * H * H

    def chain():
        pk.ops.linear(x, ws[0], bs[0], relu_in=True, out=h)
        pk.ops.linear(h, ws[1], bs[1], relu_in=True, residual=x, out=y)

    def chain_interp():
        chain()
        pk.ops.interp_add(y, interp[0], interp[1], idx, w8)

    rows = [
        ('generic chain fc_0 -> fc_1 (+residual)', timeit(chain), flop),
        ('generic chain + interp_add', timeit(chain_interp), flop),
        ('resblock fused', timeit(lambda: pk.ops.resblock(x, p0, bs[0], p1, bs[1], out=y)), flop),
        ('resblock fused + interpolation term', timeit(lambda: pk.ops.resblock(x, p0, bs[0], p1, bs[1], out=y, interp=interp)), flop),
        ('generic linear 416 -> 832', timeit(lambda: pk.ops.linear(x, wq, bq, out=aq)), flop),
        ('rowlin 416 -> 832', timeit(lambda: pk.ops.rowlin(x, pq, bq, 2 * H, out=aq)), flop),
        ('generic linear 416 -> 416 + residual', timeit(lambda: pk.ops.linear(x, ws[1], bs[1], residual=y, out=y)), flop / 2),
        ('rowlin 416 -> 416 + residual', timeit(lambda: pk.ops.rowlin(x, p3, bs[1], H, residual=y, out=y)), flop / 2),
        ('rowlin 416 -> 416 + residual + interpolation term',
         timeit(lambda: pk.ops.rowlin(x, p3, bs[1], H, residual=y, out=y, interp=interp)), flop / 2),
        ('HALF-CU resblock fused', timeit(lambda: pk.ops.resblock(x, h0, bs[0], h1, bs[1], out=y)), flop),
        ('HALF-CU resblock fused + interpolation term', timeit(lambda: pk.ops.resblock(x, h0, bs[0], h1, bs[1], out=y, interp=interp)), flop),
        ('HALF-CU rowlin 416 -> 832', timeit(lambda: pk.ops.rowlin(x, hq, bq, 2 * H, out=aq)), flop),
        ('HALF-CU rowlin 416 -> 416 + residual', timeit(lambda: pk.ops.rowlin(x, h3, bs[1], H, residual=y, out=y)), flop / 2),
        ('HALF-CU rowlin 416 -> 416 + residual + interpolation term',
         timeit(lambda: pk.ops.rowlin(x, h3, bs[1], H, residual=y, out=y, interp=interp)), flop / 2),
        ('interp_add alone', timeit(lambda: pk.ops.interp_add(y, interp[0], interp[1], idx, w8)), 0.0),
    ]
    print('rows = %d' % n)
    for name, us, fl in rows:
        print('%-52s %9.1f us  %7.1f TFLOP/s  %5.3f of fp32 MFMA peak' % (name, us, fl / us / 1e6, fl / us / 1e6 / 157.3))


if __name__ == '__main__' and not (len(sys.argv) > 1 and sys.argv[1] == 'chain'):
    main()

# (the `chain` mode of this script timed occ4d_trunk_chain_f32, the register-resident trunk chain kernel of round 3: measured
# slower end to end -- profiles/r03_time_trunk.txt, DESIGN.md 6c -- and deleted in round 4)
