"""HIP-event timing of the two fused attention kernels on one decode chunk (random neighbour lists).
Usage: python profiles/time_attn.py [n_queries] [m_abstract]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32256
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 531
    d, k = 416, 14
    rng = np.random.default_rng(0)
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).cuda()   # noqa: E731
    aq, kt, vt = T(rng.normal(size=(n, 2 * d))), T(rng.normal(size=(m, 2 * d))), T(rng.normal(size=(m, d)))
    qpos, apos = T(rng.uniform(-5, 5, size=(n, 3))), T(rng.uniform(-5, 5, size=(m, 3)))
    idx = pk.ops.knn(qpos, apos, k, metric=0)
    P1, c1 = T(rng.normal(size=(32, 3))), T(rng.normal(size=(32,)))
    wp, w2 = T(0.1 * rng.normal(size=(2 * d, 32))), T(0.03 * rng.normal(size=(d, 2 * d)))
    b2, p2, c2 = T(rng.normal(size=(d,))), T(0.1 * rng.normal(size=(d, 32))), T(rng.normal(size=(d,)))
    out_a, out_b = torch.empty((n, d), device='cuda'), torch.empty((n, d), device='cuda')

    def timeit(fn, reps=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    flop = 2.0 * n * k * (32 * 2 * d + 2 * d * d + 32 * d)
    t_old = timeit(lambda: pk.ops.pt_cross_attn(aq, qpos, apos, idx, kt, vt, P1, c1, wp, w2, b2, p2, c2, out=out_a))
    print('n = %d queries, m = %d abstract points, k = %d' % (n, m, k))
    # (crossattn16.hip, the round-2 one-workgroup-per-CU kernel of r02 / r03 runs of this script, was deleted in round 4)
    rows = [('crossattn.hip   (32x32x2, 2 channel groups)', t_old)]
    stream_p = pk.ops.pack_attn16p_stream(w2, b2, wp, p2, c2)
    out_c = torch.empty((n, d), device='cuda')
    vtc = vt + c2          # the paired kernel reads the value table with pos_mlp[2].bias folded in
    skews = [int(v) for v in os.environ.get('SKEWS', '0,2,4,6,8,12,16,24').split(',')]
    for sk in skews:
        t = timeit(lambda: pk.ops.pt_cross_attn16p(aq, qpos, apos, idx, kt, vtc, P1, c1, stream_p, out=out_c, skew=sk))
        rows.append(('crossattn16p.hip (paired workgroups), skew %2d' % sk, t))
    for name, ms in rows:
        print('%-46s %8.3f ms  %6.1f TFLOP/s executed  %.3f of fp32 MFMA peak' % (name, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3))
    print('max |difference| crossattn vs crossattn16p: %.3g' % float((out_a - out_c).abs().max()))


if __name__ == '__main__':
    main()
