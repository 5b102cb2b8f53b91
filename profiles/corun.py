"""Do kernels of two HIP streams share CUs?  Times the fused attention kernel (stream A) and a run of residual-block
kernels of about the same total length (stream B) alone and together.  If the half-CU shaped kernels co-reside, the
concurrent time approaches the sum of their matrix work instead of the sum of their times.
Usage: python profiles/corun.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402


def main():
    n, m, d, k, H = 32256, 531, 416, 14, 416
    rng = np.random.default_rng(0)
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).cuda()   # noqa: E731
    aq, kt, vt = T(rng.normal(size=(n, 2 * d))), T(rng.normal(size=(m, 2 * d))), T(rng.normal(size=(m, d)))
    qpos, apos = T(rng.uniform(-5, 5, size=(n, 3))), T(rng.uniform(-5, 5, size=(m, 3)))
    idx = pk.ops.knn(qpos, apos, k, metric=0)
    P1, c1 = T(rng.normal(size=(32, 3))), T(rng.normal(size=(32,)))
    wp, w2 = T(0.1 * rng.normal(size=(2 * d, 32))), T(0.03 * rng.normal(size=(d, 2 * d)))
    b2, p2, c2 = T(rng.normal(size=(d,))), T(0.1 * rng.normal(size=(d, 32))), T(rng.normal(size=(d,)))
    x = T(rng.normal(size=(n, H)))
    ws = [T(0.05 * rng.normal(size=(H, H))) for _ in range(2)]
    bs = [T(0.1 * rng.normal(size=(H,))) for _ in range(2)]
    out = torch.empty((n, d), device='cuda')
    y = torch.empty_like(x)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    variants = {
        'attn16p': (pk.ops.pack_attn16p_stream(w2, b2, wp, p2, c2),
                    lambda st: pk.ops.pt_cross_attn16p(aq, qpos, apos, idx, kt, vt, P1, c1, st, out=out)),
    }
    trunks = {
        'resblock4': (pk.ops.pack_trunk4_rows(ws[0]), pk.ops.pack_trunk4_cols(ws[1])),
        'resblock ': (pk.ops.pack_trunk_rows(ws[0]), pk.ops.pack_trunk_cols(ws[1])),
    }

    def timed(fa, fb, reps=5):
        ts = []
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            sA.wait_event(e0)
            sB.wait_event(e0)
            if fa:
                with torch.cuda.stream(sA):
                    fa()
                    e1.record()
            if fb:
                with torch.cuda.stream(sB):
                    fb()
                    e2.record()
            torch.cuda.synchronize()
            ts.append(max(e0.elapsed_time(e1) if fa else 0.0, e0.elapsed_time(e2) if fb else 0.0))
        return float(np.median(ts[1:]))

    for an, (st, fa_) in variants.items():
        for tn, (p0, p1) in trunks.items():
            fa = lambda: [fa_(st) for _ in range(2)]                                        # noqa: E731
            fb = lambda: [pk.ops.resblock(x, p0, bs[0], p1, bs[1], out=y) for _ in range(28)]   # noqa: E731
            ta, tb, tab = timed(fa, None), timed(None, fb), timed(fa, fb)
            print('%s x2 alone %.3f ms   %s x28 alone %.3f ms   together %.3f ms  (sum %.3f, overlap gain %.1f %%)'
                  % (an, ta, tn, tb, tab, ta + tb, 100 * (ta + tb - tab) / (ta + tb)))


if __name__ == '__main__':
    main()
