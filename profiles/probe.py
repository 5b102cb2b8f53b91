"""Small product-path workloads for rocprofv3 runs (GPU box).
    python profiles/probe.py decode [n_queries] [reps] [kind]   one decoder batch, repeated
    python profiles/probe.py encode [reps]               encoder only
    python profiles/probe.py fps                         FPS / kNN micro timings (HIP events)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402


def nets(kind='greater'):
    pa, ia, inf = pk.configs.model_args(kind, 14336)
    esd, dsd = pk.configs.synthetic_weights(pa, ia, 1830)
    enc = pk.model.PointCompletionNetV3(**pa).cuda().eval()
    dec = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    enc.load_state_dict(esd)
    dec.load_state_dict(dsd)
    pcl = pk.configs.synthetic_pcl(kind, 14336, 12, 1830).cuda()
    return enc, dec, pcl, inf


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'decode'
    with torch.no_grad():
        if mode == 'decode':
            n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
            reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
            kind = sys.argv[4] if len(sys.argv) > 4 else 'greater'
            enc, dec, pcl, inf = nets(kind)
            ab, fg, _ = enc(pcl, False)
            q = torch.from_numpy(pk.geometry.sample_implicit_points_blind_numpy(
                524288, inf['min_z'], inf['cube_bounds'], 3, kind, 4, 'grid')[:n]).cuda()
            ms = timed(lambda: dec(q, ab[0], fg[0], None), reps)
            print('decode %d queries: %.3f ms  -> %.3f M q/s' % (n, ms, n / ms / 1e3))
        elif mode == 'encode':
            reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
            enc, dec, pcl, inf = nets()
            print('encode: %.3f ms' % timed(lambda: enc(pcl, False), reps))
        elif mode == 'fps':
            rng = np.random.default_rng(0)
            for n in (14336, 4779, 1593):
                p = torch.from_numpy(rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)).cuda()
                m = -(-n // 3)
                print('fps n=%d m=%d: %.3f ms' % (n, m, timed(lambda: pk.ops.fps(p, m))))
                print('knn self n=%d k=16: %.3f ms' % (n, timed(lambda: pk.ops.knn(p, p, 16))))
                sub = p[:m].contiguous()
                print('knn down %d->%d k=12: %.3f ms' % (m, n, timed(lambda: pk.ops.knn(sub, p, 12))))


if __name__ == '__main__':
    main()
