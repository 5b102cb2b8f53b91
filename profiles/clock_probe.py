"""Sustained-load clock / power probe: loops one kernel for a few seconds while sampling rocm-smi.
Usage: python profiles/clock_probe.py resblock|linear|attn|micro [seconds]"""
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402

H = 416


def sampler(stop, rows):
    while not stop.is_set():
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showuse', '--csv'], capture_output=True,
                                 text=True, timeout=5).stdout.strip().splitlines()
            if len(out) >= 2:
                rows.append(dict(zip(out[0].split(','), out[1].split(','))))
        except Exception as e:      # noqa: BLE001
            rows.append({'error': str(e)})
        time.sleep(0.4)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'resblock'
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    n = 32256
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.normal(size=(n, H)).astype(np.float32)).cuda()
    y = torch.empty_like(x)
    w0 = torch.from_numpy((0.05 * rng.normal(size=(H, H))).astype(np.float32)).cuda()
    w1 = torch.from_numpy((0.05 * rng.normal(size=(H, H))).astype(np.float32)).cuda()
    b = torch.from_numpy((0.1 * rng.normal(size=(H,))).astype(np.float32)).cuda()
    p0, p1 = pk.ops.pack_trunk_rows(w0), pk.ops.pack_trunk_cols(w1)
    if what == 'resblock':
        fn, flop = (lambda: pk.ops.resblock(x, p0, b, p1, b, out=y)), 4.0 * n * H * H
    elif what == 'linear':
        fn, flop = (lambda: pk.ops.linear(x, w0, b, relu_in=True, out=y)), 2.0 * n * H * H
    elif what == 'attn':
        d, k, m = 416, 14, 531
        T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).cuda()   # noqa: E731
        aq, kt, vt = T(rng.normal(size=(n, 2 * d))), T(rng.normal(size=(m, 2 * d))), T(rng.normal(size=(m, d)))
        qpos, apos = T(rng.uniform(-5, 5, size=(n, 3))), T(rng.uniform(-5, 5, size=(m, 3)))
        idx = pk.ops.knn(qpos, apos, k, metric=0)
        P1, c1 = T(rng.normal(size=(32, 3))), T(rng.normal(size=(32,)))
        wp, w2 = T(0.1 * rng.normal(size=(2 * d, 32))), T(0.03 * rng.normal(size=(d, 2 * d)))
        b2, p2, c2 = T(rng.normal(size=(d,))), T(0.1 * rng.normal(size=(d, 32))), T(rng.normal(size=(d,)))
        stream = pk.ops.pack_attn16p_stream(w2, b2, wp, p2, c2)
        out = torch.empty((n, d), device='cuda')
        fn = lambda: pk.ops.pt_cross_attn16p(aq, qpos, apos, idx, kt, vt + c2, P1, c1, stream, out=out)   # noqa: E731
        flop = 2.0 * n * k * (32 * 2 * d + 2 * d * d + 32 * d)
    elif what == 'micro':
        exe = '/tmp/mfma_issue'
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', os.path.join(os.path.dirname(__file__), 'micro',
                                                                               'mfma_f32_issue.hip'), '-o', exe],
                       check=True, stderr=subprocess.DEVNULL)
        fn, flop = None, 0.0
    else:
        raise SystemExit('unknown workload ' + what)
    rows, stop = [], threading.Event()
    th = threading.Thread(target=sampler, args=(stop, rows))
    th.start()
    t0 = time.time()
    launches = 0
    if fn is None:
        while time.time() - t0 < secs:
            subprocess.run([exe], stdout=subprocess.DEVNULL)
    else:
        while time.time() - t0 < secs:
            for _ in range(50 if what == 'attn' else 200):
                fn()
            torch.cuda.synchronize()
            launches += 50 if what == 'attn' else 200
    dt = time.time() - t0
    stop.set()
    th.join()
    if launches:
        print('%s: %d launches in %.2f s -> %.1f us per launch, %.1f TFLOP/s' % (what, launches, dt, dt / launches * 1e6,
                                                                                 flop * launches / dt / 1e12))
    keys = [k for k in (rows[0] if rows else {}) if any(s in k.lower() for s in ('sclk', 'power', 'use', 'mclk', 'fclk'))]
    for r in rows:
        print('  '.join('%s=%s' % (k.strip(), r.get(k, '').strip()) for k in keys) or str(r))


if __name__ == '__main__':
    main()
