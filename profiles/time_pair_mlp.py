"""occ4d_pt_pair_mlp_f32 on one recompute chunk (32768 queries x 14 neighbours), by phase skew and with ablation builds
(-DOCC4D_PM_ABL_NOA: no `a` stores, -DOCC4D_PM_ABL_NOLOGITS / -DOCC4D_PM_ABL_NOPE: no epilogue stores; timing only).
Usage: python profiles/time_pair_mlp.py [extra -D flags]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import occlusions4d_amd as pk  # noqa: E402

CSRC = os.path.join(ROOT, 'occlusions-4d_amd', 'csrc')


def main():
    n, m, d, k = 32768, 1062, 416, 14
    rng = np.random.default_rng(0)
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).cuda()   # noqa: E731
    aq, kt = T(rng.normal(size=(n, 2 * d))), T(rng.normal(size=(m, 2 * d)))
    r = torch.relu(T(rng.normal(size=(n * k, 32))))
    idx = torch.from_numpy(rng.integers(0, m, size=(n, k)).astype(np.int32)).cuda()
    wp, w2 = T(0.1 * rng.normal(size=(2 * d, 32))), T(0.03 * rng.normal(size=(d, 2 * d)))
    b2, p2, c2 = T(rng.normal(size=(d,))), T(0.1 * rng.normal(size=(d, 32))), T(rng.normal(size=(d,)))
    stream = pk.ops.pack_attn16p_stream(w2, b2, wp, p2, c2)
    a = torch.empty((n * k, 2 * d), device='cuda')
    lg = torch.empty((n * k, d), device='cuda')
    pe = torch.empty((n * k, d), device='cuda')
    so = '/tmp/pair_mlp_var.so'
    # the variant = this file rebuilt with the extra flags, linked against the library's other objects (build/*.o):
    # crossattn16p.hip calls into path.hip / memops.hip (phase skew, CU count), so it cannot be linked alone
    objs = [os.path.join(ROOT, 'occlusions-4d_amd', 'build', f) for f in sorted(os.listdir(os.path.join(ROOT, 'occlusions-4d_amd', 'build')))
            if f.endswith('.o') and f != 'crossattn16p.o']
    obj = '/tmp/pair_mlp_var.o'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
                    '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-fno-honor-nans'] + sys.argv[1:] +
                   ['-c', os.path.join(CSRC, 'crossattn16p.hip'), '-o', obj], check=True)
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so, obj] + objs, check=True)
    lib = C.CDLL(so)
    fn = lib.occ4d_pt_pair_mlp_f32
    fn.restype = C.c_int
    fn.argtypes = pk._lib.SIGNATURES['occ4d_pt_pair_mlp_f32'][1]
    flops = 2.0 * n * k * (32 * 2 * d + 2 * d * d + 32 * d)
    for skew in (0, 3, 6, 12):
        run = lambda: fn(aq.data_ptr(), 2 * d, kt.data_ptr(), 2 * d, r.data_ptr(), idx.data_ptr(), c2.data_ptr(),   # noqa: E731
                         stream.data_ptr(), a.data_ptr(), lg.data_ptr(), pe.data_ptr(), n, m, k, d, skew, None)
        for _ in range(2):
            assert run() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print('%s skew %2d: %.3f ms  %.1f TFLOP/s  (%.3f of 157.3)' % (' '.join(sys.argv[1:]) or 'product build', skew, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3), flush=True)


main()
