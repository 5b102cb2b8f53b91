"""Phase accounting of crossattn16.hip (debug build with s_memtime stamps): prologue / hidden-block loop / epilogue
cycles per wave.  Usage: python profiles/stamp_attn.py [extra -D flags]"""
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
    n, m, d, k = 32256, 531, 416, 14
    rng = np.random.default_rng(0)
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).cuda()   # noqa: E731
    aq, kt, vt = T(rng.normal(size=(n, 2 * d))), T(rng.normal(size=(m, 2 * d))), T(rng.normal(size=(m, d)))
    qpos, apos = T(rng.uniform(-5, 5, size=(n, 3))), T(rng.uniform(-5, 5, size=(m, 3)))
    idx = pk.ops.knn(qpos, apos, k, metric=0)
    P1, c1 = T(rng.normal(size=(32, 3))), T(rng.normal(size=(32,)))
    wp, w2 = T(0.1 * rng.normal(size=(2 * d, 32))), T(0.03 * rng.normal(size=(d, 2 * d)))
    b2, p2, c2 = T(rng.normal(size=(d,))), T(0.1 * rng.normal(size=(d, 32))), T(rng.normal(size=(d,)))
    stream = pk.ops.pack_attn16_stream(w2, b2, wp, p2, c2)
    nwg = (n + 8) // 9
    out = torch.zeros((n + nwg, d), device='cuda')
    so = '/tmp/ca16_stamp.so'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
                    '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-DOCC4D_CA16_STAMP'] + sys.argv[1:] +
                   [os.path.join(CSRC, 'crossattn16.hip'), os.path.join(CSRC, 'error.hip'), '-o', so], check=True,
                   stderr=subprocess.DEVNULL)
    lib = C.CDLL(so)
    fn = lib.occ4d_pt_cross_attn16_f32
    fn.restype = C.c_int
    fn.argtypes = pk._lib.SIGNATURES['occ4d_pt_cross_attn16_f32'][1]
    run = lambda: fn(aq.data_ptr(), 2 * d, qpos.data_ptr(), 3, apos.data_ptr(), 3, idx.data_ptr(), kt.data_ptr(), 2 * d,   # noqa: E731
                     vt.data_ptr(), d, P1.data_ptr(), c1.data_ptr(), stream.data_ptr(),
                     out.data_ptr(), d, n, m, k, d, float(np.sqrt(np.float32(d))), None)
    for _ in range(2):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    st = out[n:, :32].cpu().numpy().reshape(nwg, 8, 4)[:, :, :3]
    tot = st.sum(-1)
    print('kernel %.3f ms; per wave (mean over %d workgroups): prologue %.0f  loop %.0f  epilogue %.0f  total %.0f cycles'
          % (e0.elapsed_time(e1), nwg, st[:, :, 0].mean(), st[:, :, 1].mean(), st[:, :, 2].mean(), tot.mean()))
    print('ideal loop: 26 blocks x 224 MFMAs x 32 cycles x 2 waves per SIMD = %d cycles; epilogue MFMAs 208 x 32 x 2 = %d'
          % (26 * 224 * 64, 208 * 64))
    print('14 rounds of workgroups x mean total = %.0f cycles -> at the measured wall time the shader clock was %.2f GHz'
          % (14 * tot.mean(), 14 * tot.mean() / (e0.elapsed_time(e1) * 1e6)))
    for w in range(8):
        print('wave %d: prologue %.0f  loop %.0f  epilogue %.0f' % (w, st[:, w, 0].mean(), st[:, w, 1].mean(), st[:, w, 2].mean()))


if __name__ == '__main__':
    main()
