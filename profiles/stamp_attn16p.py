"""Phase accounting of crossattn16p.hip (debug build with absolute s_memtime stamps + HW_ID per wave): how long a
pass's prologue / hidden-stage loop / epilogue take, where the workgroups sit (XCC / SE / CU / wave slot) and how far
the two workgroups of a CU are out of phase.  Usage: python profiles/stamp_attn16p.py [skew] [extra -D flags]"""
import collections
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
    skew = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n, m, d, k = 32256, 531, 416, 14
    rng = np.random.default_rng(0)
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).cuda()   # noqa: E731
    aq, kt, vt = T(rng.normal(size=(n, 2 * d))), T(rng.normal(size=(m, 2 * d))), T(rng.normal(size=(m, d)))
    qpos, apos = T(rng.uniform(-5, 5, size=(n, 3))), T(rng.uniform(-5, 5, size=(m, 3)))
    idx = pk.ops.knn(qpos, apos, k, metric=0)
    P1, c1 = T(rng.normal(size=(32, 3))), T(rng.normal(size=(32,)))
    wp, w2 = T(0.1 * rng.normal(size=(2 * d, 32))), T(0.03 * rng.normal(size=(d, 2 * d)))
    b2, p2, c2 = T(rng.normal(size=(d,))), T(0.1 * rng.normal(size=(d, 32))), T(rng.normal(size=(d,)))
    stream = pk.ops.pack_attn16p_stream(w2, b2, wp, p2, c2)
    vt = vt + c2           # the kernel reads the value table with pos_mlp[2].bias folded in
    nwg = (n + 8) // 9
    out = torch.zeros((n + nwg, d), device='cuda')
    so = '/tmp/ca16p_stamp.so'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
                    '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-DOCC4D_CA16P_STAMP', '-fno-honor-nans'] + sys.argv[2:] +
                   [os.path.join(CSRC, 'crossattn16p.hip'), os.path.join(CSRC, 'error.hip'), '-o', so], check=True,
                   stderr=subprocess.DEVNULL)
    lib = C.CDLL(so)
    fn = lib.occ4d_pt_cross_attn16p_f32
    fn.restype = C.c_int
    fn.argtypes = pk._lib.SIGNATURES['occ4d_pt_cross_attn16p_f32'][1]
    run = lambda: fn(aq.data_ptr(), 2 * d, qpos.data_ptr(), 3, apos.data_ptr(), 3, idx.data_ptr(), kt.data_ptr(), 2 * d,   # noqa: E731
                     vt.data_ptr(), d, P1.data_ptr(), c1.data_ptr(), stream.data_ptr(),
                     out.data_ptr(), d, n, m, k, d, float(np.sqrt(np.float32(d))), skew, None)
    for _ in range(2):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    raw = out[n:].cpu().numpy().view(np.uint64).reshape(nwg, d // 2)[:, :40].reshape(nwg, 4, 10)
    ts = raw[:, :, :8].astype(np.int64)
    hw = raw[:, :, 8].astype(np.int64)
    xcc = raw[:, :, 9].astype(np.int64) & 15
    rt = (raw[:, :, 9].astype(np.int64) >> 8).astype(np.float64)      # s_memrealtime (100 MHz) ticks of the workgroup
    t0 = ts[:, :, 0].min()
    ts = ts - t0
    print('kernel %.3f ms, skew %d; span of the stamps %.0f cycles' % (e0.elapsed_time(e1), skew, ts[:, :, 7].max()))
    names = ['start->loop A', 'loop A', 'epilogue A', 'A->loop B', 'loop B', 'epilogue B', 'tail']
    seg = [ts[:, :, 1] - ts[:, :, 0], ts[:, :, 2] - ts[:, :, 1], ts[:, :, 3] - ts[:, :, 2], ts[:, :, 4] - ts[:, :, 3],
           ts[:, :, 5] - ts[:, :, 4], ts[:, :, 6] - ts[:, :, 5], ts[:, :, 7] - ts[:, :, 6]]
    tot = ts[:, :, 7] - ts[:, :, 0]
    for nm, sg in zip(names, seg):
        print('  %-14s mean %9.0f  (%.1f %% of a workgroup)  first round %9.0f  later rounds %9.0f'
              % (nm, sg.mean(), 100 * sg.mean() / tot.mean(), sg[:512].mean(), sg[512:].mean()))
    print('  shader clock: %.0f cycles in %.1f us of s_memrealtime -> %.3f GHz' % (tot.mean(), rt.mean() / 100.0, tot.mean() / (rt.mean() * 10.0)))
    print('  workgroup total mean %.0f cycles; ideal MFMA time of a workgroup sharing its SIMDs with one other: '
          '2 passes x 6032 MFMAs x 32 cycles x 2 = %d' % (tot.mean(), 2 * 6032 * 64))
    wave_id, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    print('wave slots seen (wave 0 of every workgroup):', dict(collections.Counter(wave_id[:, 0].tolist())))
    print('SIMD of waves 0..3 (first 4 workgroups):', simd[:4].tolist())
    # co-residents: same (xcc, se, sh, cu); phase = distance between their loop-A ends while both run
    place = collections.defaultdict(list)
    for b in range(nwg):
        place[(int(xcc[b, 0]), int(se[b, 0]), int(sh[b, 0]), int(cu[b, 0]))].append(b)
    print('distinct CUs seen: %d; workgroups per CU: min %d max %d' % (len(place), min(map(len, place.values())),
                                                                        max(map(len, place.values()))))
    phases = []
    for wgs in place.values():
        wgs.sort(key=lambda b: ts[b, 0, 0])
        for i, b in enumerate(wgs):
            for b2_ in wgs[i + 1:]:
                if ts[b2_, 0, 0] < ts[b, 0, 7] and ts[b, 0, 0] < ts[b2_, 0, 7]:
                    phases.append(abs(int(ts[b, 0, 2]) - int(ts[b2_, 0, 2])))
    ph = np.array(phases)
    print('co-resident pairs: %d; |loop-A end offset| mean %.0f  median %.0f  p10 %.0f  p90 %.0f cycles'
          % (len(ph), ph.mean(), np.median(ph), np.percentile(ph, 10), np.percentile(ph, 90)))
    b = 0
    print('workgroup 0 stamps (wave 0):', ts[b, 0].tolist(), 'hw_id %08x' % hw[b, 0])
    print('workgroup 1 stamps (wave 0):', ts[1, 0].tolist(), 'hw_id %08x' % hw[1, 0])


if __name__ == '__main__':
    main()
