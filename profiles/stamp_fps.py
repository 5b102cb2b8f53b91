"""Per-wave cycle accounting of one step of the pruned FPS kernel (debug build with s_memtime stamps; 1 tick = 1 shader
cycle, profiles/micro/memtime_calib.hip).  Usage: python profiles/stamp_fps.py [n m] [extra -D flags]"""
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
    args = [a for a in sys.argv[1:] if not a.startswith('-')]
    flags = [a for a in sys.argv[1:] if a.startswith('-')]
    sizes = [(int(args[0]), int(args[1]))] if len(args) >= 2 else [(14336, 4779), (4779, 1593), (2049, 683)]
    out = '/tmp/fps_stamp.so'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
                    '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-DOCC4D_FPSB_STAMP', '-DOCC4D_FPS_STAMP'] + flags +
                   [os.path.join(CSRC, 'fps_bucket.hip'), os.path.join(CSRC, 'fps.hip'), os.path.join(CSRC, 'error.hip'),
                    '-o', out], check=True, stderr=subprocess.DEVNULL)
    lib = C.CDLL(out)
    fn = lib.occ4d_fps_f32
    fn.restype = C.c_int
    fn.argtypes = pk._lib.SIGNATURES['occ4d_fps_f32'][1]
    names = ['box test', 'bucket updates', 'wave max', 'index + coordinates + floor', 'publish+barrier', 'candidates from LDS',
             'candidate round', 'accepted from LDS']
    exhaustive_names = ['distance update', 'wave max + index + coordinates', 'publish+barrier', 'block winner', '-', '-', '-', '-']
    pruned_names = names
    for n, m in sizes:
        names = exhaustive_names if (os.environ.get('OCC4D_FPS_PRUNE') == '0' or n < 1536) else pruned_names
        level = pk.configs.synthetic_pcl('greater', n, 12)[0].cuda()[:, :3].contiguous()
        sel = torch.empty(m, dtype=torch.int32, device='cuda')
        order = torch.zeros(m + 2 + 8 * 16 * 2, dtype=torch.int32, device='cuda')
        for _ in range(2):
            assert fn(level.data_ptr(), 3, n, m, sel.data_ptr(), order.data_ptr(), None) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(level.data_ptr(), 3, n, m, sel.data_ptr(), order.data_ptr(), None)
        e1.record()
        torch.cuda.synchronize()
        s0 = (m + 1) & ~1
        if names is pruned_names:
            raw = order[s0:s0 + 256].cpu().numpy().view(np.int64).reshape(-1, 16)[:8]
            print('rounds %d: %.2f samples per round' % (raw[0, 8], (m - 1) / max(raw[0, 8], 1)))
        else:
            raw = order[s0:s0 + 128].cpu().numpy().view(np.int64).reshape(-1, 8)[:8]
        st = raw[:, :8].astype(np.float64) / (m - 1)
        print('n=%d m=%d: %.3f ms, %.3f us/step; cycles per step (mean over steps), per wave:' %
              (n, m, e0.elapsed_time(e1), 1e3 * e0.elapsed_time(e1) / m))
        nw = 4 if (n <= 7168 and (os.environ.get('OCC4D_FPS_PRUNE') == '0' or n < 1536)) else 8
        st = st[:nw]
        for w in range(nw):
            print('  wave %d: ' % w + '  '.join('%s %.0f' % (names[i], st[w, i]) for i in range(8)) +
                  '  total %.0f' % st[w].sum())
        print('  mean  : ' + '  '.join('%s %.0f' % (names[i], st[:, i].mean()) for i in range(8)) +
              '  total %.0f' % st.sum(1).mean())


main()
