"""Ablation of the fused vector-attention kernel on the GPU box: rebuilds crossattn.hip with
OCC4D_ABLATE_* switches into scratch libraries and times one 32768-query launch of each.
(Ablated variants compute wrong results on purpose; only their timing is read.)"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import occlusions4d_amd as pk  # noqa: E402

CSRC = os.path.join(ROOT, 'occlusions-4d_amd', 'csrc')
VARIANTS = {'base': [], 'p1_1': ['-DOCC4D_CA_P1=1'], 'p1_3': ['-DOCC4D_CA_P1=3'], 'p1_4': ['-DOCC4D_CA_P1=4'], 'p2_2': ['-DOCC4D_CA_P2=2'],
            'p2_5': ['-DOCC4D_CA_P2=5'], 'p1_4_p2_5': ['-DOCC4D_CA_P1=4', '-DOCC4D_CA_P2=5'], 'nopipe': ['-DOCC4D_CA_NO_PIPE'], 'tail8': ['-DOCC4D_CA_TAIL=8'], 'tail16': ['-DOCC4D_CA_TAIL=16'], 'tail32': ['-DOCC4D_CA_TAIL=32'], 'noload': ['-DOCC4D_ABLATE_NOLOAD'], 'nobar': ['-DOCC4D_ABLATE_NOBAR', '-DOCC4D_ABLATE_NOLOAD'],
            'noinit': ['-DOCC4D_ABLATE_NOINIT'], 'noepi': ['-DOCC4D_ABLATE_NOEPI'], 'epi_nov': ['-DOCC4D_ABLATE_EPI_NOV'], 'epi_nope': ['-DOCC4D_ABLATE_EPI_NOPE'],
            'epi_nov_nope': ['-DOCC4D_ABLATE_EPI_NOV', '-DOCC4D_ABLATE_EPI_NOPE'],
            'all': ['-DOCC4D_ABLATE_NOLOAD', '-DOCC4D_ABLATE_NOBAR', '-DOCC4D_ABLATE_NOINIT', '-DOCC4D_ABLATE_NOEPI']}


def build(name, flags):
    out = '/tmp/ca_%s.so' % name
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
           '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC] + flags + \
          [os.path.join(CSRC, 'crossattn.hip'), os.path.join(CSRC, 'error.hip'), '-o', out]
    subprocess.run(cmd, check=True)
    return out


def main():
    n, m, k, d = 32256, 531, 14, 416
    g = torch.Generator(device='cuda').manual_seed(0)
    R = lambda *s: torch.randn(*s, device='cuda', generator=g)
    aq, kt, vt = R(n, 2 * d), R(m, 2 * d), R(m, d)
    qpos, apos = R(n, 4), R(m, 3)
    idx = torch.randint(0, m, (n, k), device='cuda', dtype=torch.int32)
    P1, c1, wp, w2, b2, p2, c2 = R(32, 3), R(32), R(2 * d, 32) * 0.1, R(d, 2 * d) * 0.03, R(d), R(d, 32) * 0.1, R(d)
    agg = torch.empty(n, d, device='cuda')
    P = lambda t: C.c_void_p(t.data_ptr())
    only = sys.argv[1:]
    for name, flags in VARIANTS.items():
        if only and name not in only:
            continue
        lib = C.CDLL(build(name, flags))
        fn = lib.occ4d_pt_cross_attn_f32
        fn.restype = C.c_int
        fn.argtypes = pk._lib.SIGNATURES['occ4d_pt_cross_attn_f32'][1]

        def run():
            rc = fn(P(aq), 2 * d, P(qpos), 4, P(apos), 3, P(idx), P(kt), 2 * d, P(vt), d, P(P1), P(c1), P(wp), P(w2),
                    P(b2), P(p2), P(c2), P(agg), d, n, m, k, d, 20.396078, None)
            assert rc == 0
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        print('%-8s %.3f ms' % (name, e0.elapsed_time(e1) / 5), flush=True)


if __name__ == '__main__':
    main()
