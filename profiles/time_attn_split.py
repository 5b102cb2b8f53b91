"""The split-precision attention kernel (csrc/crossattn_bf16x6.hip, template over the scheme) stand-alone on one decode
chunk, with timing-only ablation builds.

    python profiles/time_attn_split.py [scheme ...] [-DOCC4D_XA_ABL_...]

Every -D set given (comma separated inside one argument = one variant with several macros; several arguments = several
variants) rebuilds the file into /tmp and times it next to the shipped kernel; results of an ablated kernel are garbage.
Macros: NOSPLIT (no ReLU / split VALU: accumulators reinterpreted as operands), NOGEMM1, NOINIT (no per-stage Aq / Kt
global loads), NODMA (no L2 -> LDS stream after stage 0), HALFGEMM2 (every other channel tile skipped).
MFMA FLOP executed per launch: workgroups x 8 waves x (26 x 30 + 28) tiles x products x 16384."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import occlusions4d_amd as pk  # noqa: E402
from occlusions4d_amd import ops  # noqa: E402

SIG = ops._lib.SIGNATURES


def variant(defs):
    csrc, build = os.path.join(ROOT, 'occlusions-4d_amd', 'csrc'), os.path.join(ROOT, 'occlusions-4d_amd', 'build')
    src = os.path.join(csrc, 'crossattn_bf16x6.hip')
    for d in [d for d in defs if d.startswith('-SRC=')]:       # (-SRC=<file>: another revision of the kernel source)
        defs = [x for x in defs if x != d]
        src = '/tmp/xa_src_%d.hip' % (abs(hash(d)) % 100000)
        with open(d[5:]) as fi, open(src, 'w') as fo:
            text = fi.read()
            # an older revision of the file lacks entry points the rest of the library links against: stubs
            if 'occ4d_pt_cross_attn_f16x3_hidden_scale' not in text:
                text += '\nextern "C" float occ4d_pt_cross_attn_f16x3_hidden_scale(void) { return 1.f; }\n'
            if 'occ4d_pt_cross_attn_f16x3_prescaled_f32' not in text:
                text += ('extern "C" int occ4d_pt_cross_attn_f16x3_prescaled_f32(const float*, int64_t, const float*, int64_t, '
                         'const float*, int64_t, const int32_t*, const float*, int64_t, const float*, int64_t, const float*, '
                         'const float*, const float*, float*, int64_t, int, int, int, int, float, void*) { return -1; }\n')
            if 'occ4d_pack_pair_mlp_bf16x6_stream_f32' not in text:
                text += ('extern "C" int occ4d_pack_pair_mlp_bf16x6_stream_f32(const float*, const float*, const float*, float*, '
                         'void*) { return -1; }\n')
            fo.write(text)
    tag = '_'.join(d.replace('-DOCC4D_XA_ABL_', '') for d in defs) or os.path.basename(src)
    obj, so = '/tmp/xa_%s.o' % tag, '/tmp/xa_%s.so' % tag
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
                    '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-fno-honor-nans'] + defs +
                   ['-c', src, '-o', obj], check=True)
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so, obj] +
                   [os.path.join(build, f) for f in sorted(os.listdir(build)) if f.endswith('.o') and f != 'crossattn_bf16x6.o'],
                   check=True)
    K = C.CDLL(so)
    for name in ('occ4d_pt_cross_attn_bf16x6_f32', 'occ4d_pt_cross_attn_f16x3_f32', 'occ4d_pt_cross_attn_f16x3_prescaled_f32'):
        if hasattr(K, name):
            getattr(K, name).restype, getattr(K, name).argtypes = SIG[name]
    return tag, K


def variant_w(defs):
    """A timing-only build of csrc/crossattn_f16w.hip (-DOCC4D_XW_ABL_...)."""
    csrc, build = os.path.join(ROOT, 'occlusions-4d_amd', 'csrc'), os.path.join(ROOT, 'occlusions-4d_amd', 'build')
    tag = '_'.join(d.replace('-DOCC4D_XW_ABL_', '') for d in defs)
    obj, so = '/tmp/xw_%s.o' % tag, '/tmp/xw_%s.so' % tag
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
                    '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-fno-honor-nans'] + defs +
                   ['-c', os.path.join(csrc, 'crossattn_f16w.hip'), '-o', obj], check=True)
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so, obj] +
                   [os.path.join(build, f) for f in sorted(os.listdir(build)) if f.endswith('.o') and f != 'crossattn_f16w.o'],
                   check=True)
    return tag, C.CDLL(so)


def main():
    schemes = [a for a in sys.argv[1:] if not a.startswith('-')] or ['bf16x6', 'f16x3']
    variants = [[d for d in a.split(',')] for a in sys.argv[1:] if a.startswith('-')]
    n, m, d, k = 32256, 531, 416, 14
    rng = np.random.default_rng(0)
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).cuda()   # noqa: E731
    aq, kt, vt = T(rng.normal(size=(n, 2 * d))), T(rng.normal(size=(m, 2 * d))), T(rng.normal(size=(m, d)))
    # queries in the bench's order: consecutive points of the BASELINE grid (z fastest), so that the 18 queries of a
    # workgroup share neighbours as they do in the decode; QUERIES=random gives independent uniform queries instead
    if os.environ.get('QUERIES', 'grid') == 'grid':
        grid = pk.geometry.sample_implicit_points_blind_numpy(524288, -1.0, 5.0, 3, 'greater', 4, 'grid')
        qpos = T(grid[200000:200000 + n, :3])
    else:
        qpos = T(rng.uniform(-5, 5, size=(n, 3)))
    apos = T(np.concatenate([rng.uniform(-5, 5, size=(m, 2)), rng.uniform(-1, 5, size=(m, 1))], axis=1))
    idx = ops.knn(qpos, apos, k, metric=0)
    P1, c1 = T(rng.normal(size=(32, 3))), T(rng.normal(size=(32,)))
    wp, w2, p2 = T(0.1 * rng.normal(size=(2 * d, 32))), T(0.03 * rng.normal(size=(d, 2 * d))), T(0.1 * rng.normal(size=(d, 32)))
    out = torch.empty((n, d), device='cuda')
    L = ops._lib.lib()
    libs = [('shipped', L)] + [variant(v) for v in variants if not any('XW_' in d for d in v)]
    wgs = 2 * -(-n // 18)
    for scheme in schemes:
        if scheme == 'f16w':            # the fp16 scheme on 32 x 32 x 16 instructions (csrc/crossattn_f16w.hip)
            ws = torch.empty((int(L.occ4d_pt_cross_attn_f16w_stream_floats()),), dtype=torch.float32, device='cuda')
            ops._lib.check(L.occ4d_pack_attn_f16w_stream_f32(ops._ptr(w2), ops._ptr(wp), ops._ptr(p2), ops._ptr(ws), ops._stream()))
            flop = -(-n // 9) * 4 * (26 * 84 + 78) * 32768.0
            for tag, K in [('shipped', L)] + [variant_w(v) for v in variants if any('XW_' in d for d in v)]:
              fn_w = K.occ4d_pt_cross_attn_f16w_f32
              fn_w.restype, fn_w.argtypes = SIG['occ4d_pt_cross_attn_f16w_f32']

              def run_w():
                assert fn_w(
                    ops._ptr(aq), 2 * d, ops._ptr(qpos), 3, ops._ptr(apos), 3, ops._ptr(idx), ops._ptr(kt), 2 * d, ops._ptr(vt), d,
                    ops._ptr(P1), ops._ptr(c1), ops._ptr(ws), ops._ptr(out), d, n, m, k, d, float(np.sqrt(np.float32(d))),
                    ops._stream()) == 0
              for _ in range(3):
                run_w()
              e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
              e0.record()
              for _ in range(20):
                run_w()
              e1.record()
              torch.cuda.synchronize()
              ms = e0.elapsed_time(e1) / 20
              print('%-7s %-28s %7.3f ms   %5.0f TFLOP/s of executed 32x32x16 MFMA = %.3f of 2.5 PF (full kernel count)'
                    % (scheme, tag, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0), flush=True)
            continue
        size, pack, name = ((L.occ4d_pt_cross_attn_f16x3_stream_floats, L.occ4d_pack_attn_f16x3_stream_f32,
                             'occ4d_pt_cross_attn_f16x3_f32') if scheme == 'f16x3' else
                            (L.occ4d_pt_cross_attn_bf16x6_stream_floats, L.occ4d_pack_attn_bf16x6_stream_f32,
                             'occ4d_pt_cross_attn_bf16x6_f32'))
        pack_name = 'occ4d_pack_attn_f16x3_stream_f32' if scheme == 'f16x3' else 'occ4d_pack_attn_bf16x6_stream_f32'
        streams = {}
        for tag, K in libs:                 # (every revision packs its own stream: the layouts differ)
            fnp = getattr(K, pack_name)
            fnp.restype, fnp.argtypes = SIG[pack_name]
            ws = torch.empty((int(size()),), dtype=torch.float32, device='cuda')
            ops._lib.check(fnp(ops._ptr(w2), ops._ptr(wp), ops._ptr(p2), ops._ptr(ws), ops._stream()))
            streams[tag] = ws
        prod = 3 if scheme == 'f16x3' else 6
        flop = wgs * 8 * (26 * 30 + 28) * prod * 16384.0
        entries = [(tag, K, name, tag) for tag, K in libs]
        if scheme == 'f16x3':           # the path-level calls: init term pre-multiplied by the producers (timing: same data)
            entries += [(tag + ' prescaled', K, 'occ4d_pt_cross_attn_f16x3_prescaled_f32', tag) for tag, K in libs
                        if hasattr(K, 'occ4d_pt_cross_attn_f16x3_prescaled_f32') and not tag.startswith('xa_src')]
        for tag, K, entry, stream_of in entries:
            fn = getattr(K, entry)
            ws = streams[stream_of]

            def run():
                rc = fn(ops._ptr(aq), 2 * d, ops._ptr(qpos), 3, ops._ptr(apos), 3, ops._ptr(idx), ops._ptr(kt), 2 * d,
                        ops._ptr(vt), d, ops._ptr(P1), ops._ptr(c1), ops._ptr(ws), ops._ptr(out), d, n, m, k, d,
                        float(np.sqrt(np.float32(d))), ops._stream())
                assert rc == 0
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print('%-7s %-28s %7.3f ms   %5.0f TFLOP/s of executed 16x16x32 MFMA = %.3f of 2.5 PF (full kernel count)'
                  % (scheme, tag, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0), flush=True)


if __name__ == '__main__':
    main()
