"""Build libocc4d.so (HIP, gfx950 only) in-tree with hipcc.

    python occlusions-4d_amd/build.py [--force]

One object per .hip file (compiled in parallel), linked into
occlusions-4d_amd/libocc4d.so.  -ffp-contract=off: the kNN/FPS distance arithmetic
is pinned bit-for-bit to the reference's CPU results, so only explicit fmaf()
may fuse.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
OBJDIR = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libocc4d.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-I' + INCLUDE, '-I' + CSRC]
FLAGS += os.environ.get('OCC4D_HIPCC_EXTRA', '').split()      # experiments only (e.g. -DOCC4D_CA_NO_XCD_MAP)
# per-file flags.  -fno-honor-nans: fmaxf() on MFMA / lane-swap results otherwise gets a canonicalising v_max v, v, v
# in front of every maximum, and in these kernels every VALU instruction costs matrix time (profiles/micro/
# valu_beside_mfma.hip).  The kernels' masking uses infinities (still honoured), never NaNs.
FILE_FLAGS = {f: ['-fno-honor-nans'] for f in ('crossattn16p.hip', 'trunk4.hip', 'trunk.hip', 'wgrad16.hip',
                                                'crossattn_bf16x6.hip', 'trunk_bf16x6.hip', 'pairmlp_bf16x6.hip', 'crossattn_f16w.hip', 'resblock_f16x3.hip')}
FILE_FLAGS['wgrad16.hip'] += ['-fno-slp-vectorize']    # (packed fp32 adds come with register shuffles: VALU = matrix time)


def _hipcc():
    return shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hpp')]
    headers.append(os.path.join(INCLUDE, 'occ4d.h'))
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJDIR, src[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
            jobs.append([_hipcc()] + FLAGS + FILE_FLAGS.get(src, []) + ['-c', os.path.join(CSRC, src), '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(8, len(jobs) or 1))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
