"""Static instruction mix of a kernel between `; OCC4D_MARK <name>` comments (asm volatile markers in the source):
fp32 MFMA and plain VALU instructions share the SIMD's vector issue on gfx950 (profiles/micro/valu_beside_mfma.hip:
every VALU instruction beside a saturated v_mfma_f32_16x16x4_f32 stream costs its ~4 issue cycles in full), so
"MFMAs x 32 + VALU x 4" cycles per region is the budget the stamps are compared with.
Usage: python profiles/count_valu.py file.hip kernel_substring [-D...]"""
import collections
import re
import subprocess
import sys


def main():
    src, kern = sys.argv[1], sys.argv[2]
    asm = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-Iinclude',
                          '-Iocclusions-4d_amd/csrc', '-S', '--cuda-device-only', '-o', '-', src] + sys.argv[3:],
                         check=True, capture_output=True, text=True).stdout
    lines = asm.split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and kern in l and re.match(r'^_Z\S+:', l))
    region, counts, order = 'entry', collections.defaultdict(collections.Counter), ['entry']
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith('s_endpgm'):
            break
        m = re.search(r'OCC4D_MARK (\S+)', t)
        if m:
            region = m.group(1)
            if region not in order:
                order.append(region)
            continue
        if not t or t.startswith((';', '.', '//')) or t.endswith(':'):
            continue
        op = t.split()[0]
        if op.startswith('v_mfma'):
            k = 'mfma'
        elif op.startswith('v_'):
            k = 'valu'
        elif op.startswith('ds_'):
            k = 'lds'
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
            k = 'vmem'
        elif op.startswith('s_'):
            k = 'salu'
        else:
            k = 'other'
        counts[region][k] += 1
        if k == 'valu':
            counts[region]['valu:' + op] += 1
    for r in order:
        c = counts[r]
        print('%-14s mfma %5d  valu %5d  lds %4d  vmem %4d  salu %5d' % (r, c['mfma'], c['valu'], c['lds'], c['vmem'], c['salu']))
        top = sorted(((v, k[5:]) for k, v in c.items() if k.startswith('valu:')), reverse=True)[:10]
        print('               ' + '  '.join('%s x%d' % (k, v) for v, k in top))


if __name__ == '__main__':
    main()
