"""Summarise rocprofv3 --pmc passes (profiles/run_pmc.sh): per kernel and counter, the per-dispatch value summed over
all XCD / SE instances, averaged over the kernel's full-size dispatches; derived HBM traffic and MFMA-pipe occupancy.

    python profiles/summarize_pmc.py <pmc_dir> [--json profiles/pmc_traffic.json --kind greater]

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE / WRITE_SIZE are in KiB and FETCH_SIZE reports
half the bytes of 16-byte-per-lane streaming reads on gfx950 (MI355X_MICROARCH.md, HBM section).
MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

WANT = ('cross_attn_split_kernel', 'rowlin_split_kernel', 'cross_attn_bf16x6_kernel', 'cross_attn16p_kernel', 'cross_attn16_kernel', 'cross_attn_kernel', 'resblock_kernel', 'rowlin_kernel', 'linear_kernel<13', 'interp_add', 'knn_kernel')


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0].replace(' ', '')


def main():
    root = sys.argv[1]
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))     # kernel -> counter -> dispatch -> sum
    grid = {}
    for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row['Kernel_Name'])
                if not any(w in k for w in WANT):
                    continue
                key = (path, row['Dispatch_Id'])
                per[k][row['Counter_Name']][key] += float(row['Counter_Value'])
                grid[(k, key)] = int(row['Grid_Size'])
    out = {}
    for k in sorted(per):
        # full-size dispatches only: the largest grid of this kernel
        gmax = max(g for (kk, _), g in grid.items() if kk == k)
        vals = {}
        for c, d in per[k].items():
            sel = [v for key, v in d.items() if grid[(k, key)] == gmax]
            vals[c] = (sum(sel) / len(sel), len(sel))
        out[k] = vals
        print('%s  (grid %d threads)' % (k, gmax))
        for c in sorted(vals):
            print('    %-28s dispatches=%3d  per-dispatch=%16.1f' % (c, vals[c][1], vals[c][0]))
        if 'FETCH_SIZE' in vals and 'WRITE_SIZE' in vals:
            hbm = (2 * vals['FETCH_SIZE'][0] + vals['WRITE_SIZE'][0]) * 1024
            print('    -> HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B = %.1f MB' % (hbm / 1e6))
            vals['_hbm'] = (hbm, 0)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in vals and 'GRBM_GUI_ACTIVE' in vals:
            cyc = vals['GRBM_GUI_ACTIVE'][0] / 8
            print('    -> %.3e cycles per launch; MFMA pipe busy %.1f %%' % (cyc, 100 * vals['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (cyc * 1024)))
        if 'SQ_WAIT_ANY' in vals and 'SQ_WAVE_CYCLES' in vals:
            w = vals['SQ_WAVE_CYCLES'][0]
            print('    -> of wave cycles: waiting (s_waitcnt / barrier) %.1f %%, issue stall %.1f %%, issuing %.1f %%' % (
                100 * vals['SQ_WAIT_ANY'][0] / w, 100 * vals.get('SQ_WAIT_INST_ANY', (0, 0))[0] / w,
                100 * vals.get('SQ_ACTIVE_INST_ANY', (0, 0))[0] / w))
    if '--json' in sys.argv:
        path = sys.argv[sys.argv.index('--json') + 1]
        kind = sys.argv[sys.argv.index('--kind') + 1] if '--kind' in sys.argv else 'greater'
        rec = {}
        if os.path.exists(path):
            with open(path) as f:
                rec = json.load(f)
        ca = [k for k in out if ('cross_attn16' in k or 'cross_attn_kernel<13' in k or 'cross_attn_split' in k) and '_hbm' in out[k]]
        ca.sort(key=lambda k: (0 if 'cross_attn_split' in k else 1 if 'cross_attn16p' in k else 2 if 'cross_attn16' in k else 3))
        if ca:
            v = out[ca[0]]
            rec[kind] = dict(hbm_bytes_per_launch=v['_hbm'][0], fetch_size_kib=v['FETCH_SIZE'][0],
                             write_size_kib=v['WRITE_SIZE'][0], dispatches=v['FETCH_SIZE'][1],
                             source='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, profiles/run_pmc.sh) of '
                                    '%s on one %s decode chunk (32256 queries); (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B'
                                    % (ca[0], kind))
            with open(path, 'w') as f:
                json.dump(rec, f, indent=1)
            print('wrote', path)


if __name__ == '__main__':
    main()
