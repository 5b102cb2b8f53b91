"""Timeline of one encode from a rocprofv3 --kernel-trace CSV (profiles/probe.py encode N): per kernel start / end
relative to the first kernel of the LAST encode, stream (queue) and gaps.
Usage: python profiles/encode_timeline.py <kernel_trace.csv> [n_encodes_in_trace]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    n_enc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # one encode ends with the row copies / level-id fills that lay out its abstract cloud (round 5: library kernels; up to
    # round 4 a torch.cat), the last of a run of them behind the global mean-pool: the last encode = the kernels behind
    # the second-to-last such end (encodes are issued back to back, so gaps do not delimit them)
    ends, seen_mean = [], False
    for i, r in enumerate(rows):
        name = r['Kernel_Name']
        seen_mean = seen_mean or 'mean_rows_kernel' in name
        tail = 'copy_rows_kernel' in name or 'fill_rows_kernel' in name
        nxt = rows[i + 1]['Kernel_Name'] if i + 1 < len(rows) else ''
        if seen_mean and tail and not ('copy_rows_kernel' in nxt or 'fill_rows_kernel' in nxt):
            ends.append(i)
            seen_mean = False
    assert len(ends) >= 2, 'need at least two encodes in the trace'
    seg = rows[ends[-2] + 1:ends[-1] + 1]
    t0 = int(seg[0]['Start_Timestamp'])
    end = max(int(r['End_Timestamp']) for r in seg)
    print('last encode: %d kernels, %.3f ms from first start to last end' % (len(seg), (end - t0) / 1e6))
    for r in seg:
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        print('%9.1f %9.1f us  %8.1f us  q%-3s %s' % (s, e, e - s, r.get('Queue_Id', '?'), name))


if __name__ == '__main__':
    main()
