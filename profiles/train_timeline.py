"""Where one training step's wall time goes, from a rocprofv3 --kernel-trace CSV of bench_train.py: the last step is cut
at the optimizer's kernels (the library's adamw_kernel -- torch's multi-tensor AdamW launches before round 6 -- ends a step), then per queue: kernel time, idle time between
consecutive kernels, and per kernel name its own time plus the idle time that follows it (what removing it would give
back when the stream is serial).
Usage: python profiles/train_timeline.py <kernel_trace.csv> [top] [sequence.txt | -: the step's launches in order] [back]
(back = 1: the step before the last -- bench_train.py ends with one extra step on ONE stream for its per-kernel table)"""
import collections
import csv
import sys


def short(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', 'at::')[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    opt = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]      # (round 6: the library's fused clip + AdamW)
    if not opt:
        opt = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r['Kernel_Name'] and 'Adam' in r['Kernel_Name']]
    if not opt:
        opt = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r['Kernel_Name']]
    # runs of optimizer kernels: a step ends with the last kernel of a run
    ends = [i for k, i in enumerate(opt) if k + 1 == len(opt) or opt[k + 1] - i > 50]
    assert len(ends) >= 2, 'need two steps in the trace'
    back = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    assert len(ends) >= 2 + back, 'not enough steps in the trace'
    seg = rows[ends[-2 - back] + 1:ends[-1 - back] + 1]
    t0 = int(seg[0]['Start_Timestamp'])
    t1 = max(int(r['End_Timestamp']) for r in seg)
    print('step %d from the end: %d kernels, %.2f ms first start to last end' % (back, len(seg), (t1 - t0) / 1e6))
    if len(sys.argv) > 3 and sys.argv[3] != '-':
        with open(sys.argv[3], 'w') as f:
            prev = {}
            for r in seg:
                q = r.get('Queue_Id', '?')
                s0, e0 = int(r['Start_Timestamp']), int(r['End_Timestamp'])
                f.write('%10.1f us  +%7.1f gap %8.1f us  q%-2s grid %-9s %s\n' % (
                    (s0 - t0) / 1e3, (s0 - prev.get(q, s0)) / 1e3, (e0 - s0) / 1e3, q,
                    r.get('Grid_Size_X', r.get('Grid_Size', '?')), short(r['Kernel_Name'])))
                prev[q] = e0
    queues = collections.defaultdict(list)
    for r in seg:
        queues[r.get('Queue_Id', '?')].append(r)
    for q, rs in sorted(queues.items(), key=lambda kv: -len(kv[1])):
        busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
        idle = 0
        own = collections.defaultdict(lambda: [0, 0, 0])
        for a, b in zip(rs, rs[1:] + [None]):
            d = int(a['End_Timestamp']) - int(a['Start_Timestamp'])
            g = max(0, int(b['Start_Timestamp']) - int(a['End_Timestamp'])) if b is not None else 0
            idle += g
            o = own[short(a['Kernel_Name'])]
            o[0] += 1
            o[1] += d
            o[2] += g
        span = int(rs[-1]['End_Timestamp']) - int(rs[0]['Start_Timestamp'])
        print('queue %s: %d kernels, span %.2f ms, kernel time %.2f ms, idle between kernels %.2f ms' %
              (q, len(rs), span / 1e6, busy / 1e6, idle / 1e6))
        small = [(n, o) for n, o in own.items() if o[1] / o[0] < 20e3]
        print('   kernels under 20 us on average: %d launches, %.2f ms of kernel time, %.2f ms of idle time behind them' %
              (sum(o[0] for _, o in small), sum(o[1] for _, o in small) / 1e6, sum(o[2] for _, o in small) / 1e6))
        print('   %-70s %6s %9s %9s' % ('kernel', 'n', 'time ms', 'idle after'))
        for n, o in sorted(own.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:top]:
            print('   %-70s %6d %9.3f %9.3f' % (n, o[0], o[1] / 1e6, o[2] / 1e6))


if __name__ == '__main__':
    main()
