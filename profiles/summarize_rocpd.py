"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / average.
Usage: python profiles/summarize_rocpd.py <results.db> [steps_equivalent]"""
import sqlite3
import sys


def main(path, top=25):
    cur = sqlite3.connect(path).cursor()
    q = ("select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, "
         "min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "group by s.kernel_name order by 3 desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    print('%-72s %7s %10s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', '%'))
    for r in rows[:top]:
        print('%-72s %7d %10.2f %10.1f %10.1f %10.1f %6.1f' % (r[0][:72], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
    print('total kernel time %.2f ms over %d kernels' % (tot, len(rows)))


if __name__ == '__main__':
    main(sys.argv[1])
