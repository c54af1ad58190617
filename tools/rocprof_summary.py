#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (--kernel-trace --stats) into a per-kernel summary
(CSV on stdout).  Usage: python tools/rocprof_summary.py gpurun_out/prof/r1_results.db"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute('select name, total_calls, total_duration, average, percentage from top_kernels '
                     'order by total_duration desc').fetchall()
    print('kernel,calls,total_us,avg_us,percent')
    for name, calls, tot, avg, pct in rows:
        name = name.split('(')[0].replace(',', ';')
        print('%s,%d,%d,%.0f,%.2f' % (name, calls, tot, avg, pct))


if __name__ == '__main__':
    main()
