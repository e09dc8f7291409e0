#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table.

    python tools/rocpd_stats.py gpurun_out/prof_x/x_results.db [LAST_MS] > profiles/rNN_kernel_stats.txt

Same columns as `rocprofv3 --stats` kernel_stats.csv (calls, total, average, min, max, percentage).
With --pmc the database also carries counter samples; those are averaged per kernel.
"""
import sqlite3
import sys


def main(path, last_ms=None):
    c = sqlite3.connect(path)
    where = ''
    if last_ms is not None:      # steady state only: dispatches that started in the last `last_ms` of the trace
        t_end = c.execute('select max(end) from rocpd_kernel_dispatch').fetchone()[0]
        where = 'where d.start >= %d' % (t_end - int(last_ms * 1e6))
    q = """select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start),
                  max(d.end-d.start), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           %s group by s.kernel_name order by 3 desc""" % where
    rows = list(c.execute(q))
    if last_ms is not None:
        print('# window: last %.1f ms of the trace; kernel-busy time %.2f ms' % (last_ms, sum(r[2] for r in rows) / 1e6))
    total = sum(r[2] for r in rows) or 1
    print('# source: %s' % path)
    print('%-70s %7s %14s %12s %12s %12s %7s %5s %5s %7s' % ('Name', 'Calls', 'TotalNs', 'AvgNs', 'MinNs', 'MaxNs',
                                                              'Pct', 'VGPR', 'SGPR', 'LDS'))
    for r in rows:
        print('%-70s %7d %14d %12.0f %12d %12d %7.2f %5d %5d %7d' % (r[0][:70], r[1], r[2], r[3], r[4], r[5],
                                                                      100.0 * r[2] / total, r[6] or 0, r[7] or 0, r[8] or 0))
    try:
        q = """select s.kernel_name, p.name, avg(e.value), count(*)
               from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
               join rocpd_kernel_dispatch d on e.event_id = d.event_id
               join rocpd_info_kernel_symbol s on d.kernel_id = s.id
               group by s.kernel_name, p.name order by s.kernel_name, p.name"""
        pm = list(c.execute(q))
        if pm:
            print('\n# PMC counters (average per dispatch)')
            for r in pm:
                print('%-70s %-28s %18.1f  (n=%d)' % (r[0][:70], r[1], r[2], r[3]))
    except sqlite3.Error as e:
        print('# no pmc data (%s)' % e)


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else None)
