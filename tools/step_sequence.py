#!/usr/bin/env python3
"""Ordered kernel sequence of ONE steady-state optimisation iteration from a rocprofv3 kernel trace (rocpd db):
    python tools/step_sequence.py <db>  > profiles/rNN_step_sequence.txt
The iteration is delimited by two consecutive launches of tail_adamw_kernel (the last kernel of a step)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d
                         join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""))
ends = [i for i, r in enumerate(rows) if 'tail_adamw' in r[0]]
a, b = ends[-2] + 1, ends[-1] + 1
seq = rows[a:b]
busy = sum(r[2] - r[1] for r in seq)
print('# %d kernels, busy %.3f ms, wall %.3f ms' % (len(seq), busy / 1e6, (seq[-1][2] - seq[0][1]) / 1e6))
t0 = seq[0][1]
for name, s, e in seq:
    short = name.replace('void ', '')[:110]
    print('%9.1f %8.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, short))
