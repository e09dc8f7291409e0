#!/usr/bin/env python3
"""Per-kernel PMC totals (summed over XCD/SE instances, averaged over dispatches) from a rocpd db."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
q = """select s.kernel_name, p.name, sum(e.value), count(distinct d.id), avg(d.end-d.start)
 from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id=p.id join rocpd_kernel_dispatch d on e.event_id=d.event_id
 join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name,p.name"""
print('# source: %s   (value = counter summed over all instances, per dispatch)' % sys.argv[1])
for r in c.execute(q):
    if 'lasr' in r[0]:
        print('%-44s %-24s %16.0f  dispatches=%d avg_ns=%.0f' % (r[0][9:52], r[1], r[2] / r[3], r[3], r[4]))
