#!/usr/bin/env python3
"""tools/ubench/traffic.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -> calibration factors
true bytes / (counter KiB x 1024) per access pattern.  usage: traffic_cal_json.py <true.json> <fetch.db> <write.db>"""
import json
import sqlite3
import sys

true = json.load(open(sys.argv[1]))['true_bytes']
Q = """select s.kernel_name, p.name, sum(e.value), count(distinct d.id)
 from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id=p.id join rocpd_kernel_dispatch d on e.event_id=d.event_id
 join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name, p.name"""
out = {'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over tools/ubench/traffic.hip: kernels that move a '
               'byte count known by construction through 1 GiB buffers (beyond the 256 MiB Infinity Cache), in the access patterns '
               'of the raster kernels; counter values are KiB summed over all TCC instances, per dispatch; factor = true bytes / '
               '(KiB x 1024).  A factor of 2 on a read means the counter tallies half of the bytes (the gfx950 behaviour '
               'MI355X_MICROARCH.md documents for 16 B/lane streams); a factor below 1 on a write means more bytes leave the L2 than '
               'the kernel stores (partial-line write-backs).',
       'kernels': {}}
for db in sys.argv[2:4]:
    for name, ctr, val, n in sqlite3.connect(db).execute(Q):
        key = next((k for k in true if k + '(' in name or name.startswith(k) or k in name), None)
        if key is None:
            continue
        # longest matching key (lasr_cal_read4 vs lasr_cal_read4_rect)
        key = max((k for k in true if k in name), key=len)
        kib = val / n
        d = out['kernels'].setdefault(key, {'true_bytes': true[key]})
        d[ctr.lower() + '_kib'] = kib
        d[ctr.lower() + '_bytes_over_true'] = kib * 1024 / true[key]
reads = {'lasr_cal_read16', 'lasr_cal_read4', 'lasr_cal_read8', 'lasr_cal_read4_rect', 'lasr_cal_scalar'}
for k, d in out['kernels'].items():
    ctr = 'fetch_size_kib' if k in reads else 'write_size_kib'
    if d.get(ctr):
        d['factor'] = d['true_bytes'] / (d[ctr] * 1024)
print(json.dumps(out, indent=1))
