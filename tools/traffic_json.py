#!/usr/bin/env python3
"""profiles/rNN_pmc.txt (FETCH_SIZE / WRITE_SIZE passes summarised by tools/pmc_summary.py) -> rNN_traffic.json, the file
bench.py reads for `roofline.traffic`.  usage: traffic_json.py <pmc.txt> <frames_per_launch> <note>"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import raster_source_hash          # noqa: E402  (the sources the counters were measured on)

txt, frames, note = sys.argv[1], int(sys.argv[2]), sys.argv[3]
k = {}
for line in open(txt):
    m = re.match(r'\s*\S*(sr_\w+?_kernel)\S*\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)', line)
    if not m or 'ILb1ELi3ELb1' in line:                       # skip the relaxed-math instantiation
        continue
    name, ctr, val = m.group(1), m.group(2), int(m.group(3))
    d = k.setdefault(name, {})
    d['fetch_kib' if ctr == 'FETCH_SIZE' else 'write_kib'] = val
for d in k.values():
    d['bytes'] = (d.get('fetch_kib', 0) + d.get('write_kib', 0)) * 1024
print(json.dumps({'note': note, 'frames_per_launch': frames, 'source_sha': raster_source_hash(), 'kernels': k}, indent=1))
