#!/usr/bin/env python3
"""profiles/rNN_pmc.txt (FETCH_SIZE / WRITE_SIZE passes summarised by tools/pmc_summary.py) -> rNN_traffic.json, the file
bench.py reads for `roofline.traffic`.  usage: traffic_json.py <pmc.txt> <frames_per_launch> <note> [calibration.json]

Calibration (round 4, tools/ubench/traffic.hip -> profiles/r04_traffic_calibration.json): on gfx950 rocprofv3's FETCH_SIZE tallies
HALF of the bytes of vector-memory reads at every lane width tried (4 / 8 / 16 B per lane, full lines and the backward's 11-pixel
row segments alike: 128-byte fabric requests counted as 64), ALL of the bytes of scalar-cache reads, and WRITE_SIZE all of the
bytes of vector stores (4 / 16 B per lane and the forward's 32-byte row segments).  So per kernel
    bytes = FETCH_SIZE KiB x 1024 x f_read + WRITE_SIZE KiB x 1024
with f_read = 2 where the fetch is vector loads (setup: the face vertices; backward: pixel-plane gathers, its records are the
scalar part and stay in L2 from the setup launch) and 1 where it is scalar loads (forward: the record / attribute walk; its
vector part, the 8-byte rects, is ~5 MB of unique bytes per 256 frames)."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import raster_source_hash          # noqa: E402  (the sources the counters were measured on)

# (round 6: the pair-walk forward stages its records with vector loads -- sr_forward_pairs.h -- so its fetch is tallied at half too)
READ_CLASS = {'sr_setup_kernel': 'vector', 'sr_forward_kernel': 'vector', 'sr_backward_kernel': 'vector'}

txt, frames, note = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cal = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else None
f_vec = f_sca = f_wr = None
if cal:
    ck = cal['kernels']
    f_vec = ck['lasr_cal_read4']['factor']          # == read8 == read16 == the rect pattern's per-line tally
    f_sca = ck['lasr_cal_scalar']['factor']
    f_wr = ck['lasr_cal_write_tile']['factor']
k = {}
for line in open(txt):
    m = re.match(r'\s*\S*(sr_\w+?_kernel)\S*\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)', line)
    if not m or 'ILb1ELi3ELb1' in line:                       # skip the relaxed-math instantiation
        continue
    name, ctr, val = m.group(1), m.group(2), int(m.group(3))
    if name.startswith('sr_forward_pairs'):                   # the forward kernel of launches from 10240 tiles up (ProfScope id sr_forward_kernel)
        name = 'sr_forward_kernel'
    d = k.setdefault(name, {})
    d['fetch_kib' if ctr == 'FETCH_SIZE' else 'write_kib'] = val
for name, d in k.items():
    d['raw_bytes'] = (d.get('fetch_kib', 0) + d.get('write_kib', 0)) * 1024
    if cal:
        fr = f_vec if READ_CLASS.get(name) == 'vector' else f_sca
        d['read_factor'], d['write_factor'] = round(fr, 4), round(f_wr, 4)
        d['fetch_bytes'] = int(d.get('fetch_kib', 0) * 1024 * fr)
        d['write_bytes'] = int(d.get('write_kib', 0) * 1024 * f_wr)
        d['bytes'] = d['fetch_bytes'] + d['write_bytes']
    else:
        d['bytes'] = d['raw_bytes']
print(json.dumps({'note': note, 'frames_per_launch': frames, 'source_sha': raster_source_hash(),
                  'calibration': os.path.basename(sys.argv[4]) if cal else None, 'kernels': k}, indent=1))
