"""The counter files bench.py reads for `roofline.traffic` / `roofline.valu_frac` (profiles/*_traffic.json, *_valu.json) must have
been measured on the raster sources that are in the tree: each stores the sha256 of lasr_amd/csrc/{sr_raster.hip, sr_forward_coop.h,
sr_forward_pairs.h, sr_device.h, sr_common.h, sr_backward.h, sr_backward_fast.hip, Makefile} it was produced from (tools/traffic_json.py, tools/valu_json.py).
A kernel change without a new PMC pass fails here (and bench.py then reports the figures as null with a `stale` note instead
of quoting old counters)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_latest_counter_files_match_the_raster_sources():
    import bench
    sha = bench.raster_source_hash()
    for suffix in ('_traffic.json', '_valu.json'):
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*' + suffix)))
        assert files, 'no profiles/*%s committed' % suffix
        d = json.load(open(files[-1]))
        assert d.get('source_sha') == sha, ('%s was measured on other raster sources (%s, now %s): run tools/prof/r06_final.sh on '
                                             'the GPU box and commit its output' % (os.path.basename(files[-1]), d.get('source_sha'), sha))
