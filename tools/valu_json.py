#!/usr/bin/env python3
"""profiles/rNN_pmc_sq.txt (SQ instruction counters per dispatch, tools/pmc_summary.py) -> rNN_valu.json: per raster kernel the
VALU wave-instruction mix of one launch and the issue time it implies under the per-instruction costs measured by
tools/ubench/ (profiles/r02_valu_issue.txt).  bench.py attaches it to the JSON line as `valu_issue`.
usage: valu_json.py <pmc_sq.txt> <frames_per_launch>"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import raster_source_hash          # noqa: E402  (the sources the counters were measured on)

COST = {'fp32_vgpr': 2.5, 'fp32_sgpr_operand': 4.15, 'f64': 4.3, 'transcendental': 8.3, 'other': 4.1}   # cycles @ 2.4 GHz per wave64
SGPR_SHARE = {'sr_forward_kernel': 0.05, 'sr_backward_kernel': 0.20}     # static share of fp32 mul/add/fma with an SGPR source (ISA, round 6:
#   32 of 709 in sr_forward_pairs3_kernel -- records come from LDS -- and 106 of 519 in sr_backward_kernel<true, 3>; rounds 2-5: 0.28 / 0.45)
SIMDS, GHZ = 1024, 2.4

txt, frames = sys.argv[1], int(sys.argv[2])
k = {}
for line in open(txt):
    m = re.match(r'\s*\S*(sr_\w+?_kernel)\S*\s+(SQ_\w+)\s+(\d+)', line)
    if not m or 'ILb1ELi3ELb1' in line or 'setup' in line:
        continue
    name = 'sr_forward_kernel' if m.group(1).startswith('sr_forward_pairs') else m.group(1)
    k.setdefault(name, {})[m.group(2)] = int(m.group(3))
out = {'frames_per_launch': frames, 'source_sha': raster_source_hash(), 'cycles_per_wave_instruction': COST, 'simds': SIMDS, 'clock_ghz': GHZ, 'kernels': {}}
for name, c in k.items():
    fp32 = c['SQ_INSTS_VALU_MUL_F32'] + c['SQ_INSTS_VALU_FMA_F32'] + c['SQ_INSTS_VALU_ADD_F32']
    f64 = c['SQ_INSTS_VALU_MUL_F64'] + c['SQ_INSTS_VALU_FMA_F64'] + c['SQ_INSTS_VALU_ADD_F64']
    tr = c['SQ_INSTS_VALU_TRANS_F32']
    other = c['SQ_INSTS_VALU'] - fp32 - f64 - tr
    s = SGPR_SHARE.get(name, 0.3)
    cyc = (fp32 * (1 - s) * COST['fp32_vgpr'] + fp32 * s * COST['fp32_sgpr_operand'] + f64 * COST['f64'] +
           tr * COST['transcendental'] + other * COST['other'])
    out['kernels'][name] = {
        'valu_wave_instructions': c['SQ_INSTS_VALU'], 'fp32_mul_add_fma': fp32, 'f64': f64, 'transcendental': tr, 'other': other,
        'salu': c['SQ_INSTS_SALU'], 'smem': c['SQ_INSTS_SMEM'], 'branch': c.get('SQ_INSTS_BRANCH'),
        'live_lane_fraction': c['SQ_THREAD_CYCLES_VALU'] / (64.0 * c['SQ_ACTIVE_INST_VALU']),
        'wait_any_fraction_of_wave_cycles': c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'],
        'modelled_issue_ms': cyc / SIMDS / (GHZ * 1e6)}
print(json.dumps(out, indent=1))
