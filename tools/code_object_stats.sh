#!/bin/bash
# Static facts of the gfx950 code every translation unit compiles to (no GPU needed):  bash tools/code_object_stats.sh > profiles/r05_code_objects.txt
# scratch / spill counts must be zero in the hot objects; MFMA instructions live in ops.hip (LBS forward and backward); global atomics
# only on the cold surface-texel path of sr_raster.hip and in the brute-force fp64 unit.
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
echo "# hipcc --offload-arch=gfx950 -S of lasr_amd/csrc/*.hip with the Makefile's flags; per translation unit"
printf "%-20s %8s %8s %8s %8s %8s %10s %10s\n" unit kernels scratch spills mfma atomics max_vgpr max_sgpr
for f in common sr_raster sr_backward_fast sr_fp64 ops fused glue mesh_reg post_raster tail; do
  FL="-ffp-contract=off"; [ $f = sr_backward_fast ] && FL="-ffp-contract=fast-honor-pragmas"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FL -munsafe-fp-atomics -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-memory-clause \
      -S --cuda-device-only -o $T/$f.s $R/lasr_amd/csrc/$f.hip 2>/dev/null
  printf "%-20s %8d %8d %8d %8d %8d %10d %10d\n" $f.hip $(grep -c '\.amdhsa_kernel ' $T/$f.s) $(grep -c 'scratch_\(load\|store\)' $T/$f.s) \
      $(grep -c 'vgpr_spill_count: *[1-9]' $T/$f.s) $(grep -c 'v_mfma' $T/$f.s) $(grep -c 'global_atomic' $T/$f.s) \
      $(grep '\.vgpr_count:' $T/$f.s | awk '{print $2}' | sort -n | tail -1) $(grep '\.sgpr_count:' $T/$f.s | awk '{print $2}' | sort -n | tail -1)
done
echo
echo "# raster kernels of the headline step (.vgpr_count / .sgpr_count / LDS bytes of the code object)"
for k in 'sr_forward_kernelILb1ELi3ELb0ELb1E' 'sr_forward_kernelILb1ELi9ELb0ELb1E' 'sr_forward_coop_kernelILi3ELi4E' 'sr_setup_kernel'; do
  python3 - "$T/sr_raster.s" "$k" <<'P'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r'\.group_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', t):
    if sys.argv[2] in m.group(2):
        print('%-70s vgpr %3s sgpr %3s lds %6s' % (m.group(2)[:70], m.group(4), m.group(3), m.group(1)))
        break
P
done
for k in 'sr_backward_kernelILb1ELi3E' 'sr_backward_kernelILb1ELi9E'; do
  python3 - "$T/sr_backward_fast.s" "$k" <<'P'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r'\.group_segment_fixed_size:\s+(\d+)(?:.*\n)*?\s+\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', t):
    if sys.argv[2] in m.group(2):
        print('%-70s vgpr %3s sgpr %3s lds %6s' % (m.group(2)[:70], m.group(4), m.group(3), m.group(1)))
        break
P
done
rm -rf $T
