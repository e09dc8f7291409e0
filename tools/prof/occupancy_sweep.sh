# is the headline forward launch issue-bound or latency-bound?  The one-wave kernel (42 VGPR, 4 KB LDS: 8
# waves per SIMD) rebuilt with extra LDS per workgroup so that only 6 / 4 / 2 waves fit a SIMD, same sources otherwise
# (make -C lasr_amd/csrc variant NAME=occN DEFS=-DLASR_OCC_LDS=...; built in the container, the .so files travel with the snapshot).
#   bash tools/prof/occupancy_sweep.sh > gpurun_out/r05_occupancy_sweep.txt        (one MI355X)
R=$GRAFT_REPO_ROOT; cd $R
echo "# forward kernel ms per 256-frame launch (library HIP events, bench.py --steps 20), waves per SIMD capped by LDS per workgroup"
for v in base occ6 occ4 occ2; do
  lib=$R/lasr_amd/csrc/liblasr_hip.so; [ $v != base ] && lib=$R/lasr_amd/csrc/variants/liblasr_hip_$v.so
  LASR_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['roofline']['all_kernels_avg_ms']
print('$v', 'forward %.4f' % k['sr_forward_kernel'], 'backward %.4f' % k['sr_backward_kernel'], 'step %.4f' % d['ms_per_step'])"
done
for n in 16 64; do
  for v in base occ6 occ4 occ2; do
    lib=$R/lasr_amd/csrc/liblasr_hip.so; [ $v != base ] && lib=$R/lasr_amd/csrc/variants/liblasr_hip_$v.so
    LASR_HIP_LIB=$lib LASR_SR_COOP_MAX_TILES=0 LASR_SR_COOP8_MAX_TILES=0 LASR_SR_CHOOSE_MAX_TILES=0 python bench.py --frames $n --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --steps 40 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['roofline']['all_kernels_avg_ms']
print('$n frames (one-wave kernel forced)', '$v', 'forward %.4f' % k['sr_forward_kernel'])"
  done
done
