# A/B of the round-5 backward changes on one box:  bash tools/prof/bwd_ab.sh > gpurun_out/r05_backward_ab.txt
#   old   = the library before them (commit f5b26ba's sources)
#   nowpe = launch constants from the host + v_sqrt for the per-face heights (prologue 185 -> 99 VALU instructions, 68 -> 64 VGPRs
#           = 7 -> 8 waves per SIMD), without the occupancy request for the six / nine-channel instantiations (-DLASR_BWD_WPE=0)
#   base  = shipped: + amdgpu_waves_per_eu(6 / 5) for six / nine channels (90 / 106 -> 78 / 88 VGPRs)
R=$GRAFT_REPO_ROOT; cd $R
echo "# three channels (bench.py step; kernel ms from library HIP events)"
for v in old nowpe base old base; do
  lib=$R/lasr_amd/csrc/liblasr_hip.so; [ $v != base ] && lib=$R/lasr_amd/csrc/variants/liblasr_hip_$v.so
  for args in "--frames 256" "--frames 64" "--frames 16"; do
    LASR_HIP_LIB=$lib python bench.py $args --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --no-step-profile --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['roofline']['all_kernels_avg_ms']
print('$v', '$args', 'forward %.4f' % k['sr_forward_kernel'], 'backward %.4f' % k['sr_backward_kernel'], 'step %.4f' % d['ms_per_step'])"
  done
done
echo "# nine channels: the optimisation step's render (rocprofv3 kernel trace of the replayed step, us per iteration)"
for v in old base nowpe base; do
  lib=$R/lasr_amd/csrc/liblasr_hip.so; [ $v != base ] && lib=$R/lasr_amd/csrc/variants/liblasr_hip_$v.so
  LASR_HIP_LIB=$lib python bench.py --frames 16 --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 2 --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for c in ('spot3_s0','camel_s4'):
    x=d['in_scope_step'][c]; r=x['raster']
    print('$v', c, 'forward %.2f' % r['sr_forward_kernel']['us'], 'backward %.2f' % r['sr_backward_kernel']['us'], 'raster %.1f' % x['raster_us'], 'wall %.0f' % x['wall_us'])"
done
