# Where the backward kernel's time goes: measurement builds of sr_backward.h (make -C lasr_amd/csrc variant NAME=bablN DEFS=-DLASR_BWD_ABL=N)
#   babl1 = prologue + stage 1 (reject, ring) + epilogue, stage 2 skipped      babl2 = + stage 2's distance code, nothing after it
#   babl3 = everything but the pixel-plane loads (constants instead)           base  = shipped
#   bash tools/prof/backward_breakdown.sh > gpurun_out/r05_backward_breakdown.txt
R=$GRAFT_REPO_ROOT; cd $R
echo "# backward kernel ms (library HIP events), bench.py --steps 20"
for v in base babl1 babl2 babl3 base; do
  lib=$R/lasr_amd/csrc/liblasr_hip.so; [ $v != base ] && lib=$R/lasr_amd/csrc/variants/liblasr_hip_$v.so
  for args in "--frames 256" "--frames 16"; do
    LASR_HIP_LIB=$lib python bench.py $args --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --no-step-profile --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['roofline']['all_kernels_avg_ms']
print('$v', '$args', 'backward %.4f' % k['sr_backward_kernel'])"
  done
done
