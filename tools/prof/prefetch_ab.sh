# A/B of the forward walk's scalar-cache touch of an entry's other record lines (LASR_PREFETCH, sr_raster.hip): the shipped library
# against the same sources built with -DLASR_PREFETCH=0 (make -C lasr_amd/csrc variant NAME=nopf DEFS=-DLASR_PREFETCH=0).
#   bash tools/prof/prefetch_ab.sh > gpurun_out/r05_prefetch_ab.txt
R=$GRAFT_REPO_ROOT; cd $R
echo "# kernel ms (library HIP events), bench.py; A = -DLASR_PREFETCH=0, B = shipped (touch lines 1, 2 and the attribute line with the rect load)"
for rep in 1 2; do
for v in nopf base; do
  lib=$R/lasr_amd/csrc/liblasr_hip.so; [ $v != base ] && lib=$R/lasr_amd/csrc/variants/liblasr_hip_$v.so
  for args in "--frames 256" "--frames 64" "--frames 16" "--frames 4" "--frames 64 --image-size 512"; do
    LASR_HIP_LIB=$lib python bench.py $args --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['roofline']['all_kernels_avg_ms']
print('$v', '$args', 'forward %.4f' % k['sr_forward_kernel'], 'backward %.4f' % k['sr_backward_kernel'], 'step %.4f' % d['ms_per_step'])"
  done
done
done
