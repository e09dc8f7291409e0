# where a small forward launch spends its time: bash tools/prof/small_launch_breakdown.sh  -> gpurun_out/r04_small_launch_breakdown.txt
# (full library, no-walk build LASR_ABL=2, no-store build LASR_ABL=4; forward kernel ms from the library's HIP events)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r04_small_launch_breakdown.txt; rm -f $O
for l in liblasr_hip.so variants/liblasr_hip_abl2.so variants/liblasr_hip_nostore.so; do
  LASR_HIP_LIB=$R/lasr_amd/csrc/$l python bench.py --no-cpu-baseline --no-lbs --lasr-iters 0 --steps 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read())
print('$l', {(s['frames'], s['image_size']): (s['kernel_ms']['sr_forward_kernel'], (s['segmented_opt_in'] or {}).get('forward_kernel_ms')) for s in d['sweep']})" >> $O
done
cat $O
