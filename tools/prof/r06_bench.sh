# the bench lines of the round, on a box that has not been profiled: bash tools/prof/r06_bench.sh <tag>   (one MI355X)
# (run tools/prof/r06_final.sh first and copy its counter files into profiles/: the line quotes traffic / valu_frac from them)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=${1:-r06}
cd $R
python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
python bench.py --image-size 512 --frames 64 --no-lbs --no-sweep --lasr-iters 0 > $O/${T}_bench_512.json 2>/dev/null
(cd /tmp; export TMPDIR=/tmp
 rocprofv3 --kernel-trace -d $O/prof_k -o k -- python $R/bench.py --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --steps 30 --warmup 3 > /dev/null 2>&1
 python $R/tools/rocpd_stats.py $(find $O/prof_k -name "*.db" | head -1) > $O/${T}_kernel_stats.txt; rm -rf $O/prof_k)
head -6 $O/${T}_kernel_stats.txt | cut -c1-140
python -c "
import json;d=json.load(open('$O/${T}_bench.json'))
r=d['roofline']; print(d['value'], d['ms_per_step'], r['all_kernels_avg_ms'], r['frac'], r['traffic'], r['traffic_stale'], r['valu_frac']); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['one_thread_frames_per_s']); print(d.get('optimize_py',{}).get('iters_per_s'))
for s in d['sweep']: print(s['frames'], s['image_size'], round(s['frames_per_s']), s['kernel_ms'])
for c in ('spot3_s0','camel_s4'):
    x=d['in_scope_step'][c]; print(c, x.get('raster_us'), x.get('tail_us'), x.get('other_in_scope_us'), x.get('other_in_scope_launches'), x.get('wall_us'))
print({k:(v['us_per_call'], v['backward_us_per_call'], v['trace_us']) for k,v in d['lbs']['sizes'].items()})
d=json.load(open('$O/${T}_bench_512.json')); print('512:', d['value'], d['roofline']['all_kernels_avg_ms'], d['roofline']['frac'])"
