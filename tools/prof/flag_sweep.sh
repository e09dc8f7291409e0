# compiler-flag variants of the raster translation units (make variant NAME=x COMMON="..."), benched against the shipped build;
# then the backward with the forward's records reused / rebuilt:  bash tools/prof/flag_sweep.sh
R=$GRAFT_REPO_ROOT; cd $R
bash tools/prof/ab_many.sh lasr_amd/csrc/liblasr_hip.so $(ls lasr_amd/csrc/variants/*.so)
for n in 256 64; do for rb in 0 1; do python bench.py --no-cpu-baseline --no-lbs --no-sweep --lasr-iters 0 --steps 40 --frames $n --rebuild-records $rb 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('frames $n rebuild $rb', round(d['value']), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['all_kernels_avg_ms'].items()})"; done; done
