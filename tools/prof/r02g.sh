R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_raster_parity_gpu.py -m gpu -x -q 2>&1 | grep -E "^E  |passed|failed|Error" | head -20 | cut -c1-300
for v in 0 2; do timeout 300 python bench.py --forward-variant $v --no-cpu-baseline --no-lbs --lasr-iters 0 --steps 30 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('variant $v', round(d['value']), d['ms_per_step'], d['roofline']['all_kernels_avg_ms'], d['relaxed_forward_math']['value'])"; done
