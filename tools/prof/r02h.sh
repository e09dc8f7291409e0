R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_raster_parity_gpu.py -m gpu -q 2>&1 | grep -E "^E  .*Error|passed|failed|^FAILED" | head -20 | cut -c1-300
for i in 1 2; do python bench.py --no-cpu-baseline --no-lbs --lasr-iters 0 --steps 40 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']), d['ms_per_step'], d['roofline']['all_kernels_avg_ms'])"; done
