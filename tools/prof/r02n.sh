R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_raster_parity_gpu.py -m gpu -q 2>&1 | grep -E "^E  .*Error|passed|failed|^FAILED" | head -10 | cut -c1-300
for v in 0 1 0 1; do LASR_FWD_DEFER=$v python bench.py --no-cpu-baseline --no-lbs --lasr-iters 0 --steps 40 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('defer=$v', round(d['value']), d['ms_per_step'], d['roofline']['all_kernels_avg_ms'], round(d['relaxed_forward_math']['value']))"; done
