R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E  .*Error|passed|failed|^FAILED|^ERROR" | head -20 | cut -c1-300
python tools/cosdist_bench.py > $O/r02_cosdist.json 2>/dev/null; cat $O/r02_cosdist.json | cut -c1-1500
