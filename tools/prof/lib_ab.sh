# per-kernel times of the bench step with the shipped library and every build under lasr_amd/csrc/variants/: bash tools/prof/lib_ab.sh [frames] [pattern]
R=$GRAFT_REPO_ROOT; N=${1:-256}; PAT=${2:-liblasr_hip_}
for rep in 1 2; do
for v in "" $(ls $R/lasr_amd/csrc/variants/ 2>/dev/null | grep "$PAT.*so$"); do
  if [ -n "$v" ]; then export LASR_HIP_LIB=$R/lasr_amd/csrc/variants/$v; else unset LASR_HIP_LIB; fi
  echo "${v:-shipped} $(python $R/tools/prof/pairs_check.py time-child $N 4 2>/dev/null | grep RESULT | python -c "import sys,json; d=json.loads(sys.stdin.read()[7:]); print(' '.join('%s %.4f'%(k.replace('sr_','').replace('_kernel',''),v) for k,v in d.items() if 'sr_' in k))")"
done; done
