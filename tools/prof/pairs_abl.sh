# forward-kernel time of the pair-walk kernel's measurement builds (make variant NAME=pwablN DEFS=-DLASR_PW_ABL=N): bash tools/prof/pairs_abl.sh [frames]
R=$GRAFT_REPO_ROOT; N=${1:-256}
export LASR_SR_PAIR_MIN_TILES=0
for v in "" $(ls $R/lasr_amd/csrc/variants/ 2>/dev/null | grep "liblasr_hip_pw.*so$"); do
  if [ -n "$v" ]; then export LASR_HIP_LIB=$R/lasr_amd/csrc/variants/$v; fi
  echo "${v:-shipped} $(python $R/tools/prof/pairs_check.py time-child $N 4 2>/dev/null | grep RESULT | python -c "import sys,json; d=json.loads(sys.stdin.read()[7:]); print(' '.join('%s %.4f'%(k.replace('sr_',''),v) for k,v in d.items() if 'forward' in k))")"
done
