# forward-kernel time of the shipped library and of every measurement build under lasr_amd/csrc/variants/liblasr_hip_pw*.so
# (make variant NAME=pw... DEFS=...), interleaved REPS times; prints each library's sorted times:  bash tools/prof/pairs_abl.sh [frames] [reps]
R=$GRAFT_REPO_ROOT; N=${1:-256}; REPS=${2:-3}
export LASR_SR_PAIR_MIN_TILES=0
T=$(mktemp)
for r in $(seq $REPS); do
  for v in "" $(ls $R/lasr_amd/csrc/variants/ 2>/dev/null | grep "liblasr_hip_pw.*so$"); do
    if [ -n "$v" ]; then export LASR_HIP_LIB=$R/lasr_amd/csrc/variants/$v; else unset LASR_HIP_LIB; fi
    echo "${v:-shipped} $(python $R/tools/prof/pairs_check.py time-child $N 6 2>/dev/null | grep RESULT | python -c "import sys,json; d=json.loads(sys.stdin.read()[7:]); print('%.4f %.4f' % (d['sr_forward_kernel'], d['sr_backward_kernel']))")" >> $T
  done
done
python - $T <<'P'
import sys, collections
f = collections.defaultdict(list); b = collections.defaultdict(list)
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) == 3: f[p[0]].append(float(p[1])); b[p[0]].append(float(p[2]))
for k in f: print('%-28s forward %s   backward %s' % (k, ' '.join('%.4f' % x for x in sorted(f[k])), ' '.join('%.4f' % x for x in sorted(b[k]))))
P
rm -f $T
