# HBM counter calibration for the raster kernels' access patterns: bash tools/prof/traffic_cal.sh <tag>
#   -> gpurun_out/<tag>_traffic_calibration.json (copy it into profiles/; tools/traffic_json.py applies the factors)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=${1:-r04}
cd /tmp; export TMPDIR=/tmp
BIN=$R/scratch/bin/traffic
[ -x $BIN ] || hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o $BIN $R/tools/ubench/traffic.hip
$BIN > $O/${T}_traffic_true.json
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_cal_$c
  rocprofv3 --pmc $c -d $O/pmc_cal_$c -o p -- $BIN > /dev/null 2>&1
done
python $R/tools/traffic_cal_json.py $O/${T}_traffic_true.json $(find $O/pmc_cal_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_cal_WRITE_SIZE -name "*.db" | head -1) > $O/${T}_traffic_calibration.json
rm -rf $O/pmc_cal_FETCH_SIZE $O/pmc_cal_WRITE_SIZE
cat $O/${T}_traffic_calibration.json
