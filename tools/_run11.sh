cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
rm -rf /tmp/prof_l$m
AVSI_LOSS_FROM_WAV=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l$m -- python3 $GRAFT_REPO_ROOT/bench.py --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-also > /dev/null 2>&1
echo "AVSI_LOSS_FROM_WAV=$m"
python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob('/tmp/prof_l$m/*/*kernel_stats.csv')[0])):
    if any(k in r['Name'] for k in ('frontend_kernel','l1_partial','l1_final')): print(r['Name'][28:90], r['Calls'], float(r['AverageNs'])/1e3)
PY
done
