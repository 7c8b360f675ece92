set -e
ROOT=$GRAFT_REPO_ROOT
cp $ROOT/noize_job_amd/libnoize_hip.so /tmp/base.so
cd /tmp && export TMPDIR=/tmp
for v in base mr4 mr8; do
  if [ $v = base ]; then cp /tmp/base.so $ROOT/noize_job_amd/libnoize_hip.so; else cp $ROOT/tools/_exp/libnoize_hip_$v.so $ROOT/noize_job_amd/libnoize_hip.so; fi
  echo "== $v"
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o run -- python3 $ROOT/tools/bench_next.py --only "map range" > /tmp/prof_$v.log 2>&1 || tail -5 /tmp/prof_$v.log
  python3 - /tmp/prof_$v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'map_range' in r.get('Name', ''):
            print('   ', r['Name'][:60], 'calls', r['Calls'], 'avg ns', r['AverageNs'], 'min', r['MinNs'])
PY
done
cp /tmp/base.so $ROOT/noize_job_amd/libnoize_hip.so
