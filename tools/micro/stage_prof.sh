# rocprofv3 averages of the fixed-point path's three kernels on the headline call
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o s -- python $R/tools/bench_shapes.py 8 988 988 44100 48000 0 1 1048576 0 > /tmp/sp.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('/tmp/sp/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'i8' in r['Name']: print('  ', r['Name'][:70], r['Calls'], r['AverageNs'])
PY
