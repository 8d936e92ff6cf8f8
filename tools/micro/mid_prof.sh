# rocprofv3 per-kernel averages of mid-sized device-resident calls (8 ch x 988 taps 44.1k -> 48k): what stands beside the FIR kernel
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "65536 7" "65536 0" "32768 7" "32768 0" "16384 6" "131072 0"; do
  set -- $cfg
  rm -rf /tmp/sp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o s -- python $R/tools/bench_shapes.py 8 988 988 44100 48000 0 1 $1 $2 > /tmp/sp.log 2>&1
  grep "Msamples" /tmp/sp.log
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/sp/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r['Calls'])>40: print('     ', r['Name'][:90], r['Calls'], r['AverageNs'])
PY
done
