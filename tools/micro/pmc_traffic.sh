# HBM-side traffic (fabric requests of the L2s) of the fixed-point main kernel on a shape, per launch: FETCH_SIZE x 2 (gfx950 wide-read
# correction) + WRITE_SIZE, L2 hits / misses; separate --pmc passes.  usage: bash tools/micro/pmc_traffic.sh <tag> CH TAPS FILTERS SRC DST FIXED INTERP BLOCK [KERNEL]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p3 -- python $R/tools/bench_shapes.py "$@" > $OUT/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o p4 -- python $R/tools/bench_shapes.py "$@" > $OUT/p4.log 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: [0,0.0])
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'fir_i8' not in k or 'stage' in k or 'standby' in k: continue
        a=agg[(k.split('(')[0][-40:], r['Counter_Name'])]; a[0]+=1; a[1]+=float(r['Counter_Value'])
v={}
for (k,c),(n,x) in sorted(agg.items()): print(f'   {k:42s} {c:18s} per launch {x/n:14.1f}  ({n} launches)'); v[c]=x/n
if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v: print(f'   traffic per launch: {(v["FETCH_SIZE"]*2+v["WRITE_SIZE"])*1024/1e6:.1f} MB')
PY
