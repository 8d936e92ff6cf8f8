# counters of the fixed-point main kernel in its two forms (ARTAMD_I8_WIDE=0: 32-slot tiles, staging waves; 1: 64-slot tiles)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3x/pmc_i8; mkdir -p $OUT
for w in 0 1; do
  ARTAMD_I8_WIDE=$w timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT -o w${w}_p1 -- python $R/tools/bench_shapes.py 8 988 988 44100 48000 0 1 1048576 0 > $OUT/w${w}_p1.log 2>&1
  ARTAMD_I8_WIDE=$w timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT -o w${w}_p2 -- python $R/tools/bench_shapes.py 8 988 988 44100 48000 0 1 1048576 0 > $OUT/w${w}_p2.log 2>&1
  ARTAMD_I8_WIDE=$w timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC --output-format csv -d $OUT -o w${w}_p3 -- python $R/tools/bench_shapes.py 8 988 988 44100 48000 0 1 1048576 0 > $OUT/w${w}_p3.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r3x/pmc_i8'
for f in sorted(glob.glob(out+'/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'fir_i8' not in k or 'stage' in k or 'standby' in k: continue
        a=agg[(k[:60], r['Counter_Name'])]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    print(os.path.basename(f))
    for (k,c),(n,v) in sorted(agg.items()): print(f'   {k:60s} {c:28s} per launch {v/n:14.0f}  ({n})')
PY
