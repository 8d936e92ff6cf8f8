#!/bin/bash
# Regenerate the measurement evidence of a round on the GPU box (via gpurun): bench line, rocprofv3 kernel statistics and PMC
# passes (separate runs; --pmc never together with --stats) for the headline kernel and for every other kernel the library ships.
# usage: bash tools/refresh_evidence.sh [tag]      -> gpurun_out/<tag>/...   then: python tools/roofline_report.py gpurun_out/<tag> profiles r3
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-evidence}
RND=${2:-r6}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 100 --warmup 10 > $OUT/bench100.json 2> $OUT/bench100.err
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
python $R/bench.py --steps 100 --warmup 10 --kernel 6 --no-cpu-baseline --no-other-configs > $OUT/bench_f32.json 2> $OUT/bench_f32.err
prof() {   # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $name -- "$@" > $OUT/$name.stats.log 2>&1
}
pmc() {    # name, pass, counters..., then -- command
  local name=$1 pass=$2; shift 2
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d $OUT/pmc -o ${name}_$pass -- "$@" > $OUT/${name}_$pass.pmc.log 2>&1
}
allpasses() {  # name, command...
  local name=$1; shift
  pmc $name p1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD -- "$@"
  pmc $name p3 FETCH_SIZE GRBM_GUI_ACTIVE -- "$@"
  pmc $name p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- "$@"
}
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs"
prof headline $BENCH
allpasses headline python $R/bench.py --steps 3 --warmup 1 --preroll-ms 60 --no-cpu-baseline --no-other-configs
pmc headline p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -- python $R/bench.py --steps 3 --warmup 1 --preroll-ms 60 --no-cpu-baseline --no-other-configs
prof headline_f32 $BENCH --kernel 6
allpasses headline_f32 python $R/bench.py --steps 3 --warmup 1 --preroll-ms 60 --no-cpu-baseline --no-other-configs --kernel 6
for c in fixed_D4 fixed_D32 general_E general_P matrix_P general_A matrix_B matrix_D4 matrix_D32 wide biquad biquad_serial decimate strict; do
  python $R/tools/profile_case.py $c > $OUT/case_$c.json 2> $OUT/case_$c.err
  prof $c python $R/tools/profile_case.py $c 12
  allpasses $c python $R/tools/profile_case.py $c 4
done
python $R/tools/bench_configs.py --steps 30 > $OUT/configs.jsonl 2> $OUT/configs.err
# what the counters report for a known byte count, per access width (the fp64 kernel's 8-byte loads)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o calib_fetch -- $R/tools/micro/fetch_calib > $OUT/fetch_calib.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc -o calib_write -- $R/tools/micro/fetch_calib >> $OUT/fetch_calib.log 2>&1
# the same bench line with the traffic measured in THIS lease (roofline_report.py writes the json from the passes above)
python $R/tools/roofline_report.py $OUT $OUT/report $RND > $OUT/report.log 2>&1
python $R/bench.py --pmc-json $OUT/report/${RND}_traffic.json > $OUT/bench_pmc.json 2> $OUT/bench_pmc.err
python $R/tools/bench_wide.py --block 1048576 --steps 20 > $OUT/wide.jsonl 2>/dev/null
bash $R/tools/bench_fixed_point.sh > $OUT/fixed_point_shapes.txt 2>&1
ARTAMD_HOST_TRACE=1 python $R/tools/bench_host_api.py > $OUT/host_api.txt 2>&1
python $R/tools/art_timing.py 60 600 > $OUT/art_timing.txt 2>&1
ls $OUT $OUT/prof $OUT/pmc | head -80
