# the two staging passes of the fixed-point path: units (4 frames x 1 channel) per staging thread — bytes in flight per workgroup against workgroups per launch.
# build (here): bash tools/micro/stage_units_ab.sh ; run (GPU box): bash tools/micro/stage_units_ab.sh run [outdir]
R=$(cd "$(dirname "$0")/../.." && pwd)
if [ "${1:-build}" = build ]; then
  mkdir -p $R/_abl
  for k in 2 6 8 12 16; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I $R/include -I $R/audio_resampler_amd/csrc -DI8_STAGE_UNITS=$k -c $R/audio_resampler_amd/csrc/fir_matrix_i8.hip -o $R/_abl/i8_K$k.o || exit 1
    objs=$(ls $R/audio_resampler_amd/_obj/*.o | grep -v fir_matrix_i8.hip.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/_abl/libartamd_K$k.so $objs $R/_abl/i8_K$k.o -lm -lpthread || exit 1
  done
  ls -la $R/_abl/*K*.so
else
  cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out/${2:-r6_stage_units}; mkdir -p $O
  for rep in 1 2 3; do
    for k in 4 2 6 8 12 16; do
      l=""; [ "$k" != 4 ] && l=$R/_abl/libartamd_K$k.so
      ARTAMD_LIB=$l timeout 300 python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('units $k: value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel', r['avg_kernel_ms'], 'prep', r['avg_prep_ms'])"
    done
  done > $O/ab.txt
  cat $O/ab.txt
fi
