# fir_i8_slab_kernel, TIMING-ONLY ablations: the kernel with one ingredient compiled out (wrong samples, the schedule's cost of that ingredient).
# Builds libartamd variants into _abl/ (run HERE, on the build host: hipcc), then on the GPU box: bash tools/micro/slab_ablation.sh run
R=$(cd "$(dirname "$0")/../.." && pwd)
if [ "${1:-build}" = build ]; then
  mkdir -p $R/_abl
  for v in NO_DMA NO_READ NO_MFMA NO_XCHG; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I $R/include -I $R/audio_resampler_amd/csrc -DI8_ABL_$v -c $R/audio_resampler_amd/csrc/fir_matrix_i8.hip -o $R/_abl/i8_$v.o || exit 1
    objs=$(ls $R/audio_resampler_amd/_obj/*.o | grep -v fir_matrix_i8.hip.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/_abl/libartamd_$v.so $objs $R/_abl/i8_$v.o -lm -lpthread || exit 1
  done
  ls -la $R/_abl/*.so
else
  cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out/${2:-r5_ablation}; mkdir -p $O
  for rep in 1 2; do
  for shape in "8 988 988 44100 48000 0 1 1048576" "4 988 988 44100 48000 0 1 1048576" "32 988 988 44100 48000 0 1 262144"; do
    for v in "" NO_DMA NO_READ NO_MFMA NO_XCHG; do
      l=""; [ -n "$v" ] && l=$R/_abl/libartamd_$v.so
      ARTAMD_LIB=$l timeout 120 python $R/tools/bench_shapes.py $shape 7 2>&1 | grep -v amdgpu.ids | sed "s/^/${v:-shipped}: /"
    done
  done
  done > $O/ablation.txt
  cat $O/ablation.txt
fi
