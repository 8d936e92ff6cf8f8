# fir_i8_slab_kernel, round 6: TIMING-ONLY ablation of an X-resident tile — see profiles/r6_slab_ablation.txt.
# build (here): bash tools/micro/slab_xres_ablation.sh ; run (GPU box): bash tools/micro/slab_xres_ablation.sh run [outdir]
R=$(cd "$(dirname "$0")/../.." && pwd)
VARIANTS="XRES1:-DI8_ABL_XRES=1 XRES1_NOALIGN:-DI8_ABL_XRES=1,-DI8_ABL_XRES_NOALIGN XRES2:-DI8_ABL_XRES=2 NO_DMA:-DI8_ABL_NO_DMA XRES1_STAGEONLY:-DI8_ABL_XRES=1,-DI8_ABL_NO_MFMA,-DI8_ABL_NO_READ STAGEONLY:-DI8_ABL_NO_MFMA,-DI8_ABL_NO_READ"
if [ "${1:-build}" = build ]; then
  mkdir -p $R/_abl
  for vv in $VARIANTS; do
    v=${vv%%:*}; d=$(echo ${vv#*:} | tr ',' ' ')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I $R/include -I $R/audio_resampler_amd/csrc $d -c $R/audio_resampler_amd/csrc/fir_matrix_i8.hip -o $R/_abl/i8_$v.o || exit 1
    objs=$(ls $R/audio_resampler_amd/_obj/*.o | grep -v fir_matrix_i8.hip.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/_abl/libartamd_$v.so $objs $R/_abl/i8_$v.o -lm -lpthread || exit 1
  done
  ls -la $R/_abl/*.so
else
  cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out/${2:-r6_xres_ablation}; mkdir -p $O
  for rep in 1 2 3; do
  for shape in "8 988 988 44100 48000 0 1 1048576" "4 988 988 44100 48000 0 1 1048576" "32 988 988 44100 48000 0 1 262144"; do
    for vv in shipped: $VARIANTS; do
      v=${vv%%:*}; l=""; [ "$v" != shipped ] && l=$R/_abl/libartamd_$v.so
      ARTAMD_LIB=$l timeout 120 python $R/tools/bench_shapes.py $shape 7 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /"
    done
  done
  done > $O/ablation.txt
  cat $O/ablation.txt
fi
