# fir_i8_slab_kernel trims (round 5): the tree's library against _abl/libartamd_base.so (HEAD before them), alternating runs on one box, then the bit-identity tests
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5_trims}; mkdir -p $O
for rep in 1 2 3; do
for shape in "8 988 988 44100 48000 0 1 1048576" "8 988 988 44100 48000 0 1 524288" "8 988 988 44100 48000 0 1 262144" "4 988 988 44100 48000 0 1 1048576" "32 988 988 44100 48000 0 1 262144" "16 988 988 44100 48000 0 1 524288" "8 988 147 96000 44100 1 1 1048576" "8 512 512 44100 48000 0 1 1048576"; do
  for v in base tree; do
    L=$R/_abl/libartamd_$v.so; [ $v = tree ] && L=$R/audio_resampler_amd/libartamd.so
    ARTAMD_LIB=$L timeout 120 python $R/tools/bench_shapes.py $shape 7 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /"
  done
done
done > $O/ab.txt
cat $O/ab.txt
cd $R; timeout 1200 python -m pytest tests/test_gpu_slab_kernel.py tests/test_gpu_fixed_point.py tests/test_gpu_rows_cache.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
