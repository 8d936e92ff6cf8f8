# second round of the private-X experiment: step3 (X0's wait moved behind the barrier too), staging alone under both lane maps, phase traces
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5_private_x2}; mkdir -p $O
for rep in 1 2 3; do
for shape in "8 988 988 44100 48000 0 1 1048576" "16 988 988 44100 48000 0 1 524288" "4 988 988 44100 48000 0 1 1048576" "32 988 988 44100 48000 0 1 262144"; do
  for v in base step2 step3 NO_MFMA+NO_READ step2_stageonly; do
    L=$R/_abl/libartamd_$v.so; [ $v = step2 ] && L=$R/audio_resampler_amd/libartamd.so
    ARTAMD_LIB=$L timeout 120 python $R/tools/bench_shapes.py $shape 7 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /"
  done
done
done > $O/ab.txt
cat $O/ab.txt
for v in base_trace step2_trace step3_trace; do
  echo "== $v"; ARTAMD_LIB=$R/_abl/libartamd_$v.so timeout 120 python $R/tools/bench_shapes.py 8 988 988 44100 48000 0 1 1048576 7 2>&1 | grep -v amdgpu.ids
done > $O/trace.txt 2>&1
cat $O/trace.txt
cd $R; ARTAMD_LIB=$R/_abl/libartamd_step3.so timeout 600 python -m pytest tests/test_gpu_slab_kernel.py tests/test_gpu_fixed_point.py -x -q -m gpu 2>&1 | tail -5 > $O/tests_step3.txt; cat $O/tests_step3.txt
