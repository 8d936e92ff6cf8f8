# fir_i8_slab_kernel: the run behind the whole rounds as W + (L mod W) tiles cut in two at most (ARTAMD_I8_SLAB_TAIL=0, round 4) against
# L mod W tiles cut into W pieces (=1): alternating runs on one box; then the bit-identity sessions under the switch
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5_tail}; mkdir -p $O
for rep in 1 2 3; do
for shape in "8 988 988 44100 48000 0 1 1048576" "8 988 988 44100 48000 0 1 786432" "8 988 988 44100 48000 0 1 524288" "8 988 988 44100 48000 0 1 262144" "4 988 988 44100 48000 0 1 1048576" "32 988 988 44100 48000 0 1 262144" "32 988 988 44100 48000 0 1 1048576" "16 988 988 44100 48000 0 1 524288" "8 988 147 96000 44100 1 1 1048576" "8 512 512 44100 48000 0 1 1048576"; do
  for t in 0 1; do ARTAMD_I8_SLAB_TAIL=$t timeout 120 python $R/tools/bench_shapes.py $shape 7 2>&1 | grep -v amdgpu.ids | sed "s/^/tail $t: /"; done
done
done > $O/tail.txt
cat $O/tail.txt
cd $R; ARTAMD_I8_SLAB_TAIL=1 timeout 900 python -m pytest tests/test_gpu_slab_kernel.py tests/test_gpu_fixed_point.py tests/test_gpu_planar_device.py tests/test_gpu_channel_groups.py -x -q -m gpu 2>&1 | tail -15 > $O/tests_tail1.txt; cat $O/tests_tail1.txt
