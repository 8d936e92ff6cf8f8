mkdir -p gpurun_out/r4_slab7; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_slab7
for blk in 131072 262144 524288; do for mn in 1 100000; do ARTAMD_I8_SLAB_MIN=$mn timeout 120 python $R/tools/bench_shapes.py 8 988 988 44100 48000 0 1 $blk 7; done; done 2>&1 | grep -v amdgpu.ids > $O/sizes.txt
for mn in 1 100000; do ARTAMD_I8_SLAB_MIN=$mn timeout 120 python $R/tools/bench_shapes.py 32 988 988 44100 48000 0 1 262144 7;  ARTAMD_I8_SLAB_MIN=$mn timeout 120 python $R/tools/bench_shapes.py 4 988 988 44100 48000 0 1 262144 7; ARTAMD_I8_SLAB_MIN=$mn timeout 120 python $R/tools/bench_shapes.py 8 988 147 96000 44100 1 1 1048576 7; done 2>&1 | grep -v amdgpu.ids >> $O/sizes.txt
cat $O/sizes.txt
cd $R; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
