# fir_i8_slab_kernel against the 32-slot kernel over call sizes (fixed-point forced: kernel preference 7): slabs wherever they can run | never
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r4_sizes}; mkdir -p $O
for shape in "8 988 988 44100 48000 0 1 131072" "8 988 988 44100 48000 0 1 262144" "8 988 988 44100 48000 0 1 393216" "8 988 988 44100 48000 0 1 524288" "8 988 988 44100 48000 0 1 786432" "8 988 988 44100 48000 0 1 1048576" "4 988 988 44100 48000 0 1 1048576" "4 988 988 44100 48000 0 1 524288" "32 988 988 44100 48000 0 1 262144" "32 988 988 44100 48000 0 1 131072" "16 988 988 44100 48000 0 1 262144" "8 988 147 96000 44100 1 1 1048576" "8 512 512 44100 48000 0 1 1048576"; do
  for mn in 1 100000; do ARTAMD_I8_SLAB_MIN=$mn timeout 120 python $R/tools/bench_shapes.py $shape 7 2>&1 | grep -v amdgpu.ids | sed "s/^/slab_min $mn: /"; done
done > $O/sizes.txt
cat $O/sizes.txt
