cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5_trims2}; mkdir -p $O
for rep in 1 2 3; do
for shape in "8 988 988 44100 48000 0 1 1048576" "4 988 988 44100 48000 0 1 1048576" "16 988 988 44100 48000 0 1 524288"; do
  for v in base t_none t_next t_reads2 t_both2; do
    ARTAMD_LIB=$R/_abl/libartamd_$v.so timeout 120 python $R/tools/bench_shapes.py $shape 7 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /"
  done
done
done > $O/ab.txt
cat $O/ab.txt
