# the general kernel's lean tap loop against the plain one (ARTAMD_GENERAL_LEAN=0) on config E's call shape and its neighbours: alternating runs on one box
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5_cells}; mkdir -p $O
for rep in 1 2; do
for shape in "2 380 380 0 65536" "2 380 380 0 262144" "2 380 380 0 16384" "2 380 380 1 65536" "1 988 988 1 65536" "1 380 380 0 65536" "2 156 156 1 65536" "2 988 988 1 65536" "8 380 380 1 65536" "4 256 256 0 65536"; do
  for off in 1 0; do ARTAMD_GENERAL_LEAN=$off timeout 120 python $R/tools/bench_asrc.py $shape 2>&1 | grep -v amdgpu.ids | sed "s/^/lean $off: /"; done
done
done > $O/cells.txt
cat $O/cells.txt
