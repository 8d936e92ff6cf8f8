# the general kernel's all-at-once coefficient prefetch (LEAN >= 2, round 6) against round 5's lean loop (ARTAMD_GENERAL_LEAN=1) and the plain loop (=0)
# on config E's call shape and its neighbours: alternating runs on one box
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r6_lean2}; mkdir -p $O
for rep in 1 2 3; do
for shape in "2 380 380 0 65536" "2 380 380 0 262144" "2 380 380 0 16384" "2 380 380 0 4096" "2 380 380 1 65536" "1 988 988 1 65536" "1 380 380 0 65536" "2 156 156 1 65536" "2 988 988 1 65536" "2 48 48 1 65536" "1 48 48 1 4096"; do
  for mode in 2 1; do ARTAMD_GENERAL_LEAN=$mode timeout 120 python $R/tools/bench_asrc.py $shape 2>&1 | grep -v amdgpu.ids | sed "s/^/lean $mode: /"; done
done
done > $O/ab.txt
cat $O/ab.txt
cd $R; timeout 900 python -m pytest tests/test_gpu_asrc.py tests/test_gpu_general_pipe.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
