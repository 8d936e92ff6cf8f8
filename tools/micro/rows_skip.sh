# TIMING EXPERIMENT: what a call costs without the rows' work in front of it (ARTAMD_EXPERIMENT_SKIP_ROWS=1: wrong samples, right schedule)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5_rows_skip}; mkdir -p $O
for shape in "8 988 988 44100 48000 0 1 32768" "8 988 988 44100 48000 0 1 65536" "8 988 988 44100 48000 0 1 131072" "8 988 988 44100 48000 0 1 262144" "8 988 988 44100 48000 0 1 1048576" "2 380 380 44100 48000 0 1 1048576" "2 380 380 44100 48000 0 1 65536"; do
  for pref in 0 6 7; do for sk in 0 1; do ARTAMD_EXPERIMENT_SKIP_ROWS=$sk timeout 120 python $R/tools/bench_shapes.py $shape $pref 2>&1 | grep -v amdgpu.ids | sed "s/^/skip $sk: /"; done; done
done > $O/skip.txt
cat $O/skip.txt
