# fir_general_kernel: outputs per workgroup tile (ARTAMD_GENERAL_TILE caps it; an output's bits do not depend on its tile) — round 6, profiles/r6_config_e.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r6_tile_sweep}; mkdir -p $O
for rep in 1 2 3; do
for shape in "2 380 380 0 65536" "2 380 380 0 262144" "2 380 380 0 16384" "2 380 380 0 4096" "2 380 380 1 65536" "1 988 988 1 65536" "8 988 988 1 65536" "8 988 988 1 4096" "8 380 380 1 65536" "8 380 380 1 8192" "4 256 256 0 65536" "2 48 48 1 65536" "2 156 156 1 16384" "32 380 380 1 16384"; do
  for t in 16 32 48 64; do ARTAMD_KERNEL=1 ARTAMD_GENERAL_TILE=$t timeout 120 python $R/tools/bench_asrc.py $shape 2>&1 | grep -v amdgpu.ids | sed "s/^/tile $t: /"; done
done
done > $O/sweep.txt
cat $O/sweep.txt
