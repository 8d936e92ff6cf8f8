# f32 streaming kernel on BIG launches whose tile count does not fill the chip's rounds (config B: stereo x 380 taps, 1M frames = 70 tiles per XCD on 94 slots):
# the K-split kernel forced to 2 / 3 / 4 parts (ARTAMD_SPLIT_FORCE_KS) against the un-split launch — round 6, profiles/r6_f32_split.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r6_f32_split}; mkdir -p $O
for rep in 1 2 3; do
for shape in "2 380 380 44100 48000 0 1 1048576" "2 380 380 44100 48000 0 1 524288" "2 380 380 44100 48000 0 1 262144" "4 380 380 44100 48000 0 1 1048576" "8 380 380 44100 48000 0 1 524288" "1 380 380 44100 48000 0 1 1048576" "2 988 988 44100 48000 0 1 1048576" "2 156 156 44100 48000 0 1 1048576" "8 988 988 44100 48000 0 1 1048576" "32 380 380 44100 48000 0 1 262144"; do
  for ks in 0 2 3 4; do ARTAMD_SPLIT_FORCE_KS=$ks ARTAMD_NO_FIXED=1 timeout 120 python $R/tools/bench_shapes.py $shape 2>&1 | grep -v amdgpu.ids | sed "s/^/ks $ks: /"; done
done
done > $O/ab.txt
cat $O/ab.txt
