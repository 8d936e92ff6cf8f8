# calls of many ring epochs (short filters, 1M frames): library's choice, and the general kernel
for sh in "1 48 48 44100 48000 0 1 1048576" "2 48 48 44100 48000 0 1 1048576" "8 156 156 44100 48000 0 1 1048576" "16 156 156 44100 48000 0 1 524288" "2 256 256 44100 48000 0 1 1048576" "8 48 48 48000 96000 0 1 1048576"; do
  for k in 0 1; do python tools/bench_shapes.py $sh $k 2>&1 | grep -v amdgpu.ids; done
done
