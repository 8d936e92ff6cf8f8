# mid-sized device-resident calls, 8 ch x 988 taps 44.1k -> 48k: library's choice (K-split below ~768 / parts tiles) against the
# unsplit streaming kernel (6); and other shapes
for b in 4096 8192 16384 32768 65536 131072 262144; do
  for k in 0 6; do python tools/bench_shapes.py 8 988 988 44100 48000 0 1 $b $k 2>&1 | grep -v amdgpu.ids; done
done
for sh in "2 380 380 44100 48000 0 1 65536" "2 380 380 44100 48000 0 1 16384" "32 988 988 44100 48000 0 1 16384" "8 988 988 96000 44100 1 1 65536"; do
  for k in 0 6; do python tools/bench_shapes.py $sh $k 2>&1 | grep -v amdgpu.ids; done
done
