# the f32 matrix kernels with the per-quad band flush (this tree) against the library in _ab/libartamd_old.so, same box, alternating
for sh in "8 988 988 44100 48000 0 1 1048576 6" "16 156 156 44100 48000 0 1 524288 0" "2 380 380 44100 48000 0 1 1048576 0" "8 988 988 44100 48000 0 1 65536 0" "4 988 988 44100 48000 0 1 1048576 6"; do
  for l in "" $PWD/_ab/libartamd_old.so; do echo -n "$([ -z "$l" ] && echo new || echo old): "; ARTAMD_LIB=$l python tools/bench_shapes.py $sh 2>&1 | grep -v amdgpu.ids; done
done
