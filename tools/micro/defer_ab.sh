# the slab kernel's deferred arrival (default) against arrival on the spot, same box, alternating
for i in 1 2 3; do
  for z in 1 0; do echo -n "defer $z: "; ARTAMD_I8_SLAB_DEFER=$z python tools/bench_shapes.py 8 988 988 44100 48000 0 1 1048576 0 2>&1 | grep -v amdgpu.ids; done
done
for sh in "4 988 988 44100 48000 0 1 1048576" "32 988 988 44100 48000 0 1 262144" "8 988 988 44100 48000 0 1 524288" "8 988 988 96000 44100 1 1 1048576" "8 988 988 44100 88200 0 1 524288"; do
  for z in 1 0; do echo -n "defer $z: "; ARTAMD_I8_SLAB_DEFER=$z python tools/bench_shapes.py $sh 0 2>&1 | grep -v amdgpu.ids; done
done
