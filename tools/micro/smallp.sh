# 8 ch x 988 taps, 1M-frame calls at ratios with SHORT periods (outputs per period 2, 2, 1, 2) and the headline's 160, with and
# without taking several periods at a time (fir_common.hip.h: artfir_period_multiple); kernel preference 0 auto, 6 f32 streaming, 7 fixed point
for pm in 0 1; do
  echo "ARTAMD_PERIOD_MULTIPLE=$pm"
  for r in "44100 88200" "48000 32000" "192000 48000" "48000 96000" "32000 48000" "44100 48000"; do
    for k in 0 6 7; do ARTAMD_PERIOD_MULTIPLE=$pm python tools/bench_shapes.py 8 988 988 $r 1 1 1048576 $k 2>&1 | grep -v amdgpu.ids; done
  done
  for r in "44100 88200" "48000 32000"; do
    ARTAMD_PERIOD_MULTIPLE=$pm python tools/bench_shapes.py 2 256 256 $r 1 1 262144 0 2>&1 | grep -v amdgpu.ids
  done
done
