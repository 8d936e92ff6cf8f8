# f32 K-split kernel: parts per tile (ARTAMD_SPLIT_FORCE_KS, A/B only) against call size; host_rate = enqueue + drain per call without events
cd /tmp; R=$GRAFT_REPO_ROOT
for shape in "8 988" "2 380" "32 988" "8 256"; do
for b in 4096 8192 12288 16384 24576 32768 40960 49152 65536 81920; do
  line="ch/taps $shape block $b:"
  for ks in 1 2 3 4 6 8; do
    t=$(ARTAMD_SPLIT_FORCE_KS=$ks timeout 100 python $R/tools/micro/host_rate.py $shape $b 2>&1 | tail -1 | sed -n 's/.*enqueue + drain \([0-9.]*\) us.*kernel \([0-9]*\).*/\1(k\2)/p')
    line="$line  ks$ks $t"
  done
  echo "$line"
done
done
