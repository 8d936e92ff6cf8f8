# the re-fitted split rule (library's own choice) against one part (ARTAMD_SPLIT_FORCE_KS=1) and the other counts, more shapes
cd /tmp; R=$GRAFT_REPO_ROOT
for shape in "8 988" "4 988" "16 988" "8 380" "8 512" "32 512" "2 988" "1 988"; do
for b in 8192 16384 24576 32768 49152 65536 98304 131072; do
  line="ch/taps $shape block $b:"
  for ks in 0 1 2 3 4; do
    t=$(ARTAMD_SPLIT_FORCE_KS=$ks timeout 100 python $R/tools/micro/host_rate.py $shape $b 2>&1 | tail -1 | sed -n 's/.*enqueue + drain \([0-9.]*\) us.*kernel \([0-9]*\).*/\1(k\2)/p')
    line="$line  $([ $ks = 0 ] && echo rule || echo ks$ks) $t"
  done
  echo "$line"
done
done
