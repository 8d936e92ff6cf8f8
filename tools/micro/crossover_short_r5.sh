# general kernel against the matrix path for the shorter filters (round 5; the rule of artfir_takes_matrix_path below 704 taps was fitted in round 4)
cd /tmp; R=$GRAFT_REPO_ROOT
for shape in "8 380" "8 512" "2 380" "32 380" "8 256" "4 512"; do
for b in 8192 12288 16384 24576 32768 49152 65536; do
  line="ch/taps $shape block $b:"
  for pref in 0 1 2; do
    t=$(timeout 100 python $R/tools/micro/host_rate.py $shape $b $pref 2>&1 | tail -1 | sed -n 's/.*enqueue + drain \([0-9.]*\) us.*kernel \([0-9]*\).*/\1(k\2)/p')
    line="$line  $([ $pref = 0 ] && echo auto || ([ $pref = 1 ] && echo general || echo matrix)) $t"
  done
  echo "$line"
done
done
