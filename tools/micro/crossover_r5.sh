# general kernel (pref 1) against the matrix path (pref 2) around the dispatch rule's crossover, after the K-split re-fit (round 5); "auto" = the library's choice
cd /tmp; R=$GRAFT_REPO_ROOT
for shape in "1 988" "2 988" "4 988" "8 988" "16 988" "32 988" "8 768" "2 768"; do
for b in 2048 3072 4096 6144 8192 12288 16384 24576 32768 49152; do
  line="ch/taps $shape block $b:"
  for pref in 0 1 2; do
    t=$(timeout 100 python $R/tools/micro/host_rate.py $shape $b $pref 2>&1 | tail -1 | sed -n 's/.*enqueue + drain \([0-9.]*\) us.*kernel \([0-9]*\).*/\1(k\2)/p')
    line="$line  $([ $pref = 0 ] && echo auto || ([ $pref = 1 ] && echo general || echo matrix)) $t"
  done
  echo "$line"
done
done
