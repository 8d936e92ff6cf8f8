# fixed-point path (pref 7) against the f32 streaming path (pref 6) and the library's choice (0) around the rule of artfir_planes_bytes (round 5)
cd /tmp; R=$GRAFT_REPO_ROOT
for shape in "8 988" "4 988" "16 988" "32 988" "8 768" "8 512"; do
for b in 49152 65536 81920 98304 131072 196608 262144; do
  line="ch/taps $shape block $b:"
  for pref in 0 6 7; do
    t=$(timeout 100 python $R/tools/micro/host_rate.py $shape $b $pref 2>&1 | tail -1 | sed -n 's/.*enqueue + drain \([0-9.]*\) us.*/\1/p')
    line="$line  $([ $pref = 0 ] && echo auto || ([ $pref = 6 ] && echo f32 || echo fixed)) $t"
  done
  echo "$line"
done
done
