# general kernel, lean loop with the first round's coefficient loads issued before the staging (ARTAMD_GENERAL_LEAN=2) against the shipped lean loop (=1): alternating runs on one box
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r6_lean_first}; mkdir -p $O
for rep in 1 2 3; do
for shape in "2 380 380 0 65536" "2 380 380 0 262144" "2 380 380 0 16384" "2 380 380 1 65536" "1 380 380 0 65536" "2 156 156 1 65536" "2 988 988 1 65536" "2 48 48 1 65536"; do
  for mode in 2 1; do ARTAMD_GENERAL_LEAN=$mode timeout 120 python $R/tools/bench_asrc.py $shape 2>&1 | grep -v amdgpu.ids | sed "s/^/lean $mode: /"; done
done
done > $O/ab.txt
python - <<PY
import re,statistics,collections
d=collections.defaultdict(list)
for l in open("$O/ab.txt"):
    m=re.match(r"lean (\d): ch (\d+) T (\d+) F \d+ interp (\d) block (\d+) .*fir kernel ([0-9.]+) ms",l)
    if m: d[(int(m.group(2)),int(m.group(3)),int(m.group(4)),int(m.group(5)),int(m.group(1)))].append(float(m.group(6))*1000)
for k in sorted(d): print(k, "%.1f us"%statistics.median(d[k]), d[k])
PY
cd $R; ARTAMD_GENERAL_LEAN=2 timeout 900 python -m pytest tests/test_gpu_asrc.py -x -q -m gpu 2>&1 | tail -2
