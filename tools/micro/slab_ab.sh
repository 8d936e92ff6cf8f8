# generic alternating A/B of libartamd variants under _abl/ on the slab kernel's shapes: bash tools/micro/slab_ab.sh OUTDIR VARIANT...
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2 3 4; do
for shape in "8 988 988 44100 48000 0 1 1048576" "4 988 988 44100 48000 0 1 1048576" "32 988 988 44100 48000 0 1 262144" "8 988 988 44100 48000 0 1 524288"; do
  for v in "$@"; do ARTAMD_LIB=$R/_abl/libartamd_$v.so timeout 120 python $R/tools/bench_shapes.py $shape 7 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /"; done
done
done > $O/ab.txt
python - <<PY
import re,statistics,collections
d=collections.defaultdict(list)
for l in open("$O/ab.txt"):
    m=re.match(r"(\w+): ch (\d+) .* block (\d+) .*step ([0-9.]+) ms\s+fir kernel ([0-9.]+) ms",l)
    if m: d[(int(m.group(2)),int(m.group(3)),m.group(1))].append((float(m.group(5))*1000,float(m.group(4))*1000))
for k in sorted(d): print(k, "kernel %.1f us  step %.1f us"%(statistics.median(x[0] for x in d[k]), statistics.median(x[1] for x in d[k])), [round(x[0],1) for x in d[k]])
PY
