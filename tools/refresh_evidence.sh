set -u
R=$PWD
mkdir -p gpurun_out/final2
python bench.py --steps 100 --warmup 10 > gpurun_out/final2/bench.json 2> gpurun_out/final2/bench.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final2/prof -o final -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/final2/prof.log 2>&1 )
bash tools/pmc.sh final2/pmc > gpurun_out/final2/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/final2/pmc > gpurun_out/final2/pmc_summary.txt 2>&1
python tools/bench_wide.py --block 1048576 --steps 20 > gpurun_out/final2/wide.jsonl 2>/dev/null
python tools/bench_configs.py > gpurun_out/final2/configs.jsonl 2> gpurun_out/final2/configs.err
find gpurun_out/final2 -name "*stats*" | head; tail -c 600 gpurun_out/final2/bench.json
