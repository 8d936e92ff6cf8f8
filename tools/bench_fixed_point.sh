#!/bin/bash
# Where the fixed-point matrix kernel pays: device-resident throughput of several shapes and call sizes with the f32 streaming
# kernel pinned (kernel preference 6), the fixed-point kernel forced (7) and the library's own choice (0).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { for k in 6 7 0; do python $R/tools/bench_shapes.py "$@" $k 2>/dev/null | sed 's/tiles\/wg auto//'; done; echo; }
for blk in 32768 65536 131072 262144 1048576; do run 8 988 988 44100 48000 0 1 $blk; done
for blk in 65536 262144 1048576; do run 4 988 988 44100 48000 0 1 $blk; done
run 32 988 988 44100 48000 0 1 262144
run 8 988 988 96000 44100 1 1 1048576
for blk in 262144 1048576; do run 2 380 380 44100 48000 0 1 $blk; done
run 1 380 380 44100 48000 0 1 1048576
run 16 156 156 44100 48000 0 1 262144
run 2 64 160 48000 44100 0 0 1048576
