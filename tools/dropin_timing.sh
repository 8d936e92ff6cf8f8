#!/bin/bash
# wall-clock of the reference's own artest program: reference DSP (its Makefile flags, -m = worker threads) vs libartamd.so
cd "$(dirname "$0")/.."
R=oracle/_ref
t() { local s=$(date +%s.%N); "$@" >/dev/null 2>/tmp/err.txt; local e=$(date +%s.%N); echo "$(echo "$e - $s" | bc -l | cut -c1-6) s   $*   [$(grep -o 'output (-w2): count = *[0-9]*' /tmp/err.txt)]"; }
for args in "-4 -c8 -n60 -s44100 -d48000" "-4 -c8 -n60 -s44100 -d48000 -b65536" "-4 -e -l -c8 -n60 -s96000 -d44100 -b65536 -o16" "-3 -c2 -n120 -s44100 -d48000 -b65536"; do
  t $R/artest_make $args
  t $R/artest_make -m $args
  t $R/artest_amd $args
done
