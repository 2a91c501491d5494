#!/bin/bash
# Kernel trace of a short bench run and its overlap summary.
set -e
out=gpurun_out/trace_bench
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -o bench -- python3 bench.py --steps 60 --warmup 30 --no-cpu-baseline --no-latency --no-extra-legs > $out/bench.json 2> $out/err.log
python3 tools/kernel_overlap.py $out/t 0.25 0.50
find $out -name "*.csv" -size +6M -delete
find $out -name "*.db" -delete
