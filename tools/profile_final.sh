#!/bin/bash
# Round-end evidence: kernel trace + stats of the default bench command and of the one-group variant (whose launches are
# the ones the roofline entry is measured on), HBM counters in runs of their own, instruction counts per wave.
set -e
tag=${1:-r1_h}
bash tools/profile_bench.sh $tag
out=gpurun_out/prof_$tag
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_g1 -o bench -- python3 bench.py --steps 40 --warmup 8 --repeats 1 --no-extra-legs --no-cpu-baseline --no-latency --groups 1 > $out/trace_g1_bench.json 2> $out/trace_g1.err
find $out/trace_g1 -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_groups1.csv \;
echo groups1 done
bash tools/pmc_bench_insts.sh
cp gpurun_out/pmc_bench_insts/summary.txt $out/pmc_insts.txt
find $out -name "*.csv" -size +4M -delete
find $out -name "*.db" -delete
ls -la $out
