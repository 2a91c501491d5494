#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box: kernel trace + stats, then HBM read and write counters in runs of
# their own.  Results land under gpurun_out/prof_<tag>/.
set -e
tag=${1:-r1}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
args="--steps 40 --warmup 8 --repeats 1 --no-extra-legs --no-cpu-baseline --no-latency"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 bench.py $args > $out/trace_bench.json 2> $out/trace.err
echo trace done
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o bench -- python3 bench.py $args > $out/pmc_fetch_bench.json 2> $out/pmc_fetch.err
echo fetch done
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o bench -- python3 bench.py $args > $out/pmc_write_bench.json 2> $out/pmc_write.err
echo write done
python3 tools/summarize_pmc.py $out/pmc_fetch $out/pmc_write > $out/pmc_summary.json
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
# keep the merge-back small
find $out -name "*.csv" -size +4M -delete
find $out -name "*.db" -delete
ls -la $out
