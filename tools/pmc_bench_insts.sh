#!/bin/bash
# Instruction counts per wave of every kernel of the bench workload (one counter pass over a short bench run).
set -e
out=gpurun_out/pmc_bench_insts
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY"
timeout -s KILL 250 rocprofv3 --pmc $C --output-format csv -d $out/a -o p -- python3 bench.py --steps 20 --warmup 4 --repeats 1 --no-extra-legs --no-cpu-baseline --no-latency --groups 1 > $out/bench.json 2> $out/err.log
python3 tools/summarize_insts.py $out/a | cut -c1-400 > $out/summary.txt
cat $out/summary.txt
find $out -name "*.db" -delete
find $out -name "*.csv" -size +6M -delete
