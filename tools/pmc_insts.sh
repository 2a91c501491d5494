#!/bin/bash
# Instruction-mix counters of the detector / pyramid / tracking kernels (one rocprofv3 counter pass per tool run).
# Results: gpurun_out/pmc_insts/<tool>/..._counter_collection.csv, summarized per kernel by tools/summarize_insts.py
set -e
out=gpurun_out/pmc_insts
mkdir -p $out
export TMPDIR=/tmp
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $out/fast -o p -- python3 tools/fast_bench.py > $out/fast.log 2> $out/fast.err
echo fast done
timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $out/lk -o p -- python3 tools/lk_bench.py > $out/lk.log 2> $out/lk.err
echo lk done
python3 tools/summarize_insts.py $out/fast $out/lk > $out/summary.txt
cat $out/summary.txt
find $out -name "*.csv" -size +4M -delete
find $out -name "*.db" -delete
