#!/bin/bash
# Detector micro-benchmark, then one counter pass over it (instruction mix per wave).
set -e
out=gpurun_out/pmc_fast
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 120 python3 tools/fast_bench.py > $out/time.log 2>&1
cat $out/time.log
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $out/fast -o p -- python3 tools/fast_bench.py > $out/fast.log 2> $out/fast.err
python3 tools/summarize_insts.py $out/fast | cut -c1-400
find $out -name "*.db" -delete
