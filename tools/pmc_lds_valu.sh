#!/bin/bash
# LDS bank-conflict rate and vector-ALU busy rate of every kernel of the bench workload: one counter pass over a short
# one-group bench run (64-stream launches), summarized per kernel by tools/summarize_lds_valu.py.
set -e
out=gpurun_out/pmc_lds_valu
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE"
timeout -s KILL 250 rocprofv3 --pmc $C --output-format csv -d $out/a -o p -- python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-latency --groups 1 > $out/bench.json 2> $out/err.log
python3 tools/summarize_lds_valu.py $out/a > $out/summary.txt
cat $out/summary.txt
find $out -name "*.db" -delete
find $out -name "*.csv" -size +6M -delete
