#!/bin/bash
# Detector micro-benchmark under two counter passes: instruction mix, then wave residency / LDS activity.
set -e
out=gpurun_out/pmc_fast2
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
C1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
timeout -s KILL 200 rocprofv3 --pmc $C1 --output-format csv -d $out/a -o p -- python3 tools/fast_bench.py > $out/a.log 2> $out/a.err
python3 tools/summarize_insts.py $out/a | grep -E "fast_|level" | cut -c1-420
C2="SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
timeout -s KILL 200 rocprofv3 --pmc $C2 --output-format csv -d $out/b -o p -- python3 tools/fast_bench.py > $out/b.log 2> $out/b.err
python3 tools/summarize_insts.py $out/b | grep -E "fast_|level" | cut -c1-420
find $out -name "*.db" -delete
