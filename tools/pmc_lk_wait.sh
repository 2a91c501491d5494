#!/bin/bash
# Where the tracking kernel's waves spend their time: one counter pass over tools/lk_bench.py.
set -e
out=gpurun_out/pmc_lk_wait
mkdir -p $out
export TMPDIR=/tmp
C="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $out/a -o p -- python3 tools/lk_bench.py > $out/a.log 2> $out/a.err
python3 tools/summarize_insts.py $out/a | grep -E "lk_kernel|fast_detect" | cut -c1-420
C="SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $out/b -o p -- python3 tools/lk_bench.py > $out/b.log 2> $out/b.err
python3 tools/summarize_insts.py $out/b | grep -E "lk_kernel|fast_detect" | cut -c1-420
find $out -name "*.db" -delete
