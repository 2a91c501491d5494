#!/bin/bash
# Instruction mix and wave residency of experimental detector builds (libraries under feature_tracker_amd/_lib/exp).
set -e
out=gpurun_out/pmc_fast_variants
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
C1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
for v in "$@"; do
  export FT_EXP_LIB=feature_tracker_amd/_lib/exp/libft_device_$v.so
  timeout -s KILL 200 rocprofv3 --pmc $C1 --output-format csv -d $out/$v -o p -- python3 tools/fast_bench.py > $out/$v.log 2> $out/$v.err
  echo "variant $v"
  python3 tools/summarize_insts.py $out/$v | grep -E "fast_detect" | cut -c60-420
done
find $out -name "*.db" -delete
