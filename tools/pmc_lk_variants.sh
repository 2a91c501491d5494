#!/bin/bash
# Instruction counts per wave of experimental builds of the tracking kernel (libraries under feature_tracker_amd/_lib/exp).
set -e
out=gpurun_out/pmc_lk_variants
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
for v in "$@"; do
  export FT_EXP_LIB=feature_tracker_amd/_lib/exp/libft_device_$v.so
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/$v -o p -- python3 tools/lk_bench.py iter0 > $out/$v.log 2>&1
  echo "variant $v"; python3 tools/summarize_insts.py $out/$v | grep lk_kernel | cut -c60-330
done
find $out -name "*.db" -delete
