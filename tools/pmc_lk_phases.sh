#!/bin/bash
# Vector instructions per wave of the tracking kernel with 0, 1 and all Newton iterations: the difference between the
# first two is one iteration per level, the first alone is the per-level set-up (staging, template, gradient matrix).
set -e
out=gpurun_out/pmc_lk_phases
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
for m in iter0 iter1 full; do
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/$m -o p -- python3 tools/lk_bench.py $m > $out/$m.log 2>&1
  echo "mode $m"; python3 tools/summarize_insts.py $out/$m | grep lk_kernel | cut -c60-330
done
find $out -name "*.db" -delete
