#!/bin/bash
# LK kernel check after a change: bit-exact parity tests of the tracking kernel, its time decomposition and its
# instruction counts per wave.  Run on the GPU box:  bash tools/lk_quick.sh <tag>
tag=${1:-x}
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "lk or golden or full_size or properties or ragged" --timeout 120 --timeout-method thread > gpurun_out/lkq_${tag}_test.log 2>&1
tail -2 gpurun_out/lkq_${tag}_test.log
timeout -k 10 120 python tools/lk_bench.py > gpurun_out/lkq_${tag}.log 2>&1 && cat gpurun_out/lkq_${tag}.log
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_lkq_$tag -o p -- python3 tools/lk_bench.py full > /dev/null 2>&1
python3 tools/summarize_insts.py gpurun_out/pmc_lkq_$tag | grep lk_kernel
find gpurun_out/pmc_lkq_$tag -name "*.csv" -size +1M -delete
