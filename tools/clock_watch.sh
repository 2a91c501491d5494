#!/bin/bash
# Samples the GPU clocks while a longer bench run is in flight.
mkdir -p gpurun_out
timeout -k 10 150 python bench.py --steps 3000 --warmup 8 --no-cpu-baseline --no-latency > gpurun_out/b_long.json 2> gpurun_out/b_long.err &
pid=$!
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -3; rocm-smi --showuse --showpower 2>/dev/null | grep -E "GPU use|Power" | head -2; echo; sleep 1; done
wait $pid
echo bench rc $?
