#!/bin/bash
# Pipelined-feed depth (FT_PIPE_DEPTH) against the three configurations: one JSON line per (depth, workload).
set -e
out=gpurun_out/pipe_depth.jsonl
: > $out
for d in 2 3 4; do
  for wl in cfg5 cfg3 cfg4; do
    FT_PIPE_DEPTH=$d timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --repeats 3 --no-extra-legs --no-cpu-baseline --no-latency 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(depth=$d, wl='$wl', ms=d['ms_per_step'], rep=d['repeats']['values'])))" >> $out
  done
done
cat $out
