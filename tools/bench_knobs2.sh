#!/bin/bash
# Throughput against stream groups with more hardware queues per process (GPU box).
export GPU_MAX_HW_QUEUES=${1:-8}
for args in "--groups 3" "--groups 4" "--groups 5 --host-threads 10" "--groups 6 --host-threads 12" "--groups 8 --host-threads 16"; do
  timeout -k 10 120 python bench.py --steps 100 --repeats 3 --no-extra-legs --no-latency --no-cpu-baseline $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$args'.ljust(34), [round(v/1e6,1) for v in d['repeats']['values']], 'groups', d['stream_groups'], {k: round(v*1e3) for k,v in d['host_stage_ms'].items()})"
done
