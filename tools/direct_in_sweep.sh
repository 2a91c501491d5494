# single-stream latency and small-batch throughput against the size up to which a tracking batch reads its input block in
# place from pinned host memory (FT_LK_DIRECT_MAX; 0 = always fetched into device memory first)
for dm in ${SWEEP_DIRECT:-0 1024 2048}; do
FT_LK_DIRECT_MAX=$dm timeout -k 10 200 python3 bench.py --streams 8 --no-extra-legs --no-cpu-baseline --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('direct_max',$dm,'8 streams %.2f M'%(j['value']/1e6), [round(v/1e6,2) for v in j['repeats']['values']], 'latency p50 %.4f pinned %.4f'%(j['latency']['p50_ms'], j['latency']['pinned_frames']['p50_ms']), flush=True)" || exit 1
FT_LK_DIRECT_MAX=$dm timeout -k 10 200 python3 bench.py --streams 16 --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('direct_max',$dm,'16 streams %.2f M'%(j['value']/1e6), [round(v/1e6,2) for v in j['repeats']['values']], flush=True)" || exit 1
done
