# host frames staged in device memory through the large BAR (default where there is one) against pinned host staging read
# over the bus (FT_STAGE_BAR=0): single-stream latency and the PCIe-inclusive throughput of the 64-stream benchmark
for sb in 0 1; do
FT_STAGE_BAR=$sb timeout -k 10 300 python3 bench.py --streams 8 --no-extra-legs --no-cpu-baseline --steps 50 --warmup 50 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('stage_bar',$sb,'latency p50 %.4f p95 %.4f pinned %.4f'%(j['latency']['p50_ms'], j['latency']['p95_ms'], j['latency']['pinned_frames']['p50_ms']), flush=True)" || exit 1
FT_STAGE_BAR=$sb timeout -k 10 300 python3 bench.py --host-frames --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('stage_bar',$sb,'host frames %.2f M'%(j['value']/1e6), [round(v/1e6,2) for v in j['repeats']['values']], 'ms/step %.4f'%j['ms_per_step'], flush=True)" || exit 1
done
