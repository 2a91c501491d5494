# the tracking batch's input block written by the host straight into device memory (default where the device has a large
# BAR) against the fetch kernel / in-place reads from pinned memory (FT_LK_BAR_INPUT=0)
for bi in 0 1; do
FT_LK_BAR_INPUT=$bi timeout -k 10 300 python3 bench.py --no-extra-legs --no-cpu-baseline --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('bar_input',$bi,'cfg4 %.2f M'%(j['value']/1e6), [round(v/1e6,2) for v in j['repeats']['values']], 'latency p50 %.4f pinned %.4f'%(j['latency']['p50_ms'], j['latency']['pinned_frames']['p50_ms']), flush=True)" || exit 1
for s in 8 32; do
FT_LK_BAR_INPUT=$bi timeout -k 10 200 python3 bench.py --streams $s --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('bar_input',$bi,'$s streams %.2f M'%(j['value']/1e6), [round(v/1e6,2) for v in j['repeats']['values']], flush=True)" || exit 1
done
for wl in cfg3 cfg5; do
FT_LK_BAR_INPUT=$bi timeout -k 10 200 python3 bench.py --workload $wl --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 50 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('bar_input',$bi,'$wl %.2f M'%(j['value']/1e6),'ms/step %.4f'%j['ms_per_step'], flush=True)" || exit 1
done; done
