# detector + pyramid level 1 -> 2 and ranking + level 2 -> 3 in shared launches (default) against five separate launches (FT_FRONT_FUSE=0)
for ff in 0 1; do
FT_FRONT_FUSE=$ff timeout -k 10 300 python3 bench.py --no-extra-legs --no-cpu-baseline --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('front_fuse',$ff,'cfg4 %.2f M'%(j['value']/1e6), [round(v/1e6,2) for v in j['repeats']['values']], 'ms/step %.4f'%j['ms_per_step'], 'latency p50 %.4f pinned %.4f'%(j['latency']['p50_ms'], j['latency']['pinned_frames']['p50_ms']), flush=True)" || exit 1
for s in 8 32; do
FT_FRONT_FUSE=$ff timeout -k 10 200 python3 bench.py --streams $s --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('front_fuse',$ff,'$s streams %.2f M'%(j['value']/1e6), [round(v/1e6,2) for v in j['repeats']['values']], flush=True)" || exit 1
done
FT_FRONT_FUSE=$ff timeout -k 10 300 python3 bench.py --workload cfg4_win15 --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('front_fuse',$ff,'cfg4_win15 %.2f M'%(j['value']/1e6), flush=True)" || exit 1
done
