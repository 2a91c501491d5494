# tracking kernels compiled for the 21 x 21 and 15 x 15 windows against the general ones (FT_LK_FIXED_WINDOWS=0)
for fw in 0 1; do for wl in cfg4 cfg4_win15; do
FT_LK_FIXED_WINDOWS=$fw timeout -k 10 200 python3 bench.py --workload $wl --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('fixed_windows',$fw,'$wl','value %.2f M'%(j['value']/1e6),'ms/step %.4f'%j['ms_per_step'], [round(v/1e6,2) for v in j['repeats']['values']], 'lk %.1f us'%(j['kernels']['lk_track']['avg_ms']*1000), flush=True)" || exit 1
done; done
