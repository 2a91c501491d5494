# tracked features/s against the number of streams on one GPU and the number of stream groups (0 = library default)
for s in ${SWEEP_STREAMS:-8 16 32}; do for g in ${SWEEP_GROUPS:-0 1 2 3 4}; do
timeout -k 10 120 python3 bench.py --streams $s --groups $g --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('streams',$s,'groups',$g,'used',j['stream_groups'],'value %.1f M'%(j['value']/1e6),'ms/step %.4f'%j['ms_per_step'], [round(v/1e6,1) for v in j['repeats']['values']], flush=True)" || exit 1
done; done
