# tracked features/s of the 64-stream benchmark against the number of database writer threads per stream group
for nw in ${SWEEP_WRITERS:-1 2 3 4}; do
FT_DB_WRITERS=$nw timeout -k 10 200 python3 bench.py --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('writers',$nw,'value %.2f M'%(j['value']/1e6),'ms/step %.4f'%j['ms_per_step'], [round(v/1e6,2) for v in j['repeats']['values']], flush=True)" || exit 1
done
