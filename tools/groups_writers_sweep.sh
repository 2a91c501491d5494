# more stream groups with fewer database writers each: is the collapse beyond four groups the CPU share or the device?
for cfg in "4 2" "4 1" "5 1" "6 1" "8 1"; do set -- $cfg
GPU_MAX_HW_QUEUES=16 FT_DB_WRITERS=$2 timeout -k 10 200 python3 bench.py --groups $1 --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('groups',$1,'writers',$2,'value %.2f M'%(j['value']/1e6),'ms/step %.4f'%j['ms_per_step'], [round(v/1e6,2) for v in j['repeats']['values']], j['config'].get('hw_queues'), flush=True)" || exit 1
done
