# 64 streams: fewer stream groups, each with a front queue of its own (FT_FRONT_QUEUE=1), against four groups on one queue each
for cfg in "4 0" "2 1" "2 0" "1 1" "3 1"; do set -- $cfg
FT_FRONT_QUEUE=$2 timeout -k 10 200 python3 bench.py --groups $1 --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('groups',$1,'front_queue',$2,'value %.2f M'%(j['value']/1e6),'ms/step %.4f'%j['ms_per_step'], [round(v/1e6,2) for v in j['repeats']['values']], flush=True)" || exit 1
done
