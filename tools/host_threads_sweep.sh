# helper threads per stream group (bench --host-threads) against database writers per group (FT_DB_WRITERS): the box's
# CPU share is 16, and every thread beyond it costs more than it brings
for cfg in ${SWEEP_CFGS:-"0 2" "12 1" "12 2" "16 1"}; do set -- $cfg
FT_DB_WRITERS=$2 FT_GROUP_TIMING=1 timeout -k 10 200 python3 bench.py --host-threads $1 --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 100 2>gpurun_out/ht.err | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('host_threads',$1,'writers',$2,'value %.2f M'%(j['value']/1e6),'ms/step %.4f'%j['ms_per_step'], [round(v/1e6,2) for v in j['repeats']['values']], {k:round(v*1000,1) for k,v in j['host_stage_ms'].items()}, flush=True)" || exit 1
grep "group 0+16" gpurun_out/ht.err | tail -1
done
