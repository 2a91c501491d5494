# pipelined throughput with and without the second device queue for ingest, pyramids and detector (FT_FRONT_QUEUE=0:
# never; unset: batches of one or two stream groups use it while they are fed pipelined)
for fq in 0 default; do
  if [ $fq = 0 ]; then export FT_FRONT_QUEUE=0; else unset FT_FRONT_QUEUE; fi
  for wl in ${SWEEP_WORKLOADS:-cfg3 cfg5 cfg2}; do
    timeout -k 10 200 python3 bench.py --workload $wl --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 50 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('front_queue','$fq','$wl','value %.2f M'%(j['value']/1e6),'ms/step %.4f'%j['ms_per_step'], [round(v/1e6,2) for v in j['repeats']['values']], flush=True)" || exit 1
  done
  for s in ${SWEEP_STREAMS:-8 16}; do
    timeout -k 10 200 python3 bench.py --streams $s --no-extra-legs --no-cpu-baseline --no-latency --steps 100 --warmup 100 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('front_queue','$fq','cfg4 x',$s,'value %.2f M'%(j['value']/1e6),'ms/step %.4f'%j['ms_per_step'], [round(v/1e6,2) for v in j['repeats']['values']], flush=True)" || exit 1
  done
done
