#!/bin/bash
# Round evidence on the GPU box: kernel trace + stats of the driver's command and of the one-group runs (whose launches are
# the ones the roofline entry is measured on) for the 21x21 and the 15x15 window, then the counter passes (HBM read, HBM
# write, instruction counts, LDS conflicts / vector-ALU busy), each in a run of its own.  Results under gpurun_out/prof_<tag>/.
set -e
tag=${1:-r3_a}
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
one="--steps 20 --warmup 4 --repeats 1 --no-extra-legs --no-cpu-baseline --no-latency --groups 1"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 bench.py --steps 20 --warmup 5 > $out/bench_under_rocprof.json 2> $out/trace.err
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
echo trace done
# the one-group runs keep every kind of front work in launches of its own (FT_FRONT_FUSE=0), so that the counters of the
# pyramid passes and of the detector stay apart; the device functions are the same as in the shared launches
export FT_FRONT_FUSE=0
for wl in cfg4 cfg4_win15; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_g1_$wl -o bench -- python3 bench.py --workload $wl $one > $out/bench_groups1_${wl}_under_rocprof.json 2> $out/trace_g1_$wl.err
  find $out/trace_g1_$wl -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_groups1_$wl.csv \;
  echo trace groups1 $wl done
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$wl -o p -- python3 bench.py --workload $wl $one > /dev/null 2> $out/fetch_$wl.err
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write_$wl -o p -- python3 bench.py --workload $wl $one > /dev/null 2> $out/write_$wl.err
  echo hbm $wl done
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/insts_$wl -o p -- python3 bench.py --workload $wl $one > /dev/null 2> $out/insts_$wl.err
  python3 tools/summarize_insts.py $out/insts_$wl | cut -c1-400 > $out/pmc_insts_$wl.txt
  echo insts $wl done
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/ldsvalu_$wl -o p -- python3 bench.py --workload $wl $one > /dev/null 2> $out/ldsvalu_$wl.err
  python3 tools/summarize_lds_valu.py $out/ldsvalu_$wl > $out/pmc_lds_valu_$wl.txt
  echo lds/valu $wl done
  python3 tools/pmc_per_unit.py $wl 64 $out/fetch_$wl $out/write_$wl $out/insts_$wl > $out/pmc_per_unit_$wl.json
done
find $out -name "*.db" -delete
find $out -name "*.csv" -size +2M -delete
ls -la $out
