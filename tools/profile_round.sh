#!/bin/bash
# One profiling pass over the bench workloads (run on the GPU box from the repository root):
#   the default bench line, rocprofv3 kernel statistics (two lanes and one), SQ counters per kernel, memory-side read
#   and write bytes (separate --pmc passes), for the atrium (bench default), the 10-million-triangle soup and the
#   sun/sky configuration; the bench lines of the latter two with their CPU baselines.
# usage: tools/profile_round.sh TAG     -> gpurun_out/TAG/
out=gpurun_out/${1:-r3}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="bench.py --no-cpu --steps 1 --warmup 0"
S="bench.py --no-cpu --workload soup --spp 16 --steps 1 --warmup 0"
K="bench.py --no-cpu --workload sunsky --steps 1 --warmup 0"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout -k 10 400 python3 -u bench.py > $out/bench.json 2> $out/bench.err || exit 1
tail -1 $out/bench.json | cut -c1-200
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o wf -- python3 bench.py --no-cpu --steps 2 --warmup 1 > $out/stats.log 2>&1 || exit 1
python3 tools/stream_gaps.py $(find $out/stats -name '*.db' | head -1) > $out/gaps.txt || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats1 -o wf -- python3 bench.py --no-cpu --steps 2 --warmup 1 --lanes 1 > $out/stats1.log 2>&1 || exit 1
for L in "" 1; do
  timeout -s KILL 200 rocprofv3 --pmc $SQ -d $out/sq$L -o a -- python3 $B ${L:+--lanes 1} > $out/sq$L.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/fetch$L -o f -- python3 $B ${L:+--lanes 1} > $out/fetch$L.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/write$L -o w -- python3 $B ${L:+--lanes 1} > $out/write$L.log 2>&1 || exit 1
done
timeout -k 10 500 python3 -u bench.py --workload soup --spp 16 --steps 3 --warmup 1 > $out/soup_bench.json 2> $out/soup_bench.err || exit 1
tail -1 $out/soup_bench.json | cut -c1-200
timeout -k 10 400 python3 -u bench.py --workload sunsky --steps 3 --warmup 1 > $out/sunsky_bench.json 2> $out/sunsky_bench.err || exit 1
tail -1 $out/sunsky_bench.json | cut -c1-200
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/soup_stats1 -o wf -- python3 $S --lanes 1 > $out/soup_stats1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/soup_stats -o wf -- python3 $S > $out/soup_stats.log 2>&1 || exit 1
for L in "" 1; do
  timeout -s KILL 300 rocprofv3 --pmc $SQ -d $out/soup_sq$L -o a -- python3 $S ${L:+--lanes 1} > $out/soup_sq$L.log 2>&1 || exit 1
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/soup_fetch$L -o f -- python3 $S ${L:+--lanes 1} > $out/soup_fetch$L.log 2>&1 || exit 1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/soup_write$L -o w -- python3 $S ${L:+--lanes 1} > $out/soup_write$L.log 2>&1 || exit 1
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/sunsky_stats1 -o wf -- python3 $K --lanes 1 > $out/sunsky_stats1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/sunsky_stats -o wf -- python3 $K > $out/sunsky_stats.log 2>&1 || exit 1
for L in "" 1; do
  timeout -s KILL 300 rocprofv3 --pmc $SQ -d $out/sunsky_sq$L -o a -- python3 $K ${L:+--lanes 1} > $out/sunsky_sq$L.log 2>&1 || exit 1
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/sunsky_fetch$L -o f -- python3 $K ${L:+--lanes 1} > $out/sunsky_fetch$L.log 2>&1 || exit 1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/sunsky_write$L -o w -- python3 $K ${L:+--lanes 1} > $out/sunsky_write$L.log 2>&1 || exit 1
done
for d in stats stats1 soup_stats soup_stats1 sq fetch write sq1 fetch1 write1 soup_sq soup_fetch soup_write soup_sq1 soup_fetch1 soup_write1 sunsky_stats sunsky_stats1 sunsky_sq sunsky_fetch sunsky_write sunsky_sq1 sunsky_fetch1 sunsky_write1; do
  db=$(find $out/$d -name '*.db' | head -1)
  case $d in stats|stats1|soup_stats|soup_stats1|sunsky_stats|sunsky_stats1) python3 profiles/summarize_rocpd.py $db > $out/$d.txt;; *) python3 profiles/pmc_summary.py $db > $out/$d.txt;; esac
  rm -rf $out/$d
done
ls -la $out
