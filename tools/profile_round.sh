#!/bin/bash
# One profiling pass over the bench workloads (run on the GPU box from the repository root):
#   the default bench line, rocprofv3 kernel statistics, SQ counters per kernel, memory-side read and write bytes
#   (separate --pmc passes), for the atrium (bench default) and for the 10-million-triangle soup.
# usage: tools/profile_round.sh TAG     -> gpurun_out/TAG/
out=gpurun_out/${1:-r2}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="bench.py --no-cpu --steps 1 --warmup 0"
S="bench.py --workload soup --spp 16 --steps 1 --warmup 0"
timeout -k 10 400 python3 -u bench.py > $out/bench.json 2> $out/bench.err || exit 1
tail -1 $out/bench.json | cut -c1-300
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o wf -- python3 bench.py --no-cpu --steps 2 --warmup 1 > $out/stats.log 2>&1 || exit 1
python3 tools/stream_gaps.py $(find $out/stats -name '*.db' | head -1) > $out/gaps.txt || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats1 -o wf -- python3 bench.py --no-cpu --steps 2 --warmup 1 --lanes 1 > $out/stats1.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/sq -o a -- python3 $B > $out/sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o f -- python3 $B > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/write -o w -- python3 $B > $out/write.log 2>&1 || exit 1
timeout -k 10 400 python3 -u bench.py --workload soup --spp 16 --steps 3 --warmup 1 > $out/soup_bench.json 2> $out/soup_bench.err || exit 1
tail -1 $out/soup_bench.json | cut -c1-300
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/soup_sq -o a -- python3 $S > $out/soup_sq.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/soup_fetch -o f -- python3 $S > $out/soup_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/soup_write -o w -- python3 $S > $out/soup_write.log 2>&1 || exit 1
for d in stats stats1 sq fetch write soup_sq soup_fetch soup_write; do
  db=$(find $out/$d -name '*.db' | head -1)
  if [ "$d" = stats ] || [ "$d" = stats1 ]; then python3 profiles/summarize_rocpd.py $db > $out/$d.txt; else python3 profiles/pmc_summary.py $db > $out/$d.txt; fi
  rm -rf $out/$d
done
ls -la $out
