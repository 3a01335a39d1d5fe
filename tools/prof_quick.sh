#!/bin/bash
# Quick look at one build (run on the GPU box from the repository root): the bench line, kernel statistics with one
# lane (where kernel times add up) and with the default two, and the SQ counters per kernel.
# usage: tools/prof_quick.sh TAG [extra bench arguments]     -> gpurun_out/TAG/
out=gpurun_out/${1:-quick}; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 python3 -u bench.py --no-cpu --steps 5 --warmup 1 "$@" > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats1 -o wf -- python3 bench.py --no-cpu --steps 2 --warmup 1 --lanes 1 "$@" > $out/stats1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o wf -- python3 bench.py --no-cpu --steps 2 --warmup 1 "$@" > $out/stats.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/sq -o a -- python3 bench.py --no-cpu --steps 1 --warmup 0 "$@" > $out/sq.log 2>&1 || exit 1
for d in stats stats1 sq; do
  db=$(find $out/$d -name '*.db' | head -1)
  if [ "$d" = sq ]; then python3 profiles/pmc_summary.py $db > $out/$d.txt; else python3 profiles/summarize_rocpd.py $db > $out/$d.txt; fi
  rm -rf $out/$d
done
