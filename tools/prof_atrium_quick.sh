#!/bin/bash
# Quick counter passes over the bench's default workload only (kernel statistics, SQ counters, memory-side read and
# write bytes, each in a run of its own): tools/prof_atrium_quick.sh TAG -> gpurun_out/TAG/{stats,sq,fetch,write}.txt
out=gpurun_out/${1:-quick}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="bench.py --no-cpu --steps 1 --warmup 0"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o wf -- python3 bench.py --no-cpu --steps 2 --warmup 1 > $out/stats.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc $SQ -d $out/sq -o a -- python3 $B > $out/sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o f -- python3 $B > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/write -o w -- python3 $B > $out/write.log 2>&1 || exit 1
for d in stats sq fetch write; do
  db=$(find $out/$d -name '*.db' | head -1)
  case $d in stats) python3 profiles/summarize_rocpd.py $db > $out/$d.txt;; *) python3 profiles/pmc_summary.py $db > $out/$d.txt;; esac
  rm -rf $out/$d
done
