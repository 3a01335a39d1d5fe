#!/bin/bash
# One profiling pass over the bench workload: the default bench line, rocprofv3 kernel statistics, and
# two counter passes (memory-side read and write bytes). Run on the GPU box from the repository root.
set -e
out=gpurun_out/${1:-r1}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 400 python3 -u bench.py > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o wf -- python3 bench.py --no-cpu --steps 2 --warmup 1 > $out/stats.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o f -- python3 bench.py --no-cpu --steps 1 --warmup 0 > $out/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o w -- python3 bench.py --no-cpu --steps 1 --warmup 0 > $out/pmc_write.log 2>&1
find $out -name '*.db' | while read f; do ls -la $f; done
