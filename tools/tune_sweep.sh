#!/bin/bash
# usage: tune_sweep.sh "r,v,b" ...   default library over traversal tunings on the bench workload
for t in "$@"; do
  echo -n "$t: "
  timeout -k 10 200 python3 -u bench.py --no-cpu --steps 3 --warmup 1 --tune $t 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('Mrays/s %.1f  ms/step %.1f  extend_ms %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms']))"
done
