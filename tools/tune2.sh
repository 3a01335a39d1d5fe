#!/bin/bash
# traversal kernel tuning sweep on the bench workload: refill_lanes,leaf_vote,blocks_per_cu
for t in 12,16,3 12,16,4 12,16,5 8,16,3 16,16,3 20,16,3 12,8,3 12,24,3 12,32,3 8,12,4; do
  echo -n "tune $t: "
  timeout -k 10 300 python3 -u bench.py --no-cpu --steps 3 --warmup 1 --tune $t "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('Mrays/s %.1f  ms/step %.1f  extend_ms %.4f x %d' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['launches']))"
done
