#!/bin/bash
# pool-size sweep on the bench workload
for p in 16777216 33554432 67108864 134217728; do
  echo -n "pool $p: "
  timeout -k 10 300 python3 -u bench.py --no-cpu --steps 3 --warmup 1 --pool $p "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('Mrays/s %.1f  ms/step %.1f  extend_ms %.4f x %d' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['launches']))"
done
