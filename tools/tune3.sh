#!/bin/bash
for rep in 1 2; do for t in 12,16,3 12,24,3 8,24,3 8,20,3 10,28,3; do
  echo -n "tune $t: "
  timeout -k 10 300 python3 -u bench.py --no-cpu --steps 4 --warmup 1 --tune $t "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('Mrays/s %.1f  ms/step %.1f  extend_ms %.4f x %d' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['launches']))"
done; done
