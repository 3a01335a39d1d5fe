#!/bin/bash
# concurrent lanes x traversal blocks per CU
for l in 1 2 3; do for b in 3 4 5; do
  echo -n "lanes $l blocks/CU $b: "
  timeout -k 10 300 python3 -u bench.py --no-cpu --steps 3 --warmup 1 --lanes $l --tune 8,24,$b "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('Mrays/s %.1f  ms/step %.1f  extend_ms %.4f x %d' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['launches']))"
done; done
