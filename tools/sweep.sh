#!/bin/bash
# sweep of runtime tuning parameters on the bench workload: each argument is "pool:tune"
for v in "$@"; do
  pool=${v%%:*}; tune=${v##*:}
  echo -n "pool=$pool tune=$tune: "
  timeout -k 10 200 python3 -u bench.py --no-cpu --steps 1 --warmup 1 --spp 32 --pool $pool --tune $tune 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('Mrays/s %.1f  ms/step %.1f  extend_ms %.4f x %d  (extend total %.0f ms)' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['launches'], r['kernel_ms']*r['launches']))"
done
