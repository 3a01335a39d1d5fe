#!/bin/bash
# usage: ab_tune.sh VARIANT "r,v,b" ...   bench of one library variant over traversal tunings
v=$1; shift
for t in "$@"; do
  echo -n "$v $t: "
  PTAMD_LIB=$GRAFT_REPO_ROOT/cuda-pathtracer_amd/variants/libptamd_$v.so timeout -k 10 200 python3 -u bench.py --no-cpu --steps 2 --warmup 1 --tune $t 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('Mrays/s %.1f  ms/step %.1f  extend_ms %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms']))"
done
