#!/bin/bash
# A/B of library variants on the bench workload: tools/ab.sh "bench args" variant...
args=$1; shift
for v in "$@"; do
  echo -n "$v: "
  PTAMD_LIB=$( [ -f $GRAFT_REPO_ROOT/cuda-pathtracer_amd/variants/libptamd_$v.so ] && echo $GRAFT_REPO_ROOT/cuda-pathtracer_amd/variants/libptamd_$v.so || echo $GRAFT_REPO_ROOT/cuda-pathtracer_amd/libptamd.so ) timeout -k 10 300 python3 -u bench.py --no-cpu --steps 3 --warmup 1 $args 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('Mrays/s %.1f  ms/step %.1f  extend_ms %.4f x %d  nodes/ray %.3f tris/ray %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['launches'], r['own_walk_per_ray']['wide_nodes'], r['own_walk_per_ray']['tri_tests']))" || exit 1
done
