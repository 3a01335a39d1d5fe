#!/bin/bash
# Bench line of the default build and of every experimental build under cuda-pathtracer_amd/variants (build_variant.sh).
# usage: tools/ab_variants.sh [bench arguments]
run() { PTAMD_LIB=$1 timeout -k 10 150 python3 bench.py --no-cpu --steps 4 --warmup 1 "${@:2}" > /tmp/ab.json 2>/dev/null || { echo "$1 failed"; return 1; }
  python3 -c "
import json;d=json.load(open('/tmp/ab.json'));print('%-28s %8.1f Mrays/s %8.2f ms/step  extend %.3f ms' % ('$(basename $1)', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"; }
run $PWD/cuda-pathtracer_amd/libptamd.so "$@" || exit 1
for so in cuda-pathtracer_amd/variants/*.so; do run $PWD/$so "$@" || exit 1; done
run $PWD/cuda-pathtracer_amd/libptamd.so "$@"
