#!/bin/bash
# usage: trace_variant.sh NAME [bench args]   kernel statistics of one library variant on the bench workload
set -e
v=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export PTAMD_LIB=$GRAFT_REPO_ROOT/cuda-pathtracer_amd/variants/libptamd_$v.so
rm -rf gpurun_out/tv_$v
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/tv_$v -o wf -- python3 bench.py --no-cpu --steps 1 --warmup 0 "$@" > gpurun_out/tv_$v.log 2>&1
python3 profiles/summarize_rocpd.py $(find gpurun_out/tv_$v -name "*.db") | cut -c1-60,73-200 | head -12
