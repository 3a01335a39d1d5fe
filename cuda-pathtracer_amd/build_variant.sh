#!/bin/bash
# usage: build_variant.sh NAME "-DPT_X=.. -DPT_Y=.."   -> variants/libptamd_NAME.so (experimental builds for A/B timing)
set -e
cd "$(dirname "$0")"
mkdir -p variants build_var
NAME=$1; FLAGS=$2
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-variable -Wno-unused-but-set-variable -Wno-unused-function"
$H $FLAGS -c csrc/pt_wavefront.hip -o build_var/wf_$NAME.o
$H $FLAGS -c csrc/pt_kernels.hip -o build_var/k_$NAME.o
$H $FLAGS -c csrc/pt_capi.hip -o build_var/c_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libptamd_$NAME.so build_var/k_$NAME.o build_var/wf_$NAME.o build/pt_gpu_build.o build_var/c_$NAME.o build/pt_bvh.o build/pt_loader.o build/pt_jpeg.o -lz -lgomp -ldl -lpthread
