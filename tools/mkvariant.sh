#!/bin/bash
# usage: tools/mkvariant.sh NAME [extra hipcc flags]   builds cuda-pathtracer_amd/variants/libptamd_NAME.so with the
# kernels compiled under the extra flags (A/B runs load it through PTAMD_LIB, see tools/ab.sh)
set -e
name=$1; shift
cd $(dirname $0)/../cuda-pathtracer_amd
mkdir -p build_var/$name variants
F="--offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fPIC -w"
/opt/rocm/bin/hipcc $F "$@" -c csrc/pt_kernels.hip -o build_var/$name/pt_kernels.o &
/opt/rocm/bin/hipcc $F "$@" -c csrc/pt_wavefront.hip -o build_var/$name/pt_wavefront.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libptamd_$name.so build_var/$name/pt_kernels.o build_var/$name/pt_wavefront.o build/pt_gpu_build.o build/pt_capi.o build/pt_bvh.o build/pt_loader.o build/pt_jpeg.o -lz -lgomp -ldl -lpthread
ls -la variants/libptamd_$name.so
