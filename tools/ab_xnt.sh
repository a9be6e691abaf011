#!/bin/bash
# A/B build of the product library WITH the non-temporal LDS-DMA policy on the halo kernels' input tiles (FP_X_NT=1; the product
# default is 0) -> tools/_bin/libfp_xnt.so;
# run:  FP_LIB_PATH=tools/_bin/libfp_xnt.so python bench.py --no-extras --no-cpu-baseline
set -e
cd "$(dirname "$0")/../foundationpose_cpp_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-function -I../../include"
/opt/rocm/bin/hipcc $F -DFP_X_NT=1 -c fp_nn.hip -o /tmp/fp_nn_xnt.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_bin/libfp_xnt.so fp_api.o fp_geometry.o /tmp/fp_nn_xnt.o fp_mesh_loader.o fp_image_io.o -lz
echo built tools/_bin/libfp_xnt.so
