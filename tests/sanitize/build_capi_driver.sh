#!/bin/bash
# Builds tests/sanitize/capi_driver with the GPU library's host code under ASan + UBSan (device code is compiled
# normally: GPU sanitizers are not available on this pool).  The binary is self-contained apart from libz and the
# HIP runtime, travels to the GPU box with the tree, and is run there by tests/test_gpu_sanitized_capi.py.
set -e
cd "$(dirname "$0")/../.."
P=cuda_path_tracer_amd
OBJ=$(mktemp -d)
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer"
INC="-Iinclude -I$P/csrc -I$P/host"
for f in tests/sanitize/capi_driver.cpp $P/host/ray_file.cpp $P/host/bvh_build.cpp $P/host/flatten.cpp $P/host/png_write.cpp $P/host/host_capi.cpp; do
  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -ffp-contract=off $SAN $INC -c $f -o $OBJ/$(basename $f).o
done
for f in $P/csrc/kernels.hip $P/csrc/capi.hip $P/csrc/bvh_device.hip; do
  /opt/rocm/bin/hipcc -O3 -gline-tables-only -std=c++17 --offload-arch=gfx950 -ffp-contract=off $INC $SAN -fno-gpu-sanitize \
    -Wno-unused-function -c $f -o $OBJ/$(basename $f).o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $SAN -fno-gpu-sanitize $OBJ/*.o -o tests/sanitize/capi_driver -lz -pthread
rm -rf $OBJ
echo built tests/sanitize/capi_driver
