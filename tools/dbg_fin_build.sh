#!/bin/bash
# builds differentiable-wdfs_amd/lib/wdf_hip/libwdf_dbg.so: the product library with the clipper translation unit compiled -DWDF_DBG_TIMES
set -e
cd "$(dirname "$0")/../differentiable-wdfs_amd/csrc"
make > /dev/null
mkdir -p build_dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DWDF_DBG_TIMES \
    -c wdf_capi_clipper.hip -o build_dbg/wdf_capi_clipper.o 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/wdf_hip/libwdf_dbg.so build/wdf_capi.o build_dbg/wdf_capi_clipper.o build/wdf_capi_ss.o \
    build/wdf_capi_mlp.o build/wdf_capi_mlp_step.o build/wdf_capi_ss_step.o build/wdf_capi_ss_dyn.o
ls -la ../lib/wdf_hip/libwdf_dbg.so
