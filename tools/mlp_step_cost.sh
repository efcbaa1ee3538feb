#!/bin/bash
# tools/mlp_step_cost.sh build|run -- price the ingredients of the resident MLP-root step's kernels: libraries whose step
# kernels have ONE ingredient removed (csrc/wdf_mlp_step.h, WDF_DBG_STEP bits: 1 tanh, 2 LDS transposes, 4 outer-product
# MFMAs, 8 delta-chain MFMAs, 16 the forward's kappa chain), timed with tools/mlp_step_bench.py.  Results are WRONG by
# construction (the verification flags chunks at random): only the kernel times mean anything.
cd "$(dirname "$0")/.."
CS=differentiable-wdfs_amd/csrc; LIBD=differentiable-wdfs_amd/lib/wdf_hip
if [ "$1" = build ]; then
  for v in 1 2 4 8 16 6 14 15; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 \
      -DWDF_DBG_STEP=$v -c $CS/wdf_capi_mlp_step.hip -o /tmp/step_dbg$v.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIBD/libwdf_hip_dbg$v.so $CS/build/wdf_capi.o $CS/build/wdf_capi_clipper.o \
      $CS/build/wdf_capi_ss.o $CS/build/wdf_capi_mlp.o /tmp/step_dbg$v.o && echo built $v &
  done; wait
else
  for v in 0 1 2 4 8 16 6 14 15; do
    L=$PWD/$LIBD/libwdf_hip_dbg$v.so; [ $v = 0 ] && L=$PWD/$LIBD/libwdf_hip.so
    echo "== WDF_DBG_STEP=$v"; WDF_HIP_LIB=$L REPLAN=0 FREEZE=1 timeout 120 python tools/mlp_step_bench.py 2x16_pre 20 2>&1 | grep -E "kernels:"
  done
fi
