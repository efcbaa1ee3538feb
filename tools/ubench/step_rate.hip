// Compute-only rate of the one-pass step (wdf_clipper_fused.h, fused_step) and of the plain forward step: no memory
// traffic, inputs synthesised in registers, W waves per SIMD.  ns per wave-step per SIMD = what the SIMD can sustain.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -I differentiable-wdfs_amd/csrc tools/ubench/step_rate.hip -o step_rate.bin
#include "wdf_clipper_fused.h"
#include <cstdio>
using namespace wdf;

template <int MODE, typename V>
__global__ __launch_bounds__(64) void k(const float* theta, float* out, int iters)
{
    const ClipConsts c = load_consts(theta, 48000.f, 1, 1);
    V z = vsplat<V>(0.01f * threadIdx.x);
    FusedTan<V> s; s.init();
    V x = vsplat<V>(0.3f + 0.001f * threadIdx.x), t = vsplat<V>(0.1f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            V y;
            if (MODE == 0) y = fwd_step<false, true, V, true>(c, x, vsplat<V>(1.f), z);
            else y = fused_step<false, true, true, V>(c, x, vsplat<V>(1.f), t, 1e-8f, z, s);
            x = x * -0.999f;           // keeps the input moving (1 extra VALU per step)
            t = y;
            if (MODE == 1) { s.pin(); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = vget(z + s.A + s.cL + s.cV + s.cP + s.GA + s.GL + s.GV + s.GP + s.sse + t, 0);
}

template <int MODE, typename V>
void run(const char* name, int w, const float* theta, float* out)
{
    const int iters = 1000, grid = 256 * 4 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, V><<<grid, 64>>>(theta, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, V><<<grid, 64>>>(theta, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)iters * 16;
    printf("%-28s waves/SIMD=%d : %.1f ns per wave-step per SIMD (%.1f ns per sequence-step)\n", name, w, ms * 1e6 / (steps * w),
           ms * 1e6 / (steps * w) / VT<V>::N);
}

int main()
{
    float h[4] = {4.352e-9f, 0.04927f, 45000.f, 4.7e-9f}, *theta, *out;
    hipMalloc(&theta, 16); hipMemcpy(theta, h, 16, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    for (int w : {1, 2, 3, 4, 8}) {
        run<0, float>("forward step (56 VALU)", w, theta, out);
        run<1, float>("one-pass step (88 VALU)", w, theta, out);
        if (w <= 2) run<1, v2f>("one-pass step, 2 seq/lane", w, theta, out);
    }
    return 0;
}
