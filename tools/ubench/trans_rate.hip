// Is the transcendental pipe (v_exp_f32 / v_log_f32 / v_rcp_f32: quarter rate) a SIMD's own, or shared by the four SIMDs of a
// CU?  Waves of nothing but independent v_exp_f32 (or, as the control, v_fma_f32), launched so that a CU holds 1, 2, 4, 8, 16
// of them: workgroups of 64 threads (the dispatcher spreads them over CUs) and of 256 (four waves, one per SIMD of one CU).
// Prints wave-instructions per microsecond per CU.  If four waves on four SIMDs of one CU get four times one wave's rate the
// pipe is per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/trans_rate.hip -o tools/ubench/bin/trans_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void k(float* out, int iters)
{
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.001f * (threadIdx.x + 1) + 0.01f * j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (OP == 0) a[j] = __builtin_amdgcn_exp2f(a[j]) * 0.5f - 0.4f;       // 1 transcendental + 1 fma per element
                else if (OP == 1) a[j] = __builtin_fmaf(a[j], 0.999f, 0.001f) * 0.5f - 0.4f;   // control: 2 fma
                else a[j] = __builtin_amdgcn_exp2f(a[j]);                                 // transcendental only (dependent chain x 8)
            }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int grid, int block, float* out, int per_iter_ops)
{
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, block>>>(out, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<grid, block>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * block / 64.0, ops = waves * iters * 32.0 * per_iter_ops;
    printf("%-28s grid %5d x %3d (%6.0f waves, %5.2f per CU): %8.3f ms  %8.1f wave-instr/us/CU  (%6.2f cycles per instr per CU at 2.4 GHz)\n", name, grid,
           block, waves, waves / 256.0, ms, ops / (ms * 1e3) / 256.0, 2400.0 / (ops / (ms * 1e3) / 256.0));
}

int main()
{
    float* out; hipMalloc(&out, 64 << 20);
    for (int rep = 0; rep < 1; ++rep) {
        for (int g : {256, 512, 1024, 2048, 4096}) run<2>("exp2 only, 64-thread WGs", g, 64, out, 1);
        for (int g : {64, 128, 256, 512, 1024}) run<2>("exp2 only, 256-thread WGs", g, 256, out, 1);
        for (int g : {256, 512, 1024, 2048, 4096}) run<1>("fma only, 64-thread WGs", g, 64, out, 2);
        for (int g : {64, 256, 1024}) run<1>("fma only, 256-thread WGs", g, 256, out, 2);
        for (int g : {256, 1024, 4096}) run<0>("exp2 + fma, 64-thread WGs", g, 64, out, 2);
        for (int g : {64, 256, 1024}) run<0>("exp2 + fma, 256-thread WGs", g, 256, out, 2);
    }
    return 0;
}
