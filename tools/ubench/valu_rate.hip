// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for plain fp32 FMA,
// packed fp32 FMA and transcendentals, at 1/2/4 waves per SIMD, dependent vs independent chains.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate
// NOTE on reading the "fma ILP=4" rows: LLVM's SLP vectoriser turns the four independent scalar
// FMAs into v_pk_fma_f32 pairs (32 v_pk_fma per 64 source FMAs), so "ns per inst" there is per
// SOURCE operation; one issued VALU instruction (plain or packed) costs ~2 ns per SIMD on these
// boxes, a dependent one ~3.3 ns, a transcendental ~3.4 ns.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE, int ILP>
__global__ __launch_bounds__(64) void k(float* out, int iters, float seed)
{
    float a[ILP]; v2f p[ILP];
    for (int i = 0; i < ILP; ++i) { a[i] = seed + i; p[i] = v2f{seed + i, seed - i}; }
    const float c1 = 1.0000001f, c2 = 1e-9f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], c1, c2);                       // v_fma_f32
                if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], v2f{c1, c1}, v2f{c2, c2});   // v_pk_fma_f32
                if (MODE == 2) a[i] = __builtin_amdgcn_exp2f(a[i]);                       // v_exp_f32
                if (MODE == 3) a[i] = __builtin_amdgcn_rcpf(a[i]);                        // v_rcp_f32
                if (MODE == 4) a[i] = a[i] > c1 ? a[i] * c1 : a[i] + c2;                   // cmp + cndmask + mul/add mix
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < ILP; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}

template <int MODE, int ILP>
void run(const char* name, int waves_per_simd)
{
    float* d; hipMalloc(&d, (size_t)(1 << 22) * 4 + 64);
    const int iters = 2000;
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, ILP><<<grid, 64>>>(d, iters, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, ILP><<<grid, 64>>>(d, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, (char*)d + (size_t)(1 << 20) * 8, 8, hipMemcpyDeviceToHost);
    const double n_inst = (double)iters * 16 * ILP;            // per wave
    printf("%-10s ILP=%d waves/SIMD=%d : %.2f clk/inst/wave (s_memtime), wall %.3f ms -> %.2f ns per inst per SIMD\n", name, ILP,
           waves_per_simd, (double)cyc / n_inst, ms, ms * 1e6 / (n_inst * waves_per_simd));
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0, 1>("fma", 1); run<0, 4>("fma", 1); run<1, 1>("pk_fma", 1); run<1, 4>("pk_fma", 1); run<2, 1>("exp2", 1); run<2, 4>("exp2", 1); run<3, 4>("rcp", 1); run<4, 4>("cmpsel", 1); }
        if (w == 2) { run<0, 1>("fma", 2); run<0, 4>("fma", 2); run<1, 4>("pk_fma", 2); run<2, 4>("exp2", 2); }
        if (w == 4) { run<0, 1>("fma", 4); run<0, 4>("fma", 4); run<1, 4>("pk_fma", 4); run<2, 4>("exp2", 4); run<4, 4>("cmpsel", 4); }
    }
    return 0;
}
