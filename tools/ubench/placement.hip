// Where the dispatcher puts the waves of a SMALL grid (fewer waves than the chip has SIMD slots): every wave notes its
// XCC / SE / CU / SIMD (HW_ID, XCC_ID) and its start time, then spins ~30 us.  Grids shaped like the time-parallel forward's
// (tiles x chunks of single-wave workgroups), four-wave workgroups (block (64, 4)), and the same with enough LDS asked for that
// a CU holds only one workgroup.  Prints waves per SIMD and per CU, and how late the last wave started.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/placement.hip -o tools/ubench/bin/placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>

__global__ void k(unsigned long long* out, int spin)
{
    extern __shared__ float lds[];
    const unsigned long long t0 = wall_clock64();
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
    float a = threadIdx.x * 0.001f;
    for (int i = 0; i < spin; ++i) a = __builtin_fmaf(a, 0.999f, 0.001f);
    if (a == 123.456f) lds[threadIdx.x] = a;
    const size_t wave = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.y + threadIdx.y;
    if (threadIdx.x == 0) { out[3 * wave] = t0; out[3 * wave + 1] = wall_clock64(); out[3 * wave + 2] = hw | ((unsigned long long)xcc << 32); }
}

void run(const char* name, dim3 grid, dim3 block, size_t lds, unsigned long long* dout)
{
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const size_t waves = (size_t)grid.x * grid.y * block.y;
    std::vector<unsigned long long> h(3 * waves);
    k<<<grid, block, lds>>>(dout, 20000); hipDeviceSynchronize();
    k<<<grid, block, lds>>>(dout, 20000); hipDeviceSynchronize();
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, int> simd, cu;
    unsigned long long tmin = ~0ull, tlast = 0, tend = 0;
    for (size_t w = 0; w < waves; ++w) {
        const unsigned long long hw = h[3 * w + 2];
        const unsigned long long s = (hw >> 4) & 3, c = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, x = (hw >> 32) & 15;
        const unsigned long long cuk = ((x * 8 + se) * 2 + sh) * 16 + c;
        simd[cuk * 4 + s]++; cu[cuk]++;
        tmin = std::min(tmin, h[3 * w]); tlast = std::max(tlast, h[3 * w]); tend = std::max(tend, h[3 * w + 1]);
    }
    std::map<int, int> hs, hc;
    for (auto& p : simd) hs[p.second]++;
    for (auto& p : cu) hc[p.second]++;
    printf("%-44s %5zu waves: SIMDs used %4zu, waves per SIMD {", name, waves, simd.size());
    for (auto& p : hs) printf(" %d:%d", p.first, p.second);
    printf(" }; CUs used %3zu, waves per CU {", cu.size());
    for (auto& p : hc) printf(" %d:%d", p.first, p.second);
    printf(" }; last start +%.1f us, all done +%.1f us\n", (tlast - tmin) * 0.01, (tend - tmin) * 0.01);
}

int main()
{
    unsigned long long* d; hipMalloc(&d, 64 << 20);
    for (int K : {16, 32, 64, 128}) { char n[64]; snprintf(n, 64, "grid (16, %d) x 64", K); run(n, dim3(16, K), dim3(64), 0, d); }
    for (int K : {16, 32, 64, 128}) { char n[64]; snprintf(n, 64, "grid (16, %d) x (64, 4)", K / 4); run(n, dim3(16, K / 4), dim3(64, 4), 0, d); }
    for (int K : {16, 32, 64, 128}) { char n[64]; snprintf(n, 64, "grid (16, %d) x (64, 4), 96 KB LDS", K / 4); run(n, dim3(16, K / 4), dim3(64, 4), 96 * 1024, d); }
    for (int K : {64, 128}) { char n[64]; snprintf(n, 64, "grid (16, %d) x 64, 36 KB LDS", K); run(n, dim3(16, K), dim3(64), 36 * 1024, d); }
    return 0;
}
