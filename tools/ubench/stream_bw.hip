// HBM streaming microbenchmark for gfx950: what read / write bandwidth do the access shapes of the
// clipper kernels reach?  Rows of B floats, one lane per column (4 B per lane per access, 256 B per
// wave access) vs 16 B per lane (1 KB per wave access); reads only, and reads + 2 write streams.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_bw.hip -o /tmp/stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// grid (B/64, K): wave reads rows [k*L, (k+1)*L) of 3 arrays at column block blockIdx.x, 8 rows in flight
template <int NSTREAM, bool WRITE>
__global__ __launch_bounds__(64) void rows_dword(const float* a, const float* b, const float* c, float* o1, float* o2,
                                                 int64_t B, int64_t L, float* sink)
{
    const int64_t col = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.y * L;
    float acc = 0.f;
    for (int64_t t = t0; t < t0 + L; t += 8) {
        float v[3][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[0][i] = a[(t + i) * B + col];
            if (NSTREAM > 1) v[1][i] = b[(t + i) * B + col];
            if (NSTREAM > 2) v[2][i] = c[(t + i) * B + col];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float s = v[0][i];
            if (NSTREAM > 1) s += v[1][i];
            if (NSTREAM > 2) s += v[2][i];
            acc += s;
            if (WRITE) { o1[(t + i) * B + col] = s; o2[(t + i) * B + col] = acc; }
        }
    }
    if (acc == 12345.678f) *sink = acc;
}

// same bytes, 16 B per lane: arrays viewed as [T/4][B][4]
template <int NSTREAM, bool WRITE>
__global__ __launch_bounds__(64) void rows_x4(const float4* a, const float4* b, const float4* c, float4* o1, float4* o2,
                                              int64_t B, int64_t L4, float* sink)
{
    const int64_t col = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.y * L4;
    float acc = 0.f;
    for (int64_t t = t0; t < t0 + L4; t += 2) {
        float4 v[3][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v[0][i] = a[(t + i) * B + col];
            if (NSTREAM > 1) v[1][i] = b[(t + i) * B + col];
            if (NSTREAM > 2) v[2][i] = c[(t + i) * B + col];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 s = v[0][i];
            if (NSTREAM > 1) { s.x += v[1][i].x; s.y += v[1][i].y; s.z += v[1][i].z; s.w += v[1][i].w; }
            if (NSTREAM > 2) { s.x += v[2][i].x; s.y += v[2][i].y; s.z += v[2][i].z; s.w += v[2][i].w; }
            acc += s.x + s.y + s.z + s.w;
            if (WRITE) { o1[(t + i) * B + col] = s; o2[(t + i) * B + col] = float4{acc, acc, acc, acc}; }
        }
    }
    if (acc == 12345.678f) *sink = acc;
}

template <typename F>
double time_ms(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main()
{
    const int64_t B = 8192, T = 4096, n = B * T;
    float *a, *b, *c, *o1, *o2, *sink;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&o1, n * 4); hipMalloc(&o2, n * 4);
    hipMalloc(&sink, 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(c, 0, n * 4);
    for (int K : {16, 32, 64}) {
        const int64_t L = T / K;
        dim3 g(B / 64, K);
        double ms;
        ms = time_ms([&] { rows_dword<3, false><<<g, 64>>>(a, b, c, o1, o2, B, L, sink); });
        printf("K=%d  3 read streams, 4 B/lane : %.1f us  %.2f TB/s\n", K, ms * 1e3, 3.0 * n * 4 / ms / 1e9);
        ms = time_ms([&] { rows_x4<3, false><<<g, 64>>>((float4*)a, (float4*)b, (float4*)c, (float4*)o1, (float4*)o2, B, L / 4, sink); });
        printf("K=%d  3 read streams, 16 B/lane: %.1f us  %.2f TB/s\n", K, ms * 1e3, 3.0 * n * 4 / ms / 1e9);
        ms = time_ms([&] { rows_dword<1, true><<<g, 64>>>(a, b, c, o1, o2, B, L, sink); });
        printf("K=%d  1 read + 2 write, 4 B/lane : %.1f us  %.2f TB/s\n", K, ms * 1e3, 3.0 * n * 4 / ms / 1e9);
        ms = time_ms([&] { rows_x4<1, true><<<g, 64>>>((float4*)a, (float4*)b, (float4*)c, (float4*)o1, (float4*)o2, B, L / 4, sink); });
        printf("K=%d  1 read + 2 write, 16 B/lane: %.1f us  %.2f TB/s\n", K, ms * 1e3, 3.0 * n * 4 / ms / 1e9);
    }
    return 0;
}
