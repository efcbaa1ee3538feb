// HBM streaming by data LAYOUT for the time-parallel kernels' shapes (gfx950), with the inputs
// rotated through more buffers than the 256 MiB Infinity Cache holds so that every pass really
// reads HBM (a single 134 MB input re-read in a loop is served from the cache and flatters every
// variant).  Forward shape: 1 read stream + 2 write streams; reverse shape: 3 read streams.
//   rows     [T][B]: a wave touches 256 B of a row, rows 32 KB apart          (what the kernels do today)
//   tiles    [T/32][B/64][32][64]: a wave's 32-step tile is 8 KB contiguous
//   tiles16  [T/32][B/64][8][64][4]: the same tile, 16 B per lane (4 consecutive steps)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_layout.hip -o tools/ubench/bin/stream_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

enum { ROWS = 0, TILES = 1, TILES16 = 2 };

template <int LAYOUT>
__device__ __forceinline__ int64_t elem(int64_t B, int64_t tile, int64_t t, int i)   // float index of (row t + i, lane)
{
    if (LAYOUT == ROWS) return (t + i) * B + tile * 64 + threadIdx.x;
    return ((t / 32) * (B / 64) + tile) * 2048 + i * 64 + threadIdx.x;
}

// NR read streams (a[0..NR)), NW write streams; 32-row tiles, next tile's loads issued mid-tile
template <int LAYOUT, int NR, int NW, bool NT>
__global__ __launch_bounds__(64) void stream(const float* __restrict__ a0, const float* __restrict__ a1,
                                             const float* __restrict__ a2, float* __restrict__ o0,
                                             float* __restrict__ o1, int64_t B, int64_t L, float* sink)
{
    const int64_t tile = blockIdx.x, t0 = (int64_t)blockIdx.y * L;
    const float* __restrict__ a[3] = {a0, a1, a2};
    float* __restrict__ o[2] = {o0, o1};
    float acc = 0.f;
    if (LAYOUT == TILES16) {
        v4f n[NR][8], v[NR][8];
#pragma unroll
        for (int s = 0; s < NR; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                n[s][i] = *reinterpret_cast<const v4f*>(a[s] + ((t0 / 32) * (B / 64) + tile) * 2048 + (i * 64 + threadIdx.x) * 4);
        for (int64_t t = t0; t < t0 + L; t += 32) {
#pragma unroll
            for (int s = 0; s < NR; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[s][i] = n[s][i];
            if (t + 32 < t0 + L) {
#pragma unroll
                for (int s = 0; s < NR; ++s)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        n[s][i] = *reinterpret_cast<const v4f*>(a[s] + (((t + 32) / 32) * (B / 64) + tile) * 2048 + (i * 64 + threadIdx.x) * 4);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v4f s4 = v[0][i];
#pragma unroll
                for (int s = 1; s < NR; ++s) s4 += v[s][i];
                acc = acc * 0.5f + s4.x + s4.y + s4.z + s4.w;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    v4f* q = reinterpret_cast<v4f*>(o[w] + ((t / 32) * (B / 64) + tile) * 2048 + (i * 64 + threadIdx.x) * 4);
                    const v4f val = s4 + (float)w * acc;
                    if (NT) __builtin_nontemporal_store(val, q); else *q = val;
                }
            }
        }
    } else {
        float n[NR][32], v[NR][32];
#pragma unroll
        for (int s = 0; s < NR; ++s)
#pragma unroll
            for (int i = 0; i < 32; ++i) n[s][i] = a[s][elem<LAYOUT>(B, tile, t0, i)];
        for (int64_t t = t0; t < t0 + L; t += 32) {
#pragma unroll
            for (int s = 0; s < NR; ++s)
#pragma unroll
                for (int i = 0; i < 32; ++i) v[s][i] = n[s][i];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (i == 16 && t + 32 < t0 + L) {
#pragma unroll
                    for (int s = 0; s < NR; ++s)
#pragma unroll
                        for (int ii = 0; ii < 32; ++ii) n[s][ii] = a[s][elem<LAYOUT>(B, tile, t + 32, ii)];
                }
                float sv = v[0][i];
#pragma unroll
                for (int s = 1; s < NR; ++s) sv += v[s][i];
                acc = acc * 0.5f + sv;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    float* q = o[w] + elem<LAYOUT>(B, tile, t, i);
                    const float val = w ? acc : sv;
                    if (NT) __builtin_nontemporal_store(val, q); else *q = val;
                }
            }
        }
    }
    if (acc == 12345.678f) *sink = acc;
}

int main()
{
    const int64_t B = 8192, T = 4096, n = B * T;
    const int NBUF = 9;                                   // 9 x 134 MB inputs: three passes never meet in the cache
    std::vector<float*> in(NBUF);
    for (auto& p : in) { hipMalloc(&p, n * 4); hipMemset(p, 0, n * 4); }
    float *o0, *o1, *sink;
    hipMalloc(&o0, n * 4); hipMalloc(&o1, n * 4); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int rot = 0;
    auto run = [&](const char* label, int K, double mb, auto launch) {
        float best = 1e9f, sum = 0.f;
        const int reps = 10;
        for (int i = 0; i < reps + 2; ++i) {
            const float* a0 = in[rot % NBUF]; const float* a1 = in[(rot + 1) % NBUF]; const float* a2 = in[(rot + 2) % NBUF];
            rot += 3;
            hipEventRecord(e0);
            launch(a0, a1, a2);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("K=%-2d %-44s best %6.1f us (%5.2f TB/s)  mean %6.1f us (%5.2f TB/s)\n", K, label, best * 1e3, mb / best / 1e3,
               sum / reps * 1e3, mb / (sum / reps) / 1e3);
    };
    const double MB = n * 4 / 1e6;
    for (int K : {16, 32}) {
        const int64_t L = T / K;
        dim3 g(B / 64, K);
#define L3(...) [&](const float* a0, const float* a1, const float* a2) { __VA_ARGS__; }
        run("read 1 stream rows", K, MB, L3((stream<ROWS, 1, 0, true><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
        run("read 1 stream tiles", K, MB, L3((stream<TILES, 1, 0, true><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
        run("read 1 stream tiles16", K, MB, L3((stream<TILES16, 1, 0, true><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
        run("fwd 1R+2W rows plain stores", K, 3 * MB, L3((stream<ROWS, 1, 2, false><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
        run("fwd 1R+2W rows nt", K, 3 * MB, L3((stream<ROWS, 1, 2, true><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
        run("fwd 1R+2W tiles nt", K, 3 * MB, L3((stream<TILES, 1, 2, true><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
        run("fwd 1R+2W tiles16 nt", K, 3 * MB, L3((stream<TILES16, 1, 2, true><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
        run("fwd 1R+2W tiles16 plain", K, 3 * MB, L3((stream<TILES16, 1, 2, false><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
        run("fwd 1R+1W rows nt (no y)", K, 2 * MB, L3((stream<ROWS, 1, 1, true><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
        run("fwd 1R+1W tiles16 nt (no y)", K, 2 * MB, L3((stream<TILES16, 1, 1, true><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
        run("bwd 3R rows", K, 3 * MB, L3((stream<ROWS, 3, 0, true><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
        run("bwd 3R tiles", K, 3 * MB, L3((stream<TILES, 3, 0, true><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
        run("bwd 3R tiles16", K, 3 * MB, L3((stream<TILES16, 3, 0, true><<<g, 64>>>(a0, a1, a2, o0, o1, B, L, sink))));
#undef L3
    }
    // a plain float4 copy of the same bytes for reference (read 134 MB, write 134 MB)
    return 0;
}
