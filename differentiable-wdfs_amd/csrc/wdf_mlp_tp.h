// wdf_mlp_tp.h -- time-parallel variants of the MLP-root clipper row kernels (wdf_mlp_row.h), gfx950.
//
// The reference's training set is 1340 sequences of 2048 samples (clipper_pot.py:58,232): with one
// 16-lane row per sequence that is 335 waves on a 1024-SIMD chip, each running 2048 dependent steps.
// The time axis is cut into K chunks, grid (ceil(B/4), K): wave (w, k) owns steps [k L, (k+1) L) of its
// four sequences.  L is a multiple of 16 (the row kernels' block of steps).
//
// FORWARD -- speculate and verify, as the diode-pair clipper does (wdf_clipper.h): chunk k starts W
// steps early from z = 0, W per WAVE (wrow[w], a multiple of 16): the warm-up must outlast the RC
// network's memory, which depends on the pot resistance (|1 - 2p| per step, p = Rc/(R+Rc)), and in the
// reference's data set a sequence has one pot value and neighbours in the batch come from the same
// recording (dataimport.py:96, clipper_pot.py:61-80) -- a 10 kOhm wave needs 48 steps where a
// 99.1 kOhm one needs 416.  The state a chunk arrives with and the state it ends with are recorded;
// mlp_tp_verify_kernel compares them across every boundary and raises a per-wave gate where one
// misses by more than tol; the sequential kernel is then launched GATED and re-runs exactly those
// waves (clipper_mlp_row_fwd_kernel, `gate`).  The learned root has no contraction guarantee of its
// own -- the verification is what makes the result the sequential one to tol.
//
// REVERSE -- exact, and parallel over ALL steps.  The adjoint recurrence is
//     g_b2n[n] = gz[n+1] + g[n]/2 ,   gz[n] = kappa[n] g_b2n[n] + g[n]/2 ,   kappa = Da - p (1 + Da)
// where Da = d(-MLP)/da at step n depends on the FORWARD state only (the stash).  So:
//   (A) clipper_mlp_row_kappa_kernel   every step independently: network forward + input Jacobian -> kappa[T][B];
//   (B) mlp_adjoint_scan_kernel        one lane per sequence runs the scalar recurrence (2 FMAs per step)
//                                      and writes g_b2n[T][B];
//   (C) clipper_mlp_row_wgrad_tp_kernel  every step independently again: network forward + layer deltas,
//                                      weight-gradient and {R, C} sums with the now known g_b2n.
// No truncation, no speculation: the same sums as clipper_mlp_row_bwd_w_kernel in another order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wdf_mlp_row.h"

namespace wdf {

struct MlpTpStatus {
    int n_bad;        // (sequence, chunk) pairs whose arrival state missed by more than tol
    float max_miss;   // largest |zwarm - zend| (bit pattern compared as int: values >= 0)
    int gated_waves;  // 4-sequence waves with a miss: re-run from their predecessors' end states (chunk-local repair)
    int sequential_waves;   // ... of those, the waves that still missed and went to the sequential kernel
};

// zwarm, zend: [K][B].  wrow: per-wave warm-up steps (multiple of 16) or nullptr (W for all).
// KAP: the owned steps also evaluate the network's input Jacobian and store kappa[n] = Da - p (1 + Da) [T][B] -- what
// pass (A) of the reverse sweep would recompute from the stash with a second network evaluation per step.
template <int NL, bool DYN_R, bool KAP>
__global__ __launch_bounds__(64) void clipper_mlp_row_fwd_tp_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta2,
    const float* __restrict__ w, int H, float fs, float* __restrict__ y, float* __restrict__ zstash,
    const float* __restrict__ z0, float* __restrict__ zT, float* __restrict__ zwarm, float* __restrict__ zend,
    const int* __restrict__ wrow, MlpTpStatus* __restrict__ status, int64_t B, int64_t T, int64_t L, int64_t W, int64_t L0,
    float* __restrict__ kappa, const float* __restrict__ zinit, const unsigned* __restrict__ gate)
{
    // gate (or null): the chunk-local repair pass -- only the waves the verification flagged run (again)
    if (gate != nullptr && gate[blockIdx.x] == 0u) return;
    // zinit [K][B] (or null: z = 0): the state chunk k > 0 starts its warm-up from -- the previous call's state at the
    // same sample, when the caller trains on the same batch (the verification still decides).
    // L0: length of chunk 0, the only chunk without a warm-up: chunk k > 0 owns [L0 + (k-1) L, L0 + k L).  With L0 = L + W
    // every wave runs about the same number of steps (the host balances it, wdf_capi_mlp.hip); L0 = L: equal chunks.
    if (status && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *status = MlpTpStatus{0, 0.0f, 0, 0};   // the verify kernel adds
    const int lane = threadIdx.x, j = lane & 15;
    const int64_t b_raw = (int64_t)blockIdx.x * 4 + (lane >> 4);
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const int64_t k = blockIdx.y;
    const int64_t t0 = k == 0 ? 0 : L0 + (k - 1) * L;
    const int64_t t1 = k == 0 ? (L0 < T ? L0 : T) : ((t0 + L < T) ? t0 + L : T);
    const int64_t Wq = wrow ? (int64_t)wrow[blockIdx.x] : W;
    const int64_t tw = (k > 0 && t0 > Wq) ? t0 - Wq : 0;
    const MlpClipConsts c = mlp_load_consts(theta2, fs);
    const RowWeights<NL> Wt = row_load_weights<NL>(w, H, j, KAP);
    const float* __restrict__ xp = x + b * T;
    const float* __restrict__ rp = DYN_R ? r + b * T : nullptr;
    float z = tw == 0 ? (z0 ? z0[b] : 0.0f) : (zinit ? zinit[k * B + b] : 0.0f);
    float act[NL];
    for (int64_t tb = tw; tb < t1; tb += 16) {
        if (tb == t0 && j == 0) zwarm[k * B + b] = z;          // the state this chunk arrives with
        const bool owned = tb >= t0;
        float kp = 0.0f;                                        // (KAP) lane i of the row keeps step i's kappa
        const int64_t tj = tb + j < T ? tb + j : T - 1;
        const float xblk = xp[tj];
        const float rblk = DYN_R ? rp[tj] : 1.0f;
        const int n = t1 - tb < 16 ? (int)(t1 - tb) : 16;
        float xs[16], rs[16];
        row_spread(xblk, lane, xs);
        if constexpr (DYN_R) row_spread(rblk, lane, rs);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i >= n) break;
            float p, Rp, lr;
            mlp_step_coeffs<DYN_R>(c, DYN_R ? rs[i] : 1.0f, p, Rp, lr);
            const float b_diff = z - xs[i];
            const float b_temp = -p * b_diff;
            const float a = z + b_temp;
            const float zn = b_temp - row_mlp_fwd<NL>(Wt, a, lr, act);     // b_root = -MLP
            if constexpr (KAP) {
                if (owned) {                                             // (wave-uniform)
                    float da, dlr;
                    row_mlp_grad_in<NL>(Wt, act, da, dlr);
                    const float Da = -da;
                    const float kv = Da - p * (1.0f + Da);
                    kp = (j == i) ? kv : kp;
                }
            }
            if (owned && j == 0) {
                const int64_t o = (tb + i) * B + b;
                if (zstash) zstash[o] = z;
                y[o] = 0.5f * (zn + z);
            }
            z = zn;
        }
        if constexpr (KAP) {
            if (owned && j < n) kappa[(tb + j) * B + b] = kp;
        }
    }
    if (j == 0) {
        zend[k * B + b] = z;
        if (zT && t1 == T) zT[b] = z;
    }
}

// One lane per sequence: every chunk boundary of the sequence; gate[b / 4] = 1 where any of a wave's four
// sequences missed (the gated sequential kernel re-runs that wave), 0 otherwise.
// only (or null): the second verification, after the chunk-local repair -- just the waves flagged there are looked
// at (the others did not run again) and the count goes to status->sequential_waves.
static __global__ __launch_bounds__(64) void mlp_tp_verify_kernel(const float* __restrict__ zwarm, const float* __restrict__ zend,
                                                                  int64_t B, int64_t K, float tol, unsigned* __restrict__ gate,
                                                                  MlpTpStatus* __restrict__ status,
                                                                  const unsigned* __restrict__ only = nullptr)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    float miss = 0.0f;
    int nbad = 0;
    if (only == nullptr || only[b >> 2] != 0u) {
        for (int64_t k = 1; k < K; ++k) {
            const float m = fabsf(zwarm[k * B + b] - zend[(k - 1) * B + b]);
            miss = fmaxf(miss, m);
            nbad += !(m <= tol) ? 1 : 0;
        }
    }
    int any = nbad > 0 ? 1 : 0;
    any |= __shfl_xor(any, 1, 64);
    any |= __shfl_xor(any, 2, 64);
    if ((b_raw & 3) == 0 && b_raw < B) gate[b_raw >> 2] = (unsigned)any;
    float wmax = miss;
    int wbad = b_raw < B ? nbad : 0, wg = ((b_raw & 3) == 0 && b_raw < B) ? any : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        wmax = fmaxf(wmax, __shfl_down(wmax, off, 64));
        wbad += __shfl_down(wbad, off, 64);
        wg += __shfl_down(wg, off, 64);
    }
    if (threadIdx.x == 0) {
        if (only != nullptr) {
            if (wg) atomicAdd(&status->sequential_waves, wg);
        } else {
            if (wmax > 0.0f) atomicMax(reinterpret_cast<int*>(&status->max_miss), __float_as_int(wmax));
            if (wbad) atomicAdd(&status->n_bad, wbad);
            if (wg) atomicAdd(&status->gated_waves, wg);
        }
    }
}

// (A) kappa[n] for every step of chunk (w, k): independent steps, no recurrence.
template <int NL, bool DYN_R>
__global__ __launch_bounds__(64) void clipper_mlp_row_kappa_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta2,
    const float* __restrict__ w, int H, float fs, const float* __restrict__ zstash, float* __restrict__ kappa,
    int64_t B, int64_t T, int64_t L, const unsigned* __restrict__ gate = nullptr)
{
    if (gate != nullptr && gate[blockIdx.x] == 0u) return;      // behind a forward that stored kappa itself: re-run waves only
    const int lane = threadIdx.x, j = lane & 15;
    const int64_t b_raw = (int64_t)blockIdx.x * 4 + (lane >> 4);
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const int64_t t0 = (int64_t)blockIdx.y * L, t1 = (t0 + L < T) ? t0 + L : T;
    const MlpClipConsts c = mlp_load_consts(theta2, fs);
    const RowWeights<NL> Wt = row_load_weights<NL>(w, H, j, true);
    const float* __restrict__ xp = x + b * T;
    const float* __restrict__ rp = DYN_R ? r + b * T : nullptr;
    float act[NL];
    for (int64_t tb = t0; tb < t1; tb += 16) {
        const int n = t1 - tb < 16 ? (int)(t1 - tb) : 16;
        const int64_t tj = j < n ? tb + j : tb;
        const float xblk = xp[tj];
        const float rblk = DYN_R ? rp[tj] : 1.0f;
        const float zblk = zstash[tj * B + b];
        float xs[16], rs[16], zz[16];
        row_spread(xblk, lane, xs);
        row_spread(zblk, lane, zz);
        if constexpr (DYN_R) row_spread(rblk, lane, rs);
        float kp = 0.0f;                                        // lane i of the row keeps step i's kappa
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i >= n) break;
            float p, Rp, lr;
            mlp_step_coeffs<DYN_R>(c, DYN_R ? rs[i] : 1.0f, p, Rp, lr);
            const float z = zz[i];
            const float a = fmaf(-p, z - xs[i], z);
            (void)row_mlp_fwd<NL>(Wt, a, lr, act);
            float da, dlr;
            row_mlp_grad_in<NL>(Wt, act, da, dlr);
            const float Da = -da;                                // b_root = -MLP
            const float kv = Da - p * (1.0f + Da);
            kp = (j == i) ? kv : kp;
        }
        if (j < n) kappa[(tb + j) * B + b] = kp;
    }
}

// (B) the scalar adjoint recurrence, one lane per sequence, last step first.  gb2n may alias kappa.
static __global__ __launch_bounds__(64) void mlp_adjoint_scan_kernel(const float* kappa, const float* __restrict__ gy,
                                                                     float* gb2n, int64_t B, int64_t T)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    float gz = 0.0f;
    int64_t t = T;
    for (; t >= 8; t -= 8) {                                    // 16 loads in flight, then 8 dependent steps
        float kv[8], gv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { kv[i] = kappa[(t - 1 - i) * B + b]; gv[i] = gy[(t - 1 - i) * B + b]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float g2 = fmaf(0.5f, gv[i], gz);
            gb2n[(t - 1 - i) * B + b] = g2;
            gz = fmaf(kv[i], g2, 0.5f * gv[i]);
        }
    }
    for (; t >= 1; --t) {
        const float kv = kappa[(t - 1) * B + b], gv = gy[(t - 1) * B + b];
        const float g2 = fmaf(0.5f, gv, gz);
        gb2n[(t - 1) * B + b] = g2;
        gz = fmaf(kv, g2, 0.5f * gv);
    }
}

// (B) in chunks of kScanChunk steps -- the recurrence is affine in the adjoint that comes in from the future,
//   gz_out = kappa gz_in + (kappa + 1) gy / 2 per step, so a chunk is a map gz_out = A gz_in + C:
//   (B1) every (sequence, chunk) composes its map; (B2) every (sequence, chunk) composes the maps of the chunks
//   after it (at most T / kScanChunk - 1 FMAs) for its incoming adjoint and runs its own steps, writing g_b2n.
// 2048 dependent steps per lane become 128 + 15 + 128; the arrays are read twice (the sweep is latency bound, not
// bandwidth bound).  Same values as the one-lane scan up to the rounding of the incoming adjoint.  gb2n may alias kappa.
constexpr int kScanChunk = 128;

static __global__ __launch_bounds__(64) void mlp_adjoint_chunk_map_kernel(const float* __restrict__ kappa,
                                                                          const float* __restrict__ gy, float2* __restrict__ map,
                                                                          int64_t B, int64_t T)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const int64_t t0 = (int64_t)blockIdx.y * kScanChunk;
    int64_t t = (t0 + kScanChunk < T) ? t0 + kScanChunk : T;
    float A = 1.0f, Cc = 0.0f;
    for (; t >= t0 + 8; t -= 8) {
        float kv[8], gv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { kv[i] = kappa[(t - 1 - i) * B + b]; gv[i] = gy[(t - 1 - i) * B + b]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            Cc = fmaf(kv[i], Cc, 0.5f * (kv[i] + 1.0f) * gv[i]);
            A *= kv[i];
        }
    }
    for (; t > t0; --t) {
        const float kv = kappa[(t - 1) * B + b], gv = gy[(t - 1) * B + b];
        Cc = fmaf(kv, Cc, 0.5f * (kv + 1.0f) * gv);
        A *= kv;
    }
    if (b_raw < B) map[(int64_t)blockIdx.y * B + b] = make_float2(A, Cc);
}

static __global__ __launch_bounds__(64) void mlp_adjoint_scan_chunked_kernel(const float* kappa, const float* __restrict__ gy,
                                                                             const float2* __restrict__ map, float* gb2n,
                                                                             int64_t B, int64_t T)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const int64_t t0 = (int64_t)blockIdx.y * kScanChunk;
    int64_t t = (t0 + kScanChunk < T) ? t0 + kScanChunk : T;
    float gz = 0.0f;                                            // the adjoint that arrives from the chunks after this one
    for (int kk = (int)gridDim.y - 1; kk > (int)blockIdx.y; --kk) {
        const float2 m = map[(int64_t)kk * B + b];
        gz = fmaf(m.x, gz, m.y);
    }
    for (; t >= t0 + 8; t -= 8) {
        float kv[8], gv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { kv[i] = kappa[(t - 1 - i) * B + b]; gv[i] = gy[(t - 1 - i) * B + b]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float g2 = fmaf(0.5f, gv[i], gz);
            gb2n[(t - 1 - i) * B + b] = g2;
            gz = fmaf(kv[i], g2, 0.5f * gv[i]);
        }
    }
    for (; t > t0; --t) {
        const float kv = kappa[(t - 1) * B + b], gv = gy[(t - 1) * B + b];
        const float g2 = fmaf(0.5f, gv, gz);
        gb2n[(t - 1) * B + b] = g2;
        gz = fmaf(kv, g2, 0.5f * gv);
    }
}

// (C) weight-gradient and {R, C} sums of chunk (w, k) with g_b2n known.  Outputs per wave
// (index blockIdx.y * gridDim.x + blockIdx.x), the layout of clipper_mlp_row_bwd_w_kernel's.
template <int NL, bool DYN_R>
__global__ __launch_bounds__(64) void clipper_mlp_row_wgrad_tp_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta2,
    const float* __restrict__ w, int H, float fs, const float* __restrict__ zstash, const float* __restrict__ gb2n,
    float* __restrict__ wsw, double* __restrict__ ws, int64_t B, int64_t T, int64_t L)
{
    const int lane = threadIdx.x, j = lane & 15;
    const int64_t b_raw = (int64_t)blockIdx.x * 4 + (lane >> 4);
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const int64_t t0 = (int64_t)blockIdx.y * L, t1 = (t0 + L < T) ? t0 + L : T;
    const int64_t part = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const MlpClipConsts c = mlp_load_consts(theta2, fs);
    const RowWeights<NL> Wt = row_load_weights<NL>(w, H, j, true);
    const float* __restrict__ xp = x + b * T;
    const float* __restrict__ rp = DYN_R ? r + b * T : nullptr;
    RowGrads<NL> acc;
    acc.k0a = acc.k0l = acc.b0 = acc.wo = acc.bo = 0.0f;
#pragma unroll
    for (int l = 0; l < NL - 1; ++l) {
        acc.bias[l] = 0.0f;
        acc.mid[l] = mfma_v4f{0.0f, 0.0f, 0.0f, 0.0f};
    }
    double dLr = 0.0, dP = 0.0;
    float act[NL];
    for (int64_t tb = t0; tb < t1; tb += 16) {
        const int n = t1 - tb < 16 ? (int)(t1 - tb) : 16;
        const int64_t tj = j < n ? tb + j : tb;
        const float xblk = xp[tj];
        const float rblk = DYN_R ? rp[tj] : 1.0f;
        const float zblk = zstash[tj * B + b];
        const float gblk = (live && j < n) ? gb2n[tj * B + b] : 0.0f;   // shadow rows / steps past the end add nothing
        float xs[16], rs[16], zz[16], gs[16];
        row_spread(xblk, lane, xs);
        row_spread(zblk, lane, zz);
        row_spread(gblk, lane, gs);
        if constexpr (DYN_R) row_spread(rblk, lane, rs);
        float sLr = 0.0f, sP = 0.0f;                            // fp32 within a block, fp64 across blocks
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i >= n) break;
            float p, Rp, lr;
            mlp_step_coeffs<DYN_R>(c, DYN_R ? rs[i] : 1.0f, p, Rp, lr);
            const float z = zz[i], g_b2n = gs[i];
            const float b_diff = z - xs[i];
            const float a = fmaf(-p, b_diff, z);
            (void)row_mlp_fwd<NL>(Wt, a, lr, act);
            float da, dlr;
            row_mlp_grad_all<NL>(Wt, act, a, lr, -g_b2n, acc, da, dlr);
            const float g_a = -g_b2n * da;                       // b_root = -MLP
            const float g_lr = -g_b2n * dlr;
            const float g_p = -(g_b2n + g_a) * b_diff;
            if constexpr (DYN_R) {
                sP = fmaf(Rp, fmaf(g_p, p, g_lr), sP);
            } else {
                sP += g_p;
                sLr += g_lr;
            }
        }
        dLr += (double)sLr;
        dP += (double)sP;
    }
    if (!live || j != 0) { dLr = dP = 0.0; }
    dLr = wave_sum(dLr); dP = wave_sum(dP);
    if (threadIdx.x == 0) {
        double* o = ws + part * 4;
        o[0] = dLr; o[1] = 0.0; o[2] = dP; o[3] = 0.0;
    }
    const int count = 3 * H + (NL - 1) * (H * H + H) + H + 1;
    float* __restrict__ o = wsw + part * count;
    const bool writer = lane < 16;
    const float vk0a = rows_sum(acc.k0a), vk0l = rows_sum(acc.k0l), vb0 = rows_sum(acc.b0), vwo = rows_sum(acc.wo),
                vbo = rows_sum(acc.bo);
    if (writer && j < H) {
        o[j] = vk0a; o[H + j] = vk0l; o[2 * H + j] = vb0;
        o[3 * H + (NL - 1) * (H * H + H) + j] = vwo;
    }
    if (lane == 0) o[count - 1] = vbo;
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        const float vb = rows_sum(acc.bias[l - 1]);
        if (writer && j < H) o[3 * H + (l - 1) * (H * H + H) + H * H + j] = vb;
    }
    row_store_mid<NL>(acc, o, H, lane);
}

// Fixed-order sum of many per-wave weight-gradient partials [nblk][count]: block (64, 16) -- thread (i, s)
// adds partials s, s + 16, ... of weight i in double, the 16 slices are then added in order.
static __global__ __launch_bounds__(1024) void mlp_wgrad_reduce_wide_kernel(const float* __restrict__ ws, int nblk, int count,
                                                                            float* __restrict__ gw)
{
    __shared__ double sh[16][64];
    const int i = blockIdx.x * 64 + threadIdx.x, sl = threadIdx.y;
    double acc = 0.0;
    if (i < count) {
        // eight loads in flight per thread (the adds were one dependent chain of nblk / 16 strided loads: 37 us for
        // 2010 partials); still a fixed order
        double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int b = sl;
        for (; b + 7 * 16 < nblk; b += 8 * 16) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ws[(int64_t)(b + 16 * q) * count + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] += (double)v[q];
        }
        for (; b < nblk; b += 16) a[0] += (double)ws[(int64_t)b * count + i];
        acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    sh[sl][threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0 && i < count) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sh[q][threadIdx.x];
        gw[i] = (float)t;
    }
}

}  // namespace wdf
