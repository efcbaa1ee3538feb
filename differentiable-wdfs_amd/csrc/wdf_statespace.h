// wdf_statespace.h -- generic WDF tree + one root, lowered to a state-space recursion.
//
// Every adaptor / one-port of wdf_py/lib/tf_wdf.py (Series :129-155, Parallel :158-192,
// Inverter :195-214, Resistor :62-88, Capacitor :91-126, ResistiveVoltageSource :31-59) is
// LINEAR in the waves; the only nonlinearity sits at the root.  So one time step of any tree
// (lpf.py:39-46, voltage_divider.py:35-42, clipper_pot.py:113-124) is exactly
//
//     a  = ca . z + da . x                  wave sent up into the root   (reflected() sweep)
//     b  = root(a)                          IdealVoltageSource :23-28 / diode pair / MLP
//     z' = A z + Bx x + E b                 new capacitor states         (incident() sweep)
//     y  = cy . z + dy . x + fy b           voltage() of the probed element :8-10
//
// with z the capacitor states (Capacitor.z), x the source voltages of this sample, and the
// small matrices functions of the component values only.  The Python host derives them by
// running its tf_wdf-compatible elements once on unit vectors (differentiably, so dL/dR and
// dL/dC chain through them); this kernel runs the recursion for B sequences x T samples with
// one lane per sequence, the matrices in SGPRs and the states in VGPRs.  An ideal-source
// root (b = -a + 2 Vs) is folded into the matrices by the host: root kind NONE.
//
// Layouts: x [B][T][NI] (the reference's input[:, i, c]), y [T][B], state stash [T][NS][B],
// z0 / zT [NS][B].  coef (device, fp32):
//     A[NS][NS] | Bx[NS][NI] | E[NS] | ca[NS] | da[NI] | cy[NS] | dy[NI] | fy
// rootp (device, fp32) for the diode pair: {Is, nVt, R_port}.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wdf_omega.h"

namespace wdf {

enum { kRootNone = 0, kRootDiode = 2 };

template <int NS, int NI>
struct SSCoef {
    static constexpr int kN = NS * NS + NS * NI + NS + NS + NI + NS + NI + 1;
    static constexpr int oA = 0, oB = oA + NS * NS, oE = oB + NS * NI, oCa = oE + NS, oDa = oCa + NS,
                         oCy = oDa + NI, oDy = oCy + NS, oFy = oDy + NI;
    float v[kN];
    __device__ __forceinline__ void load(const float* __restrict__ p)
    {
#pragma unroll
        for (int i = 0; i < kN; ++i) v[i] = p[i];
    }
};

struct SSDiode {
    float L, V, Is, Rport;
    DiodeStatic d;
    __device__ __forceinline__ void load(const float* __restrict__ rp, int n_up, int n_down)
    {
        Is = rp[0]; V = rp[1]; Rport = rp[2];
        L = logf(Rport * Is / V);
        d = make_diode_static(V, n_up, n_down);
    }
};

// loads kBlkSS steps x NI channels of lane b's row into v[step][chan]
constexpr int kBlkSS = 8;

template <int NI, bool VEC4>
__device__ __forceinline__ void ss_load_block(const float* __restrict__ x, int64_t b, int64_t T, int64_t t0,
                                              float (&v)[kBlkSS][NI])
{
    const float* p = x + (b * T + t0) * NI;
    if constexpr (VEC4) {
        const float4* q = reinterpret_cast<const float4*>(p);
        float tmp[kBlkSS * NI];
#pragma unroll
        for (int i = 0; i < kBlkSS * NI / 4; ++i) {
            const float4 f = q[i];
            tmp[4 * i] = f.x; tmp[4 * i + 1] = f.y; tmp[4 * i + 2] = f.z; tmp[4 * i + 3] = f.w;
        }
#pragma unroll
        for (int k = 0; k < kBlkSS; ++k)
#pragma unroll
            for (int c = 0; c < NI; ++c) v[k][c] = tmp[k * NI + c];
    } else {
#pragma unroll
        for (int k = 0; k < kBlkSS; ++k)
#pragma unroll
            for (int c = 0; c < NI; ++c) v[k][c] = p[k * NI + c];
    }
}

template <int NS, int NI, int ROOT, bool SYM>
__device__ __forceinline__ float ss_fwd_step(const SSCoef<NS, NI>& c, const SSDiode& dp, const float (&x)[NI],
                                             float (&z)[NS > 0 ? NS : 1])
{
    using C = SSCoef<NS, NI>;
    float b = 0.0f;
    if constexpr (ROOT == kRootDiode) {
        float a = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; ++s) a = fmaf(c.v[C::oCa + s], z[s], a);
#pragma unroll
        for (int i = 0; i < NI; ++i) a = fmaf(c.v[C::oDa + i], x[i], a);
        b = diode_pair<SYM>(a, dp.L, dp.d).b;
    }
    float y = c.v[C::oFy] * b;
#pragma unroll
    for (int s = 0; s < NS; ++s) y = fmaf(c.v[C::oCy + s], z[s], y);
#pragma unroll
    for (int i = 0; i < NI; ++i) y = fmaf(c.v[C::oDy + i], x[i], y);
    float zn[NS > 0 ? NS : 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float acc = c.v[C::oE + s] * b;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) acc = fmaf(c.v[C::oA + s * NS + s2], z[s2], acc);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc = fmaf(c.v[C::oB + s * NI + i], x[i], acc);
        zn[s] = acc;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) z[s] = zn[s];
    return y;
}

template <int NS, int NI, int ROOT, bool SYM, bool VEC4>
__global__ __launch_bounds__(64) void ss_fwd_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                    const float* __restrict__ rootp, int n_up, int n_down,
                                                    float* __restrict__ y, float* __restrict__ zstash,
                                                    const float* __restrict__ z0, float* __restrict__ zT,
                                                    int64_t B, int64_t T, const unsigned* __restrict__ gate = nullptr)
{
    constexpr int NSa = NS > 0 ? NS : 1;
    if (gate != nullptr && gate[blockIdx.x] == 0u) return;      // (behind the time-parallel forward: flagged waves only)
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;     // dead lanes shadow the last sequence
    SSCoef<NS, NI> c;
    c.load(coef);
    SSDiode dp = {};
    if constexpr (ROOT == kRootDiode) dp.load(rootp, n_up, n_down);
    float z[NSa];
#pragma unroll
    for (int s = 0; s < NS; ++s) z[s] = z0 ? z0[s * B + b] : 0.0f;
    float* __restrict__ yp = y + b;
    float* __restrict__ zp = zstash + b;
    const bool STASH = (zstash != nullptr) && NS > 0;       // wave-uniform

    const int64_t nfull = T / kBlkSS;
    float xc[kBlkSS][NI], xn[kBlkSS][NI];
    if (nfull > 0) ss_load_block<NI, VEC4>(x, b, T, 0, xn);
    for (int64_t blk = 0; blk < nfull; ++blk) {
#pragma unroll
        for (int k = 0; k < kBlkSS; ++k)
#pragma unroll
            for (int i = 0; i < NI; ++i) xc[k][i] = xn[k][i];
        if (blk + 1 < nfull) ss_load_block<NI, VEC4>(x, b, T, (blk + 1) * kBlkSS, xn);
#pragma unroll
        for (int k = 0; k < kBlkSS; ++k) {
            if (STASH) {
#pragma unroll
                for (int s = 0; s < NS; ++s) zp[s * B] = z[s];
                zp += (int64_t)NS * B;
            }
            *yp = ss_fwd_step<NS, NI, ROOT, SYM>(c, dp, xc[k], z);
            yp += B;
        }
    }
    for (int64_t t = nfull * kBlkSS; t < T; ++t) {
        float xt[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) xt[i] = x[(b * T + t) * NI + i];
        if (STASH) {
#pragma unroll
            for (int s = 0; s < NS; ++s) zp[s * B] = z[s];
            zp += (int64_t)NS * B;
        }
        *yp = ss_fwd_step<NS, NI, ROOT, SYM>(c, dp, xt, z);
        yp += B;
    }
    if (zT) {
#pragma unroll
        for (int s = 0; s < NS; ++s) zT[s * B + b] = z[s];
    }
}

// ---- exact time-parallel forward for LINEAR trees (root kind NONE) ------------------------------
// With an ideal-source root folded in, a step is z' = A z + Bx x, y = cy . z + dy . x: the state at a
// chunk boundary is an affine function of the state at the previous one,
//     z(t0 + L) = A^L z(t0) + (response of the chunk's own inputs from a zero state),
// so no speculation is needed (lpf.py:30-49 and voltage_divider.py:27-46 are such trees):
//   1. ss_lin_zero_state_kernel  every chunk runs its inputs from z = 0 and keeps only the end state;
//   2. ss_lin_starts_kernel      one lane per sequence walks the K chunks, z_start[k+1] = A^L z_start[k] + end0[k]
//                                (A^L by repeated squaring in double, once per lane);
//   3. ss_lin_chunk_kernel       every chunk runs again from its exact start state and writes y and the stash.
// Twice the arithmetic of the sequential kernel, K times the parallelism; same result up to fp32 rounding.
// n steps x NI channels of lane b's row, N at a time (N = 32, one channel: a whole 128-byte line of the row per burst)
template <int NI, bool VEC4, int N>
__device__ __forceinline__ void ss_load_burst(const float* __restrict__ x, int64_t b, int64_t T, int64_t t0, float (&v)[N][NI])
{
    const float* p = x + (b * T + t0) * NI;
    if constexpr (VEC4) {
        const float4* q = reinterpret_cast<const float4*>(p);
        float tmp[N * NI];
#pragma unroll
        for (int i = 0; i < N * NI / 4; ++i) {
            const float4 f = q[i];
            tmp[4 * i] = f.x; tmp[4 * i + 1] = f.y; tmp[4 * i + 2] = f.z; tmp[4 * i + 3] = f.w;
        }
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int c = 0; c < NI; ++c) v[k][c] = tmp[k * NI + c];
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int c = 0; c < NI; ++c) v[k][c] = p[k * NI + c];
    }
}

constexpr int kLinBurst = 32;        // steps per burst of the chunked linear forward (chunk starts are multiples of it)

template <int NS, int NI, bool VEC4>
__global__ __launch_bounds__(64) void ss_lin_zero_state_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                               float* __restrict__ zend0, int64_t B, int64_t T, int64_t L)
{
    constexpr int NSa = NS > 0 ? NS : 1;
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const int64_t k = blockIdx.y, t0 = k * L, t1 = (t0 + L < T) ? t0 + L : T;
    if (t1 == T) return;                                      // (nothing comes after the last chunk)
    SSCoef<NS, NI> c;
    c.load(coef);
    const SSDiode dp = {};
    float z[NSa];
#pragma unroll
    for (int s = 0; s < NSa; ++s) z[s] = 0.0f;
    // (a chunk that is not the last is L steps, L a multiple of the burst) bursts of 32 steps, the next one in flight
    float xc[kLinBurst][NI], xn[kLinBurst][NI];
    ss_load_burst<NI, VEC4, kLinBurst>(x, b, T, t0, xn);
    for (int64_t t = t0; t < t1; t += kLinBurst) {
#pragma unroll
        for (int q = 0; q < kLinBurst; ++q)
#pragma unroll
            for (int i = 0; i < NI; ++i) xc[q][i] = xn[q][i];
        if (t + kLinBurst < t1) ss_load_burst<NI, VEC4, kLinBurst>(x, b, T, t + kLinBurst, xn);
#pragma unroll
        for (int q = 0; q < kLinBurst; ++q) (void)ss_fwd_step<NS, NI, kRootNone, true>(c, dp, xc[q], z);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) zend0[(k * NS + s) * B + b] = z[s];
}

// zstart [K][NS][B]: exact state at the start of every chunk (zstart[0] = z0 or 0)
template <int NS>
__global__ __launch_bounds__(64) void ss_lin_starts_kernel(const float* __restrict__ coef, const float* __restrict__ zend0,
                                                           const float* __restrict__ z0, float* __restrict__ zstart,
                                                           int64_t B, int64_t K, int64_t L)
{
    constexpr int NSa = NS > 0 ? NS : 1;
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    double P[NSa][NSa], M[NSa][NSa];                          // P = A^L
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) { M[i][j] = (double)coef[i * NS + j]; P[i][j] = i == j ? 1.0 : 0.0; }
    for (int64_t e = L; e > 0; e >>= 1) {
        double R[NSa][NSa];
        if (e & 1) {
#pragma unroll
            for (int i = 0; i < NS; ++i)
#pragma unroll
                for (int j = 0; j < NS; ++j) { double a = 0.0; for (int q = 0; q < NS; ++q) a += P[i][q] * M[q][j]; R[i][j] = a; }
#pragma unroll
            for (int i = 0; i < NS; ++i)
#pragma unroll
                for (int j = 0; j < NS; ++j) P[i][j] = R[i][j];
        }
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int j = 0; j < NS; ++j) { double a = 0.0; for (int q = 0; q < NS; ++q) a += M[i][q] * M[q][j]; R[i][j] = a; }
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int j = 0; j < NS; ++j) M[i][j] = R[i][j];
    }
    double z[NSa];
#pragma unroll
    for (int s = 0; s < NS; ++s) z[s] = z0 ? (double)z0[s * B + b] : 0.0;
    for (int64_t k = 0; k < K; ++k) {
#pragma unroll
        for (int s = 0; s < NS; ++s) zstart[(k * NS + s) * B + b] = (float)z[s];
        double zn[NSa];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            double a = (double)zend0[(k * NS + i) * B + b];
#pragma unroll
            for (int j = 0; j < NS; ++j) a += P[i][j] * z[j];
            zn[i] = a;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) z[s] = zn[s];
    }
}

template <int NS, int NI, bool VEC4>
__global__ __launch_bounds__(64) void ss_lin_chunk_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                          const float* __restrict__ zstart, float* __restrict__ y,
                                                          float* __restrict__ zstash, float* __restrict__ zT, int64_t B,
                                                          int64_t T, int64_t L)
{
    constexpr int NSa = NS > 0 ? NS : 1;
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const int64_t k = blockIdx.y, t0 = k * L, t1 = (t0 + L < T) ? t0 + L : T;
    SSCoef<NS, NI> c;
    c.load(coef);
    const SSDiode dp = {};
    float z[NSa];
#pragma unroll
    for (int s = 0; s < NSa; ++s) z[s] = (NS > 0) ? zstart[(k * NS + (s < NS ? s : 0)) * B + b] : 0.0f;
    const bool STASH = (zstash != nullptr) && NS > 0;
    // bursts of 32 steps (t0 is a multiple of the burst), the next one in flight under this one's steps; then the tail
    const int64_t tfull = t1 - (t1 - t0) % kLinBurst;
    float xc[kLinBurst][NI], xn[kLinBurst][NI];
    if (t0 < tfull) ss_load_burst<NI, VEC4, kLinBurst>(x, b, T, t0, xn);
    for (int64_t t = t0; t < tfull; t += kLinBurst) {
#pragma unroll
        for (int q = 0; q < kLinBurst; ++q)
#pragma unroll
            for (int i = 0; i < NI; ++i) xc[q][i] = xn[q][i];
        if (t + kLinBurst < tfull) ss_load_burst<NI, VEC4, kLinBurst>(x, b, T, t + kLinBurst, xn);
#pragma unroll
        for (int q = 0; q < kLinBurst; ++q) {
            if (STASH) {
#pragma unroll
                for (int s = 0; s < NS; ++s) zstash[((t + q) * NS + s) * B + b] = z[s];
            }
            y[(t + q) * B + b] = ss_fwd_step<NS, NI, kRootNone, true>(c, dp, xc[q], z);
        }
    }
    for (int64_t t = tfull; t < t1; ++t) {
        float xt[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) xt[i] = x[(b * T + t) * NI + i];
        if (STASH) {
#pragma unroll
            for (int s = 0; s < NS; ++s) zstash[(t * NS + s) * B + b] = z[s];
        }
        y[t * B + b] = ss_fwd_step<NS, NI, kRootNone, true>(c, dp, xt, z);
    }
    if (zT && t1 == T) {
#pragma unroll
        for (int s = 0; s < NS; ++s) zT[s * B + b] = z[s];
    }
}

// ---- reverse sweep -----------------------------------------------------------------------
// With g = dL/dy[n] and lam[s] = dL/dz'[s] (state after step n):
//   gb = fy g + E . lam ;  ga = gb D_a(a)
//   lam'[s'] = cy[s'] g + sum_s A[s][s'] lam[s] + ca[s'] ga
//   dA[s][s'] += lam[s] z[s'] ; dBx[s][i] += lam[s] x[i] ; dE[s] += lam[s] b
//   dca[s] += ga z[s] ; dda[i] += ga x[i] ; dcy[s] += g z[s] ; ddy[i] += g x[i] ; dfy += g b
//   diode: dL += gb D_L ; dV += gb D_V   (D's partials as in wdf_clipper.h)
// Accumulators per lane: SSCoef::kN coefficient gradients + 2 root sums, fp32 inside a
// block of 8 steps, fp64 across blocks.
template <int NS, int NI, int ROOT, bool SYM>
__device__ __forceinline__ void ss_bwd_step(const SSCoef<NS, NI>& c, const SSDiode& dp, const float (&x)[NI],
                                            const float (&z)[NS > 0 ? NS : 1], float g,
                                            float (&lam)[NS > 0 ? NS : 1], float (&acc)[SSCoef<NS, NI>::kN + 2])
{
    using C = SSCoef<NS, NI>;
    float b = 0.0f, Da = 0.0f, DL = 0.0f, DV = 0.0f;
    if constexpr (ROOT == kRootDiode) {
        float a = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; ++s) a = fmaf(c.v[C::oCa + s], z[s], a);
#pragma unroll
        for (int i = 0; i < NI; ++i) a = fmaf(c.v[C::oDa + i], x[i], a);
        const DiodeOut o = diode_pair<SYM>(a, dp.L, dp.d);
        b = o.b;
        const float w0p = o.w0 * fast_rcp(1.0f + o.w0);
        const float w1p = o.w1 * fast_rcp(1.0f + o.w1);
        const float l2 = o.lam * o.lam;
        const float sp = w0p + w1p;
        Da = fmaf(-2.0f * l2, sp, 1.0f);
        DL = -dp.d.two_v * o.lam * (o.m0 * w0p - o.m1 * w1p);
        DV = fmaf(2.0f * l2 * a, sp * fast_rcp(dp.V), -2.0f * o.lam * (o.m0 * o.w0 - o.m1 * o.w1));
    }
    float gb = c.v[C::oFy] * g;
#pragma unroll
    for (int s = 0; s < NS; ++s) gb = fmaf(c.v[C::oE + s], lam[s], gb);
    const float ga = gb * Da;
    // coefficient gradients
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) acc[C::oA + s * NS + s2] = fmaf(lam[s], z[s2], acc[C::oA + s * NS + s2]);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[C::oB + s * NI + i] = fmaf(lam[s], x[i], acc[C::oB + s * NI + i]);
        acc[C::oE + s] = fmaf(lam[s], b, acc[C::oE + s]);
        acc[C::oCa + s] = fmaf(ga, z[s], acc[C::oCa + s]);
        acc[C::oCy + s] = fmaf(g, z[s], acc[C::oCy + s]);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        acc[C::oDa + i] = fmaf(ga, x[i], acc[C::oDa + i]);
        acc[C::oDy + i] = fmaf(g, x[i], acc[C::oDy + i]);
    }
    acc[C::oFy] = fmaf(g, b, acc[C::oFy]);
    acc[C::kN + 0] = fmaf(gb, DL, acc[C::kN + 0]);
    acc[C::kN + 1] = fmaf(gb, DV, acc[C::kN + 1]);
    // adjoint of the state
    float ln[NS > 0 ? NS : 1];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
        float v = fmaf(c.v[C::oCa + s2], ga, c.v[C::oCy + s2] * g);
#pragma unroll
        for (int s = 0; s < NS; ++s) v = fmaf(c.v[C::oA + s * NS + s2], lam[s], v);
        ln[s2] = v;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) lam[s] = ln[s];
}

// ws: double[gridDim.x][kN + 2] per-wave partial sums
template <int NS, int NI, int ROOT, bool SYM, bool VEC4>
__global__ __launch_bounds__(64) void ss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                    const float* __restrict__ rootp, int n_up, int n_down,
                                                    const float* __restrict__ zstash, const float* __restrict__ gy,
                                                    double* __restrict__ ws, float* __restrict__ gz0, int64_t B,
                                                    int64_t T)
{
    using C = SSCoef<NS, NI>;
    constexpr int NSa = NS > 0 ? NS : 1;
    constexpr int NACC = C::kN + 2;
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    C c;
    c.load(coef);
    SSDiode dp = {};
    if constexpr (ROOT == kRootDiode) dp.load(rootp, n_up, n_down);

    double tot[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) tot[i] = 0.0;
    float lam[NSa];
#pragma unroll
    for (int s = 0; s < NSa; ++s) lam[s] = 0.0f;

    const int64_t nfull = T / kBlkSS;
    for (int64_t t = T - 1; t >= nfull * kBlkSS; --t) {      // tail first (highest t)
        float acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
        float xt[NI], zt[NSa];
#pragma unroll
        for (int i = 0; i < NI; ++i) xt[i] = x[(b * T + t) * NI + i];
#pragma unroll
        for (int s = 0; s < NS; ++s) zt[s] = zstash[(t * NS + s) * B + b];
        ss_bwd_step<NS, NI, ROOT, SYM>(c, dp, xt, zt, gy[t * B + b], lam, acc);
#pragma unroll
        for (int i = 0; i < NACC; ++i) tot[i] += acc[i];
    }
    float xc[kBlkSS][NI];
    for (int64_t blk = nfull - 1; blk >= 0; --blk) {
        const int64_t t0 = blk * kBlkSS;
        ss_load_block<NI, VEC4>(x, b, T, t0, xc);
        float zc[kBlkSS][NSa], gc[kBlkSS];
#pragma unroll
        for (int k = 0; k < kBlkSS; ++k) {
            gc[k] = gy[(t0 + k) * B + b];
#pragma unroll
            for (int s = 0; s < NS; ++s) zc[k][s] = zstash[((t0 + k) * NS + s) * B + b];
        }
        float acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int k = kBlkSS - 1; k >= 0; --k) ss_bwd_step<NS, NI, ROOT, SYM>(c, dp, xc[k], zc[k], gc[k], lam, acc);
#pragma unroll
        for (int i = 0; i < NACC; ++i) tot[i] += acc[i];
    }
    if (live && gz0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) gz0[s * B + b] = lam[s];
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        double v = live ? tot[i] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (threadIdx.x == 0) ws[(int64_t)blockIdx.x * NACC + i] = v;
    }
}

// Fixed-order sum of the per-wave partials; writes gcoef[kN] and, for the diode root,
// groot[3] = dL/d{Is, nVt, R_port}  (L = log(R_port Is / nVt)).
static __global__ __launch_bounds__(64) void ss_grad_reduce_kernel(const double* __restrict__ ws, int nparts, int nacc,
                                                            int ncoef, const float* __restrict__ rootp,
                                                            float* __restrict__ gcoef, float* __restrict__ groot)
{
    // one lane per accumulator column (nacc <= 64)
    const int i = threadIdx.x;
    double s = 0.0;
    if (i < nacc) {
        int p = 0;
        for (; p + 8 <= nparts; p += 8) {                      // eight loads in flight, added in index order
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ws[(int64_t)(p + q) * nacc + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; p < nparts; ++p) s += ws[(int64_t)p * nacc + i];
    }
    if (i < ncoef) gcoef[i] = (float)s;
    const double sL = __shfl(s, ncoef, 64), sV = __shfl(s, ncoef + 1, 64);
    if (i == 0 && groot && rootp) {
        const double Is = rootp[0], V = rootp[1], Rp = rootp[2];
        groot[0] = (float)(sL / Is);
        groot[1] = (float)(sV - sL / V);
        groot[2] = (float)(sL / Rp);
    }
}

// =========================================================================================================
// Time-parallel kernels for trees with a NONLINEAR root (HPFDiodeClipper.h:28-32's Parallel(R, Series(Vs, C)) + diode pair,
// multi-state trees): B sequences are B / 64 waves on a 1024-SIMD chip, each one long dependent chain; the time axis is
// cut into K chunks like the clipper kernels' (wdf_clipper.h).
//
// Forward -- speculate, verify, re-run what missed.  The tree + diode map contracts the state (the reference itself
// discards the first 50 outputs, clipper_pot.py:232), so chunk k starts W steps early from z = 0, records the state it
// arrives with and the state it ends with; ss_tp_verify_kernel compares the boundaries per 64-sequence wave and gates
// the SEQUENTIAL kernel, launched behind it, onto the waves with a miss (normally none: the gated launch leaves at once).
// Outputs are within tol of the sequential kernel's or ARE the sequential kernel's.  W comes from the host (the spectral
// radius of the step's Jacobian at both ends of the diode's slope, lowering.plan_ss_time_parallel).
struct SsTpStatus { int n_bad; float max_miss; int gated_waves; int pad; };

template <int NS, int NI, bool SYM, bool VEC4>
__global__ __launch_bounds__(64) void ss_fwd_tp_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                       const float* __restrict__ rootp, int n_up, int n_down,
                                                       float* __restrict__ y, float* __restrict__ zstash,
                                                       const float* __restrict__ z0, float* __restrict__ zT,
                                                       float* __restrict__ zwarm, float* __restrict__ zend,
                                                       SsTpStatus* __restrict__ status, int64_t B, int64_t T, int64_t L, int64_t W,
                                                       const float* __restrict__ zinit)
{
    // zinit [K][NS][B] (or null: z = 0): the states chunk k > 0 starts its warm-up from -- an earlier call's states at the
    // same samples when the caller re-visits the batch with slightly different coefficients (the verification decides).
    static_assert(NS >= 1, "a tree without states has nothing to speculate about");
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *status = SsTpStatus{0, 0.0f, 0, 0};   // (the verify kernel adds)
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const int64_t k = blockIdx.y, t0 = k * L, t1 = (t0 + L < T) ? t0 + L : T;
    const int64_t tw = (t0 > W) ? t0 - W : 0;
    SSCoef<NS, NI> c;
    c.load(coef);
    SSDiode dp = {};
    dp.load(rootp, n_up, n_down);
    float z[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) z[s] = tw == 0 ? (z0 ? z0[s * B + b] : 0.0f) : (zinit ? zinit[(k * NS + s) * B + b] : 0.0f);
    // the lane streams its own row of x in 8-step blocks (16-byte loads), one block ahead; tw, t0 and L are multiples of 8
    const float* __restrict__ xp = x + b * T * NI;
    const int64_t tfull = t1 - (t1 - tw) % kBlkSS;
    float xc[kBlkSS][NI], xn[kBlkSS][NI];
    if (tw < tfull) ss_load_block<NI, VEC4>(x, b, T, tw, xn);
    const bool STASH = zstash != nullptr;
    for (int64_t t = tw; t < tfull; t += kBlkSS) {
#pragma unroll
        for (int q = 0; q < kBlkSS; ++q)
#pragma unroll
            for (int i = 0; i < NI; ++i) xc[q][i] = xn[q][i];
        if (t + kBlkSS < tfull) ss_load_block<NI, VEC4>(x, b, T, t + kBlkSS, xn);
        if (t < t0) {                                           // warm-up block: nothing stored
#pragma unroll
            for (int q = 0; q < kBlkSS; ++q) (void)ss_fwd_step<NS, NI, kRootDiode, SYM>(c, dp, xc[q], z);
            continue;
        }
        if (t == t0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) zwarm[(k * NS + s) * B + b] = z[s];
        }
#pragma unroll
        for (int q = 0; q < kBlkSS; ++q) {
            if (STASH) {
#pragma unroll
                for (int s = 0; s < NS; ++s) zstash[((t + q) * NS + s) * B + b] = z[s];
            }
            y[(t + q) * B + b] = ss_fwd_step<NS, NI, kRootDiode, SYM>(c, dp, xc[q], z);
        }
    }
    if (tfull <= t0) {                                          // (a chunk shorter than a block: only the last one can be)
#pragma unroll
        for (int s = 0; s < NS; ++s) zwarm[(k * NS + s) * B + b] = z[s];
    }
    for (int64_t t = (tfull > t0 ? tfull : t0); t < t1; ++t) { // tail of the last chunk (T % 8)
        float xt[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) xt[i] = xp[t * NI + i];
        if (STASH) {
#pragma unroll
            for (int s = 0; s < NS; ++s) zstash[(t * NS + s) * B + b] = z[s];
        }
        y[t * B + b] = ss_fwd_step<NS, NI, kRootDiode, SYM>(c, dp, xt, z);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) zend[(k * NS + s) * B + b] = z[s];
    if (zT && t1 == T) {
#pragma unroll
        for (int s = 0; s < NS; ++s) zT[s * B + b] = z[s];
    }
}

// gate[wave] = 1 where one of the wave's sequences arrived at a chunk more than tol away from where its predecessor ended
static __global__ __launch_bounds__(64) void ss_tp_verify_kernel(const float* __restrict__ zwarm, const float* __restrict__ zend,
                                                                 int ns, int64_t B, int64_t K, float tol, unsigned* __restrict__ gate,
                                                                 SsTpStatus* __restrict__ status)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    float miss = 0.0f;
    int nbad = 0;
    for (int64_t k = 1; k < K; ++k)
        for (int s = 0; s < ns; ++s) {
            const float m = fabsf(zwarm[(k * ns + s) * B + b] - zend[((k - 1) * ns + s) * B + b]);
            miss = fmaxf(miss, m);
            nbad += !(m <= tol) ? 1 : 0;
        }
    if (b_raw >= B) nbad = 0;
    const bool any = __builtin_amdgcn_ballot_w64(nbad != 0) != 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        miss = fmaxf(miss, __shfl_down(miss, off, 64));
        nbad += __shfl_down(nbad, off, 64);
    }
    if (threadIdx.x == 0) {
        gate[blockIdx.x] = any ? 1u : 0u;
        if (nbad) atomicAdd(&status->n_bad, nbad);
        atomicMax(reinterpret_cast<int*>(&status->max_miss), __float_as_int(miss));
        if (any) atomicAdd(&status->gated_waves, 1);
    }
}

// Reverse sweep -- EXACT.  The adjoint recurrence is linear in the adjoint LAM that enters a chunk from the future:
// lam_n = Phi_n LAM + beta_n (Phi: NS x NS), and every accumulator is affine in LAM: acc = P LAM + q.  A chunk runs the step's
// linear part 1 + NS times (the actual adjoint with dL/dy, and one homogeneous run per unit vector of LAM; the root --
// omega, its partials -- is evaluated once per step) and leaves the record {Phi, beta, P, q}; ss_bwd_tp_combine_kernel walks
// a sequence's chunks from the last to the first (LAM_{k-1} = Phi_k LAM_k + beta_k), adds up P LAM + q in double and leaves
// the per-wave partial sums ss_grad_reduce_kernel expects.  Only the summation order differs from ss_bwd_kernel.
template <int NS, int NI> struct SsTpRec { static constexpr int NACC = SSCoef<NS, NI>::kN + 2, N = NS * NS + NS + NACC * (NS + 1); };

// the linear part of ss_bwd_step for one adjoint vector (g = 0 for the homogeneous runs)
template <int NS, int NI>
__device__ __forceinline__ void ss_bwd_linear(const SSCoef<NS, NI>& c, const float (&x)[NI], const float (&z)[NS], float b, float Da,
                                              float DL, float DV, float g, float (&lam)[NS], float (&acc)[SSCoef<NS, NI>::kN + 2])
{
    using C = SSCoef<NS, NI>;
    float gb = c.v[C::oFy] * g;
#pragma unroll
    for (int s = 0; s < NS; ++s) gb = fmaf(c.v[C::oE + s], lam[s], gb);
    const float ga = gb * Da;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) acc[C::oA + s * NS + s2] = fmaf(lam[s], z[s2], acc[C::oA + s * NS + s2]);
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[C::oB + s * NI + i] = fmaf(lam[s], x[i], acc[C::oB + s * NI + i]);
        acc[C::oE + s] = fmaf(lam[s], b, acc[C::oE + s]);
        acc[C::oCa + s] = fmaf(ga, z[s], acc[C::oCa + s]);
        acc[C::oCy + s] = fmaf(g, z[s], acc[C::oCy + s]);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        acc[C::oDa + i] = fmaf(ga, x[i], acc[C::oDa + i]);
        acc[C::oDy + i] = fmaf(g, x[i], acc[C::oDy + i]);
    }
    acc[C::oFy] = fmaf(g, b, acc[C::oFy]);
    acc[C::kN + 0] = fmaf(gb, DL, acc[C::kN + 0]);
    acc[C::kN + 1] = fmaf(gb, DV, acc[C::kN + 1]);
    float ln[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
        float v = fmaf(c.v[C::oCa + s2], ga, c.v[C::oCy + s2] * g);
#pragma unroll
        for (int s = 0; s < NS; ++s) v = fmaf(c.v[C::oA + s * NS + s2], lam[s], v);
        ln[s2] = v;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) lam[s] = ln[s];
}

// rec: float [K][SsTpRec::N][B] = {Phi[NS][NS] (row r = lam component, column = LAM component), beta[NS], P[NS][NACC], q[NACC]}
template <int NS, int NI, int ROOT, bool SYM, bool VEC4>
__global__ __launch_bounds__(64) void ss_bwd_tp_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                       const float* __restrict__ rootp, int n_up, int n_down,
                                                       const float* __restrict__ zstash, const float* __restrict__ gy,
                                                       float* __restrict__ rec, int64_t B, int64_t T, int64_t L)
{
    using C = SSCoef<NS, NI>;
    constexpr int NACC = C::kN + 2;
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const int64_t k = blockIdx.y, t0 = k * L, t1 = (t0 + L < T) ? t0 + L : T;
    C c;
    c.load(coef);
    SSDiode dp = {};
    if constexpr (ROOT == kRootDiode) dp.load(rootp, n_up, n_down);
    // run 0: the actual adjoint (beta, q); runs 1..NS: homogeneous, LAM = e_{r-1} (Phi's column, P's row)
    float lam[NS + 1][NS], acc[NS + 1][NACC];
#pragma unroll
    for (int r = 0; r <= NS; ++r) {
#pragma unroll
        for (int s = 0; s < NS; ++s) lam[r][s] = (r > 0 && s == r - 1) ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[r][i] = 0.0f;
    }
    const float* __restrict__ xp = x + b * T * NI;
    auto one_step = [&](const float (&xt)[NI], const float (&zt)[NS], float g) {
        float bb = 0.0f, Da = 0.0f, DL = 0.0f, DV = 0.0f;
        if constexpr (ROOT == kRootDiode) {
            float a = 0.0f;
#pragma unroll
            for (int s = 0; s < NS; ++s) a = fmaf(c.v[C::oCa + s], zt[s], a);
#pragma unroll
            for (int i = 0; i < NI; ++i) a = fmaf(c.v[C::oDa + i], xt[i], a);
            const DiodeOut o = diode_pair<SYM>(a, dp.L, dp.d);
            bb = o.b;
            const float w0p = o.w0 * fast_rcp(1.0f + o.w0), w1p = o.w1 * fast_rcp(1.0f + o.w1);
            const float l2 = o.lam * o.lam, sp = w0p + w1p;
            Da = fmaf(-2.0f * l2, sp, 1.0f);
            DL = -dp.d.two_v * o.lam * (o.m0 * w0p - o.m1 * w1p);
            DV = fmaf(2.0f * l2 * a, sp * fast_rcp(dp.V), -2.0f * o.lam * (o.m0 * o.w0 - o.m1 * o.w1));
        }
#pragma unroll
        for (int r = 0; r <= NS; ++r) ss_bwd_linear<NS, NI>(c, xt, zt, bb, Da, DL, DV, r == 0 ? g : 0.0f, lam[r], acc[r]);
    };
    const int64_t tfull = t1 - (t1 - t0) % kBlkSS;              // t0 and L are multiples of 8: only the last chunk has a tail
    for (int64_t t = t1 - 1; t >= tfull; --t) {                 // tail first (highest t)
        float xt[NI], zt[NS];
#pragma unroll
        for (int i = 0; i < NI; ++i) xt[i] = xp[t * NI + i];
#pragma unroll
        for (int s = 0; s < NS; ++s) zt[s] = zstash[(t * NS + s) * B + b];
        one_step(xt, zt, gy[t * B + b]);
    }
    float xc[kBlkSS][NI], xn[kBlkSS][NI], zc[kBlkSS][NS], zn[kBlkSS][NS], gc[kBlkSS], gn[kBlkSS];
    auto load_blk = [&](int64_t tb, float (&xv)[kBlkSS][NI], float (&zv)[kBlkSS][NS], float (&gv)[kBlkSS]) {
        ss_load_block<NI, VEC4>(x, b, T, tb, xv);
#pragma unroll
        for (int q = 0; q < kBlkSS; ++q) {
            gv[q] = gy[(tb + q) * B + b];
#pragma unroll
            for (int s = 0; s < NS; ++s) zv[q][s] = zstash[((tb + q) * NS + s) * B + b];
        }
    };
    // the chunk's own sums (run 0) leave fp32 every 32 steps, as the sequential sweep's do every 8: a chunk is hundreds to
    // thousands of steps long, and fp32 all the way would lose sqrt(L) .. L x 6e-8 of them
    double q64[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) q64[i] = 0.0;
    int since = 0;
    if (tfull > t0) load_blk(tfull - kBlkSS, xn, zn, gn);
    for (int64_t tb = tfull - kBlkSS; tb >= t0; tb -= kBlkSS) {  // 8-step blocks, the next one (earlier in time) in flight
        if (++since == 4) {
            since = 0;
#pragma unroll
            for (int i = 0; i < NACC; ++i) { q64[i] += (double)acc[0][i]; acc[0][i] = 0.0f; }
        }
#pragma unroll
        for (int q = 0; q < kBlkSS; ++q) {
            gc[q] = gn[q];
#pragma unroll
            for (int i = 0; i < NI; ++i) xc[q][i] = xn[q][i];
#pragma unroll
            for (int s = 0; s < NS; ++s) zc[q][s] = zn[q][s];
        }
        if (tb - kBlkSS >= t0) load_blk(tb - kBlkSS, xn, zn, gn);
#pragma unroll
        for (int q = kBlkSS - 1; q >= 0; --q) one_step(xc[q], zc[q], gc[q]);
    }
    float* __restrict__ o = rec + (size_t)k * SsTpRec<NS, NI>::N * B + b;
    int e = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 1; r <= NS; ++r) o[(size_t)(e++) * B] = lam[r][s];       // Phi[s][r-1]
#pragma unroll
    for (int s = 0; s < NS; ++s) o[(size_t)(e++) * B] = lam[0][s];             // beta
#pragma unroll
    for (int r = 1; r <= NS; ++r)
#pragma unroll
        for (int i = 0; i < NACC; ++i) o[(size_t)(e++) * B] = acc[r][i];       // P[r-1][i]
#pragma unroll
    for (int i = 0; i < NACC; ++i) o[(size_t)(e++) * B] = (float)(q64[i] + (double)acc[0][i]);   // q
}

// ws: double[gridDim.x][NACC] per-wave partial sums, as ss_bwd_kernel leaves them
template <int NS, int NI>
__global__ __launch_bounds__(64) void ss_bwd_tp_combine_kernel(const float* __restrict__ rec, double* __restrict__ ws,
                                                               float* __restrict__ gz0, int64_t B, int64_t K)
{
    constexpr int NACC = SSCoef<NS, NI>::kN + 2, NREC = SsTpRec<NS, NI>::N;
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    double LAM[NS], tot[NACC];
#pragma unroll
    for (int s = 0; s < NS; ++s) LAM[s] = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) tot[i] = 0.0;
    for (int64_t k = K - 1; k >= 0; --k) {
        const float* __restrict__ o = rec + (size_t)k * NREC * B + b;
        float v[NREC];
#pragma unroll
        for (int i = 0; i < NREC; ++i) v[i] = o[(size_t)i * B];
        const float* Phi = v;
        const float* beta = v + NS * NS;
        const float* P = beta + NS;
        const float* q = P + NS * NACC;
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            double a = (double)q[i];
#pragma unroll
            for (int r = 0; r < NS; ++r) a += (double)P[r * NACC + i] * LAM[r];
            tot[i] += a;
        }
        double nl[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            double a = (double)beta[s];
#pragma unroll
            for (int r = 0; r < NS; ++r) a += (double)Phi[s * NS + r] * LAM[r];
            nl[s] = a;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) LAM[s] = nl[s];
    }
    if (live && gz0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) gz0[s * B + b] = (float)LAM[s];
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        double v = live ? tot[i] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (threadIdx.x == 0) ws[(int64_t)blockIdx.x * NACC + i] = v;
    }
}

}  // namespace wdf
