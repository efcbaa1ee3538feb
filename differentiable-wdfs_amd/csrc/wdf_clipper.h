// wdf_clipper.h -- diode-clipper sequence kernels for gfx950 (MI355X).
//
// Circuit (clipper_pot.py:94-101 topology with the analytic root of the north star):
//     Vs = ResistiveVoltageSource(R)   C = Capacitor(C, fs)   P1 = Parallel(Vs, C)
//     root = diode pair on P1
// One wavefront lane owns one training sequence: the whole tree state is the capacitor
// state z (1 VGPR); adaptor coefficients are loop invariants (or 4 VALU ops per step when a
// per-sample resistance channel is present, clipper_pot.py:116-117).  Inputs are read
// batch-major [B][T] straight from the layout the reference scripts use (input[:, i]):
// each lane streams its own row with 16-byte loads one 8-step block ahead, so a 128-byte
// line is fetched from HBM once and consumed from L1/L2 over the next 32 steps.  Outputs
// and the state stash are time-major [T][B] (TensorArray.stack() layout), so every store
// is one fully coalesced 256-byte wave transaction.  No LDS, no MFMA: the path is a scalar
// recurrence, bounded by dependent-op latency per step.
//
// Per step (tf_wdf.py line numbers):
//   b_diff = z - x                     Parallel.reflected  :185-192 (b1 = Vs :57-59, b2 = z :124-126)
//   b_temp = -p b_diff ; a = z + b_temp
//   b      = diode_pair(a)             diode_pretraining.py:39-60
//   z'     = b + b_temp                Parallel.incident :179-183 -> Capacitor.incident :120-122
//   y      = (z' + z) / 2              voltage(C) :8-10  (clipper_pot.py:123)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wdf_omega.h"

namespace wdf {

constexpr int kBlk = 8;   // time steps per register block (2 x 16-byte loads per lane)

struct ClipConsts {
    float Is, V;     // diode: saturation current, n*Vt
    float G2;        // 1/Rc = 2 C fs                     Capacitor.calc_impedance :114-115
    float lIV;       // log(Is / V)
    float p, Rp, L;  // static-R only: G1/(G1+G2), 1/(G1+G2), log(Rp Is / V)   :168-177
    DiodeStatic d;
};

__device__ __forceinline__ ClipConsts load_consts(const float* __restrict__ theta, float fs, int n_up, int n_down)
{
    ClipConsts c;
    const float Is = theta[0], V = theta[1], R = theta[2], C = theta[3];
    c.Is = Is;
    c.V = V;
    c.G2 = C * (2.0f * fs);
    const float G1 = 1.0f / R;
    const float G = G1 + c.G2;
    c.Rp = 1.0f / G;
    c.p = G1 / G;
    c.L = logf(c.Rp * Is / V);
    c.lIV = logf(Is / V);
    c.d = make_diode_static(V, n_up, n_down);
    return c;
}

// adaptor coefficients for this step
template <bool DYN_R>
__device__ __forceinline__ void step_coeffs(const ClipConsts& c, float rin, float& p, float& Rp, float& L)
{
    if constexpr (DYN_R) {                       // set_resistance + calc_impedance every step
        const float G1 = fast_rcp(rin);
        Rp = fast_rcp(G1 + c.G2);
        p = G1 * Rp;
        L = fast_log(Rp) + c.lIV;
    } else {
        p = c.p;
        Rp = c.Rp;
        L = c.L;
    }
}

// ---- block loads ----------------------------------------------------------------------
// v[k] = x[b][t0 + k] (batch-major) or x[t0 + k][b] (time-major); full blocks only.
template <bool TIME_MAJOR, bool VEC4>
__device__ __forceinline__ void load_block(const float* __restrict__ x, int64_t b, int64_t B, int64_t T,
                                           int64_t t0, float (&v)[kBlk])
{
    if constexpr (TIME_MAJOR) {
#pragma unroll
        for (int k = 0; k < kBlk; ++k) v[k] = x[(t0 + k) * B + b];
    } else if constexpr (VEC4) {
        const float4* p = reinterpret_cast<const float4*>(x + b * T + t0);
        const float4 a = p[0], c = p[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
#pragma unroll
        for (int k = 0; k < kBlk; ++k) v[k] = x[b * T + t0 + k];
    }
}

template <bool TIME_MAJOR>
__device__ __forceinline__ float load_one(const float* __restrict__ x, int64_t b, int64_t B, int64_t T, int64_t t)
{
    return TIME_MAJOR ? x[t * B + b] : x[b * T + t];
}

// =========================================================================================
// forward
// =========================================================================================
template <bool DYN_R, bool SYM>
__device__ __forceinline__ float fwd_step(const ClipConsts& c, float xin, float rin, float& z)
{
    float p, Rp, L;
    step_coeffs<DYN_R>(c, rin, p, Rp, L);
    const float b_diff = z - xin;
    const float b_temp = -p * b_diff;
    const float a = z + b_temp;
    const DiodeOut o = diode_pair<SYM>(a, L, c.d);
    const float zn = o.b + b_temp;
    const float y = 0.5f * (zn + z);
    z = zn;
    return y;
}

template <bool DYN_R, bool SYM, bool TIME_MAJOR, bool VEC4, bool STASH>
__global__ __launch_bounds__(64) void clipper_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta,
    float fs, int n_up, int n_down, float* __restrict__ y, float* __restrict__ zstash,
    const float* __restrict__ z0, float* __restrict__ zT, int64_t B, int64_t T)
{
    // Lanes past the end of the batch shadow the last sequence: they compute and store the
    // same values to the same addresses as its owner, so no store needs an exec-mask branch.
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const ClipConsts c = load_consts(theta, fs, n_up, n_down);
    float z = z0 ? z0[b] : 0.0f;                // reset(): clipper_pot.py:110-111
    float* __restrict__ yp = y + b;             // walks down column b of the [T][B] outputs
    float* __restrict__ zp = STASH ? zstash + b : nullptr;

    const int64_t nfull = T / kBlk;
    float xc[kBlk], xn[kBlk], rc[kBlk], rn[kBlk];
#pragma unroll
    for (int k = 0; k < kBlk; ++k) { xc[k] = xn[k] = 0.0f; rc[k] = rn[k] = 1.0f; }
    if (nfull > 0) {
        load_block<TIME_MAJOR, VEC4>(x, b, B, T, 0, xn);
        if constexpr (DYN_R) load_block<TIME_MAJOR, VEC4>(r, b, B, T, 0, rn);
    }
    for (int64_t blk = 0; blk < nfull; ++blk) {
        const int64_t t0 = blk * kBlk;
#pragma unroll
        for (int k = 0; k < kBlk; ++k) { xc[k] = xn[k]; if constexpr (DYN_R) rc[k] = rn[k]; }
        if (blk + 1 < nfull) {                  // prefetch the next block while this one computes
            load_block<TIME_MAJOR, VEC4>(x, b, B, T, t0 + kBlk, xn);
            if constexpr (DYN_R) load_block<TIME_MAJOR, VEC4>(r, b, B, T, t0 + kBlk, rn);
        }
#pragma unroll
        for (int k = 0; k < kBlk; ++k) {
            if constexpr (STASH) { *zp = z; zp += B; }
            *yp = fwd_step<DYN_R, SYM>(c, xc[k], rc[k], z);
            yp += B;
        }
    }
    for (int64_t t = nfull * kBlk; t < T; ++t) {   // tail (T % 8 steps)
        const float xin = load_one<TIME_MAJOR>(x, b, B, T, t);
        const float rin = DYN_R ? load_one<TIME_MAJOR>(r, b, B, T, t) : 1.0f;
        if constexpr (STASH) { *zp = z; zp += B; }
        *yp = fwd_step<DYN_R, SYM>(c, xin, rin, z);
        yp += B;
    }
    if (zT) zT[b] = z;
}

// =========================================================================================
// reverse sweep
// =========================================================================================
// Adjoint of fwd_step.  With g = dL/dy[n] and gz = dL/dz' (state after step n):
//   g_b2n = gz + g/2                  (z' = b + b_temp ; y = (z' + z)/2)
//   g_a   = g_b2n Da ; g_L = g_b2n DL ; g_V = g_b2n DV      (b = D(a; L, V))
//   g_bt  = g_b2n + g_a               (a = z + b_temp)
//   g_p   = -g_bt b_diff              (b_temp = -p b_diff)
//   gz   <- g/2 + g_a - p g_bt        (through y, a and b_diff = z - x)
// D's partials use omega' = omega/(1+omega):
//   Da = 1 - 2 lam^2 (w0' + w1')
//   DL = -2 V lam (mu0 w0' - mu1 w1')
//   DV = -2 lam (mu0 w0 - mu1 w1) + 2 lam^2 a/V (w0' + w1')      (at fixed L)
struct StepGrads {
    float sL, sV, sP;   // dL/dL, dL/dV|_L, and dL/dp (static R) or Rp (g_p p + g_L) (per-sample R)
};

template <bool DYN_R, bool SYM>
__device__ __forceinline__ void bwd_step(const ClipConsts& c, float xin, float rin, float z, float g,
                                         float& gz, StepGrads& acc)
{
    float p, Rp, L;
    step_coeffs<DYN_R>(c, rin, p, Rp, L);
    const float b_diff = z - xin;
    const float a = fmaf(-p, b_diff, z);
    const DiodeOut o = diode_pair<SYM>(a, L, c.d);
    const float w0p = o.w0 * fast_rcp(1.0f + o.w0);
    const float w1p = o.w1 * fast_rcp(1.0f + o.w1);
    const float l2 = o.lam * o.lam;
    const float sp = w0p + w1p;
    const float Da = fmaf(-2.0f * l2, sp, 1.0f);
    const float DL = -c.d.two_v * o.lam * (o.m0 * w0p - o.m1 * w1p);
    const float DV = fmaf(2.0f * l2 * a, sp * fast_rcp(c.V), -2.0f * o.lam * (o.m0 * o.w0 - o.m1 * o.w1));
    const float g_b2n = fmaf(0.5f, g, gz);
    const float g_a = g_b2n * Da;
    const float g_L = g_b2n * DL;
    const float g_bt = g_b2n + g_a;
    const float g_p = -g_bt * b_diff;
    acc.sL += g_L;
    acc.sV = fmaf(g_b2n, DV, acc.sV);
    if constexpr (DYN_R) acc.sP = fmaf(Rp, fmaf(g_p, p, g_L), acc.sP);
    else acc.sP += g_p;
    gz = fmaf(-p, g_bt, fmaf(0.5f, g, g_a));
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ws: double[gridDim.x][4] per-wave partial sums {S_L, S_V, S_P, 0}
template <bool DYN_R, bool SYM, bool TIME_MAJOR, bool VEC4>
__global__ __launch_bounds__(64) void clipper_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta,
    float fs, int n_up, int n_down, const float* __restrict__ zstash, const float* __restrict__ gy,
    double* __restrict__ ws, float* __restrict__ gz0, int64_t B, int64_t T)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const ClipConsts c = load_consts(theta, fs, n_up, n_down);

    double dL = 0.0, dV = 0.0, dP = 0.0;
    float gz = 0.0f;

    const int64_t nfull = T / kBlk;
    for (int64_t t = T - 1; t >= nfull * kBlk; --t) {     // tail first (highest t)
        StepGrads acc = {0.0f, 0.0f, 0.0f};
        const float xin = load_one<TIME_MAJOR>(x, b, B, T, t);
        const float rin = DYN_R ? load_one<TIME_MAJOR>(r, b, B, T, t) : 1.0f;
        bwd_step<DYN_R, SYM>(c, xin, rin, zstash[t * B + b], gy[t * B + b], gz, acc);
        dL += acc.sL; dV += acc.sV; dP += acc.sP;
    }
    float xc[kBlk], xn[kBlk], rc[kBlk], rn[kBlk], zc[kBlk], zn[kBlk], gc[kBlk], gn[kBlk];
#pragma unroll
    for (int k = 0; k < kBlk; ++k) { xc[k] = xn[k] = zc[k] = zn[k] = gc[k] = gn[k] = 0.0f; rc[k] = rn[k] = 1.0f; }
    if (nfull > 0) {
        const int64_t t0 = (nfull - 1) * kBlk;
        load_block<TIME_MAJOR, VEC4>(x, b, B, T, t0, xn);
        if constexpr (DYN_R) load_block<TIME_MAJOR, VEC4>(r, b, B, T, t0, rn);
        load_block<true, false>(zstash, b, B, T, t0, zn);
        load_block<true, false>(gy, b, B, T, t0, gn);
    }
    for (int64_t blk = nfull - 1; blk >= 0; --blk) {
#pragma unroll
        for (int k = 0; k < kBlk; ++k) {
            xc[k] = xn[k]; zc[k] = zn[k]; gc[k] = gn[k];
            if constexpr (DYN_R) rc[k] = rn[k];
        }
        if (blk > 0) {
            const int64_t t0 = (blk - 1) * kBlk;
            load_block<TIME_MAJOR, VEC4>(x, b, B, T, t0, xn);
            if constexpr (DYN_R) load_block<TIME_MAJOR, VEC4>(r, b, B, T, t0, rn);
            load_block<true, false>(zstash, b, B, T, t0, zn);
            load_block<true, false>(gy, b, B, T, t0, gn);
        }
        StepGrads acc = {0.0f, 0.0f, 0.0f};     // fp32 within a block, fp64 across blocks
#pragma unroll
        for (int k = kBlk - 1; k >= 0; --k) bwd_step<DYN_R, SYM>(c, xc[k], rc[k], zc[k], gc[k], gz, acc);
        dL += acc.sL; dV += acc.sV; dP += acc.sP;
    }
    if (!live) { dL = dV = dP = 0.0; }
    if (live && gz0) gz0[b] = gz;
    dL = wave_sum(dL); dV = wave_sum(dV); dP = wave_sum(dP);
    if (threadIdx.x == 0) {
        double* o = ws + (int64_t)blockIdx.x * 4;
        o[0] = dL; o[1] = dV; o[2] = dP; o[3] = 0.0;
    }
}

// Fixed-order reduction of the per-wave partials + chain rule to {Is, nVt, R, C}:
//   L = log Rp + log Is - log V ;  G1 = 1/R ; G2 = 2 C fs ; Rp = 1/(G1+G2) ; p = G1 Rp
//   static R :  dIs = S_L/Is ; dV = S_V - S_L/V ; dR = Rp G1^2 (S_L - S_P (1-p)) ;
//               dC = -2 fs Rp (S_P p + S_L)
//   per-sample R : S_P = sum Rp_n (g_p p_n + g_L) ; dR = 0 ; dC = -2 fs S_P
__global__ __launch_bounds__(256) void clipper_grad_reduce_kernel(
    const double* __restrict__ ws, int nparts, const float* __restrict__ theta, float fs,
    int dyn_r, float* __restrict__ gtheta, int accumulate)
{
    __shared__ double sh[256][3];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        s0 += ws[(int64_t)i * 4 + 0]; s1 += ws[(int64_t)i * 4 + 1]; s2 += ws[(int64_t)i * 4 + 2];
    }
    sh[threadIdx.x][0] = s0; sh[threadIdx.x][1] = s1; sh[threadIdx.x][2] = s2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sh[threadIdx.x][0] += sh[threadIdx.x + off][0];
            sh[threadIdx.x][1] += sh[threadIdx.x + off][1];
            sh[threadIdx.x][2] += sh[threadIdx.x + off][2];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double SL = sh[0][0], SV = sh[0][1], SP = sh[0][2];
        const double Is = theta[0], V = theta[1], R = theta[2], C = theta[3];
        const double G1 = 1.0 / R, G2 = C * (2.0 * (double)fs), Rp = 1.0 / (G1 + G2), p = G1 * Rp;
        double g[4];
        g[0] = SL / Is;
        g[1] = SV - SL / V;
        if (dyn_r) {
            g[2] = 0.0;
            g[3] = -2.0 * (double)fs * SP;
        } else {
            g[2] = Rp * G1 * G1 * (SL - SP * (1.0 - p));
            g[3] = -2.0 * (double)fs * Rp * (SP * p + SL);
        }
        for (int k = 0; k < 4; ++k) gtheta[k] = (accumulate ? gtheta[k] : 0.0f) + (float)g[k];
    }
}

// ---- element-wise building blocks (parity tests) ----------------------------------------
__global__ void omega_kernel(const float* __restrict__ x, float* __restrict__ w, int32_t* __restrict__ iters, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float xi = x[i < n ? i : n - 1];
    int it = 0;
    const float wi = wright_omega<true>(xi, &it);
    if (i < n) { w[i] = wi; if (iters) iters[i] = it; }
}

__global__ void diode_pair_kernel(const float* __restrict__ a, const float* __restrict__ Rp, float Is, float nVt,
                                  int n_up, int n_down, float* __restrict__ b, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = i < n ? i : n - 1;
    const DiodeStatic d = make_diode_static(nVt, n_up, n_down);
    const float L = logf(Rp[j] * Is / nVt);
    const DiodeOut o = (n_up == n_down) ? diode_pair<false>(a[j], L, d) : diode_pair<false>(a[j], L, d);
    if (i < n) b[i] = o.b;
}

}  // namespace wdf
