// wdf_asym.h -- diode clipper with two DIFFERENT antiparallel diodes (BASELINE config 5):
// fp64 Newton on the exact Shockley pair versus the fp32 Wright-omega closed form.
//
// Tree as in wdf_clipper.h (P1 = Parallel(ResistiveVoltageSource(R), Capacitor(C, fs))).
// Root: up-diode (Is1, V1 = n1 Vt) conducts for v > 0, down-diode (Is2, V2) for v < 0:
//     i(v) = Is1 (exp(v/V1) - 1) - Is2 (exp(-v/V2) - 1),   a = v + Rp i,  b = v - Rp i.
// The reference has no such element (its pairs are N_up/N_down copies of ONE diode,
// diode_pretraining.py:46-47; chowdsp's DiodeT/DiodePairT are absent), so there is nothing
// to pin parity against: the oracle is an fp64 safeguarded Newton and
// mpmath in the tests.
//
//  NEWTON (fp64): solve v + Rp i(v) - a = 0 per lane, started from the omega closed form,
//      iterated until EVERY lane of the wave meets |dv| <= tol (|v| + V) -- the wavefront
//      ballot is the loop condition, so a wave stops as soon as its slowest sequence has
//      converged -- or max_iter is reached.
//  OMEGA (fp32): the two-diode generalisation of Werner eqn 39 (Toms917DiodePair.h:51-59):
//      b = a - 2 lam (Vf w(log(Rp Isf/Vf) + lam a/Vf) - Vr w(log(Rp Isr/Vr) - lam a/Vr)),
//      f = the diode that conducts for this sign of a, r = the other.  Like eqn 39 it neglects
//      the reverse diode's saturation current in the forward branch (error ~ Rp Is_r).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wdf_omega.h"

namespace wdf {

struct AsymConsts {
    float p, Rp;
    float Is1, V1, Is2, V2;
    float l1, l2;          // log(Rp Is1 / V1), log(Rp Is2 / V2)
};

__device__ __forceinline__ AsymConsts asym_load(const float* __restrict__ th, float fs)
{
    AsymConsts c;
    c.Is1 = th[0]; c.V1 = th[1]; c.Is2 = th[2]; c.V2 = th[3];
    const float R = th[4], C = th[5];
    const float G1 = 1.0f / R, G2 = C * (2.0f * fs), G = G1 + G2;
    c.Rp = 1.0f / G;
    c.p = G1 / G;
    c.l1 = logf(c.Rp * c.Is1 / c.V1);
    c.l2 = logf(c.Rp * c.Is2 / c.V2);
    return c;
}

__device__ __forceinline__ float asym_omega_root(const AsymConsts& c, float a)
{
    const float lam = vsign(a);
    const float aa = fabsf(a);
    const bool pos = a >= 0.0f;
    const float Vf = pos ? c.V1 : c.V2, Vr = pos ? c.V2 : c.V1;
    const float lf = pos ? c.l1 : c.l2, lr = pos ? c.l2 : c.l1;
    const float wf = wright_omega(fmaf(aa, fast_rcp(Vf), lf));
    const float wr = wright_omega(fmaf(-aa, fast_rcp(Vr), lr));
    return a - 2.0f * lam * (Vf * wf - Vr * wr);
}

// Two Newton iterations of the exact Shockley pair in fp32 (v_exp_f32: a tenth of an fp64 iteration's cost) between the closed
// form's start value (8 mV off for a germanium-like pair: it drops the reverse diode's saturation current) and the fp64
// iterations: those then start ~1e-7 from the root and one or two of them meet any tolerance down to 1e-12 instead of three or
// four.  Same damping as below.  iters counts these two as well.
__device__ __forceinline__ float asym_newton_f32(const AsymConsts& c, float a, float v, int& iters)
{
    const float iV1 = fast_rcp(c.V1), iV2 = fast_rcp(c.V2), lim = 4.0f * fminf(c.V1, c.V2);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const float e1 = fast_exp(v * iV1), e2 = fast_exp(-v * iV2);
        const float f = v + c.Rp * (c.Is1 * (e1 - 1.0f) - c.Is2 * (e2 - 1.0f)) - a;
        const float fp = fmaf(c.Rp, fmaf(c.Is1 * iV1, e1, c.Is2 * iV2 * e2), 1.0f);
        const float q = f * fast_rcp(fp);
        // (a quotient that is not a number -- v_exp_f32 overflowed on a wild start value -- moves nothing: fmaxf / fminf would turn
        //  it into a full step of -lim, in a direction nobody chose; the fp64 iterations behind this take over)
        const float dv = (q == q) ? fminf(fmaxf(q, -lim), lim) : 0.0f;
        v -= dv;
    }
    iters += 2;
    return v;
}

// returns b; *iters += Newton iterations this wave ran
__device__ __forceinline__ double asym_newton_root(const AsymConsts& c, double a, double v0, double tol, int max_iter,
                                                   int& iters)
{
    const double Rp = c.Rp, Is1 = c.Is1, Is2 = c.Is2, iV1 = 1.0 / (double)c.V1, iV2 = 1.0 / (double)c.V2;
    const double vscale = fmin((double)c.V1, (double)c.V2);
    double v = v0;
    // The exponentials at v follow the iterate: two fp64 exp at the start value, and after a step dv the factor exp(-dv / V) --
    // a 6th-order Taylor polynomial, exact to 2e-25 relative while |dv| / V < 1e-3, which is every step once the fp32
    // iterations ahead of this loop have run (asym_newton_f32: the first fp64 step is ~1e-7) -- instead of two more exp per
    // iteration and two for the current at the end (they were what an iteration costs).  A wave with a lane outside that
    // bound (a cold start value, the damping at work) takes the library exp again.
    auto step_exps = [&](double& e1, double& e2, double dv) {
        const double d1 = -dv * iV1, d2 = dv * iV2;
        // (also when a carried exponential is no longer finite: inf x polynomial stays inf, only a fresh exp recovers)
        if (__builtin_amdgcn_ballot_w64(!(fmax(fabs(d1), fabs(d2)) < 1.0e-3) || !(fabs(e1) < 1.0e300) || !(fabs(e2) < 1.0e300)) != 0) {
            e1 = exp(v * iV1); e2 = exp(-v * iV2);
        } else {
            const double c6 = 1.0 / 720.0, c5 = 1.0 / 120.0, c4 = 1.0 / 24.0, c3 = 1.0 / 6.0;
            e1 *= fma(d1, fma(d1, fma(d1, fma(d1, fma(d1, fma(d1, c6, c5), c4), c3), 0.5), 1.0), 1.0);
            e2 *= fma(d2, fma(d2, fma(d2, fma(d2, fma(d2, fma(d2, c6, c5), c4), c3), 0.5), 1.0), 1.0);
        }
    };
    double e1 = exp(v * iV1), e2 = exp(-v * iV2);
    for (int it = 0; it < max_iter; ++it) {
        const double f = v + Rp * (Is1 * (e1 - 1.0) - Is2 * (e2 - 1.0)) - a;
        const double fp = 1.0 + Rp * (Is1 * iV1 * e1 + Is2 * iV2 * e2);
        double dv = f / fp;
        // damping: never move more than a few thermal voltages (exp overshoot guard)
        const double lim = 4.0 * vscale;
        dv = fmin(fmax(dv, -lim), lim);
        v -= dv;
        step_exps(e1, e2, dv);                                    // ... now at the new v: the next iteration's, or the current's below
        ++iters;
        const bool active = fabs(dv) > tol * (fabs(v) + vscale);
        if (__builtin_amdgcn_ballot_w64(active) == 0) break;     // the whole wave has converged
    }
    const double i = Is1 * (e1 - 1.0) - Is2 * (e2 - 1.0);
    return v - Rp * i;
}

// One step of the tree around the root; S is the state type (double for NEWTON, float otherwise).
template <bool NEWTON>
struct AsymStep;
template <>
struct AsymStep<true> {
    using S = double;
    static __device__ __forceinline__ float run(const AsymConsts& c, float xin, double& z, double tol, int max_iter,
                                                int& iters)
    {
        const double b_diff = z - (double)xin;
        const double b_temp = -(double)c.p * b_diff;
        const double a = z + b_temp;
        const float bw = asym_omega_root(c, (float)a);                   // start value: v0 = (a + b)/2 of the closed form
        const double br = asym_newton_root(c, a, (double)asym_newton_f32(c, (float)a, 0.5f * ((float)a + bw), iters), tol, max_iter, iters);
        const double zn = br + b_temp;
        const float y = (float)(0.5 * (zn + z));
        z = zn;
        return y;
    }
};
template <>
struct AsymStep<false> {
    using S = float;
    static __device__ __forceinline__ float run(const AsymConsts& c, float xin, float& z, double, int, int&)
    {
        const float b_diff = z - xin;
        const float b_temp = -c.p * b_diff;
        const float a = z + b_temp;
        const float zn = asym_omega_root(c, a) + b_temp;
        const float y = 0.5f * (zn + z);
        z = zn;
        return y;
    }
};

// x [B][T] -> y [T][B]; theta6 = {Is1, V1, Is2, V2, R, C}; iters_out: optional int64[gridDim.x]
template <bool NEWTON, bool VEC4>
__global__ __launch_bounds__(64) void clipper_asym_fwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ theta6, float fs,
                                                              float* __restrict__ y, float* __restrict__ zstash,
                                                              const float* __restrict__ z0,
                                                              float* __restrict__ zT, double tol, int max_iter,
                                                              long long* __restrict__ iters_out, int64_t B, int64_t T,
                                                              const unsigned* __restrict__ gate = nullptr)
{
    using S = typename AsymStep<NEWTON>::S;
    if (gate != nullptr && gate[blockIdx.x] == 0u) return;    // sequential re-run behind a time-parallel pass: flagged waves only
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const AsymConsts c = asym_load(theta6, fs);
    float* __restrict__ yp = y + b;
    float* __restrict__ zp = zstash ? zstash + b : nullptr;   // state BEFORE each step, for clipper_asym_bwd_kernel
    int iters = 0;
    S z = z0 ? (S)z0[b] : (S)0;
    constexpr int kB = 8;
    const int64_t nfull = T / kB;
    float xc[kB], xn[kB];
#pragma unroll
    for (int k = 0; k < kB; ++k) xc[k] = xn[k] = 0.0f;
    auto load8 = [&](int64_t t0, float(&v)[kB]) {
        if constexpr (VEC4) {
            const float4* p = reinterpret_cast<const float4*>(x + b * T + t0);
            const float4 a = p[0], d = p[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
        } else {
#pragma unroll
            for (int k = 0; k < kB; ++k) v[k] = x[b * T + t0 + k];
        }
    };
    if (nfull > 0) load8(0, xn);
    for (int64_t blk = 0; blk < nfull; ++blk) {
#pragma unroll
        for (int k = 0; k < kB; ++k) xc[k] = xn[k];
        if (blk + 1 < nfull) load8((blk + 1) * kB, xn);       // one block ahead of the recursion
#pragma unroll
        for (int k = 0; k < kB; ++k) {
            if (zp) { *zp = (float)z; zp += B; }
            *yp = AsymStep<NEWTON>::run(c, xc[k], z, tol, max_iter, iters);
            yp += B;
        }
    }
    for (int64_t t = nfull * kB; t < T; ++t) {
        if (zp) { *zp = (float)z; zp += B; }
        *yp = AsymStep<NEWTON>::run(c, x[b * T + t], z, tol, max_iter, iters);
        yp += B;
    }
    if (zT) zT[b] = (float)z;
    if (iters_out && threadIdx.x == 0) iters_out[blockIdx.x] = iters;   // wave-uniform count
}

// ---- time-parallel forward ------------------------------------------------------------------------------
// 8192 sequences are 128 waves for 1024 SIMDs, and a Newton step is a long fp64 chain: the time axis is cut into K chunks
// (grid.y), chunk k starts W steps early from z = 0 (the step contracts: the diode-off rate 1 - 2p bounds the memory, as
// for the symmetric pair), records the state it arrives with and the state it ends with; asym_tp_verify_kernel compares
// them across every boundary and raises a per-wave gate where one misses by more than tol; the sequential kernel is then
// launched GATED and re-runs exactly those waves.  L and W are multiples of 8.
struct AsymTpStatus { int n_bad; float max_miss; int gated_waves; int pad; };

template <bool NEWTON>
__global__ __launch_bounds__(64) void clipper_asym_fwd_tp_kernel(const float* __restrict__ x, const float* __restrict__ theta6,
                                                                 float fs, float* __restrict__ y, float* __restrict__ zstash,
                                                                 const float* __restrict__ z0, float* __restrict__ zT,
                                                                 float* __restrict__ zwarm, float* __restrict__ zend, double tol,
                                                                 int max_iter, AsymTpStatus* __restrict__ status, int64_t B,
                                                                 int64_t T, int64_t L, int64_t W)
{
    using S = typename AsymStep<NEWTON>::S;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *status = AsymTpStatus{0, 0.0f, 0, 0};   // the verify kernel adds
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const int64_t k = blockIdx.y;
    const int64_t t0 = k * L, t1 = (t0 + L < T) ? t0 + L : T;
    const int64_t tw = (k > 0 && t0 > W) ? t0 - W : 0;
    const AsymConsts c = asym_load(theta6, fs);
    const float* __restrict__ xp = x + b * T;
    int iters = 0;
    S z = (tw == 0 && z0) ? (S)z0[b] : (S)0;
    constexpr int kB = 8;
    float xc[kB], xn[kB];
    auto load8 = [&](int64_t t, float(&v)[kB]) {
#pragma unroll
        for (int i = 0; i < kB; ++i) v[i] = xp[t + i < T ? t + i : T - 1];
    };
    load8(tw, xn);
    for (int64_t tb = tw; tb < t1; tb += kB) {
        if (tb == t0) zwarm[k * B + b] = (float)z;              // the state this chunk arrives with
#pragma unroll
        for (int i = 0; i < kB; ++i) xc[i] = xn[i];
        if (tb + kB < t1) load8(tb + kB, xn);
        const bool owned = tb >= t0;
#pragma unroll
        for (int i = 0; i < kB; ++i) {
            if (tb + i < t1) {                                   // wave-uniform (the last chunk's ragged end)
                const S zb = z;
                const float yv = AsymStep<NEWTON>::run(c, xc[i], z, tol, max_iter, iters);
                if (owned) {
                    if (zstash) zstash[(tb + i) * B + b] = (float)zb;
                    y[(tb + i) * B + b] = yv;
                }
            }
        }
    }
    zend[k * B + b] = (float)z;
    if (zT && t1 == T) zT[b] = (float)z;
}

// one lane per sequence: every chunk boundary; gate[wave] = 1 where any of the wave's 64 sequences missed
static __global__ __launch_bounds__(64) void asym_tp_verify_kernel(const float* __restrict__ zwarm, const float* __restrict__ zend,
                                                                   int64_t B, int64_t K, float tol, unsigned* __restrict__ gate,
                                                                   AsymTpStatus* __restrict__ status)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    float miss = 0.0f;
    int nbad = 0;
    for (int64_t k = 1; k < K; ++k) {
        const float m = fabsf(zwarm[k * B + b] - zend[(k - 1) * B + b]);
        miss = fmaxf(miss, m);
        nbad += !(m <= tol) ? 1 : 0;
    }
    if (b_raw >= B) nbad = 0;
    const bool any = __builtin_amdgcn_ballot_w64(nbad > 0) != 0;
    float wmax = miss;
    int wbad = nbad;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        wmax = fmaxf(wmax, __shfl_down(wmax, off, 64));
        wbad += __shfl_down(wbad, off, 64);
    }
    if (threadIdx.x == 0) {
        gate[blockIdx.x] = any ? 1u : 0u;
        if (wmax > 0.0f) atomicMax(reinterpret_cast<int*>(&status->max_miss), __float_as_int(wmax));
        if (wbad) atomicAdd(&status->n_bad, wbad);
        if (any) atomicAdd(&status->gated_waves, 1);
    }
}

// ---- reverse sweep of the Newton-mode loop ---------------------------------------------------------
// The root is defined implicitly, F(v; a) = v + Rp i(v) - a = 0, b = 2 v - a.  With F_v = 1 + Rp i'(v):
//     d b / d a  = 2 / F_v - 1                        (=: Da)
//     d b / d th = -2 F_th / F_v   for th in {Is1, V1, Is2, V2, Rp}:
//         F_Is1 = Rp (e1 - 1) ; F_V1 = -Rp Is1 e1 v / V1^2 ; F_Is2 = -Rp (e2 - 1) ; F_V2 = -Rp Is2 e2 v / V2^2 ; F_Rp = i(v)
//     (e1 = exp(v/V1), e2 = exp(-v/V2)).  The tree around it is the clipper's (wdf_clipper.h bwd_step):
//     g_b2n = gz + g/2 ; g_a = g_b2n Da ; g_bt = g_b2n + g_a ; S_p += -g_bt b_diff ; gz <- -p g_bt + g/2 + g_a.
// Per step the root is re-solved from the stashed state (fp64 Newton from the omega closed form, as the
// forward does).  ws: double[gridDim.x][8] per-wave sums {S_Is1, S_V1, S_Is2, S_V2, S_Rp, S_p, 0, 0}.
static __global__ __launch_bounds__(64) void clipper_asym_bwd_kernel(const float* __restrict__ x, const float* __restrict__ theta6,
                                                                     float fs, const float* __restrict__ zstash,
                                                                     const float* __restrict__ gy, double tol, int max_iter,
                                                                     double* __restrict__ ws, int64_t B, int64_t T)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const AsymConsts c = asym_load(theta6, fs);
    const double Rp = c.Rp, p = c.p, Is1 = c.Is1, Is2 = c.Is2, V1 = c.V1, V2 = c.V2;
    double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double gz = 0.0;
    int iters = 0;
    for (int64_t t = T - 1; t >= 0; --t) {
        const double z = (double)zstash[t * B + b], xin = (double)x[b * T + t], g = (double)gy[t * B + b];
        const double b_diff = z - xin;
        const double a = z - p * b_diff;
        const float bw = asym_omega_root(c, (float)a);
        const double br = asym_newton_root(c, a, 0.5 * (a + (double)bw), tol, max_iter, iters);
        const double v = 0.5 * (a + br);
        const double e1 = exp(v / V1), e2 = exp(-v / V2);
        const double iF = 1.0 / (1.0 + Rp * (Is1 / V1 * e1 + Is2 / V2 * e2));
        const double Da = 2.0 * iF - 1.0;
        const double g_b2n = gz + 0.5 * g;
        const double k2 = -2.0 * iF * g_b2n;                              // g_b2n d b / d th = k2 F_th
        s[0] += k2 * Rp * (e1 - 1.0);
        s[1] += k2 * (-Rp * Is1 * e1 * v / (V1 * V1));
        s[2] += k2 * (-Rp * (e2 - 1.0));
        s[3] += k2 * (-Rp * Is2 * e2 * v / (V2 * V2));
        s[4] += k2 * (Is1 * (e1 - 1.0) - Is2 * (e2 - 1.0));
        const double g_a = g_b2n * Da;
        const double g_bt = g_b2n + g_a;
        s[5] += -g_bt * b_diff;
        gz = -p * g_bt + 0.5 * g + g_a;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = live ? s[i] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (threadIdx.x == 0) ws[(int64_t)blockIdx.x * 8 + i] = v;
    }
}

// fixed-order sum over the waves + chain rule Rp = 1/(G1+G2), p = G1 Rp (G1 = 1/R, G2 = 2 C fs) -> gtheta6
static __global__ __launch_bounds__(256) void clipper_asym_grad_reduce_kernel(const double* __restrict__ ws, int nparts,
                                                                              const float* __restrict__ theta6, float fs,
                                                                              float* __restrict__ gtheta6)
{
    __shared__ double sh[256][6];
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nparts; i += 256)
#pragma unroll
        for (int q = 0; q < 6; ++q) a[q] += ws[(int64_t)i * 8 + q];
#pragma unroll
    for (int q = 0; q < 6; ++q) sh[threadIdx.x][q] = a[q];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
#pragma unroll
            for (int q = 0; q < 6; ++q) sh[threadIdx.x][q] += sh[threadIdx.x + off][q];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double R = theta6[4], C = theta6[5];
        const double G1 = 1.0 / R, G2 = C * (2.0 * (double)fs), Rp = 1.0 / (G1 + G2), p = G1 * Rp;
        const double SRp = sh[0][4], Sp = sh[0][5];
#pragma unroll
        for (int q = 0; q < 4; ++q) gtheta6[q] = (float)sh[0][q];
        gtheta6[4] = (float)(SRp * Rp * Rp * G1 * G1 - Sp * G1 * G1 * Rp * (1.0 - p));
        gtheta6[5] = (float)(-2.0 * (double)fs * (SRp * Rp * Rp + Sp * p * Rp));
    }
}

// ---- time-parallel reverse sweep (both modes) ---------------------------------------------------------------------------
// The sweep above is one dependent chain per sequence (128 waves on 1024 SIMDs at B = 8192) and re-solves the root by Newton
// at every step.  Neither is needed:
//  * Given the stash, every step's local quantities are independent of the adjoint: the root needs no re-solve -- the
//    forward's z' = b + b_temp gives b = z[t+1] + p (z[t] - x[t]) from two CONSECUTIVE stash entries (the last step takes
//    z[T] = zT), so v = (a + b)/2 costs a subtraction instead of an fp64 Newton solve (the stash is fp32: 6e-8 relative on
//    z moves exp(v/V) by ~1e-6 relative -- far inside the gradient's 2e-4 against finite differences).
//  * The adjoint recurrence is LINEAR in the adjoint entering a chunk from the future: with u = gz + g/2,
//        gz <- kappa u + g/2,  kappa = Da - p (1 + Da) ;   S_i += c_i u
//    a chunk run with the unknown entering adjoint Lam carries (m, g0): gz = m Lam + g0, and leaves
//        {P, q}: gz at its first step's exit = P Lam + q ;  {alpha_i, beta_i}: its sums = alpha_i Lam + beta_i.
//    clipper_asym_bwd_combine_kernel walks a sequence's K records last to first -- exact, no truncation.
// NEWTON mode differentiates the exact Shockley pair implicitly (formulas above).  OMEGA mode differentiates the fp32
// closed form the OMEGA forward evaluates (asym_omega_root): with xf = lf + |a|/Vf, xr = lr - |a|/Vr, w' = w/(1 + w):
//     db/da   = 1 - 2 (wf' + wr')
//     db/dVf  = -2 lam (wf - wf' (1 + |a|/Vf)) ;  db/dVr  = 2 lam (wr - wr' (1 - |a|/Vr))
//     db/dIsf = -2 lam Vf wf' / Isf            ;  db/dIsr = 2 lam Vr wr' / Isr ;   db/dRp = -2 lam (Vf wf' - Vr wr') / Rp
// (f = the diode conducting for this sign of a, r = the other).
// rec: double [K][14][B] = {P, q, alpha[6], beta[6]}.   L is a multiple of 8.
constexpr int kAsymRec = 14;

template <bool NEWTON, bool VEC4>
__global__ __launch_bounds__(64) void clipper_asym_bwd_tp_kernel(const float* __restrict__ x, const float* __restrict__ theta6,
                                                                 float fs, const float* __restrict__ zstash,
                                                                 const float* __restrict__ zT, const float* __restrict__ gy,
                                                                 double* __restrict__ rec, int64_t B, int64_t T, int64_t L)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const int64_t k = blockIdx.y;
    const int64_t t0 = k * L, t1 = (t0 + L < T) ? t0 + L : T;
    const AsymConsts c = asym_load(theta6, fs);
    const double p = c.p;
    double m = 1.0, g0 = 0.0;
    double al[6] = {0, 0, 0, 0, 0, 0}, be[6] = {0, 0, 0, 0, 0, 0};
    const float* __restrict__ xp = x + b * T;
    float znext = (t1 < T) ? zstash[t1 * B + b] : zT[b];         // the state AFTER the chunk's last step
    constexpr int kB = 8;
    // blocks of 8 steps, last to first; the chunk's ragged top block (t1 not a multiple of 8: the sequence's end) first
    int64_t tb = (t1 - 1) / kB * kB;
    float xc[kB];
    for (; tb >= t0; tb -= kB) {
        if constexpr (VEC4) {
            if (tb + kB <= T) {
                const float4* q4 = reinterpret_cast<const float4*>(xp + tb);
                const float4 u0 = q4[0], u1 = q4[1];
                xc[0] = u0.x; xc[1] = u0.y; xc[2] = u0.z; xc[3] = u0.w; xc[4] = u1.x; xc[5] = u1.y; xc[6] = u1.z; xc[7] = u1.w;
            } else {
#pragma unroll
                for (int i = 0; i < kB; ++i) xc[i] = xp[tb + i < T ? tb + i : T - 1];
            }
        } else {
#pragma unroll
            for (int i = 0; i < kB; ++i) xc[i] = xp[tb + i < T ? tb + i : T - 1];
        }
#pragma unroll
        for (int i = kB - 1; i >= 0; --i) {
            const int64_t t = tb + i;
            if (t < t1) {                                         // wave-uniform
                // The step's local partials in fp32: their inputs are fp32 already (the stash, x), 1e-6 relative is two orders
                // inside the gradient's tolerance, and fp64 exp / divide were most of this kernel (0.58 ms at 8192 x 4096).
                // The sums and the adjoint recurrence stay in fp64 below.
                const float zf = zstash[t * B + b];
                const double g = (double)gy[t * B + b];
                const float bd = zf - xc[i];
                const float a = zf - c.p * bd;                    // the forward's own fp32 root input
                float Daf, cf[5];
                if constexpr (NEWTON) {
                    const float br = fmaf(c.p, bd, znext);        // b = z[t+1] + p (z[t] - x[t]): no re-solve
                    const float v = 0.5f * (a + br);
                    const float iV1 = fast_rcp(c.V1), iV2 = fast_rcp(c.V2);
                    const float e1 = fast_exp(v * iV1), e2 = fast_exp(-v * iV2);
                    const float iF = fast_rcp(fmaf(c.Rp, fmaf(c.Is1 * iV1, e1, c.Is2 * iV2 * e2), 1.0f));
                    Daf = fmaf(2.0f, iF, -1.0f);
                    const float k2 = -2.0f * iF, k2R = k2 * c.Rp;
                    cf[0] = k2R * (e1 - 1.0f);
                    cf[1] = -k2R * c.Is1 * e1 * v * iV1 * iV1;
                    cf[2] = -k2R * (e2 - 1.0f);
                    cf[3] = -k2R * c.Is2 * e2 * v * iV2 * iV2;
                    cf[4] = k2 * fmaf(c.Is1, e1 - 1.0f, -c.Is2 * (e2 - 1.0f));
                } else {
                    const float lam = vsign(a), aa = fabsf(a);
                    const bool pos = a >= 0.0f;
                    const float Vf = pos ? c.V1 : c.V2, Vr = pos ? c.V2 : c.V1;
                    const float lf = pos ? c.l1 : c.l2, lr = pos ? c.l2 : c.l1;
                    const float Isf = pos ? c.Is1 : c.Is2, Isr = pos ? c.Is2 : c.Is1;
                    const float af = aa * fast_rcp(Vf), ar = aa * fast_rcp(Vr);
                    const float wf = wright_omega(af + lf), wr = wright_omega(lr - ar);
                    const float dwf = wf * fast_rcp(1.0f + wf), dwr = wr * fast_rcp(1.0f + wr);
                    const float l2 = 2.0f * lam;
                    Daf = fmaf(-2.0f, dwf + dwr, 1.0f);
                    const float dVf = -l2 * fmaf(-dwf, 1.0f + af, wf), dVr = l2 * fmaf(-dwr, 1.0f - ar, wr);
                    const float dIsf = -l2 * Vf * dwf * fast_rcp(Isf), dIsr = l2 * Vr * dwr * fast_rcp(Isr);
                    cf[0] = pos ? dIsf : dIsr;
                    cf[1] = pos ? dVf : dVr;
                    cf[2] = pos ? dIsr : dIsf;
                    cf[3] = pos ? dVr : dVf;
                    cf[4] = -l2 * fmaf(Vf, dwf, -Vr * dwr) * fast_rcp(c.Rp);
                }
                const double Da = (double)Daf, b_diff = (double)bd;
                double cth[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) cth[q] = (double)cf[q];
                const double h = g0 + 0.5 * g;                    // u = gz + g/2 = m Lam + h
#pragma unroll
                for (int q = 0; q < 5; ++q) { al[q] = fma(cth[q], m, al[q]); be[q] = fma(cth[q], h, be[q]); }
                const double c5 = -(1.0 + Da) * b_diff;
                al[5] = fma(c5, m, al[5]);
                be[5] = fma(c5, h, be[5]);
                const double kap = Da - p * (1.0 + Da);
                g0 = fma(kap, h, 0.5 * g);
                m *= kap;
                znext = zf;
            }
        }
    }
    double* __restrict__ r = rec + (k * kAsymRec) * B + b;
    r[0] = m;
    r[B] = g0;
#pragma unroll
    for (int q = 0; q < 6; ++q) { r[(2 + q) * B] = al[q]; r[(8 + q) * B] = be[q]; }
}

// one lane per sequence: the K records last to first (Lam = gzT or 0 enters the last chunk), then the wave's sums ->
// ws[wave][8] (clipper_asym_grad_reduce_kernel finishes); gz0 (optional): dL/dz0
static __global__ __launch_bounds__(64) void clipper_asym_bwd_combine_kernel(const double* __restrict__ rec, const float* __restrict__ gzT,
                                                                             double* __restrict__ ws, float* __restrict__ gz0,
                                                                             int64_t B, int64_t K)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    double lam = gzT ? (double)gzT[b] : 0.0;
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t k = K - 1; k >= 0; --k) {
        const double* __restrict__ r = rec + (k * kAsymRec) * B + b;
#pragma unroll
        for (int q = 0; q < 6; ++q) s[q] += fma(r[(2 + q) * B], lam, r[(8 + q) * B]);
        lam = fma(r[0], lam, r[B]);
    }
    if (gz0 && live) gz0[b] = (float)lam;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = live ? s[i] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (threadIdx.x == 0) ws[(int64_t)blockIdx.x * 8 + i] = v;
    }
}

// element-wise root, for accuracy sweeps: b[i] = root(a[i])
template <bool NEWTON>
__global__ void asym_root_kernel(const float* __restrict__ a, const float* __restrict__ theta6, float fs,
                                 double* __restrict__ b, double tol, int max_iter, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = i < n ? i : n - 1;
    const AsymConsts c = asym_load(theta6, fs);
    double r;
    if constexpr (NEWTON) {
        int it = 0;
        const float bw = asym_omega_root(c, a[j]);
        r = asym_newton_root(c, (double)a[j], 0.5 * ((double)a[j] + (double)bw), tol, max_iter, it);
    } else {
        r = (double)asym_omega_root(c, a[j]);
    }
    if (i < n) b[i] = r;
}

}  // namespace wdf
