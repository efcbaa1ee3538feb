// wdf_omega64.h -- Wright omega and the diode pair in fp64 on the device (flag WDF_PREC_F64, config C5).
//
// modules/toms917/toms917.cpp restricted to the real axis, in its own precision: the regional start value
// (:240-248 x <= -2, :253-261 -2 < x <= 1 + pi, :290-296 beyond, all orders kept), FSC iteration one
// (:347-352), and the reference's test for iteration two, |(2w^2 - 8w - 1) r^4| >= eps 72 |w+1|^6
// (:356-364) -- which in fp64, unlike fp32 (wdf_omega.h), really fires: near x = -2 and around the
// region boundaries the first iterate is only good to ~1e-9.  The second iteration sits behind ONE
// wavefront ballot: a wave whose 64 sequences all pass the test skips it.
// One region per lane is evaluated (per-lane branches): fp64 exp / log are ~100-instruction routines,
// evaluating all three and selecting, as the fp32 path does with its single-instruction transcendentals,
// would triple the cost.
#pragma once

#include <hip/hip_runtime.h>

namespace wdf {

__device__ __forceinline__ double fsc_step64(double x, double w, double& r)
{
    r = x - w - log(w);
    const double wp1 = w + 1.0;
    const double t = 2.0 * wp1 * (wp1 + (2.0 / 3.0) * r);
    const double e = r / wp1 * (t - r) / (t - 2.0 * r);
    return w * (1.0 + e);
}

// iters (optional): 0 (start value only: exp underflow), 1 or 2 FSC iterations, as toms917 would run
__device__ __forceinline__ double wright_omega64(double x, int* iters = nullptr)
{
    double w;
    if (x <= -2.0) {                                            // region 3
        const double p = exp(x);
        w = (1.0 + (-1.0 + (1.5 + (-8.0 / 3.0 + 125.0 / 24.0 * p) * p) * p) * p) * p;
    } else if (x <= 1.0 + 3.14159265358979323846) {             // region 4: series about x = 1
        const double q = x - 1.0;
        w = 0.5 + 0.5 * x + (1.0 / 16.0 + (-1.0 / 192.0 + (-1.0 / 3072.0 + 13.0 / 61440.0 * q) * q) * q) * q * q;
    } else {                                                    // region 7: series about +infinity
        const double l = log(x);
        w = ((1.0 + (-1.5 + (1.0 / 3.0) * l) * l) * l + ((-1.0 + 0.5 * l) * l + (l + (-l + x) * x) * x) * x) / (x * x * x);
    }
    const bool live = w > 0.0 && !(x != x);                     // exp underflow (x < -745): omega = 0 to fp64; NaN passes through
    double r = 0.0;
    double w1 = live ? fsc_step64(x, w, r) : w;
    const double rr = fabs(r), wp = fabs(w + 1.0);
    const bool again = live && fabs((2.0 * w1 * w1 - 8.0 * w1 - 1.0) * (rr * rr * rr * rr)) >=
                                   2.220446049250313e-16 * 72.0 * (wp * wp * wp * wp * wp * wp);
    if (__builtin_amdgcn_ballot_w64(again)) {                   // per wave
        double r2;
        const double w2 = fsc_step64(x, again ? w1 : 1.0, r2);
        w1 = again ? w2 : w1;
    }
    if (iters) *iters = live ? (again ? 2 : 1) : 0;
    return (x != x) ? x : w1;
}

// diode_pretraining.py:39-60 in fp64; L = log(Rp Is / nVt)
__device__ __forceinline__ double diode_pair64(double a, double L, double nVt, int n_up, int n_down)
{
    const double lam = (a > 0.0) ? 1.0 : ((a < 0.0) ? -1.0 : 0.0);
    const double m0 = (a >= 0.0) ? (double)n_down : (double)n_up, m1 = (a >= 0.0) ? (double)n_up : (double)n_down;
    const double aa = fabs(a);
    const double w0 = wright_omega64(L - log(m0) + aa / (m0 * nVt));
    const double w1 = wright_omega64(L - log(m1) - aa / (m1 * nVt));
    return a - 2.0 * nVt * lam * (m0 * w0 - m1 * w1);
}

static __global__ void omega64_kernel(const double* __restrict__ x, double* __restrict__ w, int32_t* __restrict__ iters, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int it = 0;
    const double wi = wright_omega64(x[i < n ? i : n - 1], &it);
    if (i < n) { w[i] = wi; if (iters) iters[i] = it; }
}

// The clipper loop of wdf_clipper.h with tree and root arithmetic in fp64 (x, r, theta, y, stash stay fp32):
// one lane per sequence, sequential in time -- the accuracy reference on the device, not a fast path.
template <bool DYN_R, bool TIME_MAJOR>
__global__ __launch_bounds__(64) void clipper_fwd_f64_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta, float fs, int n_up,
    int n_down, float* __restrict__ y, float* __restrict__ zstash, const float* __restrict__ z0, float* __restrict__ zT,
    int64_t B, int64_t T)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const double Is = theta[0], V = theta[1], R = theta[2], C = theta[3];
    const double G2 = C * (2.0 * (double)fs);
    double Rp = 1.0 / (1.0 / R + G2), p = (1.0 / R) * Rp, L = log(Rp * Is / V);
    double z = z0 ? (double)z0[b] : 0.0;
    for (int64_t t = 0; t < T; ++t) {
        const double xin = TIME_MAJOR ? x[t * B + b] : x[b * T + t];
        if constexpr (DYN_R) {
            const double G1 = 1.0 / (double)(TIME_MAJOR ? r[t * B + b] : r[b * T + t]);
            Rp = 1.0 / (G1 + G2);
            p = G1 * Rp;
            L = log(Rp * Is / V);
        }
        const double b_diff = z - xin;
        const double b_temp = -p * b_diff;
        const double a = z + b_temp;
        const double zn = diode_pair64(a, L, V, n_up, n_down) + b_temp;
        if (zstash) zstash[t * B + b] = (float)z;
        y[t * B + b] = (float)(0.5 * (zn + z));
        z = zn;
    }
    if (zT) zT[b] = (float)z;
}

// The reverse sweep of that loop in fp64 (flag WDF_PREC_F64 on wdf_clipper_bwd): the adjoint of wdf_clipper.h's bwd_step with
// the root recomputed from the stashed state by wright_omega64 and every partial in double (omega' = omega / (1 + omega)).
// Inputs, stash and dL/dy stay fp32 as the forward left them; per-wave partial sums {S_L, S_V, S_P, 0} in double, reduced by
// clipper_grad_reduce_kernel like the fp32 sweep's.  Sequential, one lane per sequence: the accuracy reference on the device
// for the gradient (tests hold it against the oracle's fp64 adjoint), not a fast path.
template <bool DYN_R, bool TIME_MAJOR>
__global__ __launch_bounds__(64) void clipper_bwd_f64_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta, float fs, int n_up,
    int n_down, const float* __restrict__ zstash, const float* __restrict__ gy, double* __restrict__ ws,
    float* __restrict__ gz0, const float* __restrict__ gzT, int64_t B, int64_t T)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const double Is = theta[0], V = theta[1], R = theta[2], C = theta[3];
    const double G2 = C * (2.0 * (double)fs);
    double Rp = 1.0 / (1.0 / R + G2), p = (1.0 / R) * Rp, L = log(Rp * Is / V);
    double dL = 0.0, dV = 0.0, dP = 0.0;
    double gz = gzT ? (double)gzT[b] : 0.0;
    for (int64_t t = T - 1; t >= 0; --t) {
        const double xin = TIME_MAJOR ? x[t * B + b] : x[b * T + t];
        if constexpr (DYN_R) {
            const double G1 = 1.0 / (double)(TIME_MAJOR ? r[t * B + b] : r[b * T + t]);
            Rp = 1.0 / (G1 + G2);
            p = G1 * Rp;
            L = log(Rp * Is / V);
        }
        const double z = zstash[t * B + b], g = gy[t * B + b];
        const double b_diff = z - xin;
        const double a = z - p * b_diff;
        const double lam = (a > 0.0) ? 1.0 : ((a < 0.0) ? -1.0 : 0.0);
        const double m0 = (a >= 0.0) ? (double)n_down : (double)n_up, m1 = (a >= 0.0) ? (double)n_up : (double)n_down;
        const double aa = fabs(a);
        const double w0 = wright_omega64(L - log(m0) + aa / (m0 * V));
        const double w1 = wright_omega64(L - log(m1) - aa / (m1 * V));
        const double w0p = w0 / (1.0 + w0), w1p = w1 / (1.0 + w1);
        const double l2 = lam * lam, sp = w0p + w1p;
        const double Da = 1.0 - 2.0 * l2 * sp;
        const double DL = -2.0 * V * lam * (m0 * w0p - m1 * w1p);
        const double DV = 2.0 * l2 * a * sp / V - 2.0 * lam * (m0 * w0 - m1 * w1);
        const double g_b2n = gz + 0.5 * g;
        const double g_a = g_b2n * Da, g_L = g_b2n * DL;
        const double g_bt = g_b2n + g_a;
        const double g_p = -g_bt * b_diff;
        dL += g_L;
        dV += g_b2n * DV;
        if constexpr (DYN_R) dP += Rp * (g_p * p + g_L);
        else dP += g_p;
        gz = 0.5 * g + g_a - p * g_bt;
    }
    if (live && gz0) gz0[b] = (float)gz;
    if (!live) { dL = dV = dP = 0.0; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        dL += __shfl_down(dL, off, 64); dV += __shfl_down(dV, off, 64); dP += __shfl_down(dP, off, 64);
    }
    if (threadIdx.x == 0) {
        double* o = ws + (int64_t)blockIdx.x * 4;
        o[0] = dL; o[1] = dV; o[2] = dP; o[3] = 0.0;
    }
}

}  // namespace wdf
