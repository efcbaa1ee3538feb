// wdf_elementwise.h -- kernels without a time recursion: the MSE + ESR loss sums and coefficients,
// and the element-wise Wright omega / diode-pair evaluations the parity tests and the pre-training
// table use.  gfx950.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wdf_omega.h"

namespace wdf {

// ---- MSE + ESR loss (clipper_pot.py:146-156,177) ---------------------------------------------
// Sums over the samples past skip_samples of one rank's [T][B] arrays: S = sum (y - t)^2 and
// E = sum y^2 (the script passes (outs, target) as (target_y, predicted_y), :248, so the energy is
// the model output's).  Grid-stride, per-block partials in double, fixed-order finish.
static __global__ __launch_bounds__(256) void loss_sums_kernel(const float* __restrict__ y, const float* __restrict__ target,
                                                        int64_t n0, int64_t n1, double* __restrict__ part)
{
    __shared__ double sh[256][2];
    double s = 0.0, e = 0.0;
    for (int64_t i = n0 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n1; i += (int64_t)gridDim.x * 256) {
        const float yv = y[i], d = yv - target[i];
        s += (double)(d * d);
        e += (double)(yv * yv);
    }
    sh[threadIdx.x][0] = s; sh[threadIdx.x][1] = e;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { sh[threadIdx.x][0] += sh[threadIdx.x + off][0]; sh[threadIdx.x][1] += sh[threadIdx.x + off][1]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = sh[0][0]; part[2 * blockIdx.x + 1] = sh[0][1]; }
}

static __global__ __launch_bounds__(256) void loss_sums_finish_kernel(const double* __restrict__ part, int nblk,
                                                               double* __restrict__ sums)
{
    __shared__ double sh[256][2];
    double s = 0.0, e = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) { s += part[2 * i]; e += part[2 * i + 1]; }
    sh[threadIdx.x][0] = s; sh[threadIdx.x][1] = e;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { sh[threadIdx.x][0] += sh[threadIdx.x + off][0]; sh[threadIdx.x][1] += sh[threadIdx.x + off][1]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sums[0] = sh[0][0]; sums[1] = sh[0][1]; }
}

// From the (global) sums: loss = S/n + sqrt(S / (E + eps) / n) and its derivative w.r.t. y,
//   dL/dy_i = ga (y_i - t_i) + gb y_i ,  ga = 2/n + 1/(esr (E+eps) n) ,  gb = -esr / (E+eps).
static __global__ void esr_coef_kernel(const double* __restrict__ sums, double n, double eps, float* __restrict__ gcoef,
                                float* __restrict__ loss)
{
    const double S = sums[0], E = sums[1] + eps;
    const double mse = S / n, esr = sqrt(S / E / n);
    gcoef[0] = (float)(2.0 / n + (esr > 0.0 ? 1.0 / (esr * E * n) : 0.0));
    gcoef[1] = (float)(-esr / E);
    loss[0] = (float)mse; loss[1] = (float)esr; loss[2] = (float)(mse + esr);
}

// dL/dy [T][B] of that loss for a reverse sweep that takes an upstream gradient (the MLP-root sweep): zero before
// skip, ga (y - t) + gb y after, with {ga, gb} read from the device (esr_coef_kernel's output).
static __global__ __launch_bounds__(256) void loss_esr_grad_kernel(const float* __restrict__ y, const float* __restrict__ target,
                                                                   const float* __restrict__ gcoef, int64_t n0, int64_t n1,
                                                                   float* __restrict__ gy)
{
    const float ga = gcoef[0], gb = gcoef[1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n1; i += (int64_t)gridDim.x * 256) {
        const float yv = y[i];
        gy[i] = i < n0 ? 0.0f : fmaf(ga, yv - target[i], gb * yv);
    }
}

// ---- element-wise building blocks (parity tests) ----------------------------------------
static __global__ void omega_kernel(const float* __restrict__ x, float* __restrict__ w, int32_t* __restrict__ iters, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float xi = x[i < n ? i : n - 1];
    int it = 0;
    const float wi = wright_omega<true>(xi, &it);
    if (i < n) { w[i] = wi; if (iters) iters[i] = it; }
}

static __global__ void diode_pair_kernel(const float* __restrict__ a, const float* __restrict__ Rp, float Is, float nVt,
                                  int n_up, int n_down, float* __restrict__ b, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = i < n ? i : n - 1;
    const DiodeStatic d = make_diode_static(nVt, n_up, n_down);
    const float L = logf(Rp[j] * Is / nVt);
    const DiodeOut o = diode_pair<false, float>(a[j], L, d);
    if (i < n) b[i] = o.b;
}

}  // namespace wdf
