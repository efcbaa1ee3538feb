// wdf_optim.h -- the optimizer step of the training loops, on the device.
//
// tf.keras.optimizers.Adam.apply_gradients as the reference scripts use it (lpf.py:79-80,93-94:
// one optimizer per component with its own learning rate; clipper_pot.py:179,183-184), followed
// by the Variable's constraint (tf_wdf.py:74 R in [180, 1e6], :104 C in [1e-13, 1]), which Keras
// applies after every update.  Update rule (TF 2.5 Adam, non-amsgrad):
//     t += 1 ; lr_t = lr sqrt(1 - b2^t) / (1 - b1^t)
//     m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; theta -= lr_t m / (sqrt(v) + eps)
// The component values live on the device (the kernels read them there), so the whole training
// step -- forward, reverse sweep, all-reduce, update -- runs without a host round trip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wdf {

// b^t for a step count t >= 0 by repeated squaring: a few fp64 multiplies instead of pow()'s hundreds of instructions
__device__ __forceinline__ double ipow(double b, int t)
{
    double r = 1.0;
    for (unsigned n = (unsigned)t; n != 0u; n >>= 1) {
        if (n & 1u) r *= b;
        b *= b;
    }
    return r;
}

// One thread per parameter; `step` is the shared iteration counter (read by all, bumped by thread 0
// after a barrier -- n <= 1024, one block).
static __global__ __launch_bounds__(1024) void adam_clip_kernel(float* __restrict__ theta, const float* __restrict__ grad,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         int32_t* __restrict__ step, const float* __restrict__ lr,
                                                         float b1, float b2, float eps, const float* __restrict__ lo,
                                                         const float* __restrict__ hi, int n)
{
    const int i = threadIdx.x;
    const int t = *step + 1;
    __syncthreads();
    if (i == 0) *step = t;
    if (i >= n) return;
    const double c1 = 1.0 - ipow((double)b1, t), c2 = 1.0 - ipow((double)b2, t);
    const float g = grad[i];
    const float mi = b1 * m[i] + (1.0f - b1) * g;
    const float vi = b2 * v[i] + (1.0f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    const float lr_t = (float)((double)lr[i] * sqrt(c2) / c1);
    float th = theta[i] - lr_t * mi / (sqrtf(vi) + eps);
    if (lo) th = fmaxf(th, lo[i]);
    if (hi) th = fminf(th, hi[i]);
    theta[i] = th;
}

// Several independent optimizers in ONE launch (lpf.py:79-80,93-94 keeps one Adam per component: five launches a step for
// the HPF clipper's five components when each goes out on its own).  One workgroup per job, adam_clip_kernel's rule.
constexpr int kAdamMultiMax = 8;
struct AdamJob {
    float* theta; const float* grad; float* m; float* v; int32_t* step; const float* lr; const float* lo; const float* hi;
    float b1, b2, eps; int n;
};
struct AdamJobs { AdamJob j[kAdamMultiMax]; };

static __global__ __launch_bounds__(256) void adam_clip_multi_kernel(const AdamJobs jobs)
{
    const AdamJob& q = jobs.j[blockIdx.x];
    const int t = *q.step + 1;
    __syncthreads();
    if (threadIdx.x == 0) *q.step = t;
    const double c1 = 1.0 - ipow((double)q.b1, t), c2 = 1.0 - ipow((double)q.b2, t);
    for (int i = threadIdx.x; i < q.n; i += blockDim.x) {
        const float g = q.grad[i];
        const float mi = q.b1 * q.m[i] + (1.0f - q.b1) * g;
        const float vi = q.b2 * q.v[i] + (1.0f - q.b2) * g * g;
        q.m[i] = mi;
        q.v[i] = vi;
        const float lr_t = (float)((double)q.lr[i] * sqrt(c2) / c1);
        float th = q.theta[i] - lr_t * mi / (sqrtf(vi) + q.eps);
        if (q.lo) th = fmaxf(th, q.lo[i]);
        if (q.hi) th = fminf(th, q.hi[i]);
        q.theta[i] = th;
    }
}

}  // namespace wdf
