// wdf_ss_step.h -- the one-pass MSE training step for LINEAR state-space trees (lpf.py:20-49,86-99: the RC lowpass with
// its ideal-source root folded in; voltage_divider.py:19-46), gfx950.
//
//     z' = A z + Bx x ;  y = cy . z + dy . x          (csrc/wdf_statespace.h with E = ca = da = fy = 0)
//
// wdf_statespace.h runs forward + stash, then a reverse sweep, and torch chains dLoss/d(coefficients) to the component
// values through the host probe (lowering.Circuit.matrices): 28 G samples/s through the element API, host-bound.  Here
// the step is the diode clipper's one-pass idea on the simpler recursion:
//
//   * the gradient is carried FORWARD: S_c = dz/dc for every state coefficient c in {A, Bx} obeys
//         S_c' = A S_c + (dz'/dc explicit) ,   dLoss/dc = sum_n g_n cy . S_c[n] ,   g_n = 2 (y_n - t_n) / N
//     (cy, dy: direct).  No stash, no second pass over the time axis, x read once per pass;
//   * chunks in time are EXACT, not speculated: the joint recursion u = (z, S_A.., S_B..) is linear,
//         u(t0 + L) = Phi u(t0) + (response of the chunk's inputs from u = 0) ,   Phi = {A^L, G_c}
//     so pass 1 runs every chunk from u = 0 (x only), one lane per sequence walks the chunks, and pass 2 runs every chunk
//     again from its exact start with the loss: 16 B per sample through HBM (x twice, target, y) -- an RC lowpass needs
//     760 steps of warm-up to forget its state, speculation would be hopeless;
//   * the chain rule to the component values happens on the device: ss_probe_kernel evaluates the probed step
//     (lib/wdf_hip/probe_tape.py: the elements' own calc_impedance / reflected / incident code, recorded once) in float64
//     with forward-mode tangents -> coefficients and their Jacobian; the step's finishing wave contracts it with the
//     coefficient gradient.  The component values never leave the device (Circuit.to_device).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wdf_statespace.h"
#include "wdf_clipper.h"      // wave_sum_dpp
#include "wdf_vec.h"
#include "wdf_optim.h"

namespace wdf {

constexpr int kProbeMaxOps = 384, kProbeMaxParams = 15, kProbeLanes = 16;   // (round 5: 7 -> 15 component values; 106 KB of the CU's 160 KB LDS)
enum { kOpConst = 0, kOpParam, kOpAdd, kOpSub, kOpMul, kOpDiv, kOpNeg, kOpRecip };

// tape: int32 [n_ops][3] = {op, a, b}; consts: double; params: the float32 block the component values live in.
// outs: node of every output (the coefficient vector, then the port resistance).  -> coef (float32 and float64) and
// jac double [n_out][n_params].  One wave: lane p carries the tangent w.r.t. parameter p (values are computed by all).
// jobs (n_jobs >= 0): optimizer updates of `params` queued by the host (wdf_optim.h, adam_clip_multi_kernel's rule) -- applied
// by this workgroup BEFORE the probe reads the values: a training step's optimizers and its probe in one launch.
static __global__ __launch_bounds__(64) void ss_probe_kernel(const int32_t* __restrict__ tape, int n_ops, const double* __restrict__ consts,
                                                             const float* params, int n_params, const int32_t* __restrict__ outs,
                                                             int n_out, float* __restrict__ coef, double* __restrict__ coef64,
                                                             double* __restrict__ jac, const AdamJobs jobs, int n_jobs)
{
    __shared__ double val[kProbeMaxOps][kProbeLanes], tan[kProbeMaxOps][kProbeLanes];
    __shared__ int ops[kProbeMaxOps * 3];
    __shared__ double leaf[kProbeMaxOps];                         // the value of every CONST / PARAM node
    // everything that does not depend on the parameters is asked for FIRST (one round trip to memory together with the
    // optimizers' operands): the tape (six words per lane in registers; longer tapes: the rest goes the plain way), the
    // first 64 output nodes
    constexpr int kPre = 6;
    int pre_ops[kPre];
#pragma unroll
    for (int r = 0; r < kPre; ++r) {
        const int i = threadIdx.x + 64 * r;
        pre_ops[r] = i < 3 * n_ops ? tape[i] : 0;
    }
    const int my_out = (int)threadIdx.x < n_out ? outs[threadIdx.x] : 0;
    if (n_jobs > 0) {                                            // eight lanes per job, all jobs side by side
        const int jb = threadIdx.x >> 3, li = threadIdx.x & 7;
        const bool mine = jb < n_jobs;
        const AdamJob q = jobs.j[mine ? jb : 0];
        const int t = *q.step + 1;
        if (mine) {
            const double c1 = 1.0 - ipow((double)q.b1, t), c2 = 1.0 - ipow((double)q.b2, t);
            for (int i = li; i < q.n; i += 8) {
                const float g = q.grad[i];
                const float mi = q.b1 * q.m[i] + (1.0f - q.b1) * g;
                const float vi = q.b2 * q.v[i] + (1.0f - q.b2) * g * g;
                q.m[i] = mi;
                q.v[i] = vi;
                const float lr_t = (float)((double)q.lr[i] * sqrt(c2) / c1);
                float th = q.theta[i] - lr_t * mi / (sqrtf(vi) + q.eps);
                if (q.lo) th = fmaxf(th, q.lo[i]);
                if (q.hi) th = fminf(th, q.hi[i]);
                __hip_atomic_store(q.theta + i, th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();                                          // (every lane has read its job's step count)
        if (mine && li == 0) *q.step = t;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the new values are out before anyone asks for them)
    }
#pragma unroll
    for (int r = 0; r < kPre; ++r) {
        const int i = threadIdx.x + 64 * r;
        if (i < 3 * n_ops) ops[i] = pre_ops[r];
    }
    for (int i = threadIdx.x + 64 * kPre; i < 3 * n_ops; i += 64) ops[i] = tape[i];
    __syncthreads();
    for (int i = threadIdx.x; i < n_ops; i += 64) {               // (all leaves fetched at once: no dependent global loads below)
        const int op = ops[3 * i], a = ops[3 * i + 1];
        leaf[i] = op == kOpConst ? consts[a] : (op == kOpParam ? (double)__hip_atomic_load(params + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0);
    }
    __syncthreads();
    const int p = threadIdx.x;
    if (p < kProbeLanes) {
        int nop = ops[0], na = ops[1], nb = ops[2];
        for (int i = 0; i < n_ops; ++i) {
            const int op = nop, a = na, b = nb;
            if (i + 1 < n_ops) { nop = ops[3 * i + 3]; na = ops[3 * i + 4]; nb = ops[3 * i + 5]; }   // (the next instruction is fetched under this one)
            double v = 0.0, t = 0.0;
            switch (op) {
            case kOpConst: v = leaf[i]; break;
            case kOpParam: v = leaf[i]; t = (a == p) ? 1.0 : 0.0; break;
            case kOpAdd: v = val[a][p] + val[b][p]; t = tan[a][p] + tan[b][p]; break;
            case kOpSub: v = val[a][p] - val[b][p]; t = tan[a][p] - tan[b][p]; break;
            case kOpMul: v = val[a][p] * val[b][p]; t = tan[a][p] * val[b][p] + val[a][p] * tan[b][p]; break;
            case kOpDiv: { const double q = val[a][p] / val[b][p]; v = q; t = (tan[a][p] - q * tan[b][p]) / val[b][p]; break; }
            case kOpNeg: v = -val[a][p]; t = -tan[a][p]; break;
            case kOpRecip: { const double r = 1.0 / val[a][p]; v = r; t = -r * r * tan[a][p]; break; }
            }
            val[i][p] = v;
            tan[i][p] = t;
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < n_out; o += 64) {               // lane o: output o and its row of the Jacobian
        const int nd = o < 64 ? my_out : outs[o];
        coef[o] = (float)val[nd][0];
        coef64[o] = val[nd][0];
        for (int q = 0; q < n_params; ++q) jac[o * n_params + q] = tan[nd][q];
    }
}

// ---- the joint recursion ------------------------------------------------------------------------------------------------
// V = float: one sequence per lane; V = v2f: two adjacent sequences (8-byte loads and stores: half the memory instructions)
template <typename V> __device__ __forceinline__ V lin_ld(const float* p) { return *reinterpret_cast<const V*>(p); }
template <typename V> __device__ __forceinline__ void lin_st(float* p, V v) { *reinterpret_cast<V*>(p) = v; }
template <typename V> __device__ __forceinline__ void lin_st_nt(float* p, V v) { __builtin_nontemporal_store(v, reinterpret_cast<V*>(p)); }
__device__ __forceinline__ double lin_hsum(float v) { return (double)v; }
__device__ __forceinline__ double lin_hsum(v2f v) { return (double)v.x + (double)v.y; }

template <int NS, int NI, typename V>
struct LinU {
    static constexpr int nA = NS * NS, nB = NS * NI;
    static constexpr int kD = NS * (1 + nA + nB);                // floats per sequence: z, S_A[nA][NS], S_B[nB][NS]
    static constexpr int kG = nA + nB + NS + NI;                 // gradient entries: A, Bx, cy, dy
    V z[NS > 0 ? NS : 1];
    V SA[nA > 0 ? nA : 1][NS > 0 ? NS : 1];
    V SB[nB > 0 ? nB : 1][NS > 0 ? NS : 1];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int s = 0; s < NS; ++s) z[s] = vsplat<V>(0.0f);
#pragma unroll
        for (int c = 0; c < nA; ++c)
#pragma unroll
            for (int s = 0; s < NS; ++s) SA[c][s] = vsplat<V>(0.0f);
#pragma unroll
        for (int c = 0; c < nB; ++c)
#pragma unroll
            for (int s = 0; s < NS; ++s) SB[c][s] = vsplat<V>(0.0f);
    }
    // [kD][B] planes at p (p already points at this lane's column)
    __device__ __forceinline__ void store(float* __restrict__ p, int64_t B) const
    {
        int o = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) lin_st<V>(p + (o++) * B, z[s]);
#pragma unroll
        for (int c = 0; c < nA; ++c)
#pragma unroll
            for (int s = 0; s < NS; ++s) lin_st<V>(p + (o++) * B, SA[c][s]);
#pragma unroll
        for (int c = 0; c < nB; ++c)
#pragma unroll
            for (int s = 0; s < NS; ++s) lin_st<V>(p + (o++) * B, SB[c][s]);
    }
    __device__ __forceinline__ void load(const float* __restrict__ p, int64_t B)
    {
        int o = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) z[s] = lin_ld<V>(p + (o++) * B);
#pragma unroll
        for (int c = 0; c < nA; ++c)
#pragma unroll
            for (int s = 0; s < NS; ++s) SA[c][s] = lin_ld<V>(p + (o++) * B);
#pragma unroll
        for (int c = 0; c < nB; ++c)
#pragma unroll
            for (int s = 0; s < NS; ++s) SB[c][s] = lin_ld<V>(p + (o++) * B);
    }
};

// one step of u = (z, S_A, S_B) with input x
template <int NS, int NI, typename V>
__device__ __forceinline__ void lin_u_step(const SSCoef<NS, NI>& c, const V (&x)[NI], LinU<NS, NI, V>& u)
{
    using C = SSCoef<NS, NI>;
    const V zero = vsplat<V>(0.0f);
    V zn[NS > 0 ? NS : 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        V a = zero;
#pragma unroll
        for (int i = 0; i < NI; ++i) a = vfma(c.v[C::oB + s * NI + i], x[i], a);
#pragma unroll
        for (int q = 0; q < NS; ++q) a = vfma(c.v[C::oA + s * NS + q], u.z[q], a);
        zn[s] = a;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) {                            // S_{A_ij}' = A S + e_i z_j
            V sn[NS > 0 ? NS : 1];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                V a = (s == i) ? u.z[j] : zero;
#pragma unroll
                for (int q = 0; q < NS; ++q) a = vfma(c.v[C::oA + s * NS + q], u.SA[i * NS + j][q], a);
                sn[s] = a;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) u.SA[i * NS + j][s] = sn[s];
        }
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {                            // S_{Bx_ij}' = A S + e_i x_j
            V sn[NS > 0 ? NS : 1];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                V a = (s == i) ? x[j] : zero;
#pragma unroll
                for (int q = 0; q < NS; ++q) a = vfma(c.v[C::oA + s * NS + q], u.SB[i * NI + j][q], a);
                sn[s] = a;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) u.SB[i * NI + j][s] = sn[s];
        }
#pragma unroll
    for (int s = 0; s < NS; ++s) u.z[s] = zn[s];
}

constexpr int kLinBlk = 8;                                         // steps whose loads are issued together

// x: [T][NI][B] (time-major: the engine's resident training set); loads kLinBlk steps of lane b
template <int NI, typename V>
__device__ __forceinline__ void lin_load_x(const float* __restrict__ x, int64_t B, int64_t b, int64_t t, int n, V (&xs)[kLinBlk][NI])
{
#pragma unroll
    for (int k = 0; k < kLinBlk; ++k)
#pragma unroll
        for (int i = 0; i < NI; ++i) xs[k][i] = (k < n) ? lin_ld<V>(x + ((t + k) * NI + i) * B + b) : vsplat<V>(0.0f);
}

template <typename V> struct LinWidth { static constexpr int w = 1; };
template <> struct LinWidth<v2f> { static constexpr int w = 2; };

// ---- pass 1: every chunk from u = 0 -> uend0 [K][kD][B] ------------------------------------------------------------------
// (V = v2f: B even; a lane holds sequences b, b + 1)
template <int NS, int NI, typename V>
__global__ __launch_bounds__(64) void ss_lin_step_zero_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                              float* __restrict__ uend0, int64_t B, int64_t T, int64_t L)
{
    constexpr int W = LinWidth<V>::w;
    const int64_t b_raw = ((int64_t)blockIdx.x * 64 + threadIdx.x) * W;
    const int64_t b = b_raw < B ? b_raw : B - W;
    const int64_t k = blockIdx.y, t0 = k * L, t1 = (t0 + L < T) ? t0 + L : T;
    if (t1 == T) return;                                          // (nothing comes after the last chunk)
    SSCoef<NS, NI> c;
    c.load(coef);
    LinU<NS, NI, V> u;
    u.zero();
    for (int64_t tb = t0; tb < t1; tb += kLinBlk) {
        const int n = t1 - tb < kLinBlk ? (int)(t1 - tb) : kLinBlk;
        V xs[kLinBlk][NI];
        lin_load_x<NI, V>(x, B, b, tb, n, xs);
#pragma unroll
        for (int i = 0; i < kLinBlk; ++i)
            if (i < n) lin_u_step<NS, NI, V>(c, xs[i], u);
    }
    if (b_raw < B) u.store(uend0 + (k * LinU<NS, NI, V>::kD) * B + b, B);
}

// u <- the exact joint state at the start of chunk k (zero initial state), from the zero-state chunk ends uend0 [K][kD][B]
// Phi: a chunk run from z = e_j (S = 0, no input) ends in z = A^L e_j, S_{A c} = G_c e_j; S_B's homogeneous part is A^L too.
template <int NS, int NI, typename V>
__device__ __forceinline__ void lin_walk_to(const SSCoef<NS, NI>& c, const float* __restrict__ uend0, int64_t B, int64_t b,
                                            int64_t k, int64_t L, LinU<NS, NI, V>& u)
{
    using U = LinU<NS, NI, V>;
    using H = LinU<NS, NI, float>;
    float AL[NS > 0 ? NS : 1][NS > 0 ? NS : 1], G[U::nA > 0 ? U::nA : 1][NS > 0 ? NS : 1][NS > 0 ? NS : 1];
    const float zero_x[NI] = {};
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        H h;
        h.zero();
        h.z[j] = 1.0f;
        for (int64_t t = 0; t < L; ++t) lin_u_step<NS, NI, float>(c, zero_x, h);
#pragma unroll
        for (int s = 0; s < NS; ++s) AL[s][j] = h.z[s];
#pragma unroll
        for (int cc = 0; cc < U::nA; ++cc)
#pragma unroll
            for (int s = 0; s < NS; ++s) G[cc][s][j] = h.SA[cc][s];
    }
    U p;
    p.load(uend0 + b, B);
    for (int64_t q = 0; q < k; ++q) {
        U pn = p;
        if (q + 1 < k) pn.load(uend0 + ((q + 1) * U::kD) * B + b, B);   // (the next chunk's end is on its way while this one is applied)
        U n;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            V a = p.z[s];
#pragma unroll
            for (int r = 0; r < NS; ++r) a = vfma(AL[s][r], u.z[r], a);
            n.z[s] = a;
        }
#pragma unroll
        for (int cc = 0; cc < U::nA; ++cc)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                V a = p.SA[cc][s];
#pragma unroll
                for (int r = 0; r < NS; ++r) a = vfma(AL[s][r], u.SA[cc][r], vfma(G[cc][s][r], u.z[r], a));
                n.SA[cc][s] = a;
            }
#pragma unroll
        for (int cc = 0; cc < U::nB; ++cc)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                V a = p.SB[cc][s];
#pragma unroll
                for (int r = 0; r < NS; ++r) a = vfma(AL[s][r], u.SB[cc][r], a);
                n.SB[cc][s] = a;
            }
        u = n;
        p = pn;
    }
}

// ---- pass 2: every chunk from its exact start: y, the squared error, the coefficient gradient; the last wave finishes ------
// The chunk's exact start is made by the wave itself: Phi = {A^L, G_c} from L homogeneous steps, then the walk over the chunks
// before it (their zero-state ends: pass 1) -- no separate walk launch.
// part: double [waves][kG + 1] = {gA.., gBx.., gcy.., gdy.., SSE}; ticket: one word, left 0.
// jac: double [ncoef (+1)][n_params] of ss_probe_kernel (rows in SSCoef order).  out: float [1 + n_params] = {SSE, dLoss/dparam}.
#ifndef WDF_LIN_BLK
#define WDF_LIN_BLK 8
#endif
constexpr int kLinBlk2 = WDF_LIN_BLK;                            // pass 2's block (steps whose loads are issued together)
template <int NI, typename V>
__device__ __forceinline__ void lin_load_x2(const float* __restrict__ x, int64_t B, int64_t b, int64_t t, int n, V (&xs)[kLinBlk2][NI])
{
#pragma unroll
    for (int k = 0; k < kLinBlk2; ++k)
#pragma unroll
        for (int i = 0; i < NI; ++i) xs[k][i] = (k < n) ? lin_ld<V>(x + ((t + k) * NI + i) * B + b) : vsplat<V>(0.0f);
}

template <int NS, int NI, typename V>
__global__ __launch_bounds__(64) void ss_lin_step_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                         const float* __restrict__ uend0, const float* __restrict__ target,
                                                         float gscale, float* __restrict__ y, double* __restrict__ part,
                                                         unsigned* __restrict__ ticket, const double* __restrict__ jac, int n_params,
                                                         float* __restrict__ out, float* __restrict__ loss_out, float* __restrict__ gcoef_out, int64_t B, int64_t T,
                                                         int64_t L, const float* __restrict__ z0, float* __restrict__ zT)
{
    using U = LinU<NS, NI, V>;
    using C = SSCoef<NS, NI>;
    constexpr int W = LinWidth<V>::w;
    const V zero = vsplat<V>(0.0f);
    const int64_t b_raw = ((int64_t)blockIdx.x * 64 + threadIdx.x) * W;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - W;
    const int64_t k = blockIdx.y, t0 = k * L, t1 = (t0 + L < T) ? t0 + L : T;
    SSCoef<NS, NI> c;
    c.load(coef);
    U u;
    u.zero();
    // z0 [NS][B] (or NULL): the capacitor states this call starts from -- lpf.py:30-49 never resets C1, so an epoch starts where
    // the last one ended; a constant of this call's tape (the tangents start at 0), as the reference's stored tensor is
    if (NS > 0 && z0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) u.z[s] = lin_ld<V>(z0 + s * B + b);
    }
    if (NS > 0 && k > 0) lin_walk_to<NS, NI, V>(c, uend0, B, b, k, L, u);
    const float gs = live ? gscale : 0.0f, lv = live ? 1.0f : 0.0f;
    double acc[U::kG + 1];
#pragma unroll
    for (int i = 0; i <= U::kG; ++i) acc[i] = 0.0;
    // 8-step blocks, the next one's loads in flight under this one's arithmetic (a wave that loads, waits, computes keeps
    // half as many bytes on their way)
    V xn[kLinBlk2][NI], tn[kLinBlk2];
    auto load_blk = [&](int64_t ts) {
        const int n = t1 - ts < kLinBlk2 ? (int)(t1 - ts) : kLinBlk2;
        lin_load_x2<NI, V>(x, B, b, ts, n, xn);
#pragma unroll
        for (int i = 0; i < kLinBlk2; ++i) tn[i] = (i < n) ? lin_ld<V>(target + (ts + i) * B + b) : zero;
    };
    load_blk(t0);
    for (int64_t tb = t0; tb < t1; tb += 32) {                     // fp32 sums within 32 steps, fp64 across
        V f[U::kG + 1];
#pragma unroll
        for (int i = 0; i <= U::kG; ++i) f[i] = zero;
#pragma unroll 1
        for (int64_t ts = tb; ts < tb + 32 && ts < t1; ts += kLinBlk2) {
            const int n = t1 - ts < kLinBlk2 ? (int)(t1 - ts) : kLinBlk2;
            V xs[kLinBlk2][NI], tg[kLinBlk2];
#pragma unroll
            for (int i = 0; i < kLinBlk2; ++i) {
                tg[i] = tn[i];
#pragma unroll
                for (int j = 0; j < NI; ++j) xs[i][j] = xn[i][j];
            }
            if (ts + kLinBlk2 < t1) load_blk(ts + kLinBlk2);
#pragma unroll
            for (int i = 0; i < kLinBlk2; ++i) {
                if (i >= n) break;
                V yv = zero;
#pragma unroll
                for (int j = 0; j < NI; ++j) yv = vfma(c.v[C::oDy + j], xs[i][j], yv);
#pragma unroll
                for (int s = 0; s < NS; ++s) yv = vfma(c.v[C::oCy + s], u.z[s], yv);
                const V e = yv - tg[i];
                const V g = e * gs;
                if (live) lin_st_nt<V>(y + (ts + i) * B + b, yv);
                f[U::kG] = vfma(e * lv, e, f[U::kG]);
#pragma unroll
                for (int cc = 0; cc < U::nA; ++cc) {
                    V d = zero;
#pragma unroll
                    for (int s = 0; s < NS; ++s) d = vfma(c.v[C::oCy + s], u.SA[cc][s], d);
                    f[cc] = vfma(g, d, f[cc]);
                }
#pragma unroll
                for (int cc = 0; cc < U::nB; ++cc) {
                    V d = zero;
#pragma unroll
                    for (int s = 0; s < NS; ++s) d = vfma(c.v[C::oCy + s], u.SB[cc][s], d);
                    f[U::nA + cc] = vfma(g, d, f[U::nA + cc]);
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) f[U::nA + U::nB + s] = vfma(g, u.z[s], f[U::nA + U::nB + s]);
#pragma unroll
                for (int j = 0; j < NI; ++j) f[U::nA + U::nB + NS + j] = vfma(g, xs[i][j], f[U::nA + U::nB + NS + j]);
                lin_u_step<NS, NI, V>(c, xs[i], u);
            }
        }
#pragma unroll
        for (int i = 0; i <= U::kG; ++i) acc[i] += lin_hsum(f[i]);
    }
    if (NS > 0 && zT && t1 == T && live) {                          // the states the call ends in [NS][B] (never the z0 buffer)
#pragma unroll
        for (int s = 0; s < NS; ++s) lin_st<V>(zT + s * B + b, u.z[s]);
    }
    // ---- this wave's partial; the last wave of the launch adds them up in a fixed order and applies the chain rule
    const int64_t wave = (int64_t)blockIdx.y * gridDim.x + blockIdx.x, nwaves = (int64_t)gridDim.x * gridDim.y;
#pragma unroll
    for (int i = 0; i <= U::kG; ++i) {
        const double s = wave_sum_dpp(acc[i]);
        if (threadIdx.x == 0) __hip_atomic_store(part + wave * (U::kG + 1) + i, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned old = 0;
    if (threadIdx.x == 0) old = atomicAdd(ticket, 1u);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != (unsigned)(nwaves - 1)) return;
    if (threadIdx.x == 0) *ticket = 0u;
    double tot[U::kG + 1];
#pragma unroll
    for (int i = 0; i <= U::kG; ++i) {
        double s = 0.0;
        for (int64_t w = threadIdx.x; w < nwaves; w += 64)
            s += __hip_atomic_load(part + w * (U::kG + 1) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot[i] = wave_sum_dpp(s);
    }
    // coefficient index (SSCoef order) of gradient entry i
    const int p = threadIdx.x;
    if (p < n_params) {
        double gp = 0.0;
#pragma unroll
        for (int i = 0; i < U::kG; ++i) {
            const int row = i < U::nA ? C::oA + i : (i < U::nA + U::nB ? C::oB + (i - U::nA)
                                                     : (i < U::nA + U::nB + NS ? C::oCy + (i - U::nA - U::nB) : C::oDy + (i - U::nA - U::nB - NS)));
            gp += tot[i] * jac[row * n_params + p];
        }
        out[1 + p] = (float)gp;
    }
    if (p == 0) {
        out[0] = (float)tot[U::kG];
        if (loss_out) *loss_out = (float)(0.5 * (double)gscale * tot[U::kG]);   // the mean squared error when gscale = 2 / (B T)
        if (gcoef_out) {
#pragma unroll
            for (int i = 0; i < U::kG; ++i) gcoef_out[i] = (float)tot[i];
        }
    }
}

}  // namespace wdf
