// wdf_mlp.h -- diode clipper with the tanh-MLP root of clipper_pot.py, gfx950.
//
// Circuit (clipper_pot.py:94-127):  P1 = Parallel(ResistiveVoltageSource(R or per-sample r),
// Capacitor(C, fs)); the root is DenseRootModel (layers.py:42-82) fed (a, log P1.R), and the
// wave sent back down is -MLP(a, log R) (clipper_pot.py:119-121).  Per step:
//     b_diff = z - x ; b_temp = -p b_diff ; a = z + b_temp          Parallel.reflected  tf_wdf.py:185-192
//     b = -MLP(a, log Rp)                                           layers.py:76-82, DenseLayer :38-39
//     z' = b + b_temp ; y = (z' + z)/2                              tf_wdf.py:179-183, :8-10
// One lane per sequence; the network is evaluated per lane in VGPRs.  The weights (609 floats
// for a 2x16 net) are staged once per wave into LDS and read from there with wave-uniform
// addresses (broadcast ds_read, immediate offsets): left in global memory, LLVM hoists all
// of them out of the time loop into SGPRs and spills them to VGPR lanes (measured 6300
// v_readlane per step, 12x slower than the FLOPs warrant).  Networks: 2 -> H -> ... -> H -> 1 with NL
// tanh layers, H in {4, 8, 16}, NL in {3, 5}: the "2xH" and "4xH" families of
// wdf_py/diode_clipper/models (n_layers + 1 hidden layers, diode_pretraining.py:113-126).
// Weight layout (the JSON order, layers.py:31-36): per layer kernel[in][out] row-major, then
// bias[out].
//
// Reverse sweep: the per-lane kernel does the scalar adjoint recurrence and the input-
// Jacobian of the network (d out / d a, d out / d log R); it writes g_b[n] = dL/d b[n] and the
// network inputs (a[n], log R[n]).  The weight gradient is then an ordinary dense problem,
// dL/dW = -sum_n g_b[n] d MLP(a[n], lr[n]) / dW over B*T independent samples, which the host
// runs as plain GEMMs (torch -> hipBLASLt); doing it per lane would need 609 accumulators
// per lane.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wdf_clipper.h"

namespace wdf {

__device__ __forceinline__ float tanh_fast(float x)
{
    const float t = __builtin_amdgcn_exp2f(-2.0f * kLog2e * fabsf(x));     // exp(-2|x|) in (0, 1]
    const float r = (1.0f - t) * fast_rcp(1.0f + t);
    return copysignf(r, x);
}

template <int H, int NL>
struct Mlp {
    static constexpr int kW0 = 0;                       // kernel0 [2][H]
    static constexpr int kB0 = 2 * H;                   // bias0 [H]
    static constexpr int kMid = 3 * H;                  // then (NL-1) x { kernel [H][H], bias [H] }
    static constexpr int kMidStride = H * H + H;
    static constexpr int kWo = kMid + (NL - 1) * kMidStride;   // kernel_out [H][1]
    static constexpr int kBo = kWo + H;                 // bias_out [1]
    static constexpr int kCount = kBo + 1;

    // Weights live in LDS (staged once per wave) and are consumed ROW-wise: row i of a layer's
    // kernel[in][out] is H contiguous floats, fetched with H/4 ds_read_b128 (wave-uniform address:
    // broadcast) one row ahead of the H FMAs that use it.  A scheduling fence per row keeps the
    // scheduler from lifting every weight read of the unrolled network to the top of the step
    // (which needs 600 VGPRs) and keeps the next row's reads in flight under this row's FMAs.
    static __device__ __forceinline__ void load_row(const float* __restrict__ p, float (&r)[H])
    {
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
            const float4 v = reinterpret_cast<const float4*>(p)[q];
            r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
        }
    }
    // The same read, but not before `dep` has been computed: the offset goes through an empty asm
    // that also names dep, so the LDS reads of row i+1 cannot be lifted above the FMAs of row i-1.
    // (sched_barrier alone does not hold them: instruction selection is free to place loads with
    // constant addresses at the top of the block, and a 2x16 step then spills 120 VGPRs.)
    // Measured (1340 x 2048, per-sample R): 2x16 forward 12.0 -> 5.4 ms with the fence; the H <= 8
    // nets have registers to spare and lose 20% of their prefetch distance to it, so they go without.
    static constexpr bool kFence = H > 8;
    static __device__ __forceinline__ void load_row_after(const float* __restrict__ w, int off, float dep, float (&r)[H])
    {
        if constexpr (kFence) asm volatile("" : "+v"(off) : "v"(dep));
        load_row(w + off, r);
    }
    // ... not before ALL of dep[0..H) (one dependence would let the other H-1 FMAs of the row be
    // deferred behind every later load)
    static __device__ __forceinline__ void load_row_after(const float* __restrict__ w, int off, const float (&dep)[H],
                                                          float (&r)[H])
    {
        static_assert(!kFence || H == 16, "write the operand list for this width");
        if constexpr (kFence)
            asm volatile("" : "+v"(off) : "v"(dep[0]), "v"(dep[1]), "v"(dep[2]), "v"(dep[3]), "v"(dep[4]), "v"(dep[5]),
                         "v"(dep[6]), "v"(dep[7]), "v"(dep[8]), "v"(dep[9]), "v"(dep[10]), "v"(dep[11]), "v"(dep[12]),
                         "v"(dep[13]), "v"(dep[14]), "v"(dep[15]));
        load_row(w + off, r);
    }

    // out = MLP(a, lr); act[l][i] keeps the tanh outputs of layer l
    static __device__ __forceinline__ float fwd(const float* __restrict__ w, float a, float lr, float (&act)[NL][H])
    {
        float acc[H], kr[H], kn[H];
        // layer 0: acc = bias0 + a k0[0][:] + lr k0[1][:]
        load_row(w + kB0, acc);
        load_row(w + kW0, kr);
        load_row(w + kW0 + H, kn);
#pragma unroll
        for (int o = 0; o < H; ++o) acc[o] = fmaf(lr, kn[o], fmaf(a, kr[o], acc[o]));
        load_row_after(w, kMid, acc, kr);                     // first row of the next layer (or of the output layer)
#pragma unroll
        for (int o = 0; o < H; ++o) act[0][o] = tanh_fast(acc[o]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            load_row_after(w, kMid + (l - 1) * kMidStride + H * H, act[l - 1], acc);   // bias
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const int koff = kMid + (l - 1) * kMidStride;
                if (i + 1 < H) load_row_after(w, koff + (i + 1) * H, acc, kn);
                else load_row_after(w, koff + kMidStride, acc, kn);   // next layer's first row / the output kernel
#pragma unroll
                for (int o = 0; o < H; ++o) acc[o] = fmaf(act[l - 1][i], kr[o], acc[o]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int o = 0; o < H; ++o) kr[o] = kn[o];
            }
#pragma unroll
            for (int o = 0; o < H; ++o) act[l][o] = tanh_fast(acc[o]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // kr now holds the output kernel [H][1] (contiguous right after the last hidden layer)
        float out = w[kBo];
#pragma unroll
        for (int i = 0; i < H; ++i) out = fmaf(act[NL - 1][i], kr[i], out);
        return out;
    }

    // d out / d a and d out / d lr from the kept activations:
    //   d_L = Wo (.) (1 - h_L^2) ;  d_{l-1}[j] = (sum_i W_l[j][i] d_l[i]) (1 - h_{l-1}[j]^2) ;  row j of W_l is contiguous
    static __device__ __forceinline__ void grad_in(const float* __restrict__ w, const float (&act)[NL][H], float& da,
                                                   float& dlr)
    {
        float d[H], dn[H], kr[H], kn[H];
        load_row(w + kWo, kr);
#pragma unroll
        for (int i = 0; i < H; ++i) d[i] = kr[i] * fmaf(-act[NL - 1][i], act[NL - 1][i], 1.0f);
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            const int koff = kMid + (l - 1) * kMidStride;
            load_row_after(w, koff, d[0], kr);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < H; ++j) {
                if (j + 1 < H) load_row_after(w, koff + (j + 1) * H, j == 0 ? d[0] : dn[j - 1], kn);
                // four partial sums: a 16-long dependent FMA chain would cost 16 x 7 cycles
                float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
                for (int i = 0; i < H; i += 4) {
                    s0 = fmaf(kr[i], d[i], s0); s1 = fmaf(kr[i + 1], d[i + 1], s1);
                    s2 = fmaf(kr[i + 2], d[i + 2], s2); s3 = fmaf(kr[i + 3], d[i + 3], s3);
                }
                dn[j] = ((s0 + s1) + (s2 + s3)) * fmaf(-act[l - 1][j], act[l - 1][j], 1.0f);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < H) {
#pragma unroll
                    for (int i = 0; i < H; ++i) kr[i] = kn[i];
                }
            }
#pragma unroll
            for (int j = 0; j < H; ++j) d[j] = dn[j];
        }
        load_row(w + kW0, kr);
        load_row(w + kW0 + H, kn);
        da = 0.0f;
        dlr = 0.0f;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            da = fmaf(kr[i], d[i], da);
            dlr = fmaf(kn[i], d[i], dlr);
        }
    }

    // Weight-gradient accumulation for ONE sample: acc += dout * d MLP / d W restricted to `part`
    // (wave-uniform).  part 0: acc[0..3H) = kernel0 rows (a, lr) + bias0, acc[3H..4H] = output
    // kernel + output bias.  Mid layer l in 1..NL-1 is split into kSplit row groups (H = 16: two
    // halves, so that the accumulators fit in 256 VGPRs): part = 1 + (l-1) kSplit + g holds rows
    // [g kRows, (g+1) kRows) of the kernel [in][out] in acc[0..kRows*H) and, for g = 0, the bias
    // in acc[kRows*H .. kRows*H + H).
    static constexpr int kSplit = H > 8 ? 2 : 1;
    static constexpr int kRows = H / kSplit;
    static constexpr int kAcc = kRows * H + H;
    static constexpr int kParts = 1 + (NL - 1) * kSplit;
    static_assert(kAcc >= 4 * H + 1, "part 0 must fit");

    template <int G>
    static __device__ __forceinline__ void wgrad_rows(const float (&h)[H], const float (&d)[H], float (&acc)[kAcc])
    {
#pragma unroll
        for (int jj = 0; jj < kRows; ++jj)
#pragma unroll
            for (int i = 0; i < H; ++i) acc[jj * H + i] = fmaf(h[G * kRows + jj], d[i], acc[jj * H + i]);
        if constexpr (G == 0) {
#pragma unroll
            for (int i = 0; i < H; ++i) acc[kRows * H + i] += d[i];
        }
    }

    static __device__ __forceinline__ void wgrad(const float* __restrict__ w, const float (&act)[NL][H], float a, float lr,
                                                 float dout, int part, float (&acc)[kAcc])
    {
        float d[H], dn[H], kr[H], kn[H];
        load_row(w + kWo, kr);
#pragma unroll
        for (int i = 0; i < H; ++i) d[i] = dout * kr[i] * fmaf(-act[NL - 1][i], act[NL - 1][i], 1.0f);
        if (part == 0) {
#pragma unroll
            for (int i = 0; i < H; ++i) acc[3 * H + i] = fmaf(dout, act[NL - 1][i], acc[3 * H + i]);
            acc[4 * H] += dout;
        }
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            if (part == 1 + (l - 1) * kSplit) wgrad_rows<0>(act[l - 1], d, acc);
            if constexpr (kSplit == 2) {
                if (part == 2 + (l - 1) * kSplit) wgrad_rows<1>(act[l - 1], d, acc);
            }
            const float* __restrict__ k = w + kMid + (l - 1) * kMidStride;
            load_row(k, kr);
#pragma unroll
            for (int j = 0; j < H; ++j) {
                if (j + 1 < H) load_row(k + (j + 1) * H, kn);
                float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
                for (int i = 0; i < H; i += 4) {
                    s0 = fmaf(kr[i], d[i], s0); s1 = fmaf(kr[i + 1], d[i + 1], s1);
                    s2 = fmaf(kr[i + 2], d[i + 2], s2); s3 = fmaf(kr[i + 3], d[i + 3], s3);
                }
                dn[j] = ((s0 + s1) + (s2 + s3)) * fmaf(-act[l - 1][j], act[l - 1][j], 1.0f);
                if (j + 1 < H) {
#pragma unroll
                    for (int i = 0; i < H; ++i) kr[i] = kn[i];
                }
            }
#pragma unroll
            for (int j = 0; j < H; ++j) d[j] = dn[j];
        }
        if (part == 0) {
#pragma unroll
            for (int i = 0; i < H; ++i) {
                acc[i] = fmaf(a, d[i], acc[i]);
                acc[H + i] = fmaf(lr, d[i], acc[H + i]);
                acc[2 * H + i] += d[i];
            }
        }
    }
};

struct MlpClipConsts {
    float G2;         // 2 C fs
    float p, Rp, lr;  // static R: G1/(G1+G2), 1/(G1+G2), log Rp
};

__device__ __forceinline__ MlpClipConsts mlp_load_consts(const float* __restrict__ theta2, float fs)
{
    MlpClipConsts c;
    const float R = theta2[0], C = theta2[1];
    c.G2 = C * (2.0f * fs);
    const float G1 = 1.0f / R, G = G1 + c.G2;
    c.Rp = 1.0f / G;
    c.p = G1 / G;
    c.lr = logf(c.Rp);
    return c;
}

template <bool DYN_R>
__device__ __forceinline__ void mlp_step_coeffs(const MlpClipConsts& c, float rin, float& p, float& Rp, float& lr)
{
    if constexpr (DYN_R) {                    // set_resistance + calc_impedance every step (clipper_pot.py:116-117)
        const float G1 = fast_rcp(rin);
        Rp = fast_rcp(G1 + c.G2);
        p = G1 * Rp;
        lr = fast_log(Rp);                    // tf.math.log(P1.R), clipper_pot.py:119
    } else {
        p = c.p; Rp = c.Rp; lr = c.lr;
    }
}

// x, r: [B][T]; y, zstash: [T][B]; theta2 = {R, C}; w: flat weights (Mlp<H,NL>::kCount floats)
template <int H, int NL, bool DYN_R>
__global__ __launch_bounds__(64) void clipper_mlp_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta2,
    const float* __restrict__ w_in, float fs, float* __restrict__ y, float* __restrict__ zstash,
    const float* __restrict__ z0, float* __restrict__ zT, int64_t B, int64_t T)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const MlpClipConsts c = mlp_load_consts(theta2, fs);
    __shared__ __attribute__((aligned(16))) float w[Mlp<H, NL>::kCount + 4];
    for (int i = threadIdx.x; i < Mlp<H, NL>::kCount; i += 64) w[i] = w_in[i];
    __syncthreads();
    float z = z0 ? z0[b] : 0.0f;                           // reset(): clipper_pot.py:110-111
    float* __restrict__ yp = y + b;
    float* __restrict__ zp = zstash ? zstash + b : nullptr;
    const float* __restrict__ xp = x + b * T;
    const float* __restrict__ rp = DYN_R ? r + b * T : nullptr;
    float act[NL][H];
    for (int64_t t = 0; t < T; ++t) {
        asm volatile("" ::: "memory");      // re-read the weights from LDS every step (see the header comment)
        float p, Rp, lr;
        mlp_step_coeffs<DYN_R>(c, DYN_R ? rp[t] : 1.0f, p, Rp, lr);
        const float b_diff = z - xp[t];
        const float b_temp = -p * b_diff;
        const float a = z + b_temp;
        const float broot = -Mlp<H, NL>::fwd(w, a, lr, act);
        const float zn = broot + b_temp;
        if (zp) { *zp = z; zp += B; }
        *yp = 0.5f * (zn + z);
        yp += B;
        z = zn;
    }
    if (zT) zT[b] = z;
}

// Reverse sweep.  Outputs: gb [T][B] = dL/d b_root, ain [T][B] = a, lrin [T][B] = log Rp (only
// when DYN_R), and per-wave partials ws: double[gridDim.x][4] = {S_lr, 0, S_P, 0} in the
// clipper convention (static R: S_P = sum g_p, S_lr = sum g_lr; per-sample R:
// S_P = sum Rp (g_p p + g_lr)).
template <int H, int NL, bool DYN_R>
__global__ __launch_bounds__(64) void clipper_mlp_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta2,
    const float* __restrict__ w_in, float fs, const float* __restrict__ zstash, const float* __restrict__ gy,
    float* __restrict__ gb, float* __restrict__ ain, float* __restrict__ lrin, double* __restrict__ ws,
    int64_t B, int64_t T)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const MlpClipConsts c = mlp_load_consts(theta2, fs);
    __shared__ __attribute__((aligned(16))) float w[Mlp<H, NL>::kCount + 4];
    for (int i = threadIdx.x; i < Mlp<H, NL>::kCount; i += 64) w[i] = w_in[i];
    __syncthreads();
    const float* __restrict__ xp = x + b * T;
    const float* __restrict__ rp = DYN_R ? r + b * T : nullptr;
    double dLr = 0.0, dP = 0.0;
    float gz = 0.0f;
    float act[NL][H];
    for (int64_t t = T - 1; t >= 0; --t) {
        asm volatile("" ::: "memory");      // re-read the weights from LDS every step
        float p, Rp, lr;
        mlp_step_coeffs<DYN_R>(c, DYN_R ? rp[t] : 1.0f, p, Rp, lr);
        const float z = zstash[t * B + b];
        const float b_diff = z - xp[t];
        const float a = fmaf(-p, b_diff, z);
        (void)Mlp<H, NL>::fwd(w, a, lr, act);
        float da, dlr;
        Mlp<H, NL>::grad_in(w, act, da, dlr);
        const float Da = -da, Dlr = -dlr;                    // b_root = -MLP
        const float g = gy[t * B + b];
        const float g_b2n = fmaf(0.5f, g, gz);
        const float g_a = g_b2n * Da;
        const float g_lr = g_b2n * Dlr;
        const float g_bt = g_b2n + g_a;
        const float g_p = -g_bt * b_diff;
        gb[t * B + b] = g_b2n;
        ain[t * B + b] = a;
        if constexpr (DYN_R) {
            lrin[t * B + b] = lr;
            dP += (double)(Rp * fmaf(g_p, p, g_lr));
        } else {
            dP += (double)g_p;
            dLr += (double)g_lr;
        }
        gz = fmaf(-p, g_bt, fmaf(0.5f, g, g_a));
    }
    if (!live) { dLr = dP = 0.0; }
    dLr = wave_sum(dLr); dP = wave_sum(dP);
    if (threadIdx.x == 0) {
        double* o = ws + (int64_t)blockIdx.x * 4;
        o[0] = dLr; o[1] = 0.0; o[2] = dP; o[3] = 0.0;
    }
}

// gtheta2 = dL/d{R, C}:  static R: lr = log Rp, so with S_L := S_lr the clipper formulas apply
//   dR = Rp G1^2 (S_lr - S_P (1-p)) ; dC = -2 fs Rp (S_P p + S_lr);  per-sample R: dR = 0, dC = -2 fs S_P
static __global__ __launch_bounds__(256) void clipper_mlp_grad_reduce_kernel(const double* __restrict__ ws, int nparts,
                                                                      const float* __restrict__ theta2, float fs,
                                                                      int dyn_r, float* __restrict__ gtheta2)
{
    __shared__ double sh[256][2];
    double s0 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) { s0 += ws[(int64_t)i * 4 + 0]; s2 += ws[(int64_t)i * 4 + 2]; }
    sh[threadIdx.x][0] = s0; sh[threadIdx.x][1] = s2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sh[threadIdx.x][0] += sh[threadIdx.x + off][0];
            sh[threadIdx.x][1] += sh[threadIdx.x + off][1];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double SL = sh[0][0], SP = sh[0][1];
        const double R = theta2[0], C = theta2[1];
        const double G1 = 1.0 / R, G2 = C * (2.0 * (double)fs), Rp = 1.0 / (G1 + G2), p = G1 * Rp;
        if (dyn_r) {
            gtheta2[0] = 0.0f;
            gtheta2[1] = (float)(-2.0 * (double)fs * SP);
        } else {
            gtheta2[0] = (float)(Rp * G1 * G1 * (SL - SP * (1.0 - p)));
            gtheta2[1] = (float)(-2.0 * (double)fs * Rp * (SP * p + SL));
        }
    }
}

// out[n] = MLP(a[n], lr[n]) over S independent samples: DenseRootModel.incident/reflected on a
// table (layers.py:76-82), the forward of diode_pretraining.py's fit (:113-126, :159-160).
template <int H, int NL>
__global__ __launch_bounds__(64) void mlp_eval_kernel(const float* __restrict__ ain, const float* __restrict__ lrin,
                                                      const float* __restrict__ w_in, float* __restrict__ out, int64_t S)
{
    using M = Mlp<H, NL>;
    __shared__ __attribute__((aligned(16))) float w[M::kCount + 4];
    for (int i = threadIdx.x; i < M::kCount; i += 64) w[i] = w_in[i];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 64;
    float act[NL][H];
    for (int64_t n0 = (int64_t)blockIdx.x * 64; n0 < S; n0 += stride) {
        const int64_t n_raw = n0 + threadIdx.x;
        const int64_t n = n_raw < S ? n_raw : S - 1;
        const float v = M::fwd(w, ain[n], lrin[n], act);
        if (n_raw < S) out[n] = v;
    }
}

// One EPOCH of the pre-training fit in one launch (diode_pretraining.py:159-160: Keras fit with
// Adam, mini-batches of <= 64 table points, loss = MSE + ESR of :136-155): the mini-batch loop is
// launch-bound when driven from the host (25 launches of ~1 us of work each per step), so the
// whole sequential loop runs inside one workgroup.  Weights, Adam moments and the gradient live in
// LDS; wave p of the block owns layer part p (Mlp::wgrad), every wave evaluates the same batch
// (one table point per lane), reduces its part's gradient across lanes, and after a barrier all
// threads apply the Adam update.  xa / xl / ys are the table in the order to visit (the host
// reshuffles per epoch).  esr_n: the ESR normaliser N (:145, the script's global N = 1000).
template <int H, int NL>
__global__ __launch_bounds__((64 * Mlp<H, NL>::kParts)) void mlp_fit_epoch_kernel(
    const float* __restrict__ xa, const float* __restrict__ xl, const float* __restrict__ ys, int64_t S, int batch,
    float* __restrict__ w_io, float* __restrict__ m_io, float* __restrict__ v_io, int32_t* __restrict__ step, float lr,
    float b1, float b2, float eps_adam, float esr_n, float eps_energy, double* __restrict__ loss_sum)
{
    using M = Mlp<H, NL>;
    constexpr int kThreads = 64 * M::kParts;
    __shared__ __attribute__((aligned(16))) float w[M::kCount + 4];
    __shared__ float mo[M::kCount], vo[M::kCount], g[M::kCount];
    for (int i = threadIdx.x; i < M::kCount; i += kThreads) { w[i] = w_io[i]; mo[i] = m_io[i]; vo[i] = v_io[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int part = threadIdx.x >> 6;
    const int l = part == 0 ? 0 : 1 + (part - 1) / M::kSplit, grp = part == 0 ? 0 : (part - 1) % M::kSplit;
    const int layer = M::kMid + (l - 1) * M::kMidStride;
    const int t_start = *step;
    double pb1 = pow((double)b1, (double)t_start), pb2 = pow((double)b2, (double)t_start), loss_acc = 0.0;
    int t = t_start;
    float act[NL][H];
    for (int64_t b0 = 0; b0 < S; b0 += batch) {
        asm volatile("" ::: "memory");                          // the weights in LDS changed
        const int nb = (int)((S - b0 < batch) ? S - b0 : batch);
        const bool live = lane < nb;
        const int64_t n = b0 + (live ? lane : 0);
        const float a = xa[n], lrin = xl[n], y = live ? ys[n] : 0.0f;
        const float out = M::fwd(w, a, lrin, act);
        const float d = live ? out - y : 0.0f;
        float sse = d * d, energy = y * y;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sse += __shfl_xor(sse, off, 64); energy += __shfl_xor(energy, off, 64); }
        energy += eps_energy;
        const float inv_nb = 1.0f / (float)nb;
        const float esr = sqrtf(sse / energy / esr_n);
        // d loss / d out = 2 d / nb + d / (esr energy N)
        const float gout = d * (2.0f * inv_nb + (esr > 0.0f ? 1.0f / (esr * energy * esr_n) : 0.0f));
        float acc[M::kAcc];
#pragma unroll
        for (int i = 0; i < M::kAcc; ++i) acc[i] = 0.0f;
        M::wgrad(w, act, a, lrin, gout, part, acc);
#pragma unroll
        for (int i = 0; i < M::kAcc; ++i) {
            float s = acc[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            int dst;
            if (part == 0) dst = i < 3 * H ? i : (i <= 4 * H ? M::kWo + (i - 3 * H) : -1);
            else if (i < M::kRows * H) dst = layer + grp * M::kRows * H + i;
            else dst = grp == 0 ? layer + H * H + (i - M::kRows * H) : -1;
            if (lane == 0 && dst >= 0) g[dst] = s;
        }
        __syncthreads();
        t += 1;
        pb1 *= (double)b1;
        pb2 *= (double)b2;
        const float lr_t = (float)((double)lr * sqrt(1.0 - pb2) / (1.0 - pb1));
        for (int i = threadIdx.x; i < M::kCount; i += kThreads) {
            const float gi = g[i];
            const float mi = b1 * mo[i] + (1.0f - b1) * gi;
            const float vi = b2 * vo[i] + (1.0f - b2) * gi * gi;
            mo[i] = mi;
            vo[i] = vi;
            w[i] -= lr_t * mi / (sqrtf(vi) + eps_adam);
        }
        loss_acc += (double)(sse * inv_nb + esr);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < M::kCount; i += kThreads) { w_io[i] = w[i]; m_io[i] = mo[i]; v_io[i] = vo[i]; }
    if (threadIdx.x == 0) { *step = t; *loss_sum = loss_acc; }
}

static __global__ __launch_bounds__(256) void mlp_wgrad_reduce_kernel(const float* __restrict__ ws, int nblk, int count,
                                                               float* __restrict__ gw)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)ws[(int64_t)b * count + i];
    gw[i] = (float)s;
}

}  // namespace wdf
