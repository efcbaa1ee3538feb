// wdf_mlp_row.h -- the MLP-root clipper with one 16-lane DPP row per sequence, gfx950.
//
// wdf_mlp.h gives every sequence ONE lane, which evaluates the whole network serially: a 2x16 net
// is ~600 dependent FMAs per time step, and the reference's 1340 training sequences
// (clipper_pot.py:58) fill 21 waves of a 1024-SIMD chip.  Here a sequence owns a ROW of 16 lanes,
// lane j = hidden neuron j (widths 4 and 8 are zero-padded to 16):
//   * a hidden layer is 16 FMAs per lane, the activations of the previous layer arriving as the
//     DPP operand of the FMA itself (v_fmac_f32_dpp ... row_ror:s) -- no LDS, no weight traffic:
//     lane j keeps its 16 weights per layer, ordered by rotation, in VGPRs;
//   * the output layer is one product per lane and a 4-step row reduction;
//   * a wave carries 4 sequences, so the same batch is 16x more waves with a ~6x shorter
//     dependent chain per step.
// The clipper arithmetic (tf_wdf.py:179-192) is replicated in the 16 lanes of a row; lane 0 of
// the row stores.  Same math as wdf_mlp.h up to summation order inside a layer.
// Which lane a rotation reads from is not assumed: the lane index itself is sent through the
// same DPP control, and the weights are gathered with the index that comes back.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wdf_mlp.h"

namespace wdf {

template <int S>
__device__ __forceinline__ float row_rot(float v)
{
    if constexpr (S == 0) return v;
    else return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + S, 0xf, 0xf, true));
}
template <int S>
__device__ __forceinline__ int row_rot_i(int v)
{
    if constexpr (S == 0) return v;
    else return __builtin_amdgcn_update_dpp(0, v, 0x120 + S, 0xf, 0xf, true);
}

// sum over the 16 lanes of a row, result in every lane: rotations by 8, 4, 2, 1
__device__ __forceinline__ float row_sum(float v)
{
    v += row_rot<8>(v);
    v += row_rot<4>(v);
    v += row_rot<2>(v);
    v += row_rot<1>(v);
    return v;
}

// The weights one lane (neuron j) holds.  mid[l][s] multiplies the activation that rotation s
// delivers to this lane, i.e. W_l[src(s)][j] with src(s) = row_rot_i<s>(j).
template <int NL>
struct RowWeights {
    float k0a, k0l, b0;            // layer 0: out_j = b0 + a k0a + lr k0l          (kernel0 [2][H], bias0)
    float mid[NL - 1][16];         // layers 1..NL-1, by rotation
    float tmid[NL - 1][16];        // the transposed access for the reverse sweep: W_l[j][src(s)]
    float bias[NL - 1];
    float wo, bo;                  // output layer [H][1], bias
};

template <int S, int NL>
__device__ __forceinline__ void row_load_rot(RowWeights<NL>& W, const float* __restrict__ w, int H, int j, bool with_t)
{
    const int src = row_rot_i<S>(j);                           // the neuron whose activation rotation S brings here
    const bool ok = j < H && src < H;
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        const float* __restrict__ k = w + 3 * H + (l - 1) * (H * H + H);      // kernel_l [in][out]
        W.mid[l - 1][S] = ok ? k[src * H + j] : 0.0f;
        W.tmid[l - 1][S] = (with_t && ok) ? k[j * H + src] : 0.0f;
    }
    if constexpr (S + 1 < 16) row_load_rot<S + 1, NL>(W, w, H, j, with_t);
}

template <int NL>
__device__ __forceinline__ RowWeights<NL> row_load_weights(const float* __restrict__ w, int H, int j, bool with_t)
{
    RowWeights<NL> W;
    const bool live = j < H;
    W.k0a = live ? w[j] : 0.0f;
    W.k0l = live ? w[H + j] : 0.0f;
    W.b0 = live ? w[2 * H + j] : 0.0f;
#pragma unroll
    for (int l = 1; l < NL; ++l) W.bias[l - 1] = live ? w[3 * H + (l - 1) * (H * H + H) + H * H + j] : 0.0f;
    const int kWo = 3 * H + (NL - 1) * (H * H + H);
    W.wo = live ? w[kWo + j] : 0.0f;
    W.bo = w[kWo + H];
    row_load_rot<0, NL>(W, w, H, j, with_t);
    return W;
}

// acc += sum_s wt[s] * rot_s(h): four partial sums keep the dependent chain at 4 FMAs.  Written as
// v_fmac_f32_dpp -- the rotation is an operand modifier of the FMA itself -- because the compiler
// leaves the builtin as v_mov_b32_dpp + v_fmac (twice the instructions).  The leading s_nop covers
// the VALU-write -> DPP-read hazard on h (2 wait states), which the hazard recogniser does not see
// inside an asm block.
__device__ __forceinline__ float row_matvec(const float (&wt)[16], float h, float init)
{
    float p0 = init, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f32_e32 %0, %4, %5\n\t"
        "v_fmac_f32_dpp %1, %4, %6 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %4, %7 row_ror:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %4, %8 row_ror:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %0, %4, %9 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %4, %10 row_ror:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %4, %11 row_ror:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %4, %12 row_ror:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %0, %4, %13 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %4, %14 row_ror:9 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %4, %15 row_ror:10 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %4, %16 row_ror:11 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %0, %4, %17 row_ror:12 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %4, %18 row_ror:13 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %4, %19 row_ror:14 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %4, %20 row_ror:15 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
        : "v"(h), "v"(wt[0]), "v"(wt[1]), "v"(wt[2]), "v"(wt[3]), "v"(wt[4]), "v"(wt[5]), "v"(wt[6]), "v"(wt[7]),
          "v"(wt[8]), "v"(wt[9]), "v"(wt[10]), "v"(wt[11]), "v"(wt[12]), "v"(wt[13]), "v"(wt[14]), "v"(wt[15]));
    return (p0 + p1) + (p2 + p3);
}

// out = MLP(a, lr) in every lane of the row; act[l] = this lane's activation in layer l
template <int NL>
__device__ __forceinline__ float row_mlp_fwd(const RowWeights<NL>& W, float a, float lr, float (&act)[NL])
{
    act[0] = tanh_fast(fmaf(lr, W.k0l, fmaf(a, W.k0a, W.b0)));
#pragma unroll
    for (int l = 1; l < NL; ++l) act[l] = tanh_fast(row_matvec(W.mid[l - 1], act[l - 1], W.bias[l - 1]));
    return row_sum(W.wo * act[NL - 1]) + W.bo;
}

// v[i] = the value lane i of this lane's row holds, i = 0..15: the 16 ds_bpermutes are issued together
// at the top of a 16-step block, off the recurrence's critical path.
__device__ __forceinline__ void row_spread(float blk, int lane, float (&v)[16])
{
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __shfl(blk, (lane & 48) | i, 64);
}

// x, r: [B][T]; y, zstash: [T][B]; w: flat weights of a 2 -> H -> ... -> H -> 1 net, H <= 16
template <int NL, bool DYN_R>
__global__ __launch_bounds__(64) void clipper_mlp_row_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta2,
    const float* __restrict__ w, int H, float fs, float* __restrict__ y, float* __restrict__ zstash,
    const float* __restrict__ z0, float* __restrict__ zT, int64_t B, int64_t T, const unsigned* __restrict__ gate)
{
    // gate: the repair launch of the time-parallel forward (wdf_mlp_tp.h) -- only flagged waves run
    if (gate != nullptr && gate[blockIdx.x] == 0u) return;
    const int lane = threadIdx.x, j = lane & 15;
    const int64_t b_raw = (int64_t)blockIdx.x * 4 + (lane >> 4);
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const MlpClipConsts c = mlp_load_consts(theta2, fs);
    const RowWeights<NL> W = row_load_weights<NL>(w, H, j, false);
    const float* __restrict__ xp = x + b * T;
    const float* __restrict__ rp = DYN_R ? r + b * T : nullptr;
    float z = z0 ? z0[b] : 0.0f;
    float act[NL];
    for (int64_t t0 = 0; t0 < T; t0 += 16) {
        // lane j of the row fetches sample t0 + j: one 64-byte read per row and 16 steps
        const int64_t tj = t0 + j < T ? t0 + j : T - 1;
        const float xblk = xp[tj];
        const float rblk = DYN_R ? rp[tj] : 1.0f;
        const int n = T - t0 < 16 ? (int)(T - t0) : 16;
        float xs[16], rs[16];
        row_spread(xblk, lane, xs);
        if constexpr (DYN_R) row_spread(rblk, lane, rs);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i >= n) break;
            const float xin = xs[i];
            float p, Rp, lr;
            mlp_step_coeffs<DYN_R>(c, DYN_R ? rs[i] : 1.0f, p, Rp, lr);
            const float b_diff = z - xin;
            const float b_temp = -p * b_diff;
            const float a = z + b_temp;
            const float zn = b_temp - row_mlp_fwd<NL>(W, a, lr, act);       // b_root = -MLP
            if (j == 0) {
                const int64_t o = (t0 + i) * B + b;
                if (zstash) zstash[o] = z;
                y[o] = 0.5f * (zn + z);
            }
            z = zn;
        }
    }
    if (zT && j == 0) zT[b] = z;
}

// d out / d a and d out / d lr from the kept activations (every lane of the row gets both):
//   delta_L[j] = wo[j] (1 - h_L[j]^2) ;  delta_{l-1}[j] = (sum_i W_l[j][i] delta_l[i]) (1 - h_{l-1}[j]^2)
// -- the transposed product uses the same rotations with the weights gathered the other way round.
template <int NL>
__device__ __forceinline__ void row_mlp_grad_in(const RowWeights<NL>& W, const float (&act)[NL], float& da, float& dlr)
{
    float d = W.wo * fmaf(-act[NL - 1], act[NL - 1], 1.0f);
#pragma unroll
    for (int l = NL - 1; l >= 1; --l) d = row_matvec(W.tmid[l - 1], d, 0.0f) * fmaf(-act[l - 1], act[l - 1], 1.0f);
    da = row_sum(W.k0a * d);
    dlr = row_sum(W.k0l * d);
}

// Reverse sweep, same outputs as clipper_mlp_bwd_kernel: gb, ain (and lrin when DYN_R) [T][B], and
// per-WAVE partials ws: double[gridDim.x][4] = {S_lr, 0, S_P, 0} over the wave's 4 sequences.
template <int NL, bool DYN_R>
__global__ __launch_bounds__(64) void clipper_mlp_row_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta2,
    const float* __restrict__ w, int H, float fs, const float* __restrict__ zstash, const float* __restrict__ gy,
    float* __restrict__ gb, float* __restrict__ ain, float* __restrict__ lrin, double* __restrict__ ws,
    int64_t B, int64_t T)
{
    const int lane = threadIdx.x, j = lane & 15;
    const int64_t b_raw = (int64_t)blockIdx.x * 4 + (lane >> 4);
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const MlpClipConsts c = mlp_load_consts(theta2, fs);
    const RowWeights<NL> W = row_load_weights<NL>(w, H, j, true);
    const float* __restrict__ xp = x + b * T;
    const float* __restrict__ rp = DYN_R ? r + b * T : nullptr;
    double dLr = 0.0, dP = 0.0;
    float gz = 0.0f;
    float act[NL];
    for (int64_t t1 = T; t1 > 0; t1 -= 16) {                  // blocks of 16 steps, walked backwards
        const int64_t t0 = t1 >= 16 ? t1 - 16 : 0;
        const int n = (int)(t1 - t0);
        // lane j of the row fetches what step t0 + j needs: x, r from the row of the sequence,
        // state and dL/dy from the time-major arrays
        const int64_t tj = j < n ? t0 + j : t0;
        const float xblk = xp[tj];
        const float rblk = DYN_R ? rp[tj] : 1.0f;
        const float zblk = zstash[tj * B + b];
        const float gblk = gy[tj * B + b];
        float xs[16], rs[16], zz[16], gs[16];
        row_spread(xblk, lane, xs);
        row_spread(zblk, lane, zz);
        row_spread(gblk, lane, gs);
        if constexpr (DYN_R) row_spread(rblk, lane, rs);
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            if (i >= n) continue;
            const float xin = xs[i], z = zz[i], g = gs[i];
            float p, Rp, lr;
            mlp_step_coeffs<DYN_R>(c, DYN_R ? rs[i] : 1.0f, p, Rp, lr);
            const float b_diff = z - xin;
            const float a = fmaf(-p, b_diff, z);
            (void)row_mlp_fwd<NL>(W, a, lr, act);
            float da, dlr;
            row_mlp_grad_in<NL>(W, act, da, dlr);
            const float Da = -da, Dlr = -dlr;                    // b_root = -MLP
            const float g_b2n = fmaf(0.5f, g, gz);
            const float g_a = g_b2n * Da;
            const float g_lr = g_b2n * Dlr;
            const float g_bt = g_b2n + g_a;
            const float g_p = -g_bt * b_diff;
            if (j == 0) {
                const int64_t o = (t0 + i) * B + b;
                gb[o] = g_b2n;
                ain[o] = a;
                if constexpr (DYN_R) lrin[o] = lr;
            }
            if constexpr (DYN_R) {
                dP += (double)(Rp * fmaf(g_p, p, g_lr));
            } else {
                dP += (double)g_p;
                dLr += (double)g_lr;
            }
            gz = fmaf(-p, g_bt, fmaf(0.5f, g, g_a));
        }
    }
    if (!live || j != 0) { dLr = dP = 0.0; }                   // one lane per sequence contributes
    dLr = wave_sum(dLr); dP = wave_sum(dP);
    if (threadIdx.x == 0) {
        double* o = ws + (int64_t)blockIdx.x * 4;
        o[0] = dLr; o[1] = 0.0; o[2] = dP; o[3] = 0.0;
    }
}

// ---- reverse sweep with the weight gradient folded in ------------------------------------------
// dL/dW = -sum_n g_b2n[n] dMLP(a[n], lr[n])/dW.  Lane j owns neuron j, so it also owns the gradient
// of every weight INTO neuron j: with G = -g_b2n and delta_l[j] = d out / d pre_l[j] (the values
// the input-Jacobian chain passes through anyway),
//   g wo[j] += G h_L[j] ; g bias_l[j] += G delta_l[j] ; g W_l[src(s)][j] += G delta_l[j] rot_s(h_{l-1})
//   g k0a[j] += G delta_0[j] a ; g k0l[j] += G delta_0[j] lr ; g b0[j] += G delta_0[j]
// -- independent accumulations that fill issue slots of a latency-bound step.  No g_b / a / log R
// arrays are written and no second pass over the samples is needed.
typedef float mfma_v4f __attribute__((ext_vector_type(4)));

// D = A B + C of v_mfma_f32_16x16x4_f32: A[i][k] in lane 16 k + i, B[k][n] in lane 16 k + n, D[i][n] in VGPR v of
// lane 16 g + n with i = 4 g + v
__device__ __forceinline__ mfma_v4f mfma4(float a, float b, mfma_v4f c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// The hidden layers' kernel gradients are outer products summed over steps AND over the wave's four sequences:
// with lane 16 q + j holding (G delta)_q[j] as A[j][q] and h_q[j] as B[q][j], ONE MFMA per layer and step adds
// sum_q (G delta)_q[i] h_q[n] to mid[l] = dK_l[in n][out i] in the D layout (was 16 DPP-FMAs per lane into 16
// accumulators per layer, plus a sum over the four rows at the end).
template <int NL>
struct RowGrads {
    float k0a, k0l, b0, wo, bo;
    float bias[NL - 1];
    mfma_v4f mid[NL - 1];
};

template <int NL>
__device__ __forceinline__ void row_mlp_grad_all(const RowWeights<NL>& W, const float (&act)[NL], float a, float lr,
                                                 float G, RowGrads<NL>& acc, float& da, float& dlr)
{
    acc.wo = fmaf(G, act[NL - 1], acc.wo);
    acc.bo += G;
    float d = W.wo * fmaf(-act[NL - 1], act[NL - 1], 1.0f);      // delta_L
#pragma unroll
    for (int l = NL - 1; l >= 1; --l) {
        const float gd = G * d;
        acc.bias[l - 1] += gd;
        acc.mid[l - 1] = mfma4(gd, act[l - 1], acc.mid[l - 1]);
        d = row_matvec(W.tmid[l - 1], d, 0.0f) * fmaf(-act[l - 1], act[l - 1], 1.0f);
    }
    const float gd0 = G * d;
    acc.k0a = fmaf(gd0, a, acc.k0a);
    acc.k0l = fmaf(gd0, lr, acc.k0l);
    acc.b0 += gd0;
    da = row_sum(W.k0a * d);
    dlr = row_sum(W.k0l * d);
}

// sum over the 4 rows of the wave (lanes j, j+16, j+32, j+48), result in every lane
__device__ __forceinline__ float rows_sum(float v)
{
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// mid[l] (D layout: VGPR v of lane 16 g + n = dK_l[in n][out 4 g + v], the wave's four sequences already summed)
// -> the flat weight order, kernel_l [in][out]
template <int NL>
__device__ __forceinline__ void row_store_mid(const RowGrads<NL>& acc, float* __restrict__ o, int H, int lane)
{
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int l = 1; l < NL; ++l) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = 4 * g + v;
            if (n < H && i < H) o[3 * H + (l - 1) * (H * H + H) + n * H + i] = acc.mid[l - 1][v];
        }
    }
}

// Outputs: wsw float[gridDim.x][count] = this wave's weight-gradient partial in the flat weight
// order (every entry written exactly once per wave), ws double[gridDim.x][4] as above.
template <int NL, bool DYN_R>
__global__ __launch_bounds__(64) void clipper_mlp_row_bwd_w_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta2,
    const float* __restrict__ w, int H, float fs, const float* __restrict__ zstash, const float* __restrict__ gy,
    float* __restrict__ wsw, double* __restrict__ ws, int64_t B, int64_t T)
{
    const int lane = threadIdx.x, j = lane & 15;
    const int64_t b_raw = (int64_t)blockIdx.x * 4 + (lane >> 4);
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const MlpClipConsts c = mlp_load_consts(theta2, fs);
    const RowWeights<NL> W = row_load_weights<NL>(w, H, j, true);
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
    float gz = 0.0f;
    float act[NL];
    for (int64_t t1 = T; t1 > 0; t1 -= 16) {
        const int64_t t0 = t1 >= 16 ? t1 - 16 : 0;
        const int n = (int)(t1 - t0);
        const int64_t tj = j < n ? t0 + j : t0;
        const float xblk = xp[tj];
        const float rblk = DYN_R ? rp[tj] : 1.0f;
        const float zblk = zstash[tj * B + b];
        const float gblk = live ? gy[tj * B + b] : 0.0f;        // shadow rows of the last wave add nothing
        float xs[16], rs[16], zz[16], gs[16];
        row_spread(xblk, lane, xs);
        row_spread(zblk, lane, zz);
        row_spread(gblk, lane, gs);
        if constexpr (DYN_R) row_spread(rblk, lane, rs);
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            if (i >= n) continue;
            const float xin = xs[i], z = zz[i], g = gs[i];
            float p, Rp, lr;
            mlp_step_coeffs<DYN_R>(c, DYN_R ? rs[i] : 1.0f, p, Rp, lr);
            const float b_diff = z - xin;
            const float a = fmaf(-p, b_diff, z);
            (void)row_mlp_fwd<NL>(W, a, lr, act);
            const float g_b2n = fmaf(0.5f, g, gz);
            float da, dlr;
            row_mlp_grad_all<NL>(W, act, a, lr, -g_b2n, acc, da, dlr);
            const float g_a = -g_b2n * da;                       // b_root = -MLP
            const float g_lr = -g_b2n * dlr;
            const float g_bt = g_b2n + g_a;
            const float g_p = -g_bt * b_diff;
            if constexpr (DYN_R) {
                dP += (double)(Rp * fmaf(g_p, p, g_lr));
            } else {
                dP += (double)g_p;
                dLr += (double)g_lr;
            }
            gz = fmaf(-p, g_bt, fmaf(0.5f, g, g_a));
        }
    }
    if (!live || j != 0) { dLr = dP = 0.0; }
    dLr = wave_sum(dLr); dP = wave_sum(dP);
    if (threadIdx.x == 0) {
        double* o = ws + (int64_t)blockIdx.x * 4;
        o[0] = dLr; o[1] = 0.0; o[2] = dP; o[3] = 0.0;
    }
    // weight-gradient partial of this wave (gy was zeroed for shadow rows, so they contribute 0)
    const int count = 3 * H + (NL - 1) * (H * H + H) + H + 1;
    float* __restrict__ o = wsw + (int64_t)blockIdx.x * count;
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

}  // namespace wdf
