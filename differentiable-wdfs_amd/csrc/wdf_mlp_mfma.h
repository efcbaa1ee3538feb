// wdf_mlp_mfma.h -- the MLP-root forward (time-parallel chunks or one sequential chunk) and the weight-gradient pass of
// the reverse sweep with the hidden layers on the matrix cores, gfx950.
//
// The row kernels (wdf_mlp_row.h) give a sequence 16 lanes and do a 16 x 16 layer as 16 DPP-FMAs per lane:
// ~45 VALU instructions per layer for the 4 sequences of a wave.  Here a wave carries 16 sequences and a
// hidden layer is FOUR v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate -- the parity tolerance of this path
// leaves no room for bf16) with NO data movement between layers:
//
//   D = A B + C of that instruction:  A[i][k] in lane 16 k + i,  B[k][n] in lane 16 k + n,
//                                     D[i][n] in VGPR v of lane 16 g + n with i = 4 g + v.
//   Activations live in the D layout: unit u = 4 g + v of sequence n in VGPR v of lane 16 g + n.  VGPR v of
//   that layout IS a valid B operand -- B[k][n] = h[4 k + v][n] -- for the slice of the contraction that
//   holds the input units {v, 4 + v, 8 + v, 12 + v}; the weights are loaded once in the matching order
//   (A_v[i][k] = K[in 4 k + v][out i]), so the next layer is acc = bias; acc = mfma(A_v, h_v, acc), v = 0..3.
//   The output layer is 4 FMAs per lane and one MFMA against a matrix of ones (the sum over the four lane
//   groups, delivered to every lane); the input Jacobian (kappa) is the same chain with transposed weights.
//
// The clipper arithmetic (tf_wdf.py:179-192) is replicated in the four lane groups; group g stores the steps
// i = g (mod 4) of every four, so a store instruction has all 64 lanes busy.  Widths 4 and 8 are zero-padded
// to 16 (tanh(0) = 0: the padding contributes nothing).
//
// Measured (tools/mlp_fwd_probe.py, 2x16 net, T = 2048): a step is a dependent chain of 4 (NL-1) + 1 MFMAs and NL
// tanh over FOUR registers each (256 activations per layer on 64 lanes; exp2 and rcp issue at quarter rate):
// 0.55 us per step against the row kernel's 0.31 us -- so while the batch leaves SIMDs idle (the reference's 1340
// sequences) the row kernel wins, and from ~12k sequences on this one does: 37.8 against 24.9 G samples/s at
// 65536 x 2048, 26.1 against 15.4 with kappa.  The host picks by batch size (wdf_capi_mlp.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wdf_mlp_tp.h"

namespace wdf {

template <int NL>
struct MfmaWeights {
    float k0a[4], k0l[4], b0[4];   // layer 0, unit 4 g + v
    float a[NL - 1][4];            // slice v of layer l:  A[i][k] = K_l[in 4 k + v][out i]
    float at[NL - 1][4];           // transposed (kappa):  A[i][k] = K_l[in i][out 4 k + v]
    float bias[NL - 1][4];
    float wo[4], bo;
};

// w: the flat weights of a 2 -> H -> ... -> H -> 1 net (kernel0 [2][H], bias0, (NL-1) x {kernel [in][out], bias}, out)
template <int NL>
__device__ __forceinline__ MfmaWeights<NL> mfma_load_weights(const float* __restrict__ w, int H, int lane, bool with_t)
{
    MfmaWeights<NL> W;
    const int i = lane & 15, k = lane >> 4;
    const int kWo = 3 * H + (NL - 1) * (H * H + H);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int u = 4 * k + v;
        const bool live = u < H;
        W.k0a[v] = live ? w[u] : 0.0f;
        W.k0l[v] = live ? w[H + u] : 0.0f;
        W.b0[v] = live ? w[2 * H + u] : 0.0f;
        W.wo[v] = live ? w[kWo + u] : 0.0f;
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            const float* __restrict__ kern = w + 3 * H + (l - 1) * (H * H + H);
            const bool ok = live && i < H;
            W.a[l - 1][v] = ok ? kern[u * H + i] : 0.0f;
            W.at[l - 1][v] = (with_t && ok) ? kern[i * H + u] : 0.0f;
            W.bias[l - 1][v] = live ? kern[H * H + u] : 0.0f;
        }
    }
    W.bo = w[kWo + H];
    return W;
}

// (mfma_v4f, mfma4: wdf_mlp_row.h)

// out[n] = MLP(a[n], lr[n]) in every lane of sequence n; act[l] = the lane's four activations of layer l
template <int NL>
__device__ __forceinline__ float mfma_mlp_fwd(const MfmaWeights<NL>& W, float a, float lr, mfma_v4f (&act)[NL])
{
#pragma unroll
    for (int v = 0; v < 4; ++v) act[0][v] = tanh_fast(fmaf(lr, W.k0l[v], fmaf(a, W.k0a[v], W.b0[v])));
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        mfma_v4f acc = {W.bias[l - 1][0], W.bias[l - 1][1], W.bias[l - 1][2], W.bias[l - 1][3]};
#pragma unroll
        for (int v = 0; v < 4; ++v) acc = mfma4(W.a[l - 1][v], act[l - 1][v], acc);
#pragma unroll
        for (int v = 0; v < 4; ++v) act[l][v] = tanh_fast(acc[v]);
    }
    const float part = fmaf(W.wo[3], act[NL - 1][3],
                            fmaf(W.wo[2], act[NL - 1][2], fmaf(W.wo[1], act[NL - 1][1], W.wo[0] * act[NL - 1][0])));
    const mfma_v4f s = mfma4(1.0f, part, mfma_v4f{W.bo, W.bo, W.bo, W.bo});     // sum over the four lane groups
    return s[0];
}

// d out / d a from the kept activations (every lane of the sequence gets it):
//   delta_L = wo (1 - h_L^2);  delta_{l-1}[i] = (sum_j K_l[i][j] delta_l[j]) (1 - h_{l-1}[i]^2);  da = sum_j k0a[j] delta_0[j]
template <int NL>
__device__ __forceinline__ float mfma_mlp_grad_a(const MfmaWeights<NL>& W, const mfma_v4f (&act)[NL])
{
    mfma_v4f d;
#pragma unroll
    for (int v = 0; v < 4; ++v) d[v] = W.wo[v] * fmaf(-act[NL - 1][v], act[NL - 1][v], 1.0f);
#pragma unroll
    for (int l = NL - 1; l >= 1; --l) {
        mfma_v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int v = 0; v < 4; ++v) acc = mfma4(W.at[l - 1][v], d[v], acc);
#pragma unroll
        for (int v = 0; v < 4; ++v) d[v] = acc[v] * fmaf(-act[l - 1][v], act[l - 1][v], 1.0f);
    }
    const float part = fmaf(W.k0a[3], d[3], fmaf(W.k0a[2], d[2], fmaf(W.k0a[1], d[1], W.k0a[0] * d[0])));
    const mfma_v4f s = mfma4(1.0f, part, mfma_v4f{0.0f, 0.0f, 0.0f, 0.0f});
    return s[0];
}

// The arguments of clipper_mlp_row_fwd_tp_kernel (wdf_mlp_tp.h); grid (ceil(B / 16), K).  wrow stays one entry
// per FOUR sequences (the verify kernel's and the sequential kernel's wave): a wave here takes the largest of its four.
template <int NL, bool DYN_R, bool KAP>
__global__ __launch_bounds__(64) void clipper_mlp_mfma_fwd_tp_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta2,
    const float* __restrict__ w, int H, float fs, float* __restrict__ y, float* __restrict__ zstash,
    const float* __restrict__ z0, float* __restrict__ zT, float* __restrict__ zwarm, float* __restrict__ zend,
    const int* __restrict__ wrow, MlpTpStatus* __restrict__ status, int64_t B, int64_t T, int64_t L, int64_t W, int64_t L0,
    float* __restrict__ kappa, const float* __restrict__ zinit, const unsigned* __restrict__ gate)
{
    if (gate != nullptr) {                                      // the chunk-local repair pass: flagged waves only
        const int64_t nw = (B + 3) / 4;
        unsigned any = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t wi = (int64_t)blockIdx.x * 4 + q;
            any |= gate[wi < nw ? wi : nw - 1];
        }
        if (any == 0u) return;
    }
    // zwarm / zend / status may be null: the plain sequential call (one chunk, nothing to verify)
    if (status && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *status = MlpTpStatus{0, 0.0f, 0, 0};   // the verify kernel adds
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    const int64_t b_raw = (int64_t)blockIdx.x * 16 + n;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const int64_t k = blockIdx.y;
    const int64_t t0 = k == 0 ? 0 : L0 + (k - 1) * L;
    const int64_t t1 = k == 0 ? (L0 < T ? L0 : T) : ((t0 + L < T) ? t0 + L : T);
    int64_t Wq = W;
    if (wrow) {
        const int64_t nw = (B + 3) / 4;
        Wq = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t wi = (int64_t)blockIdx.x * 4 + q;
            const int64_t wv = wrow[wi < nw ? wi : nw - 1];
            Wq = wv > Wq ? wv : Wq;
        }
    }
    const int64_t tw = (k > 0 && t0 > Wq) ? t0 - Wq : 0;
    const MlpClipConsts c = mlp_load_consts(theta2, fs);
    const MfmaWeights<NL> Wt = mfma_load_weights<NL>(w, H, lane, KAP);
    const float* __restrict__ xp = x + b * T;
    const float* __restrict__ rp = DYN_R ? r + b * T : nullptr;
    float z = tw == 0 ? (z0 ? z0[b] : 0.0f) : (zinit ? zinit[k * B + b] : 0.0f);
    mfma_v4f act[NL];
    for (int64_t tb = tw; tb < t1; tb += 16) {
        if (zwarm && tb == t0 && g == 0 && live) zwarm[k * B + b] = z;   // the state this chunk arrives with
        const bool owned = tb >= t0;
        const int nst = t1 - tb < 16 ? (int)(t1 - tb) : 16;
        float xs[16], rs[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t ti = tb + i < T ? tb + i : T - 1;
            xs[i] = xp[ti];
            rs[i] = DYN_R ? rp[ti] : 1.0f;
        }
        float yk = 0.0f, zk = 0.0f, kk = 0.0f;                  // group g keeps the steps i = g (mod 4) of every four
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i < nst) {                                      // (wave-uniform)
                float p, Rp, lr;
                mlp_step_coeffs<DYN_R>(c, rs[i], p, Rp, lr);
                const float b_diff = z - xs[i];
                const float b_temp = -p * b_diff;
                const float a = z + b_temp;
                const float zn = b_temp - mfma_mlp_fwd<NL>(Wt, a, lr, act);     // b_root = -MLP
                if (owned) {
                    const bool mine = (i & 3) == g;
                    yk = mine ? 0.5f * (zn + z) : yk;
                    zk = mine ? z : zk;
                    if constexpr (KAP) {
                        const float Da = -mfma_mlp_grad_a<NL>(Wt, act);
                        kk = mine ? Da - p * (1.0f + Da) : kk;
                    }
                }
                z = zn;
            }
            if ((i & 3) == 3 && owned && live && (i - 3 + g) < nst) {
                const int64_t o = (tb + (i - 3 + g)) * B + b;
                y[o] = yk;
                if (zstash) zstash[o] = zk;
                if constexpr (KAP) kappa[o] = kk;
            }
        }
    }
    if (g == 0 && live) {
        if (zend) zend[k * B + b] = z;
        if (zT && t1 == T) zT[b] = z;
    }
}

// ---- (C) of the reverse sweep on the matrix cores ---------------------------------------------------------------
// The weight-gradient pass with 16 sequences per wave: network forward and layer deltas as above; the kernel
// gradient dK_l[in n][out i] = sum over steps and sequences of (G delta_l)[i][s] h_{l-1}[n][s] contracts over the
// SEQUENCE index, which the D layout keeps in the lane -- so both factors are transposed first; register r of the
// transposed pair then is a valid (A, B) for the sequences {r, 4 + r, 8 + r, 12 + r}, and four MFMAs add the outer
// products of all 16 sequences.  The transposes go through LDS (a 16 x 17 tile each, 4 writes + 4 reads per lane;
// the block is one wave).  -DWDF_MLP_WGRAD_MFMA_TRANSPOSE keeps the first version, the matrix cores transposing by
// themselves: with register v of X (units {4 g + v}) as the A operand, A[s][g] = X[4 g + v][s], and the constant
// selector E_v[g][n] = (n == 4 g + v) as B, four accumulated MFMAs give D[s][n] = X[n][s] exactly -- neat, but 16 of
// that version's 43 MFMAs per step sat in the one pipe the kernel is bound by (training step 0.596 vs 0.554 ms).
__device__ __forceinline__ mfma_v4f mfma_transpose(const mfma_v4f& X, const float (&E)[4])
{
    mfma_v4f t = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int v = 0; v < 4; ++v) t = mfma4(X[v], E[v], t);
    return t;
}

// Same outputs as clipper_mlp_row_wgrad_tp_kernel (wdf_mlp_tp.h): wsw float[parts][count], ws double[parts][4],
// part = blockIdx.y * gridDim.x + blockIdx.x; grid (ceil(B / 16), chunks of L steps).
template <int NL, bool DYN_R>
__global__ __launch_bounds__(64) void clipper_mlp_mfma_wgrad_tp_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta2,
    const float* __restrict__ w, int H, float fs, const float* __restrict__ zstash, const float* __restrict__ gb2n,
    float* __restrict__ wsw, double* __restrict__ ws, int64_t B, int64_t T, int64_t L)
{
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    const int64_t b_raw = (int64_t)blockIdx.x * 16 + n;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const int64_t t0 = (int64_t)blockIdx.y * L, t1 = (t0 + L < T) ? t0 + L : T;
    const int64_t part = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const MlpClipConsts c = mlp_load_consts(theta2, fs);
    const MfmaWeights<NL> Wt = mfma_load_weights<NL>(w, H, lane, true);
    float E[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) E[v] = (n == 4 * g + v) ? 1.0f : 0.0f;
#ifndef WDF_MLP_WGRAD_MFMA_TRANSPOSE
    __shared__ float tbuf[2][16 * 17];
#endif
    const float* __restrict__ xp = x + b * T;
    const float* __restrict__ rp = DYN_R ? r + b * T : nullptr;
    float gk0a[4] = {0, 0, 0, 0}, gk0l[4] = {0, 0, 0, 0}, gb0[4] = {0, 0, 0, 0}, gwo[4] = {0, 0, 0, 0}, gbo = 0.0f;
    float gbias[NL - 1][4];
    mfma_v4f gK[NL - 1];
#pragma unroll
    for (int l = 0; l < NL - 1; ++l) {
        gK[l] = mfma_v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int v = 0; v < 4; ++v) gbias[l][v] = 0.0f;
    }
    double dLr = 0.0, dP = 0.0;
    mfma_v4f act[NL];
    for (int64_t tb = t0; tb < t1; tb += 16) {
        const int nst = t1 - tb < 16 ? (int)(t1 - tb) : 16;
        float xs[16], rs[16], zz[16], gs[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t ti = i < nst ? tb + i : tb;
            xs[i] = xp[ti];
            rs[i] = DYN_R ? rp[ti] : 1.0f;
            zz[i] = zstash[ti * B + b];
            gs[i] = (live && i < nst) ? gb2n[ti * B + b] : 0.0f;  // shadow sequences / steps past the end add nothing
        }
        float sLr = 0.0f, sP = 0.0f;                            // fp32 within a block, fp64 across blocks
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i >= nst) break;
            float p, Rp, lr;
            mlp_step_coeffs<DYN_R>(c, rs[i], p, Rp, lr);
            const float z = zz[i], g_b2n = gs[i], G = -g_b2n;   // b_root = -MLP
            const float b_diff = z - xs[i];
            const float a = fmaf(-p, b_diff, z);
            (void)mfma_mlp_fwd<NL>(Wt, a, lr, act);
            gbo += G;
            mfma_v4f d;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                gwo[v] = fmaf(G, act[NL - 1][v], gwo[v]);
                d[v] = Wt.wo[v] * fmaf(-act[NL - 1][v], act[NL - 1][v], 1.0f);
            }
#pragma unroll
            for (int l = NL - 1; l >= 1; --l) {
                mfma_v4f gd;
#pragma unroll
                for (int v = 0; v < 4; ++v) { gd[v] = G * d[v]; gbias[l - 1][v] += gd[v]; }
#ifndef WDF_MLP_WGRAD_MFMA_TRANSPOSE
                // the two transposes through LDS (one 16 x 17 tile each; the block is one wave)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    tbuf[0][(4 * g + v) * 17 + n] = gd[v];
                    tbuf[1][(4 * g + v) * 17 + n] = act[l - 1][v];
                }
                __syncthreads();
                mfma_v4f gdT, hT;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    gdT[q] = tbuf[0][n * 17 + 4 * g + q];
                    hT[q] = tbuf[1][n * 17 + 4 * g + q];
                }
                __syncthreads();
#else
                const mfma_v4f gdT = mfma_transpose(gd, E), hT = mfma_transpose(act[l - 1], E);
#endif
#pragma unroll
                for (int q = 0; q < 4; ++q) gK[l - 1] = mfma4(gdT[q], hT[q], gK[l - 1]);
                mfma_v4f nd = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int v = 0; v < 4; ++v) nd = mfma4(Wt.at[l - 1][v], d[v], nd);
#pragma unroll
                for (int v = 0; v < 4; ++v) d[v] = nd[v] * fmaf(-act[l - 1][v], act[l - 1][v], 1.0f);
            }
            float pa = 0.0f, pl = 0.0f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float gd0 = G * d[v];
                gk0a[v] = fmaf(gd0, a, gk0a[v]);
                gk0l[v] = fmaf(gd0, lr, gk0l[v]);
                gb0[v] += gd0;
                pa = fmaf(Wt.k0a[v], d[v], pa);
                pl = fmaf(Wt.k0l[v], d[v], pl);
            }
            const float da = mfma4(1.0f, pa, mfma_v4f{0.0f, 0.0f, 0.0f, 0.0f})[0];
            const float dlr = mfma4(1.0f, pl, mfma_v4f{0.0f, 0.0f, 0.0f, 0.0f})[0];
            const float g_a = G * da, g_lr = G * dlr;
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
    if (!live || g != 0) { dLr = dP = 0.0; }                    // (the four lane groups carry the same sequences)
    dLr = wave_sum(dLr); dP = wave_sum(dP);
    if (threadIdx.x == 0) {
        double* o = ws + part * 4;
        o[0] = dLr; o[1] = 0.0; o[2] = dP; o[3] = 0.0;
    }
    // per-lane partials are per (unit 4 g + v, sequence n): sum over the 16 sequences of the lane group
    const int count = 3 * H + (NL - 1) * (H * H + H) + H + 1;
    float* __restrict__ o = wsw + part * count;
    const float vbo = row_sum(gbo);
    if (lane == 0) o[count - 1] = vbo;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int u = 4 * g + v;
        const float a0 = row_sum(gk0a[v]), a1 = row_sum(gk0l[v]), a2 = row_sum(gb0[v]), a3 = row_sum(gwo[v]);
        if (n == 0 && u < H) {
            o[u] = a0; o[H + u] = a1; o[2 * H + u] = a2;
            o[3 * H + (NL - 1) * (H * H + H) + u] = a3;
        }
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            const float vb = row_sum(gbias[l - 1][v]);
            if (n == 0 && u < H) o[3 * H + (l - 1) * (H * H + H) + H * H + u] = vb;
        }
    }
    // gK[l] (D layout: VGPR v of lane 16 g + n = dK_l[in n][out 4 g + v], all 16 sequences summed) -> kernel_l [in][out]
#pragma unroll
    for (int l = 1; l < NL; ++l) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = 4 * g + v;
            if (n < H && i < H) o[3 * H + (l - 1) * (H * H + H) + n * H + i] = gK[l - 1][v];
        }
    }
}

}  // namespace wdf
