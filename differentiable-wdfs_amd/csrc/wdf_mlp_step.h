// wdf_mlp_step.h -- the RESIDENT training step of the MLP-root pot clipper (clipper_pot.py:94-127,141-177,245-269:
// ClipperModel.forward, MSE + ESR past skip_samples, tape.gradient to the DenseRootModel weights, Adam), gfx950.
//
// wdf_mlp_tp.h / wdf_mlp_mfma.h run that step as ~20 launches steered from the host.  Here it is five, all steered
// on the device (so one HIP graph replays it):
//
//   (1) mlp_step_fwd_kernel<MODE 0>   the forward in verified time chunks, 16 sequences per wave on the matrix cores.
//         A work item is (column of 16 sequences, chunk [t0, t1)): the chunk count is PER COLUMN -- a column of
//         99.1 kOhm sequences forgets its state in ~260 steps, a 10 kOhm one in ~30 (|1 - 2p| per step), so slow
//         columns get more, shorter chunks and every wave ends up with about the same number of steps.
//         A chunk starts W steps early from the state the PREVIOUS call had at that sample (snapshots of every
//         16th state, by absolute time, two sets ping-pong); W is per column and moves by 16 steps from what the
//         verification of the last call measured (below).  Owned steps also produce kappa = dz'/dz, the two loss
//         sums, and per 16-step block the adjoint recurrence's map (A, C1, C2): the adjoint entering the block from
//         the future leaves it as  A gz + ga C1 + gb C2  (dLoss/dy = ga (y - t) + gb y is linear in the two
//         coefficients the loss sums decide later).
//         The last wave of a column to finish (device-scope ticket) verifies the column's chunk boundaries, flags
//         the chunks that arrived off by more than tol, adds the column's loss sums and steers the column's W:
//         besides the arrival miss it looks at the miss 16, 32 and 48 steps BEFORE arrival (what a shorter warm-up
//         would have given), so W shrinks only while there is measured slack and grows before a miss happens.
//   (2) <MODE 1>  flagged chunks again, from the state their predecessor ended in, no warm-up; the column's last
//         finisher checks that the re-run chunks end where they ended before (else the successors' starts are stale).
//   (3) <MODE 2>  what still fails: the whole column sequentially (rare: a wrong result is not an option).
//   (4) mlp_step_wgrad_kernel   exact reverse sweep: a wave takes (column, chunk), composes the block maps of
//         everything after its chunk into the adjoint that enters it, then walks its steps LAST TO FIRST running the
//         scalar adjoint recurrence next to the network's forward / delta chain / outer products (wdf_mlp_mfma.h).
//         No dL/dy array, no adjoint array, no scan launches.
//   (5) mlp_step_reduce_adam_kernel   fixed-order sum of the waves' partial gradients, Adam, the three loss values,
//         call bookkeeping (snapshot parity).
//
// Multi-rank: (1)-(3), then the rank's two loss sums -> all-reduce -> (4) with the global sums -> (5) without Adam ->
// all-reduce of the gradient -> wdf_adam_step.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wdf_mlp_mfma.h"
#include "wdf_optim.h"

namespace wdf {

constexpr int kStepPre = 3;            // states recorded 16, 32, 48 steps before a chunk's first owned step

struct MlpStepCtl {                    // 128 bytes, device; the host reads it, wdf_clipper_mlp_step_state_init writes it
    int call;                          // training steps done
    int parity;                        // snapshot set the next forward READS (it writes the other one)
    int have_snap;                     // 0: no snapshots yet -> cold chunks (z = 0, cold16 units of warm-up)
    int cold16;                        // cold warm-up, 16-step units
    int w_min, w_max;                  // bounds of the per-column warm-up (units)
    int slack;                         // shrink only when the miss `slack` units before arrival is inside the tolerance (1..3)
    int cool_miss, cool_shrink;        // calls to wait after a miss / after a shrink before shrinking again
    float tol;
    float grow_at;                     // arrival miss above grow_at * tol: one unit more (before it becomes a miss)
    float shrink_at;                   // ... the slack miss must be below shrink_at * tol
    int freeze;                        // 1: the controller leaves W alone (profiling one configuration)
    float repair_at;                   // a re-run chunk has met the old trajectory when they agree to repair_at * tol
    int pad0[2];
    // the last call's verdict [parity of the call]: {bad boundaries, max miss bits, columns with a flagged chunk,
    // columns that went sequential}
    int status[2][4];
    int total_flagged, total_sequential;   // since init
    int pad1[6];
};
static_assert(sizeof(MlpStepCtl) == 128, "MlpStepCtl");

struct MlpStepItem { int col, k, t0, t1; };     // chunk k of column col owns steps [t0, t1), both multiples of 16
struct MlpStepCol { int first, K; };            // the column's items are first .. first + K - 1, in time order

struct MlpStepArgs {
    const float* x;        // [B][T]
    const float* p;        // [B][T] (DYN_R) or null: p = G1/(G1+G2) per sample (wdf_clipper_mlp_step_prepare)
    const float* lr;       // [B][T] (DYN_R) or null: log(1/(G1+G2))
    const float* theta2;   // {R, C}: static R only
    const float* w;        // flat weights
    const float* target;   // [T][B]
    float* y;              // [T][B]
    float* zstash;         // [T][B]
    float* kappa;          // [T][B]
    float* maps;           // [T/16][3][B]
    float* snap;           // [2][T/16][B]
    float* zwarm;          // [items][16]   state at t0
    float* zend;           // [items][16]   state at t1
    float* zpre;           // [items][kStepPre][16]
    float* lossblk;        // [T/16][cols][2]  the two loss sums of every 16-step block of a column
    double* colsum;        // [cols][2]
    const MlpStepItem* items;
    const MlpStepCol* cols;
    int* wcol;             // [cols] warm-up units in use
    int* cool;             // [cols]
    int* wpeak;            // [cols] the largest warm-up the column has run with since the plan was installed
    unsigned* ticket;      // [cols]   (left 0)
    unsigned* ticket2;     // [cols]   (left 0)
    unsigned* flag;        // [items]  boundary missed -> MODE 1 re-runs the item
    unsigned* nflag;       // [cols]   flagged items of the column
    unsigned* colseq;      // [cols]   MODE 2 re-runs the column
    float* colmiss;        // [cols][4] the last verification's arrival miss and the misses 1, 2, 3 units before arrival
    unsigned* hwid;        // [items][2] where the item's wave ran (HW_ID, XCC_ID): placement diagnostics
    MlpStepCtl* ctl;
    int64_t B, T, skip;
    int H, n_items, n_cols;
    float fs;
};

// WDF_DBG_STEP (tools/mlp_step_cost.sh; never defined in the product build): kernels with one ingredient removed, to price it
#ifndef WDF_DBG_STEP
#define WDF_DBG_STEP 0
#endif

// tanh costs a third of the forward and a quarter of the reverse sweep (tools/mlp_step_cost.sh).  wdf_mlp.h's tanh_fast is
//     t = exp2(-2 log2(e) |x|) ;  tanh(x) = copysign((1 - t) rcp(1 + t), x)
// -- relative accuracy near 0 (1 - t is exact there), which the chunk verification's 4e-6 needs: the cheaper
// 2 rcp(1 + exp2(s)) - 1 carries an ABSOLUTE error of ~2 ulp of 1 in every small activation, and the warm-ups the
// controller settles on grow by a third with it (measured).  What can go is the multiplication: the factor 2 log2(e) is
// folded into the layer's weights and bias when they are loaded, the MFMA chain delivers s = 2 log2(e) x.
constexpr float kTanhScale = 2.0f * kLog2e;

// act(pre-activation as the MFMA chain delivers it: scaled by kTanhScale for tanh)
template <int ACT> __device__ __forceinline__ float step_act(float s)
{
    if constexpr (WDF_DBG_STEP & 1) return s * 0.5f;             // (pricing tanh)
    if constexpr (ACT == 1) return fmaxf(s, 0.0f);               // relu (layers.py:63-65)
    else {
        const float t = __builtin_amdgcn_exp2f(-fabsf(s));        // exp(-2 |x|) in (0, 1]
        return copysignf((1.0f - t) * fast_rcp(1.0f + t), s);
    }
}
// derivative of the activation from its OUTPUT h
template <int ACT> __device__ __forceinline__ float step_dact(float h)
{
    if constexpr (ACT == 1) return h > 0.0f ? 1.0f : 0.0f;
    else return fmaf(-h, h, 1.0f);
}

// The weights a lane holds (layout: wdf_mlp_mfma.h).  Forward copies carry the activation's scale; the copies the
// derivative chains use do not.  Biases and the output bias sit in 4-register tuples: they ARE the C operand of a
// layer's first MFMA.
template <int NL>
struct StepWeights {
    float k0a[4], k0l[4], b0[4];   // layer 0 (scaled), unit 4 g + v
    float k0a_u[4];                // ... unscaled (d out / d a)
    float a[NL - 1][4];            // slice v of layer l (scaled):  A[i][k] = K_l[in 4 k + v][out i]
    float at[NL - 1][4];           // transposed, unscaled:         A[i][k] = K_l[in i][out 4 k + v]
    mfma_v4f bias[NL - 1];         // (scaled)
    float wo[4];
    mfma_v4f bo;
};

template <int NL, int ACT>
__device__ __forceinline__ StepWeights<NL> step_load_weights(const float* __restrict__ w, int H, int lane)
{
    StepWeights<NL> W;
    const float S = (ACT == 0) ? kTanhScale : 1.0f;
    const int i = lane & 15, k = lane >> 4;
    const int kWo = 3 * H + (NL - 1) * (H * H + H);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int u = 4 * k + v;
        const bool live = u < H;
        W.k0a_u[v] = live ? w[u] : 0.0f;
        W.k0a[v] = S * W.k0a_u[v];
        W.k0l[v] = live ? S * w[H + u] : 0.0f;
        W.b0[v] = live ? S * w[2 * H + u] : 0.0f;
        W.wo[v] = live ? w[kWo + u] : 0.0f;
#pragma unroll
        for (int l = 1; l < NL; ++l) {
            const float* __restrict__ kern = w + 3 * H + (l - 1) * (H * H + H);
            const bool ok = live && i < H;
            W.a[l - 1][v] = ok ? S * kern[u * H + i] : 0.0f;
            W.at[l - 1][v] = ok ? kern[i * H + u] : 0.0f;
            W.bias[l - 1][v] = live ? S * kern[H * H + u] : 0.0f;
        }
    }
    const float bo = w[kWo + H];
    W.bo = mfma_v4f{bo, bo, bo, bo};
    return W;
}

template <int NL, int ACT>
__device__ __forceinline__ float step_mlp_fwd(const StepWeights<NL>& W, float a, float lr, mfma_v4f (&act)[NL])
{
#pragma unroll
    for (int v = 0; v < 4; ++v) act[0][v] = step_act<ACT>(fmaf(lr, W.k0l[v], fmaf(a, W.k0a[v], W.b0[v])));
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        mfma_v4f acc = mfma4(W.a[l - 1][0], act[l - 1][0], W.bias[l - 1]);
#pragma unroll
        for (int v = 1; v < 4; ++v) acc = mfma4(W.a[l - 1][v], act[l - 1][v], acc);
#pragma unroll
        for (int v = 0; v < 4; ++v) act[l][v] = step_act<ACT>(acc[v]);
    }
    const float part = fmaf(W.wo[3], act[NL - 1][3],
                            fmaf(W.wo[2], act[NL - 1][2], fmaf(W.wo[1], act[NL - 1][1], W.wo[0] * act[NL - 1][0])));
    const mfma_v4f s = mfma4(1.0f, part, W.bo);                  // sum over the four lane groups
    return s[0];
}

template <int NL, int ACT>
__device__ __forceinline__ float step_mlp_grad_a(const StepWeights<NL>& W, const mfma_v4f (&act)[NL])
{
    mfma_v4f d;
#pragma unroll
    for (int v = 0; v < 4; ++v) d[v] = W.wo[v] * step_dact<ACT>(act[NL - 1][v]);
#pragma unroll
    for (int l = NL - 1; l >= 1; --l) {
        mfma_v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int v = 0; v < 4; ++v) acc = mfma4(W.at[l - 1][v], d[v], acc);
#pragma unroll
        for (int v = 0; v < 4; ++v) d[v] = acc[v] * step_dact<ACT>(act[l - 1][v]);
    }
    const float part = fmaf(W.k0a_u[3], d[3], fmaf(W.k0a_u[2], d[2], fmaf(W.k0a_u[1], d[1], W.k0a_u[0] * d[0])));
    const mfma_v4f s = mfma4(1.0f, part, mfma_v4f{0.0f, 0.0f, 0.0f, 0.0f});
    return s[0];
}

__device__ __forceinline__ void step_publish(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float step_published(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void step_publish(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double step_published(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// LDS traffic of ONE wave (a wave's LDS instructions complete in order; the fence keeps the compiler from moving them)
__device__ __forceinline__ void step_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the wave's 64 lanes, fixed order, result in every lane (DPP inside the rows, v_readlane across them)
__device__ __forceinline__ float wave_sum_f32(float v)
{
    auto mv = [](float x, auto ctrl) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false)); };
    v += mv(v, std::integral_constant<int, 0xB1>{});
    v += mv(v, std::integral_constant<int, 0x4E>{});
    v += mv(v, std::integral_constant<int, 0x141>{});
    v += mv(v, std::integral_constant<int, 0x140>{});
    auto rl = [](float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); };
    return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
}

// sum over the four lane groups (lanes n, 16 + n, 32 + n, 48 + n), to every lane
__device__ __forceinline__ float groups_sum(float v)
{
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Per-wave running values of a span
struct StepSpan {
    float z;
};

// One block of 16 steps.  OWNED: outputs, kappa, loss sums, the block's adjoint map; otherwise a warm-up block: state only.
// Straight-line code (the whole block is one basic block): the scheduler places step i's kappa chain beside step i + 1's
// forward chain -- they are independent.
template <int NL, bool DYN_R, int ACT, bool OWNED>
__device__ __forceinline__ void mlp_step_block(const MlpStepArgs& A, const StepWeights<NL>& Wt, const MlpClipConsts& c, int64_t tb,
                                               int col, int64_t b, bool live, int g, const float* __restrict__ xp,
                                               const float* __restrict__ pp, const float* __restrict__ lp, StepSpan& sp)
{
    const int64_t B = A.B;
    float xs[16], ps[16], ls[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 xv = *reinterpret_cast<const float4*>(xp + tb + 4 * q);
        xs[4 * q] = xv.x; xs[4 * q + 1] = xv.y; xs[4 * q + 2] = xv.z; xs[4 * q + 3] = xv.w;
        if constexpr (DYN_R) {
            const float4 pv = *reinterpret_cast<const float4*>(pp + tb + 4 * q);
            const float4 lv = *reinterpret_cast<const float4*>(lp + tb + 4 * q);
            ps[4 * q] = pv.x; ps[4 * q + 1] = pv.y; ps[4 * q + 2] = pv.z; ps[4 * q + 3] = pv.w;
            ls[4 * q] = lv.x; ls[4 * q + 1] = lv.y; ls[4 * q + 2] = lv.z; ls[4 * q + 3] = lv.w;
        }
    }
    float tt[4] = {0.0f, 0.0f, 0.0f, 0.0f};                     // group g looks after the steps i = g (mod 4)
    if constexpr (OWNED) {
#pragma unroll
        for (int q = 0; q < 4; ++q) tt[q] = A.target[(tb + 4 * q + g) * B + b];
    }
    float z = sp.z;
    float yk = 0.0f, zk = 0.0f, kk = 0.0f;
    float Am = 1.0f, C1 = 0.0f, C2 = 0.0f, sS = 0.0f, sE = 0.0f;
    mfma_v4f act[NL];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float p = DYN_R ? ps[i] : c.p, lr = DYN_R ? ls[i] : c.lr;
        const float b_diff = z - xs[i];
        const float b_temp = -p * b_diff;
        const float a = z + b_temp;
        const float zn = b_temp - step_mlp_fwd<NL, ACT>(Wt, a, lr, act);     // b_root = -MLP (clipper_pot.py:121)
        if constexpr (OWNED) {
            const float Da = (WDF_DBG_STEP & 16) ? act[NL - 1][0] : -step_mlp_grad_a<NL, ACT>(Wt, act);   // (16: pricing the kappa chain)
            const float kap = Da - p * (1.0f + Da);
            const float yv = 0.5f * (zn + z);
            if ((i & 3) == g) {
                yk = yv; zk = z; kk = kap;
                const float msk = (tb + i >= A.skip) ? 1.0f : 0.0f;
                const float d = yv - tt[i >> 2];
                const float wv = msk * 0.5f * Am * (kap + 1.0f);
                sS = fmaf(msk * d, d, sS);
                sE = fmaf(msk * yv, yv, sE);
                C1 = fmaf(wv, d, C1);
                C2 = fmaf(wv, yv, C2);
            }
            Am *= kap;
            if ((i & 3) == 3 && live) {
                const int64_t o = (tb + (i - 3 + g)) * B + b;
                A.y[o] = yk;
                A.zstash[o] = zk;
                A.kappa[o] = kk;
            }
        }
        z = zn;
    }
    sp.z = z;
    if constexpr (OWNED) {
        C1 = groups_sum(C1);
        C2 = groups_sum(C2);
        if (g == 0 && live) {
            float* __restrict__ m = A.maps + (tb >> 4) * 3 * B + b;
            m[0] = Am; m[B] = C1; m[2 * B] = C2;
        }
        // the block's two loss sums (every lane holds its own steps of its own sequence); published: the column's last
        // finisher of this launch adds the blocks up
        const float bS = wave_sum_f32(live ? sS : 0.0f), bE = wave_sum_f32(live ? sE : 0.0f);
        if ((threadIdx.x & 63) == 0) {
            float* __restrict__ lb = A.lossblk + ((tb >> 4) * A.n_cols + col) * 2;
            step_publish(lb, bS);
            step_publish(lb + 1, bE);
        }
    }
}

// One span of one column: steps [tw, t0) warm up, [t0, t1) are owned (outputs written).  Returns the end state.
// snap_w: the snapshot set this call writes.  PUBLISH: boundary data for a verifier in the same launch.
template <int NL, bool DYN_R, int ACT, bool PUBLISH>
__device__ __forceinline__ float mlp_step_span(const MlpStepArgs& A, const StepWeights<NL>& Wt, const MlpClipConsts& c,
                                               int item, int col, int64_t tw, int64_t t0, int64_t t1, float z,
                                               float* __restrict__ snap_w)
{
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int64_t B = A.B, T = A.T;
    const int64_t b_raw = (int64_t)col * 16 + n;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const float* __restrict__ xp = A.x + b * T;
    const float* __restrict__ pp = DYN_R ? A.p + b * T : nullptr;
    const float* __restrict__ lp = DYN_R ? A.lr + b * T : nullptr;
    StepSpan sp{z};
    for (int64_t tb = tw; tb < t0; tb += 16) {                   // ---- warm-up blocks
        const int64_t j = (t0 - tb) >> 4;                        // 16-step units before the first owned step
        if (g == 0 && j <= kStepPre) step_publish(A.zpre + ((int64_t)item * kStepPre + (j - 1)) * 16 + n, sp.z);
        mlp_step_block<NL, DYN_R, ACT, false>(A, Wt, c, tb, col, b, live, g, xp, pp, lp, sp);
    }
    if (g == 0) {
        if constexpr (PUBLISH) step_publish(A.zwarm + (int64_t)item * 16 + n, sp.z);
        else A.zwarm[(int64_t)item * 16 + n] = sp.z;
    }
    for (int64_t tb = t0; tb < t1; tb += 16) {                   // ---- owned blocks
        // the state a later call may start from (and this call's verifier compares shorter warm-ups against)
        if (g == 0 && live) step_publish(snap_w + (tb >> 4) * B + b, sp.z);
        mlp_step_block<NL, DYN_R, ACT, true>(A, Wt, c, tb, col, b, live, g, xp, pp, lp, sp);
    }
    return sp.z;
}

// the column's loss sums: its blocks in time order, fp64, by the whole wave (fixed order) -> colsum[col]
__device__ __forceinline__ void mlp_step_column_loss(const MlpStepArgs& A, int col)
{
    const int lane = threadIdx.x & 63;
    const int64_t nblk = A.T >> 4;
    double s = 0.0, e = 0.0;
    for (int64_t j = lane; j < nblk; j += 64) {
        const float* lb = A.lossblk + (j * A.n_cols + col) * 2;
        s += (double)step_published(lb);
        e += (double)step_published(lb + 1);
    }
    s = wave_sum_dpp(s);
    e = wave_sum_dpp(e);
    if (lane == 0) { A.colsum[2 * col] = s; A.colsum[2 * col + 1] = e; }
}

// MODE 0: the chunked forward.  MODE 1: flagged chunks again from their predecessors' end states.  MODE 2: flagged
// columns sequentially.  Grid: MODE 0 / 1: one block (wave) per item; MODE 2: one per column.
// Workgroups are FOUR waves (four items; MODE 2: four columns): the dispatcher spreads a workgroup's waves over the four
// SIMDs of its CU, whereas single-wave workgroups land wherever the previous kernel left each CU's SIMD pointer -- with
// one wave per SIMD wanted, 30 doubled-up SIMDs of 1024 cost the forward +55 % (tools/mlp_step_placement.py).
template <int NL, bool DYN_R, int ACT, int MODE>
__global__ __launch_bounds__(256) void mlp_step_fwd_kernel(const MlpStepArgs A)
{
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int unit = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);   // item (MODE 0, 1) or column (MODE 2) of this wave
    if (unit >= (MODE == 2 ? A.n_cols : A.n_items)) return;
    // the two repair launches leave at once unless their item / column is flagged (the common case): ONE load decides, before
    // anything else is asked for (the control block, the item and its column are two more dependent round trips)
    if constexpr (MODE == 1) { if (A.flag[unit] == 0u) return; }
    if constexpr (MODE == 2) { if (A.colseq[unit] == 0u) return; }
    MlpStepCtl* ctl = A.ctl;
    const int par = ctl->parity;                                   // this call reads set `par`, writes the other
    const int64_t nblk = A.T >> 4;
    float* snap_w = A.snap + (int64_t)(par ^ 1) * nblk * A.B;
    const float* snap_r = A.snap + (int64_t)par * nblk * A.B;
    int* st = ctl->status[ctl->call & 1];
    if constexpr (MODE == 2) {
        const int col = unit;
        if (A.colseq[col] == 0u) return;
        const MlpStepCol cinfo = A.cols[col];
        const MlpClipConsts c = DYN_R ? MlpClipConsts{} : mlp_load_consts(A.theta2, A.fs);
        const StepWeights<NL> Wt = step_load_weights<NL, ACT>(A.w, A.H, lane);
        // one item after the other, each from the state the previous one ended in
        float z = 0.0f;                                            // reset(): clipper_pot.py:110-111
        for (int k = 0; k < cinfo.K; ++k) {
            const MlpStepItem it = A.items[cinfo.first + k];
            z = mlp_step_span<NL, DYN_R, ACT, false>(A, Wt, c, cinfo.first + k, col, it.t0, it.t0, it.t1, z, snap_w);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mlp_step_column_loss(A, col);
        if (lane == 0) {
            A.colseq[col] = 0u;
            atomicAdd(&st[3], 1);
            atomicAdd(&ctl->total_sequential, 1);
        }
        return;
    } else {
        const int item = unit;
        const MlpStepItem it = A.items[item];
        const MlpStepCol cinfo = A.cols[it.col];
        if constexpr (MODE == 0) {
            if (item == 0 && lane == 0) {                          // the NEXT call's verdict words start clean
                int* nx = ctl->status[(ctl->call + 1) & 1];
                nx[0] = 0; nx[1] = 0; nx[2] = 0; nx[3] = 0;
            }
        } else {
            if (A.flag[item] == 0u) return;
        }
        if constexpr (MODE == 0) {
            if (lane == 0) {
                A.hwid[2 * item] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
                A.hwid[2 * item + 1] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
            }
        }
        const MlpClipConsts c = DYN_R ? MlpClipConsts{} : mlp_load_consts(A.theta2, A.fs);
        const StepWeights<NL> Wt = step_load_weights<NL, ACT>(A.w, A.H, lane);
        const int64_t b_raw = (int64_t)it.col * 16 + n;
        const bool live = b_raw < A.B;
        const int64_t b = live ? b_raw : A.B - 1;
        int64_t tw = it.t0;
        float z = 0.0f;
        if constexpr (MODE == 0) {
            if (it.k > 0) {
                const int warm = ctl->have_snap;
                const int64_t W = 16 * (int64_t)(warm ? A.wcol[it.col] : ctl->cold16);
                tw = it.t0 > W ? it.t0 - W : 0;
                z = (warm && tw > 0) ? snap_r[(tw >> 4) * A.B + b] : 0.0f;
            }
            z = mlp_step_span<NL, DYN_R, ACT, true>(A, Wt, c, item, it.col, tw, it.t0, it.t1, z, snap_w);
            if (g == 0) step_publish(A.zend + (int64_t)item * 16 + n, z);
        } else {
            // The repair: the flagged chunk again from the state its predecessor ended in -- but only as far as it takes:
            // after every block the new state is held against the OLD trajectory (the stash row of the next step, still
            // untouched); once they agree to repair_at x tol the rest of the chunk stands (a miss of a few tol is forgotten within
            // a few dozen steps).  A chunk that reaches its end still off by more than that carries on into its successor's
            // steps -- unless the successor is being re-run itself (from a state that is now stale): then the column goes
            // sequential.
            const float* __restrict__ xp = A.x + b * A.T;
            const float* __restrict__ pp = DYN_R ? A.p + b * A.T : nullptr;
            const float* __restrict__ lp = DYN_R ? A.lr + b * A.T : nullptr;
            StepSpan sp{A.zend[(int64_t)(item - 1) * 16 + n]};    // (k > 0: only boundaries are flagged)
            const float eps_c = ctl->repair_at * ctl->tol;       // (two fp32 runs of this path sit 1-3e-6 apart whatever they started from)
            int cur = item;                                       // the item whose steps are being rewritten
            int64_t t_end = it.t1;
            for (int64_t tb = it.t0; tb < A.T; tb += 16) {
                if (tb == t_end) {                                // ran through a whole chunk without meeting the old trajectory
                    const bool last = cur + 1 >= cinfo.first + cinfo.K;
                    if (!last && A.flag[cur + 1] != 0u) { if (lane == 0) A.colseq[it.col] = 1u; break; }
                    cur += 1;
                    t_end = A.items[cur].t1;
                }
                if (g == 0 && live) step_publish(snap_w + (tb >> 4) * A.B + b, sp.z);
                const float z_old_next = (tb + 16 < A.T) ? A.zstash[(tb + 16) * A.B + b] : sp.z;
                const float z_end_old = A.zend[(int64_t)cur * 16 + n];
                mlp_step_block<NL, DYN_R, ACT, true>(A, Wt, c, tb, it.col, b, live, g, xp, pp, lp, sp);
                // (the stash row of step tb + 16 holds the state BEFORE that step; at a chunk's end the recorded end state)
                const float ref = (tb + 16 == t_end) ? z_end_old : z_old_next;
                const bool off = live && !(fabsf(sp.z - ref) <= eps_c);
                if (tb + 16 < A.T && __ballot(off) == 0ull) break;
            }
        }
        // ---- the column's last finisher
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's outputs and boundary states have landed
        unsigned* tk = (MODE == 0 ? A.ticket : A.ticket2) + it.col;
        const unsigned expect = MODE == 0 ? (unsigned)cinfo.K : A.nflag[it.col];
        unsigned old = 0;
        if (lane == 0) old = atomicAdd(tk, 1u);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old != expect - 1u) return;
        if (lane == 0) *tk = 0u;                                   // left clean for the next launch
        mlp_step_column_loss(A, it.col);
        const float tol = ctl->tol;
        if constexpr (MODE == 0) {
            // every boundary of the column: lane (q, n) takes the boundaries k = 1 + q, 5 + q, ...
            float m0 = 0.0f, mp[kStepPre] = {0.0f, 0.0f, 0.0f};
            int nbad = 0;
            const int warm = ctl->have_snap;
            const int wc = warm ? A.wcol[it.col] : ctl->cold16;
            for (int k = 1 + g; k < cinfo.K; k += 4) {
                const int64_t ik = cinfo.first + k;
                const float zw = step_published(A.zwarm + ik * 16 + n);
                const float ze = step_published(A.zend + (ik - 1) * 16 + n);
                const float m = live ? fabsf(zw - ze) : 0.0f;
                const bool bad = !(m <= tol);
                const unsigned long long any = __ballot(bad) >> (16 * g) & 0xffffull;   // (the 16 lanes of this boundary)
                if (n == 0) A.flag[ik] = any ? 1u : 0u;
                nbad += (any && n == 0) ? 1 : 0;
                m0 = fmaxf(m0, m);
                const int64_t t0k = A.items[ik].t0;
#pragma unroll
                for (int j = 1; j <= kStepPre; ++j) {
                    // the miss a warm-up j units shorter would have arrived with (needs j <= W and the sample to exist)
                    float mj = j <= wc ? 0.0f : 3.0e38f;         // (a warm-up that reaches back to t = 0 is exact)
                    if (j <= wc && t0k - 16 * j > 0) {
                        const float zp = step_published(A.zpre + (ik * kStepPre + (j - 1)) * 16 + n);
                        const float zt = step_published(snap_w + ((t0k >> 4) - j) * A.B + b);
                        mj = live ? fabsf(zp - zt) : 0.0f;
                    }
                    mp[j - 1] = fmaxf(mp[j - 1], mj);
                }
            }
            m0 = wave_max_dpp(m0);
#pragma unroll
            for (int j = 0; j < kStepPre; ++j) mp[j] = wave_max_dpp(mp[j]);
            nbad = wave_sum_dpp(nbad);
            if (lane == 0) {
                if (cinfo.K > 0) A.flag[cinfo.first] = 0u;
                A.nflag[it.col] = (unsigned)nbad;
                A.colmiss[4 * it.col] = m0;
#pragma unroll
                for (int j = 0; j < kStepPre; ++j) A.colmiss[4 * it.col + 1 + j] = mp[j];
                if (nbad) { atomicAdd(&st[0], nbad); atomicAdd(&st[2], 1); atomicAdd(&ctl->total_flagged, nbad); }
                if (m0 > 0.0f) atomicMax(&st[1], __float_as_int(m0));
                // ---- steer the column's warm-up for the next call
                if (warm && !ctl->freeze && cinfo.K > 1) {
                    int W = A.wcol[it.col], cl = A.cool[it.col];
                    const float ms = mp[ctl->slack - 1];
                    if (nbad) { W += 2; cl = ctl->cool_miss; }
                    // (a slow column's miss doubles per 16 steps less, and in a swing of the weights it doubles per call: one
                    //  unit more as soon as TWO units less would miss keeps pace with it)
                    else if (m0 > ctl->grow_at * tol || mp[1] > tol) { W += 1; cl = cl > ctl->cool_shrink ? cl : ctl->cool_shrink; }
                    else if (cl > 0) { cl -= 1; }
                    else if (mp[kStepPre - 1] <= ctl->shrink_at * tol && W - 2 >= ctl->w_min) { W -= 2; cl = ctl->cool_shrink; }   // ample slack
                    else if (ms <= ctl->shrink_at * tol && W > ctl->w_min) { W -= 1; cl = ctl->cool_shrink; }
                    W = W > ctl->w_max ? ctl->w_max : W;
                    if (wc > A.wpeak[it.col]) A.wpeak[it.col] = wc;
                    A.wcol[it.col] = W;
                    A.cool[it.col] = cl;
                }
            }
        } else {
            if (lane == 0) {
                for (int k = 1; k < cinfo.K; ++k) A.flag[cinfo.first + k] = 0u;
                A.nflag[it.col] = 0u;
            }
        }
    }
}

// ---- the weight gradient's per-wave accumulators: the forward / delta chain / outer products of ONE step of 16 sequences
// (lane n + 16 g: sequence n, units 4 g + v), shared by the resident step's reverse sweep and by the sample-list pass below.
// LDS of one wave: [step parity][layer][gd | h][sequence][unit, row padded to 20 floats].  A wave's LDS instructions
// complete in order, so a write -> read pair of ONE wave needs no barrier, and tiles of their own per (parity, layer)
// leave the scheduler free to run a step's transposes under the neighbouring step's matrix work.
constexpr int kStepTile = 16 * 20;
template <int NL> constexpr int step_wave_lds() { return 2 * (NL - 1) * 2 * kStepTile; }   // floats per wave

template <int NL, int ACT>
struct StepGradAcc {
    float gk0a[4], gk0l[4], gb0[4], gwo[4], gbo;
    float gbias[NL - 1][4];
    mfma_v4f gK[NL - 1];

    __device__ __forceinline__ void zero()
    {
        gbo = 0.0f;
#pragma unroll
        for (int v = 0; v < 4; ++v) gk0a[v] = gk0l[v] = gb0[v] = gwo[v] = 0.0f;
#pragma unroll
        for (int l = 0; l < NL - 1; ++l) {
            gK[l] = mfma_v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int v = 0; v < 4; ++v) gbias[l][v] = 0.0f;
        }
    }

    // += G dMLP(a, lr)/dW for the 16 sequences of the wave.  tiles: the wave's LDS; parity: alternates from step to step.
    __device__ __forceinline__ void add(const StepWeights<NL>& Wt, float a, float lr, float G, float* __restrict__ tiles, int parity,
                                        int n, int g)
    {
        mfma_v4f act[NL];
        (void)step_mlp_fwd<NL, ACT>(Wt, a, lr, act);
        gbo += G;
        mfma_v4f d;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            gwo[v] = fmaf(G, act[NL - 1][v], gwo[v]);
            d[v] = Wt.wo[v] * step_dact<ACT>(act[NL - 1][v]);
        }
#pragma unroll
        for (int l = NL - 1; l >= 1; --l) {
            mfma_v4f gd;
#pragma unroll
            for (int v = 0; v < 4; ++v) { gd[v] = G * d[v]; gbias[l - 1][v] += gd[v]; }
            // the two transposes through LDS: a lane writes its four units of sequence n as one 16-byte store
            // (tile[sequence][unit]) and reads, for outer-product MFMA q, unit n of sequence 4 g + q
            float* __restrict__ tg = tiles + ((parity * (NL - 1) + (l - 1)) * 2) * kStepTile;
            float* __restrict__ th = tg + kStepTile;
            *reinterpret_cast<float4*>(tg + n * 20 + 4 * g) = make_float4(gd[0], gd[1], gd[2], gd[3]);
            *reinterpret_cast<float4*>(th + n * 20 + 4 * g) = make_float4(act[l - 1][0], act[l - 1][1], act[l - 1][2], act[l - 1][3]);
            mfma_v4f gdT, hT;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gdT[q] = tg[(4 * g + q) * 20 + n];
                hT[q] = th[(4 * g + q) * 20 + n];
            }
            if constexpr (WDF_DBG_STEP & 2) { gdT = gd; hT = act[l - 1]; }   // (pricing the transposes: the LDS traffic above goes dead)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (WDF_DBG_STEP & 4) gK[l - 1][q] += gdT[q] * hT[q];   // (pricing the outer-product MFMAs)
                else gK[l - 1] = mfma4(gdT[q], hT[q], gK[l - 1]);
            }
            mfma_v4f nd = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if constexpr (WDF_DBG_STEP & 8) nd[v] += Wt.at[l - 1][v] * d[v];    // (pricing the delta chain's MFMAs)
                else nd = mfma4(Wt.at[l - 1][v], d[v], nd);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) d[v] = nd[v] * step_dact<ACT>(act[l - 1][v]);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float gd0 = G * d[v];
            gk0a[v] = fmaf(gd0, a, gk0a[v]);
            gk0l[v] = fmaf(gd0, lr, gk0l[v]);
            gb0[v] += gd0;
        }
    }

    // the wave's partial, in the flat weight order, to o[count] (per-lane partials are per (unit 4 g + v, sequence n): summed
    // over the 16 sequences of the lane group)
    __device__ __forceinline__ void store(float* __restrict__ o, int H, int n, int g, int lane) const
    {
        const int count = 3 * H + (NL - 1) * (H * H + H) + H + 1;
        // (the four lane groups carry the same sequences: gbo is the same in all of them)
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
#pragma unroll
        for (int l = 1; l < NL; ++l) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = 4 * g + v;
                if (n < H && i < H) o[3 * H + (l - 1) * (H * H + H) + n * H + i] = gK[l - 1][v];
            }
        }
    }
};

// ---- (4) the reverse sweep: adjoint recurrence + weight gradient, one wave per (column, chunk of Lw steps) ----------
// sums: the two GLOBAL loss sums {S, E} (multi-rank: after the all-reduce) or null -> this rank's colsum.
// wsw: float[workgroups][count] (a workgroup = four consecutive parts, part = chunk * n_cols + column).  adam_step (or null): bumped once here, so the
// reduce / Adam kernel behind reads the step count it has to use.
template <int NL, bool DYN_R, int ACT, int BS = 8>
__global__ __launch_bounds__(256) void mlp_step_wgrad_kernel(const MlpStepArgs A, const double* __restrict__ sums, double n_global,
                                                             double eps_energy, int64_t Lw, int Kw, float* __restrict__ wsw,
                                                             float* __restrict__ gcoef_out, int32_t* adam_step)
{
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int64_t B = A.B, T = A.T;
    const int64_t nparts = (int64_t)A.n_cols * Kw;
    const int64_t part_raw = (int64_t)blockIdx.x * 4 + wv;        // four (column, chunk) parts per workgroup, one per wave
    const bool active = part_raw < nparts;                        // (a wave past the end runs the last part again, adds nothing)
    const int64_t part = active ? part_raw : nparts - 1;
    const int col = (int)(part % A.n_cols), chunk = (int)(part / A.n_cols);
    if (adam_step && part_raw == 0 && lane == 0) *adam_step += 1;
    const int64_t b_raw = (int64_t)col * 16 + n;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const int64_t t0 = (int64_t)chunk * Lw, t1 = !active ? t0 : ((t0 + Lw < T) ? t0 + Lw : T);
    // ---- the loss coefficients: dLoss/dy = ga (y - t) + gb y past skip (clipper_pot.py:146-156,177; wdf_elementwise.h)
    double S, E;
    if (sums) { S = sums[0]; E = sums[1]; }
    else {
        double s = 0.0, e = 0.0;
        for (int cc = lane; cc < A.n_cols; cc += 64) { s += A.colsum[2 * cc]; e += A.colsum[2 * cc + 1]; }
        S = wave_sum_dpp(s); E = wave_sum_dpp(e);
    }
    E += eps_energy;
    const double esr = sqrt(S / E / n_global);
    const float ga = (float)(2.0 / n_global + (esr > 0.0 ? 1.0 / (esr * E * n_global) : 0.0));
    const float gb = (float)(-esr / E);
    if (gcoef_out && part_raw == 0 && lane == 0) { gcoef_out[0] = ga; gcoef_out[1] = gb; }
    const MlpClipConsts c = DYN_R ? MlpClipConsts{} : mlp_load_consts(A.theta2, A.fs);
    const StepWeights<NL> Wt = step_load_weights<NL, ACT>(A.w, A.H, lane);
    // ---- the adjoint that enters this chunk: the maps of every later 16-step block, last block first
    float gz = 0.0f;
    {
        const int64_t j0 = t1 >> 4, j1 = T >> 4;
        int64_t j = j1;
        for (; j - 8 >= j0; j -= 8) {
            float ma[8], m1[8], m2[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float* __restrict__ m = A.maps + (j - 1 - q) * 3 * B + b;
                ma[q] = m[0]; m1[q] = m[B]; m2[q] = m[2 * B];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) gz = fmaf(ma[q], gz, fmaf(ga, m1[q], gb * m2[q]));
        }
        for (; j > j0; --j) {
            const float* __restrict__ m = A.maps + (j - 1) * 3 * B + b;
            gz = fmaf(m[0], gz, fmaf(ga, m[B], gb * m[2 * B]));
        }
    }
    constexpr int kWaveLds = step_wave_lds<NL>();                 // the transposes' LDS tiles (StepGradAcc), per wave
    __shared__ __attribute__((aligned(16))) float tbuf_all[4 * kWaveLds];
    const float* __restrict__ xp = A.x + b * T;
    const float* __restrict__ pp = DYN_R ? A.p + b * T : nullptr;
    const float* __restrict__ lp = DYN_R ? A.lr + b * T : nullptr;
    StepGradAcc<NL, ACT> acc;
    acc.zero();
    static_assert(BS == 4 || BS == 8 || BS == 16, "staging block");
    for (int64_t tb = t1 - BS; tb >= t0; tb -= BS) {         // (chunk bounds are multiples of 16)
        float xs[BS], ps[BS], ls[BS], zz[BS], kp[BS], gs[BS];
#pragma unroll
        for (int q = 0; q < BS / 4; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(xp + tb + 4 * q);
            xs[4 * q] = xv.x; xs[4 * q + 1] = xv.y; xs[4 * q + 2] = xv.z; xs[4 * q + 3] = xv.w;
            if constexpr (DYN_R) {
                const float4 pv = *reinterpret_cast<const float4*>(pp + tb + 4 * q);
                const float4 lv = *reinterpret_cast<const float4*>(lp + tb + 4 * q);
                ps[4 * q] = pv.x; ps[4 * q + 1] = pv.y; ps[4 * q + 2] = pv.z; ps[4 * q + 3] = pv.w;
                ls[4 * q] = lv.x; ls[4 * q + 1] = lv.y; ls[4 * q + 2] = lv.z; ls[4 * q + 3] = lv.w;
            }
        }
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            const int64_t o = (tb + i) * B + b;
            zz[i] = A.zstash[o];
            kp[i] = A.kappa[o];
            const float yv = A.y[o], tv = A.target[o];
            gs[i] = (live && tb + i >= A.skip) ? fmaf(ga, yv - tv, gb * yv) : 0.0f;     // dLoss/dy[n]
        }
#pragma unroll
        for (int i = BS - 1; i >= 0; --i) {
            const float p = DYN_R ? ps[i] : c.p, lr = DYN_R ? ls[i] : c.lr;
            // adjoint recurrence (wdf_mlp_tp.h):  g_b2n[n] = gz[n+1] + g[n]/2 ;  gz[n] = kappa[n] g_b2n[n] + g[n]/2
            const float g_b2n = fmaf(0.5f, gs[i], gz);
            gz = fmaf(kp[i], g_b2n, 0.5f * gs[i]);
            const float G = live ? -g_b2n : 0.0f;                // b_root = -MLP; shadow sequences add nothing
            const float z = zz[i];
            const float a = fmaf(-p, z - xs[i], z);
            acc.add(Wt, a, lr, G, tbuf_all + wv * kWaveLds, i & 1, n, g);
        }
    }
    const int H = A.H;
    const int count = 3 * H + (NL - 1) * (H * H + H) + H + 1;
    // the wave's partial goes to LDS (its own transposes' tiles are free now), the workgroup's four are added there, in
    // wave order, and ONE partial per workgroup goes out: wsw float[workgroups][count]
    acc.store(tbuf_all + wv * kWaveLds, H, n, g, lane);
    __syncthreads();
    float* __restrict__ og = wsw + (int64_t)blockIdx.x * count;
    for (int idx = threadIdx.x; idx < count; idx += 256) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) t += tbuf_all[q * kWaveLds + idx];
        og[idx] = t;
    }
}

// ---- the same weight gradient over a LIST of S independent samples:  gw = -sum_n gb[n] dMLP(ain[n], lr[n])/dW  (lrin null:
// lr = log P1.R from theta2).  What tape.gradient returns for DenseRootModel's variables once a reverse sweep has left
// (a, log R, dL/db) of every sample (layers.py:72-82, clipper_pot.py:181-184): wdf_clipper_mlp_bwd's and wdf_ss_dyn_bwd's
// emission.  A wave takes `per_wave` consecutive samples (a multiple of 128), 16 at a time on the matrix cores; partials as
// above: wsw float[workgroups][count].
template <int NL, int BS = 8>
__global__ __launch_bounds__(256) void mlp_wgrad_list_kernel(const float* __restrict__ ain, const float* __restrict__ lrin,
                                                             const float* __restrict__ gb, const float* __restrict__ theta2, float fs,
                                                             const float* __restrict__ w, int H, int64_t S, int64_t per_wave,
                                                             float* __restrict__ wsw)
{
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int64_t s0 = ((int64_t)blockIdx.x * 4 + wv) * per_wave;
    const int64_t s1 = s0 + per_wave < S ? s0 + per_wave : S;     // (a wave past the end: s0 >= s1, adds nothing)
    const float lr_static = lrin ? 0.0f : mlp_load_consts(theta2, fs).lr;
    const StepWeights<NL> Wt = step_load_weights<NL, 0>(w, H, lane);
    constexpr int kWaveLds = step_wave_lds<NL>();
    __shared__ __attribute__((aligned(16))) float tbuf_all[4 * kWaveLds];
    StepGradAcc<NL, 0> acc;
    acc.zero();
    for (int64_t sb = s0; sb < s1; sb += 16 * BS) {
        float as[BS], ls[BS], gs[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            const int64_t idx = sb + 16 * i + n;
            const bool ok = idx < s1;
            const int64_t j = ok ? idx : s1 - 1;
            as[i] = ain[j];
            ls[i] = lrin ? lrin[j] : lr_static;
            gs[i] = ok ? -gb[j] : 0.0f;                          // L depends on b_root = -MLP
        }
#pragma unroll
        for (int i = 0; i < BS; ++i) acc.add(Wt, as[i], ls[i], gs[i], tbuf_all + wv * kWaveLds, i & 1, n, g);
    }
    const int count = 3 * H + (NL - 1) * (H * H + H) + H + 1;
    acc.store(tbuf_all + wv * kWaveLds, H, n, g, lane);
    __syncthreads();
    float* __restrict__ og = wsw + (int64_t)blockIdx.x * count;
    for (int idx = threadIdx.x; idx < count; idx += 256) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) t += tbuf_all[q * kWaveLds + idx];
        og[idx] = t;
    }
}

// ---- (5) fixed-order sum of the waves' partials, Adam, the loss values, call bookkeeping -----------------------------
// block (64, 16): thread (i, s) adds partials s, s + 16, ... of weight i in double, the 16 slices are then added in
// order (mlp_wgrad_reduce_wide_kernel's sum).  m == null: no update (multi-rank: all-reduce gw, then wdf_adam_step).
static __global__ __launch_bounds__(1024) void mlp_step_reduce_adam_kernel(
    const float* __restrict__ ws, int nblk, int count, float* __restrict__ gw, float* __restrict__ w, float* __restrict__ m,
    float* __restrict__ v, const int32_t* __restrict__ step, const float* __restrict__ lr, float b1, float b2, float eps,
    const double* __restrict__ colsum, int n_cols, const double* __restrict__ sums, double n_global, double eps_energy,
    float* __restrict__ loss3, MlpStepCtl* ctl)
{
    __shared__ double sh[16][64];
    const int i = blockIdx.x * 64 + threadIdx.x, sl = threadIdx.y;
    double acc = 0.0;
    if (i < count) {
        double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int b = sl;
        for (; b + 7 * 16 < nblk; b += 8 * 16) {
            float q8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) q8[q] = ws[(int64_t)(b + 16 * q) * count + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] += (double)q8[q];
        }
        for (; b < nblk; b += 16) a[0] += (double)ws[(int64_t)b * count + i];
        acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    sh[sl][threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0 && i < count) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sh[q][threadIdx.x];
        const float g = (float)t;
        gw[i] = g;
        if (m) {                                                   // tf.keras Adam (wdf_optim.h), step count already bumped
            const int ts = *step;
            const double c1 = 1.0 - ipow((double)b1, ts), c2 = 1.0 - ipow((double)b2, ts);
            const float mi = b1 * m[i] + (1.0f - b1) * g;
            const float vi = b2 * v[i] + (1.0f - b2) * g * g;
            m[i] = mi;
            v[i] = vi;
            const float lr_t = (float)((double)lr[i] * sqrt(c2) / c1);
            w[i] = w[i] - lr_t * mi / (sqrtf(vi) + eps);
        }
    }
    if (blockIdx.x == 0 && sl == 1) {                              // (another wave of block 0: off the reduction's path)
        double S, E;
        if (sums) { S = sums[0]; E = sums[1]; }
        else {
            double s = 0.0, e = 0.0;
            for (int cc = threadIdx.x; cc < n_cols; cc += 64) { s += colsum[2 * cc]; e += colsum[2 * cc + 1]; }
            S = wave_sum_dpp(s); E = wave_sum_dpp(e);
        }
        if (threadIdx.x == 0) {
            const double mse = S / n_global, esr = sqrt(S / (E + eps_energy) / n_global);
            if (loss3) { loss3[0] = (float)mse; loss3[1] = (float)esr; loss3[2] = (float)(mse + esr); }
            ctl->call += 1;
            ctl->parity ^= 1;
            ctl->have_snap = 1;
        }
    }
}

// this rank's two loss sums out of the column sums (multi-rank: the all-reduce's payload)
static __global__ __launch_bounds__(64) void mlp_step_sums_kernel(const double* __restrict__ colsum, int n_cols, double* __restrict__ sums)
{
    double s = 0.0, e = 0.0;
    for (int cc = threadIdx.x; cc < n_cols; cc += 64) { s += colsum[2 * cc]; e += colsum[2 * cc + 1]; }
    s = wave_sum_dpp(s); e = wave_sum_dpp(e);
    if (threadIdx.x == 0) { sums[0] = s; sums[1] = e; }
}

// p = G1/(G1 + G2), lr = log(1/(G1 + G2)) per sample of the resident pot channel: set_resistance + calc_impedance of every
// step (clipper_pot.py:116-117, tf_wdf.py:168-177) hoisted out of the training loop -- they depend on the data and on C, not
// on the weights being trained.  Same arithmetic as mlp_step_coeffs<true> (wdf_mlp.h).
static __global__ __launch_bounds__(256) void mlp_step_prepare_kernel(const float* __restrict__ r, const float* __restrict__ theta2,
                                                                      float fs, int64_t n, float* __restrict__ p, float* __restrict__ lr)
{
    const MlpClipConsts c = mlp_load_consts(theta2, fs);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float pv, Rp, lv;
        mlp_step_coeffs<true>(c, r[i], pv, Rp, lv);
        p[i] = pv;
        lr[i] = lv;
    }
}

}  // namespace wdf
