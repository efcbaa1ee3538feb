// wdf_ss_dyn.h -- the state-space recursion of wdf_statespace.h with the step's coefficients STREAMED per sample, and with
// the tanh-MLP root (layers.DenseRootModel) as a third root kind: what lifts two restrictions of the lowering --
//
//   * per-sample impedance on ANY tree: in the reference set_resistance replaces R on any ResistiveVoltageSource / Resistor
//     (tf_wdf.py:51-52,80-81) and calc_impedance may run every step (clipper_pot.py:116-117).  The adaptor coefficients
//     (tf_wdf.py:139-145,168-177) are then functions of the sample's resistance; the probed step's tape
//     (lib/wdf_hip/probe_tape.py) is run over the whole resistance channel first (wdf_ss_dyn_rows.h; long tapes: torch) -- the
//     idea of wdf_clipper_mlp_step_prepare, for any tree -- and this kernel reads one coefficient ROW per (sample, sequence);
//   * DenseRootModel terminating any tree (layers.py:72-82): b = -MLP(a, log R_port) (clipper_pot.py:119-121) evaluated in
//     16-lane rows, four sequences per wave (wdf_mlp_row.h); its weight gradient is the dense pass mlp_wgrad_list_kernel
//     (wdf_mlp_step.h, matrix cores) over (a, log R_port, dL/db).
//
// One step, as in wdf_statespace.h:   a = ca.z + da.x ;  b = root(a) ;  z' = A z + Bx x + E b ;  y = cy.z + dy.x + fy b.
// Row layout (ns <= 4 states, ni <= 2 inputs, both run-time):
//     A[ns][ns] | Bx[ns][ni] | E[ns] | ca[ns] | da[ni] | cy[ns] | dy[ni] | fy | R_port          (kN1 = wdf_ss_ncoef + 1 entries)
// addressed as  crow[i * cs + t * ts + b * bs]: per-sample rows [T][kN1][B] (cs = B, ts = kN1 B, bs = 1) or ONE static row
// (cs = 1, ts = bs = 0) through the same code.  x [B][T][ni]; y, dL/dy [T][B]; state stash [T][ns][B]; z0 / zT [ns][B].
// One lane per sequence (the network root: four sequences per wave), sequential in time or in verified chunks: the general
// path, not a fast one (the clipper topology keeps its own kernels).
//
// Reverse sweep (formulas: wdf_statespace.h ss_bwd_step): emits dL/d(row) for EVERY sample -- grow [T][kN1][B] -- because
// with per-sample rows the chain rule to the component values runs per sample too (wdf_ss_dyn_rows_bwd runs the tape
// backwards over it); diode root: the lane sums of gb D_L, gb D_V (-> dL/dIs, dL/dnVt) and dL/dR_port = gb D_L / R_port in the row;
// MLP root: gb = dL/db, a and log R_port per sample for the weight-gradient pass, dL/dR_port = gb (-dMLP/dlr) / R_port.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wdf_mlp_row.h"
#include "wdf_statespace.h"

namespace wdf {

enum { kDynRootNone = 0, kDynRootDiode = 2, kDynRootMlp = 3 };

// Who is who in a wave.  Ideal-source / diode roots: one lane per sequence.  MLP root: one 16-lane DPP ROW per sequence
// (wdf_mlp_row.h: lane j = hidden neuron j, the layer's matrix-vector product as 16 v_fmac_f32_dpp, weights in registers) --
// four sequences per wave, the tree's arithmetic replicated in the row's lanes, lane 0 of the row stores; the per-lane
// evaluation of wdf_mlp.h (~600 dependent FMAs per step) made the network root ten times slower than this.
template <int ROOT>
struct DynLanes {
    static constexpr bool kRow = ROOT == kDynRootMlp;
    static __device__ __forceinline__ int64_t seq() { return kRow ? (int64_t)blockIdx.x * 4 + (threadIdx.x >> 4) : (int64_t)blockIdx.x * 64 + threadIdx.x; }
    static __device__ __forceinline__ bool writer() { return kRow ? (threadIdx.x & 15) == 0 : true; }
    static __device__ __forceinline__ unsigned gate_index() { return kRow ? blockIdx.x >> 4 : blockIdx.x; }   // (gates are per 64 sequences)
};
// Trees of up to kDynMaxS capacitors (round 6: 8; rounds 5: 4).  The kernels are compiled for MS = 4 and MS = 8 state slots -- the
// loops over states are unrolled to MS with the run-time ns masking them -- and the C ABI picks MS = 4 whenever ns <= 4: small
// trees pay nothing for the larger ones.
constexpr int kDynMaxS = 8, kDynMaxI = 2;

template <int MS>
struct DynRowT {                    // one step's coefficients in registers (entries beyond ns / ni are zero)
    float A[MS][MS], Bx[MS][kDynMaxI], E[MS], ca[MS], da[kDynMaxI], cy[MS], dy[kDynMaxI], fy, rp;
};

struct DynLayout {                  // offsets of the row's groups for this (ns, ni)
    int oA, oB, oE, oCa, oDa, oCy, oDy, oFy, oRp, n;
    __host__ __device__ DynLayout(int ns, int ni)
    {
        oA = 0; oB = ns * ns; oE = oB + ns * ni; oCa = oE + ns; oDa = oCa + ns; oCy = oDa + ni; oDy = oCy + ns; oFy = oDy + ni;
        oRp = oFy + 1; n = oRp + 1;
    }
};

template <int MS>
__device__ __forceinline__ DynRowT<MS> dyn_load_row(const float* __restrict__ p, int64_t cs, const DynLayout& L, int ns, int ni)
{
    DynRowT<MS> r;
#pragma unroll
    for (int s = 0; s < MS; ++s) {
        const bool ls = s < ns;
#pragma unroll
        for (int s2 = 0; s2 < MS; ++s2) r.A[s][s2] = (ls && s2 < ns) ? p[(L.oA + s * ns + s2) * cs] : 0.0f;
#pragma unroll
        for (int i = 0; i < kDynMaxI; ++i) r.Bx[s][i] = (ls && i < ni) ? p[(L.oB + s * ni + i) * cs] : 0.0f;
        r.E[s] = ls ? p[(L.oE + s) * cs] : 0.0f;
        r.ca[s] = ls ? p[(L.oCa + s) * cs] : 0.0f;
        r.cy[s] = ls ? p[(L.oCy + s) * cs] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < kDynMaxI; ++i) {
        r.da[i] = i < ni ? p[(L.oDa + i) * cs] : 0.0f;
        r.dy[i] = i < ni ? p[(L.oDy + i) * cs] : 0.0f;
    }
    r.fy = p[L.oFy * cs];
    r.rp = p[L.oRp * cs];
    return r;
}

// The root of this step: b (and, for the reverse sweep, its partials).
//   diode: rootp = {Is, nVt}; L = log(R_port Is / nVt)
//   MLP:   lr = log R_port; b = -MLP(a, lr)
template <int ROOT, bool SYM, int H, int NL>
struct DynRoot {
    float Is, V;
    DiodeStatic d;
    __device__ __forceinline__ void load(const float* __restrict__ rootp, int n_up, int n_down)
    {
        if constexpr (ROOT == kDynRootDiode) {
            Is = rootp[0]; V = rootp[1];
            d = make_diode_static(V, n_up, n_down);
        }
    }
};

// x [B][T][ni] -> y [T][B]; w_in: flat MLP weights (MLP root) -- H, NL are ignored for the other roots
template <int ROOT, bool SYM, int H, int NL, int MS = 4>
__global__ __launch_bounds__(64) void ss_dyn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ crow, int64_t cs,
                                                        int64_t ts, int64_t bs, const float* __restrict__ rootp,
                                                        const float* __restrict__ w_in, int n_up, int n_down,
                                                        float* __restrict__ y, float* __restrict__ zstash,
                                                        const float* __restrict__ z0, float* __restrict__ zT, int ns, int ni,
                                                        int64_t B, int64_t T, int64_t Lc, int64_t W, float* __restrict__ zwarm,
                                                        float* __restrict__ zend, const unsigned* __restrict__ gate,
                                                        const float* __restrict__ zinit, int hidden)
{
    // Time chunks (grid.y = K; Lc = T, K = 1: the sequential recursion): chunk k owns [k Lc, (k + 1) Lc) and starts W steps early from
    // z = 0 -- or from zinit[k] -- (or at t = 0 from z0); the state it ARRIVES with at its first owned step and the state it ends with go to zwarm /
    // zend for ss_tp_verify_kernel; a second, gated launch with K = 1 re-runs the waves that missed (wdf_statespace.h's scheme).
    using LN = DynLanes<ROOT>;
    if (gate != nullptr && gate[LN::gate_index()] == 0u) return;
    const int64_t b_raw = LN::seq();
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const bool writer = LN::writer();
    const int64_t k = blockIdx.y;
    const int64_t t0 = k * Lc, t1 = (t0 + Lc < T) ? t0 + Lc : T;
    const int64_t tw = (k > 0 && t0 > W) ? t0 - W : 0;
    const DynLayout L(ns, ni);
    DynRoot<ROOT, SYM, H, NL> root;
    root.load(rootp, n_up, n_down);
    [[maybe_unused]] RowWeights<NL> RW;
    if constexpr (ROOT == kDynRootMlp) RW = row_load_weights<NL>(w_in, hidden, threadIdx.x & 15, false);
    float z[MS];
#pragma unroll
    for (int s = 0; s < MS; ++s)                            // t = 0: the caller's z0; else zinit [K][ns][B] (a training loop: the
        z[s] = s >= ns ? 0.0f                                     // previous call's state at the sample this warm-up begins) or 0
               : (tw == 0 ? (z0 ? z0[s * B + b] : 0.0f) : (zinit ? zinit[(k * ns + s) * B + b] : 0.0f));
    const float* __restrict__ xp = x + b * T * ni;
    const float* __restrict__ cp = crow + b * bs;
    [[maybe_unused]] float act[NL];
    // rows that do not change in time (ts = 0: one static row, or one row per sequence) are read ONCE, and with them what the
    // root needs of R_port -- log(R_port Is / nVt) / log R_port: a full-precision logarithm per step otherwise
    DynRowT<MS> c = dyn_load_row<MS>(cp + tw * ts, cs, L, ns, ni);
    [[maybe_unused]] float lroot = 0.0f;
    if constexpr (ROOT == kDynRootDiode) lroot = logf(c.rp * root.Is / root.V);
    if constexpr (ROOT == kDynRootMlp) lroot = logf(c.rp);
    for (int64_t t = tw; t < t1; ++t) {
        const bool owned = t >= t0;                                // wave-uniform
        if (t == t0 && zwarm != nullptr && writer) {
#pragma unroll
            for (int s = 0; s < MS; ++s)
                if (s < ns) zwarm[(k * ns + s) * B + b] = z[s];
        }
        if (ts != 0 && t != tw) {                                  // wave-uniform
            c = dyn_load_row<MS>(cp + t * ts, cs, L, ns, ni);
            if constexpr (ROOT == kDynRootDiode) lroot = logf(c.rp * root.Is / root.V);
            if constexpr (ROOT == kDynRootMlp) lroot = logf(c.rp);
        }
        float xv[kDynMaxI];
#pragma unroll
        for (int i = 0; i < kDynMaxI; ++i) xv[i] = i < ni ? xp[t * ni + i] : 0.0f;
        float a = 0.0f;
#pragma unroll
        for (int s = 0; s < MS; ++s) a = fmaf(c.ca[s], z[s], a);
#pragma unroll
        for (int i = 0; i < kDynMaxI; ++i) a = fmaf(c.da[i], xv[i], a);
        float broot = 0.0f;
        if constexpr (ROOT == kDynRootDiode) broot = diode_pair<SYM>(a, lroot, root.d).b;
        if constexpr (ROOT == kDynRootMlp) broot = -row_mlp_fwd<NL>(RW, a, lroot, act);
        float yv = c.fy * broot;
#pragma unroll
        for (int s = 0; s < MS; ++s) yv = fmaf(c.cy[s], z[s], yv);
#pragma unroll
        for (int i = 0; i < kDynMaxI; ++i) yv = fmaf(c.dy[i], xv[i], yv);
        float zn[MS];
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            float acc = c.E[s] * broot;
#pragma unroll
            for (int s2 = 0; s2 < MS; ++s2) acc = fmaf(c.A[s][s2], z[s2], acc);
#pragma unroll
            for (int i = 0; i < kDynMaxI; ++i) acc = fmaf(c.Bx[s][i], xv[i], acc);
            zn[s] = acc;
        }
        if (owned && writer) {
            if (zstash) {
#pragma unroll
                for (int s = 0; s < MS; ++s)
                    if (s < ns) zstash[(t * ns + s) * B + b] = z[s];
            }
            y[t * B + b] = yv;
        }
#pragma unroll
        for (int s = 0; s < MS; ++s) z[s] = zn[s];
    }
    if (zend != nullptr && writer) {
#pragma unroll
        for (int s = 0; s < MS; ++s)
            if (s < ns) zend[(k * ns + s) * B + b] = z[s];
    }
    if (zT && t1 == T && writer) {
#pragma unroll
        for (int s = 0; s < MS; ++s)
            if (s < ns) zT[s * B + b] = z[s];
    }
}

// grow [T][kN1][B]; ws: double[gridDim.y gridDim.x][2] = the (chunk, wave)'s {sum gb D_L, sum gb D_V} (diode root);
// gbroot / ain / lrin [T][B] (MLP root): dL/db, a, log R_port of every step for mlp_wgrad_list_kernel.
//
// Time chunks (round 5, wdf_ss_dyn_bwd_tp) -- EXACT: the adjoint recurrence is linear in the adjoint entering a chunk from the
// future, so the sweep runs in three launches over grid (waves, K):
//   MODE 1  evaluates the root's partials of every step ONCE (the expensive part: omega / the network and its input gradient),
//           leaves them in rpart [T][5][B] = {Da, b, d b/d R_port, D_L, D_V} (and a, log R_port for the network's weight
//           gradient), and carries the chunk's adjoint map: lam leaving the chunk = Phi lam entering + beta (ns homogeneous runs
//           and the particular one) -> rec [K][(ns + 1) ns][B];
//   ss_dyn_bwd_combine_kernel walks a sequence's K maps last to first -> the adjoint entering every chunk, lam_in [K][ns][B];
//   MODE 2  re-walks every chunk from its true entering adjoint reading rpart -- no root evaluation -- and emits what MODE 0 emits.
// MODE 0 (K = 1) is the sequential sweep: root evaluated and rows emitted in one walk.
constexpr int kDynRpart = 5;

template <int ROOT, bool SYM, int H, int NL, int MODE, int MS = 4>
__global__ __launch_bounds__(64) void ss_dyn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ crow, int64_t cs,
                                                        int64_t ts, int64_t bs, const float* __restrict__ rootp,
                                                        const float* __restrict__ w_in, int n_up, int n_down,
                                                        const float* __restrict__ zstash, const float* __restrict__ gy,
                                                        float* __restrict__ grow, double* __restrict__ ws,
                                                        float* __restrict__ gbroot, float* __restrict__ ain,
                                                        float* __restrict__ lrin, float* __restrict__ gz0, int ns, int ni,
                                                        int64_t B, int64_t T, int64_t Lc, float* __restrict__ rpart,
                                                        float* __restrict__ rec, const float* __restrict__ lam_in, int hidden, int acc)
{
    // acc (round 6): the rows do not change in time -- ONE static row, or one row per SEQUENCE (a pot that is constant over a
    // recording: dataimport.py:96 repeats the file's resistance down the channel) -- so dL/d(row entry) is wanted summed over the
    // chunk's steps, not per sample: grow is then [K][kN1][B] (one partial per chunk and lane, double accumulators in registers at
    // compile-time slots) instead of [T][kN1][B] -- 131 MB per call at 1340 x 2048 that only existed to be summed again.
    // (MODE 2 evaluates no root: it runs one lane per sequence whatever the root)
    using LN = DynLanes<(MODE == 2 ? kDynRootNone : ROOT)>;
    constexpr bool kEval = ROOT == kDynRootMlp && MODE != 2;      // the network is evaluated here: 16-lane rows
    const int64_t b_raw = LN::seq();
    const bool writer = LN::writer();
    const bool live = b_raw < B && writer;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    const int64_t k = blockIdx.y;
    const int64_t t0 = k * Lc, t1 = (t0 + Lc < T) ? t0 + Lc : T;
    const DynLayout L(ns, ni);
    DynRoot<ROOT, SYM, H, NL> root;
    root.load(rootp, n_up, n_down);
    [[maybe_unused]] RowWeights<NL> RW;
    if constexpr (kEval) RW = row_load_weights<NL>(w_in, hidden, threadIdx.x & 15, true);
    const float* __restrict__ xp = x + b * T * ni;
    const float* __restrict__ cp = crow + b * bs;
    float lam[MS];
#pragma unroll
    for (int s = 0; s < MS; ++s) lam[s] = (MODE == 2 && s < ns) ? lam_in[(k * ns + s) * B + b] : 0.0f;
    // MODE 1: the homogeneous runs -- hom[j] is the adjoint that enters as the unit vector e_j (dL/dy = 0)
    [[maybe_unused]] float hom[MS][MS];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int j = 0; j < MS; ++j)
#pragma unroll
            for (int s = 0; s < MS; ++s) hom[j][s] = (j == s && j < ns) ? 1.0f : 0.0f;
    }
    double sL = 0.0, sV = 0.0;
    // accumulators of the acc mode at compile-time slots: A s MS + s2 | Bx | E | ca | da | cy | dy | fy | R_port
    constexpr int gA = 0, gB = MS * MS, gE = gB + 2 * MS, gCa = gE + MS, gDa = gCa + MS, gCy = gDa + 2, gDy = gCy + MS, gFy = gDy + 2,
                  gRp = gFy + 1, gN = gRp + 1;
    [[maybe_unused]] double gacc[gN];
    if constexpr (MODE != 1) {
#pragma unroll
        for (int i = 0; i < gN; ++i) gacc[i] = 0.0;
    }
    [[maybe_unused]] float act[NL];
    DynRowT<MS> c = dyn_load_row<MS>(cp + (t1 - 1) * ts, cs, L, ns, ni);    // (rows constant in time: read once, as in the forward)
    [[maybe_unused]] float lroot = 0.0f;
    if constexpr (ROOT == kDynRootDiode && MODE != 2) lroot = logf(c.rp * root.Is / root.V);
    if constexpr (kEval) lroot = logf(c.rp);
    for (int64_t t = t1 - 1; t >= t0; --t) {
        if (ts != 0 && t != t1 - 1) {                              // wave-uniform
            c = dyn_load_row<MS>(cp + t * ts, cs, L, ns, ni);
            if constexpr (ROOT == kDynRootDiode && MODE != 2) lroot = logf(c.rp * root.Is / root.V);
            if constexpr (kEval) lroot = logf(c.rp);
        }
        float xv[kDynMaxI], z[MS];
#pragma unroll
        for (int i = 0; i < kDynMaxI; ++i) xv[i] = i < ni ? xp[t * ni + i] : 0.0f;
#pragma unroll
        for (int s = 0; s < MS; ++s) z[s] = s < ns ? zstash[(t * ns + s) * B + b] : 0.0f;
        const float g = gy[t * B + b];
        float a = 0.0f;
#pragma unroll
        for (int s = 0; s < MS; ++s) a = fmaf(c.ca[s], z[s], a);
#pragma unroll
        for (int i = 0; i < kDynMaxI; ++i) a = fmaf(c.da[i], xv[i], a);
        float broot = 0.0f, Da = 0.0f, Drp = 0.0f;                 // d b / d a, d b / d R_port
        [[maybe_unused]] float DL = 0.0f, DV = 0.0f, lr = 0.0f;
        if constexpr (MODE == 2) {
            if constexpr (ROOT != kDynRootNone) {
                const float* __restrict__ rp_ = rpart + (t * kDynRpart) * B + b;
                Da = rp_[0]; broot = rp_[B]; Drp = rp_[2 * B];
                if constexpr (ROOT == kDynRootDiode) { DL = rp_[3 * B]; DV = rp_[4 * B]; }
            }
        } else
        if constexpr (ROOT == kDynRootDiode) {
            const DiodeOut o = diode_pair<SYM>(a, lroot, root.d);
            broot = o.b;
            const float w0p = o.w0 * fast_rcp(1.0f + o.w0), w1p = o.w1 * fast_rcp(1.0f + o.w1);
            const float l2 = o.lam * o.lam, sp = w0p + w1p;
            Da = fmaf(-2.0f * l2, sp, 1.0f);
            DL = -root.d.two_v * o.lam * (o.m0 * w0p - o.m1 * w1p);
            DV = fmaf(2.0f * l2 * a, sp * fast_rcp(root.V), -2.0f * o.lam * (o.m0 * o.w0 - o.m1 * o.w1));
            Drp = DL / c.rp;                                        // L = log(R_port Is / nVt)
        }
        if constexpr (kEval) {
            lr = lroot;
            broot = -row_mlp_fwd<NL>(RW, a, lr, act);
            float da, dlr;
            row_mlp_grad_in<NL>(RW, act, da, dlr);
            Da = -da;
            Drp = -dlr / c.rp;
        }
        float gb = c.fy * g;
#pragma unroll
        for (int s = 0; s < MS; ++s) gb = fmaf(c.E[s], lam[s], gb);
        const float ga = gb * Da;
        if constexpr (MODE == 1) {
            // what MODE 2 needs of this step's root, and the network's operands; then the adjoint maps -- no rows
            if constexpr (ROOT != kDynRootNone) {
                if (writer) {
                    float* __restrict__ rp_ = rpart + (t * kDynRpart) * B + b;
                    rp_[0] = Da; rp_[B] = broot; rp_[2 * B] = Drp; rp_[3 * B] = DL; rp_[4 * B] = DV;
                }
            }
            if constexpr (ROOT == kDynRootMlp) {
                if (writer) {
                    ain[t * B + b] = a;
                    lrin[t * B + b] = lr;
                }
            }
#pragma unroll
            for (int j = 0; j < MS; ++j) {
                if (j < ns) {
                    float gbj = 0.0f;
#pragma unroll
                    for (int s = 0; s < MS; ++s) gbj = fmaf(c.E[s], hom[j][s], gbj);
                    const float gaj = gbj * Da;
                    float hn[MS];
#pragma unroll
                    for (int s2 = 0; s2 < MS; ++s2) {
                        float v = c.ca[s2] * gaj;
#pragma unroll
                        for (int s = 0; s < MS; ++s) v = fmaf(c.A[s][s2], hom[j][s], v);
                        hn[s2] = v;
                    }
#pragma unroll
                    for (int s = 0; s < MS; ++s) hom[j][s] = hn[s];
                }
            }
        } else if (writer) {
        if (acc) {                                                  // wave-uniform: summed over the chunk's steps (entries past ns / ni stay 0)
#pragma unroll
            for (int s = 0; s < MS; ++s) {
#pragma unroll
                for (int s2 = 0; s2 < MS; ++s2) gacc[gA + s * MS + s2] += (double)(lam[s] * z[s2]);
#pragma unroll
                for (int i = 0; i < kDynMaxI; ++i) gacc[gB + s * 2 + i] += (double)(lam[s] * xv[i]);
                gacc[gE + s] += (double)(lam[s] * broot);
                gacc[gCa + s] += (double)(ga * z[s]);
                gacc[gCy + s] += (double)(g * z[s]);
            }
#pragma unroll
            for (int i = 0; i < kDynMaxI; ++i) {
                gacc[gDa + i] += (double)(ga * xv[i]);
                gacc[gDy + i] += (double)(g * xv[i]);
            }
            gacc[gFy] += (double)(g * broot);
            gacc[gRp] += (double)(gb * Drp);
        } else {
        float* __restrict__ gp = grow + (t * L.n) * B + b;
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            if (s < ns) {
#pragma unroll
                for (int s2 = 0; s2 < MS; ++s2)
                    if (s2 < ns) gp[(L.oA + s * ns + s2) * B] = lam[s] * z[s2];
#pragma unroll
                for (int i = 0; i < kDynMaxI; ++i)
                    if (i < ni) gp[(L.oB + s * ni + i) * B] = lam[s] * xv[i];
                gp[(L.oE + s) * B] = lam[s] * broot;
                gp[(L.oCa + s) * B] = ga * z[s];
                gp[(L.oCy + s) * B] = g * z[s];
            }
        }
#pragma unroll
        for (int i = 0; i < kDynMaxI; ++i) {
            if (i < ni) {
                gp[(L.oDa + i) * B] = ga * xv[i];
                gp[(L.oDy + i) * B] = g * xv[i];
            }
        }
        gp[L.oFy * B] = g * broot;
        gp[L.oRp * B] = gb * Drp;
        }
        if constexpr (ROOT == kDynRootDiode) {
            sL += (double)(gb * DL);
            sV += (double)(gb * DV);
        }
        if constexpr (ROOT == kDynRootMlp) {
            gbroot[t * B + b] = gb;
            if constexpr (MODE == 0) {
                ain[t * B + b] = a;
                lrin[t * B + b] = lr;
            }
        }
        }   // MODE != 1
        float ln[MS];
#pragma unroll
        for (int s2 = 0; s2 < MS; ++s2) {
            float v = fmaf(c.ca[s2], ga, c.cy[s2] * g);
#pragma unroll
            for (int s = 0; s < MS; ++s) v = fmaf(c.A[s][s2], lam[s], v);
            ln[s2] = v;
        }
#pragma unroll
        for (int s = 0; s < MS; ++s) lam[s] = ln[s];
    }
    if constexpr (MODE == 1) {
        // rec [k][j][s][B]: j < ns the columns of Phi (the adjoint that entered as e_j), j = ns the particular run (beta)
        if (!writer) return;
        float* __restrict__ r = rec + (k * (ns + 1) * ns) * B + b;
#pragma unroll
        for (int j = 0; j < MS; ++j)
#pragma unroll
            for (int s = 0; s < MS; ++s)
                if (j < ns && s < ns) r[(j * ns + s) * B] = hom[j][s];
#pragma unroll
        for (int s = 0; s < MS; ++s)
            if (s < ns) r[(ns * ns + s) * B] = lam[s];
        return;
    }
    if (live && gz0 && t0 == 0) {
#pragma unroll
        for (int s = 0; s < MS; ++s)
            if (s < ns) gz0[s * B + b] = lam[s];
    }
    if constexpr (MODE != 1) {
        if (acc && writer && b_raw < B) {                          // the chunk's sums: grow [K][kN1][B]
            float* __restrict__ gp = grow + (k * L.n) * B + b;
#pragma unroll
            for (int s = 0; s < MS; ++s) {
                if (s < ns) {
#pragma unroll
                    for (int s2 = 0; s2 < MS; ++s2)
                        if (s2 < ns) gp[(L.oA + s * ns + s2) * B] = (float)gacc[gA + s * MS + s2];
#pragma unroll
                    for (int i = 0; i < kDynMaxI; ++i)
                        if (i < ni) gp[(L.oB + s * ni + i) * B] = (float)gacc[gB + s * 2 + i];
                    gp[(L.oE + s) * B] = (float)gacc[gE + s];
                    gp[(L.oCa + s) * B] = (float)gacc[gCa + s];
                    gp[(L.oCy + s) * B] = (float)gacc[gCy + s];
                }
            }
#pragma unroll
            for (int i = 0; i < kDynMaxI; ++i) {
                if (i < ni) {
                    gp[(L.oDa + i) * B] = (float)gacc[gDa + i];
                    gp[(L.oDy + i) * B] = (float)gacc[gDy + i];
                }
            }
            gp[L.oFy * B] = (float)gacc[gFy];
            gp[L.oRp * B] = (float)gacc[gRp];
        }
    }
    if constexpr (ROOT == kDynRootDiode) {
        if (!live) { sL = 0.0; sV = 0.0; }
        sL = wave_sum(sL);
        sV = wave_sum(sV);
        if (threadIdx.x == 0 && ws) {
            const int64_t slot = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
            ws[slot * 2 + 0] = sL;
            ws[slot * 2 + 1] = sV;
        }
    }
}

// one lane per sequence: the K chunk maps last to first -> lam_in [K][ns][B], the adjoint entering every chunk (0 enters the last)
static __global__ __launch_bounds__(64) void ss_dyn_bwd_combine_kernel(const float* __restrict__ rec, float* __restrict__ lam_in, int ns,
                                                                       int64_t B, int64_t K)
{
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = b_raw < B ? b_raw : B - 1;
    float lam[kDynMaxS];
#pragma unroll
    for (int s = 0; s < kDynMaxS; ++s) lam[s] = 0.0f;
    for (int64_t k = K - 1; k >= 0; --k) {
        const float* __restrict__ r = rec + (k * (ns + 1) * ns) * B + b;
        float nx[kDynMaxS];
#pragma unroll
        for (int s = 0; s < kDynMaxS; ++s) {
            if (s < ns) lam_in[(k * ns + s) * B + b] = lam[s];
            float v = s < ns ? r[(ns * ns + s) * B] : 0.0f;
#pragma unroll
            for (int j = 0; j < kDynMaxS; ++j)
                if (j < ns && s < ns) v = fmaf(r[(j * ns + s) * B], lam[j], v);
            nx[s] = v;
        }
#pragma unroll
        for (int s = 0; s < kDynMaxS; ++s) lam[s] = nx[s];
    }
}

}  // namespace wdf
