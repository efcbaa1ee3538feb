// wdf_ss_nl_step.h -- the one-pass MSE training step of small state-space trees with a DIODE-PAIR root (the HPF diode clipper,
// HPFDiodeClipper.h:28-32; tf_wdf.py:179-214 for the root): forward, loss and the gradient of every coefficient in ONE sweep over
// the resident training set, by carrying the forward-mode tangents of the state next to the state.
//
//   a = ca.z + da.x      b = diode_pair(a; L, V)      y = fy b + cy.z + dy.x      z' = A z + Bx x + E b
//
// A tangent S_c = dz/d theta_c (theta: the entries of A, Bx, E, ca, da, and the diode's L = log(Rp Is / nVt) and V = nVt) obeys
//   db_c = Da (ca.S_c) + beta_c          S_c' = A S_c + E db_c + (direct term of A / Bx / E entries)
//   dy_c = cy.S_c + fy db_c              dLoss/d theta_c += g dy_c,   g = gscale (y - target)
// with Da = db/da and beta_c = db/d theta_c at fixed state (Da z_s, Da x_i, DL, DV).  cy, dy, fy enter y only.
// The reference's reverse-mode tape (lpf.py:87-90: tape.gradient through the unrolled loop) gives the same numbers; this
// direction needs neither the state stash nor dLoss/dy as an array: 12 bytes per sample (x, target in; y out) instead of 28.
//
// Time parallelism.  The time axis is cut into K chunks (one wave each; a lane holds one or two sequences).  The recursion of
// the tangents is LINEAR given the state trajectory:  S_t = S0_t + Psi_t S_start  (S0: from zero, Psi_t: product of the step
// Jacobians M = A + E Da ca^T since the chunk began), so a chunk runs from S = 0, carries Psi and H = sum g (dy/dz) Psi, and the
// finishing kernel walks a sequence's chunks in order: dLoss += H . S_start, S_start <- Psi_end S_start + S0_end -- exact.
// The STATE is nonlinear: chunk k starts w steps early (warm-up) from a PREDICTED state -- the previous call's state at that
// very sample plus its tangents times the change of the coefficients since (first-order Taylor: a training loop moves the
// coefficients by ~1e-3 of their value per epoch) -- and the finishing kernel compares the state a chunk arrives with to the
// state its predecessor ended with.  A group of sequences with a miss > tol is re-run SEQUENTIALLY by its finishing wave
// (exact), so results are within tol of the sequential recursion or ARE it.  The warm-up length is steered on the device from
// the largest miss (NlStepCtl); the snapshots {z, S} for the next call are taken w_snap steps before each chunk's end.
//
// Launches per step: ss_nl_step_kernel, ss_nl_step_finish_kernel (+ the probe, wdf_ss_step.h, which also carries the
// optimizers' queued updates): three.
#pragma once
#include "wdf_ss_step.h"

namespace wdf {

struct NlStepCtl {                 // 128 bytes, device resident
    int call, parity, have_snap, w_cur;       // w_cur: this call's warm-up == where the previous call took its snapshots
    int w_snap, w_min, w_max, cool;           // w_snap: where this call takes them == the next call's warm-up
    float tol, grow_at, shrink_at;            // grow when miss > grow_at tol; shrink when miss <= shrink_at tol
    int cool_miss;                            // calls without shrinking after a miss
    int n_bad; float max_miss; int gated_groups, total_gated;   // verdict of the last call; groups re-run since the plan
    int acc_bad, acc_miss, acc_gated;         // (unused: the groups' verdicts travel in their partial sums)
    int w_used;                               // the warm-up the last call ran with
    int pad[12];
};
static_assert(sizeof(NlStepCtl) == 128, "NlStepCtl is 128 bytes");

template <int NS, int NI>
struct NlDims {
    using C = SSCoef<NS, NI>;
    static constexpr int nT = C::oCy + 2;                        // tangents: A, Bx, E, ca, da entries, then L and V
    static constexpr int nG = C::kN + 2;                         // gradient entries: every coefficient, then L and V
    static constexpr int nRec = 2 * NS + NS * NS + nT * NS + NS; // zwarm, zend, Psi_end, S0_end, H
    static constexpr int oZw = 0, oZe = NS, oPsi = 2 * NS, oS = oPsi + NS * NS, oH = oS + nT * NS;
    static constexpr int nSnap = NS + nT * NS + NS * NS;         // z, S (S0 until the finishing kernel adds Psi S_start), Psi
    static constexpr int sZ = 0, sS = NS, sPsi = NS + nT * NS;
    __host__ __device__ static constexpr int gidx(int c) { return c < C::oCy ? c : C::kN + (c - C::oCy); }
};

struct NlStepArgs {
    const float* x;            // [T][NI][B]
    const float* coef;         // SSCoef order (the probe's output)
    const float* pIs;          // the root's own values, on the device
    const float* pV;
    const float* pRp;          // the port resistance the root sees (the probe's last output)
    const float* target;       // [T][B]
    float* y;                  // [T][B]
    NlStepCtl* ctl;
    float* coef_prev;          // [kN + 2]: the coefficients, L, V of the previous call (what the snapshots were taken with)
    float* rec;                // [K][nRec][B]
    float* snap;               // [2][K][nSnap][B]
    double* gpart;             // [K][groups][nG + 1]: the chunks' own sums
    double* part;              // [groups][nG + 1]
    unsigned* ticket;
    const double* jac;         // [kN + 1][n_tree] (row kN: the port resistance)
    float* out;                // [1 + n_tree + 2]
    float* loss;               // (or null) <- gscale / 2 x the sum of squared errors: the mean squared error when gscale = 2 / (B T)
    int64_t B, T, L;
    int K, groups, n_tree, n_up, n_down;
    float gscale;
};

template <int NS, int NI, typename V, bool PSI>
struct NlU {
    using D = NlDims<NS, NI>;
    V z[NS];
    V S[D::nT][NS];
    V P[PSI ? NS : 1][NS];                                        // P[j] = Psi e_j
    __device__ __forceinline__ void start()
    {
        const V zero = vsplat<V>(0.0f);
#pragma unroll
        for (int c = 0; c < D::nT; ++c)
#pragma unroll
            for (int s = 0; s < NS; ++s) S[c][s] = zero;
        if constexpr (PSI) {
#pragma unroll
            for (int j = 0; j < NS; ++j)
#pragma unroll
                for (int s = 0; s < NS; ++s) P[j][s] = vsplat<V>(j == s ? 1.0f : 0.0f);
        }
    }
};

__device__ __forceinline__ void nl_load_diode(const NlStepArgs& a, SSDiode& dp)
{
    dp.Is = *a.pIs; dp.V = *a.pV; dp.Rport = *a.pRp;
    dp.L = logf(dp.Rport * dp.Is / dp.V);
    dp.d = make_diode_static(dp.V, a.n_up, a.n_down);
}

// the state alone (warm-up)
template <int NS, int NI, bool SYM, typename V, int FAST = 0>
__device__ __forceinline__ void nl_z_step(const SSCoef<NS, NI>& c, const SSDiode& dp, const V (&x)[NI], V (&z)[NS])
{
    using C = SSCoef<NS, NI>;
    V a = vsplat<V>(0.0f);
#pragma unroll
    for (int s = 0; s < NS; ++s) a = vfma(c.v[C::oCa + s], z[s], a);
#pragma unroll
    for (int i = 0; i < NI; ++i) a = vfma(c.v[C::oDa + i], x[i], a);
    const V b = diode_pair<SYM, V, FAST>(a, vsplat<V>(dp.L), dp.d).b;
    V zn[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        V v = b * c.v[C::oE + s];
#pragma unroll
        for (int q = 0; q < NS; ++q) v = vfma(c.v[C::oA + s * NS + q], z[q], v);
#pragma unroll
        for (int i = 0; i < NI; ++i) v = vfma(c.v[C::oB + s * NI + i], x[i], v);
        zn[s] = v;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) z[s] = zn[s];
}

// one step with tangents -> y.  gs = gscale (0 on a padding lane), lv = 1 (0 on a padding lane).
// FAST (with SYM): the kernel has checked that omega_1's argument never leaves the series-only region (diode_pair, wdf_omega.h).
template <int NS, int NI, bool SYM, typename V, bool PSI, int FAST = 0>
__device__ __forceinline__ V nl_step(const SSCoef<NS, NI>& c, const SSDiode& dp, const V (&x)[NI], V tgt, float gs, float lv,
                                     NlU<NS, NI, V, PSI>& u, V (&G)[NlDims<NS, NI>::nG], V (&H)[NS], V& sse)
{
    using C = SSCoef<NS, NI>;
    using D = NlDims<NS, NI>;
    const V zero = vsplat<V>(0.0f);
    V a = zero;
#pragma unroll
    for (int s = 0; s < NS; ++s) a = vfma(c.v[C::oCa + s], u.z[s], a);
#pragma unroll
    for (int i = 0; i < NI; ++i) a = vfma(c.v[C::oDa + i], x[i], a);
    const DiodeOutT<V> o = diode_pair<SYM, V, FAST>(a, vsplat<V>(dp.L), dp.d);
    const V b = o.b;
    // the root's partials (wdf_statespace.h, ss_bwd_tp_kernel's one_step): Da = db/da, DL = db/dL, DV = db/dV
    const V w0p = o.w0 * vrcp(o.w0 + 1.0f);
    // omega_1 <= omega(-4) = 0.018 in the series-only region: omega / (1 + omega) by its alternating series to the cubic term
    // (next term 1e-7 relative) instead of a second reciprocal (wdf_clipper_fused.h, fused_step)
    V w1p;
    if constexpr (SYM && FAST == kRootLean) w1p = vfma(-o.w1, o.w1, o.w1);      // LEAN tier (wdf_omega.h): omega_1 <= 5.6e-4, w (1 - w)
    else if constexpr (SYM && FAST) w1p = o.w1 * vfma(-o.w1, vfma(-o.w1, 1.0f - o.w1, 1.0f), 1.0f);
    else w1p = o.w1 * vrcp(o.w1 + 1.0f);
    const V sp = w0p + w1p;
    V Da, DL, DV;
    if constexpr (SYM && FAST) {
        // lam (w0p - w1p) = copysign(w0p - w1p, a): omega is increasing, and at a = 0 both are the same series of the same
        // argument (diode_pair); -2 m lam (w0 - w1) = (b - a) / V; lam^2 = (a != 0)
        const V l2 = vsel(vgt_c(vabs(a), 0.0f), 1.0f, 0.0f);
        Da = vfma(l2 * -2.0f, sp, 1.0f);
        DL = vcopysign(w0p - w1p, a) * (-dp.d.two_v * dp.d.m_dn);
        DV = vfma((l2 * 2.0f) * a, sp, b - a) * fast_rcp(dp.V);
    } else {
        const V l2 = o.lam * o.lam;
        Da = vfma(l2 * -2.0f, sp, 1.0f);
        DL = (o.lam * (o.m0 * w0p - o.m1 * w1p)) * (-dp.d.two_v);
        DV = vfma((l2 * 2.0f) * a, sp * fast_rcp(dp.V), (o.lam * (o.m0 * o.w0 - o.m1 * o.w1)) * -2.0f);
    }
    V yv = b * c.v[C::oFy];
#pragma unroll
    for (int s = 0; s < NS; ++s) yv = vfma(c.v[C::oCy + s], u.z[s], yv);
#pragma unroll
    for (int i = 0; i < NI; ++i) yv = vfma(c.v[C::oDy + i], x[i], yv);
    const V e = yv - tgt;
    const V g = e * gs;
    sse = vfma(e * lv, e, sse);
    // M = A + E (Da ca)^T: the step's Jacobian; gw = g (cy + fy Da ca): dLoss/dz through y
    V M[NS][NS], gw[NS];
    const V gf = g * c.v[C::oFy];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const V dq = Da * c.v[C::oCa + q];
        gw[q] = vfma(gf, dq, g * c.v[C::oCy + q]);
#pragma unroll
        for (int s = 0; s < NS; ++s) M[s][q] = vfma(dq, c.v[C::oE + s], vsplat<V>(c.v[C::oA + s * NS + q]));
    }
    // every tangent: G += gw.S + gf beta;  S' = M S + E beta + direct
    auto advance = [&](V (&S)[NS], V& Gc, V beta, bool has_beta, int di, V dv) {
        V acc = Gc;
#pragma unroll
        for (int s = 0; s < NS; ++s) acc = vfma(gw[s], S[s], acc);
        if (has_beta) acc = vfma(gf, beta, acc);
        Gc = acc;
        V sn[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            V v = (s == di) ? dv : zero;
            if (has_beta) v = vfma(beta, c.v[C::oE + s], v);
#pragma unroll
            for (int q = 0; q < NS; ++q) v = vfma(M[s][q], S[q], v);
            sn[s] = v;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) S[s] = sn[s];
    };
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) advance(u.S[C::oA + i * NS + j], G[C::oA + i * NS + j], zero, false, i, u.z[j]);
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) advance(u.S[C::oB + i * NI + j], G[C::oB + i * NI + j], zero, false, i, x[j]);
#pragma unroll
    for (int i = 0; i < NS; ++i) advance(u.S[C::oE + i], G[C::oE + i], zero, false, i, b);
#pragma unroll
    for (int s = 0; s < NS; ++s) advance(u.S[C::oCa + s], G[C::oCa + s], Da * u.z[s], true, -1, zero);
#pragma unroll
    for (int i = 0; i < NI; ++i) advance(u.S[C::oDa + i], G[C::oDa + i], Da * x[i], true, -1, zero);
    advance(u.S[C::oCy + 0], G[C::kN + 0], DL, true, -1, zero);
    advance(u.S[C::oCy + 1], G[C::kN + 1], DV, true, -1, zero);
    if constexpr (PSI) {
#pragma unroll
        for (int j = 0; j < NS; ++j) advance(u.P[j], H[j], zero, false, -1, zero);
    }
    // the coefficients y sees directly
#pragma unroll
    for (int s = 0; s < NS; ++s) G[C::oCy + s] = vfma(g, u.z[s], G[C::oCy + s]);
#pragma unroll
    for (int i = 0; i < NI; ++i) G[C::oDy + i] = vfma(g, x[i], G[C::oDy + i]);
    G[C::oFy] = vfma(g, b, G[C::oFy]);
    // the state
    V zn[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        V v = b * c.v[C::oE + s];
#pragma unroll
        for (int q = 0; q < NS; ++q) v = vfma(c.v[C::oA + s * NS + q], u.z[q], v);
#pragma unroll
        for (int i = 0; i < NI; ++i) v = vfma(c.v[C::oB + s * NI + i], x[i], v);
        zn[s] = v;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) u.z[s] = zn[s];
    (void)D::nT;
    return yv;
}

// ---- the chunks ------------------------------------------------------------------------------------------------------
// 256-thread workgroups (four chunks of four neighbouring groups): single-wave workgroups land unevenly on the SIMDs.
template <int NS, int NI, bool SYM, typename V, int FAST>
__device__ __forceinline__ void nl_chunk(const NlStepArgs& a, const SSCoef<NS, NI>& c, const SSDiode& dp)
{
    using C = SSCoef<NS, NI>;
    using D = NlDims<NS, NI>;
    constexpr int WD = LinWidth<V>::w;
    const V zero = vsplat<V>(0.0f);
    const int lane = threadIdx.x & 63;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= (int64_t)a.groups * a.K) return;
    const int64_t k = unit / a.groups, grp = unit % a.groups;
    const int64_t B = a.B, b_raw = (grp * 64 + lane) * WD;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - WD;
    const int64_t t0 = k * a.L, t1 = (t0 + a.L < a.T) ? t0 + a.L : a.T;
    const int w_cur = a.ctl->w_cur, w_snap = a.ctl->w_snap, have_snap = a.ctl->have_snap, par = a.ctl->parity;
    const int64_t tw = t0 > w_cur ? t0 - w_cur : 0;
    const int64_t tsnap = (k + 1 < a.K && t1 - w_snap >= t0) ? t1 - w_snap : -1;
    NlU<NS, NI, V, true> u;
    if (tw == 0 || !have_snap) {
#pragma unroll
        for (int s = 0; s < NS; ++s) u.z[s] = zero;
    } else {
        // the previous call's state at this very sample, moved along its tangents by the change of the coefficients since
        const float* sp = a.snap + (((size_t)(par ^ 1) * a.K + k) * D::nSnap) * B + b;
#pragma unroll
        for (int s = 0; s < NS; ++s) u.z[s] = lin_ld<V>(sp + (size_t)(D::sZ + s) * B);
#pragma unroll
        for (int cc = 0; cc < D::nT; ++cc) {
            const float now = cc < C::oCy ? c.v[cc] : (cc == C::oCy ? dp.L : dp.V);
            const float d = now - a.coef_prev[D::gidx(cc)];
#pragma unroll
            for (int s = 0; s < NS; ++s) u.z[s] = vfma(d, lin_ld<V>(sp + (size_t)(D::sS + cc * NS + s) * B), u.z[s]);
        }
    }
    u.start();
    const float gs = live ? a.gscale : 0.0f, lv = live ? 1.0f : 0.0f;
    V G[D::nG], H[NS], sse = zero;
    double acc[D::nG + 1];
#pragma unroll
    for (int i = 0; i < D::nG; ++i) G[i] = zero;
#pragma unroll
    for (int s = 0; s < NS; ++s) H[s] = zero;
#pragma unroll
    for (int i = 0; i <= D::nG; ++i) acc[i] = 0.0;
    float* __restrict__ rk = a.rec + ((size_t)k * D::nRec) * B + b;
    V xn[kLinBlk][NI], tn[kLinBlk];
    auto load_blk = [&](int64_t ts) {
        const int n = t1 - ts < kLinBlk ? (int)(t1 - ts) : kLinBlk;
        lin_load_x<NI, V>(a.x, B, b, ts, n, xn);
#pragma unroll
        for (int i = 0; i < kLinBlk; ++i) tn[i] = (i < n && ts >= t0) ? lin_ld<V>(a.target + (ts + i) * B + b) : zero;
    };
    load_blk(tw);
    int since = 0;
    for (int64_t ts = tw; ts < t1; ts += kLinBlk) {               // 8-step blocks, the next one in flight
        const int n = t1 - ts < kLinBlk ? (int)(t1 - ts) : kLinBlk;
        V xc[kLinBlk][NI], tc[kLinBlk];
#pragma unroll
        for (int i = 0; i < kLinBlk; ++i) {
            tc[i] = tn[i];
#pragma unroll
            for (int j = 0; j < NI; ++j) xc[i][j] = xn[i][j];
        }
        if (ts + kLinBlk < t1) load_blk(ts + kLinBlk);
        if (ts < t0) {                                            // warm-up (t0 - tw is a multiple of 8)
#pragma unroll
            for (int i = 0; i < kLinBlk; ++i) nl_z_step<NS, NI, SYM, V, FAST>(c, dp, xc[i], u.z);
            continue;
        }
        if (ts == t0 && live) {
#pragma unroll
            for (int s = 0; s < NS; ++s) lin_st<V>(rk + (size_t)(D::oZw + s) * B, u.z[s]);
        }
        if (ts == tsnap && live) {                                // for chunk k + 1 of the next call
            float* sp = a.snap + (((size_t)par * a.K + (k + 1)) * D::nSnap) * B + b;
#pragma unroll
            for (int s = 0; s < NS; ++s) lin_st<V>(sp + (size_t)(D::sZ + s) * B, u.z[s]);
#pragma unroll
            for (int cc = 0; cc < D::nT; ++cc)
#pragma unroll
                for (int s = 0; s < NS; ++s) lin_st<V>(sp + (size_t)(D::sS + cc * NS + s) * B, u.S[cc][s]);
#pragma unroll
            for (int j = 0; j < NS; ++j)
#pragma unroll
                for (int s = 0; s < NS; ++s) lin_st<V>(sp + (size_t)(D::sPsi + j * NS + s) * B, u.P[j][s]);
        }
        if (n == kLinBlk) {
            // a whole block, straight-line: the tangent arithmetic of one step fills the transcendental latencies of the
            // next step's diode pair (a padding lane repeats the last sequence: its stores write the same values again)
#pragma unroll
            for (int i = 0; i < kLinBlk; ++i) {
                const V yv = nl_step<NS, NI, SYM, V, true, FAST>(c, dp, xc[i], tc[i], gs, lv, u, G, H, sse);
                lin_st_nt<V>(a.y + (ts + i) * B + b, yv);
            }
        } else {
#pragma unroll
            for (int i = 0; i < kLinBlk; ++i) {
                if (i >= n) break;
                const V yv = nl_step<NS, NI, SYM, V, true, FAST>(c, dp, xc[i], tc[i], gs, lv, u, G, H, sse);
                if (live) lin_st_nt<V>(a.y + (ts + i) * B + b, yv);
            }
        }
        if (++since == 4) {                                       // fp32 sums within 32 steps, fp64 across
            since = 0;
#pragma unroll
            for (int i = 0; i < D::nG; ++i) { acc[i] += lin_hsum(G[i]); G[i] = zero; }
            acc[D::nG] += lin_hsum(sse);
            sse = zero;
        }
    }
#pragma unroll
    for (int i = 0; i < D::nG; ++i) acc[i] += lin_hsum(G[i]);
    acc[D::nG] += lin_hsum(sse);
    if (live) {
#pragma unroll
        for (int s = 0; s < NS; ++s) lin_st<V>(rk + (size_t)(D::oZe + s) * B, u.z[s]);
#pragma unroll
        for (int j = 0; j < NS; ++j)
#pragma unroll
            for (int s = 0; s < NS; ++s) lin_st<V>(rk + (size_t)(D::oPsi + j * NS + s) * B, u.P[j][s]);
#pragma unroll
        for (int cc = 0; cc < D::nT; ++cc)
#pragma unroll
            for (int s = 0; s < NS; ++s) lin_st<V>(rk + (size_t)(D::oS + cc * NS + s) * B, u.S[cc][s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) lin_st<V>(rk + (size_t)(D::oH + s) * B, H[s]);
    }
    double* gp = a.gpart + ((size_t)k * a.groups + grp) * (D::nG + 1);
#pragma unroll
    for (int i = 0; i <= D::nG; ++i) {
        const double s = wave_sum_dpp(acc[i]);
        if (lane == 0) gp[i] = s;
    }
}

#ifndef WDF_NL_WAVES
#define WDF_NL_WAVES 1
#endif
template <int NS, int NI, bool SYM, typename V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WDF_NL_WAVES, WDF_NL_WAVES))) void ss_nl_step_kernel(const NlStepArgs a)
{
    SSCoef<NS, NI> c;
    c.load(a.coef);
    SSDiode dp = {};
    nl_load_diode(a, dp);
    if constexpr (SYM) {
        // the series-only evaluation of omega_1 and no per-step ballot when its argument can never leave that region
        // (u1 <= L - log N: a fact about the circuit, the same for every lane)
        const float l0 = dp.L - dp.d.l_dn;
        if (l0 <= -7.5f && l0 >= -80.0f) {                     // the LEAN root tier (round 6, wdf_omega.h): every practical diode
            nl_chunk<NS, NI, SYM, V, kRootLean>(a, c, dp);
            return;
        }
        if (l0 <= kSeriesOnlyBelow) {
            nl_chunk<NS, NI, SYM, V, kRootFast>(a, c, dp);
            return;
        }
    }
    nl_chunk<NS, NI, SYM, V, kRootGeneral>(a, c, dp);
}

// ---- the finish: verify the boundaries, walk the tangents, re-run what missed, reduce, chain rule, steer -----------------
// One workgroup per group of 64 WD sequences (WD: sequences per lane of the chunk kernel -- its waves and these cover the same
// ones): wave q = (h, slot) takes sequence h of every lane's WD and the chunks k = slot (mod the tile), lane l the l-th lane's
// sequences.  A tile of chunk records goes through LDS (one round trip to memory per tile instead of one per chunk, and the
// next tile's loads are in flight while this one is walked); every thread walks the whole tile -- a few dozen FMAs -- and
// keeps the start of its own chunk.  Wave 0 finishes: the repair of a group that missed, the sums, the ticket.
template <int NS, int WD> struct NlTile { static constexpr int n = (NS == 1 ? 8 : 4) / WD; };   // (at most 512 threads: 256 VGPRs for the repair path; the records fit 64 KB of LDS)

template <int NS, int NI, bool SYM, int WD>
__global__ __launch_bounds__((64 * WD * NlTile<NS, WD>::n)) void ss_nl_step_finish_kernel(const NlStepArgs a)
{
    constexpr int kNlTile = NlTile<NS, WD>::n, kWaves = WD * kNlTile;
    using C = SSCoef<NS, NI>;
    using D = NlDims<NS, NI>;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, q = wv % kNlTile, hq = wv / kNlTile;
    const int64_t grp = blockIdx.x, B = a.B;
    const int K = a.K;
    const NlStepCtl c0 = *a.ctl;                                  // (only the launch's last wave writes it, after everyone has read)
    const float tol = c0.tol;
    const int w_snap = c0.w_snap, par = c0.parity;
    // what the launch's LAST wave needs for the chain rule, fetched now by everybody's wave 0 (a round trip to memory the tail
    // does not wait for): the probe's Jacobian column of lane p, the root's values, the coefficients
    struct { double jac[C::kN + 1]; float coef[C::kN]; double Is, V, Rp; } pre;
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i <= C::kN; ++i) pre.jac[i] = lane < a.n_tree ? a.jac[(size_t)i * a.n_tree + lane] : 0.0;
#pragma unroll
        for (int i = 0; i < C::kN; ++i) pre.coef[i] = a.coef[i];
        pre.Is = *a.pIs; pre.V = *a.pV; pre.Rp = *a.pRp;
    }
    double gpre[D::nG + 1];                                       // the chunks' own sums of this group (lane k: chunk k)
#pragma unroll
    for (int i = 0; i <= D::nG; ++i)
        gpre[i] = (wv == 0 && lane < a.K) ? a.gpart[((size_t)lane * a.groups + grp) * (D::nG + 1) + i] : 0.0;
    __shared__ float recs[WD][kNlTile][D::nRec][64];
    __shared__ double red[kWaves][D::nG + 1];
    __shared__ int rbad[kWaves];
    __shared__ float rmiss[kWaves];
    double tot[D::nG + 1];
#pragma unroll
    for (int i = 0; i <= D::nG; ++i) tot[i] = 0.0;
    float miss = 0.0f;
    int nbad = 0;
    {
        const int64_t b_raw = (grp * 64 + lane) * WD + hq;
        const bool live = b_raw < B;
        const int64_t b = live ? b_raw : B - 1;
        float carry[D::nT][NS], zcarry[NS];                       // S_start and the predecessor's end at the tile's first chunk
#pragma unroll
        for (int cc = 0; cc < D::nT; ++cc)
#pragma unroll
            for (int s = 0; s < NS; ++s) carry[cc][s] = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; ++s) zcarry[s] = 0.0f;
        // chunk k's record, and the snapshot it took for the next call
        auto fetch = [&](int k, float (&v)[D::nRec], float (&psi)[NS][NS], float (&s0)[D::nT][NS]) {
            if (k >= K) return;
            const float* rk = a.rec + ((size_t)k * D::nRec) * B + b;
#pragma unroll
            for (int i = 0; i < D::nRec; ++i) v[i] = rk[(size_t)i * B];
            if (k + 1 < K && live && w_snap <= a.L) {
                const float* sp = a.snap + (((size_t)par * K + (k + 1)) * D::nSnap) * B + b;
#pragma unroll
                for (int j = 0; j < NS; ++j)
#pragma unroll
                    for (int s = 0; s < NS; ++s) psi[j][s] = sp[(size_t)(D::sPsi + j * NS + s) * B];
#pragma unroll
                for (int cc = 0; cc < D::nT; ++cc)
#pragma unroll
                    for (int s = 0; s < NS; ++s) s0[cc][s] = sp[(size_t)(D::sS + cc * NS + s) * B];
            }
        };
        constexpr int kAhead = NS == 1 ? 2 : 1;                   // tiles whose records are fetched together (registers)
        for (int ks = 0; ks < K; ks += kNlTile * kAhead) {
        float vv[kAhead][D::nRec], psiv[kAhead][NS][NS], s0v[kAhead][D::nT][NS];
#pragma unroll
        for (int r = 0; r < kAhead; ++r) fetch(ks + r * kNlTile + q, vv[r], psiv[r], s0v[r]);
#pragma unroll
        for (int r = 0; r < kAhead; ++r) {
            const int k0 = ks + r * kNlTile;
            if (k0 >= K) break;
            const int k = k0 + q, nt = K - k0 < kNlTile ? K - k0 : kNlTile;
            if (k0 > 0) __syncthreads();                          // (the tile before this one has been read)
            float (&psic)[NS][NS] = psiv[r];
            float (&s0c)[D::nT][NS] = s0v[r];
            if (k < K) {
#pragma unroll
                for (int i = 0; i < D::nRec; ++i) recs[hq][q][i][lane] = vv[r][i];
            }
            __syncthreads();
            const bool fix = k + 1 < K && live && w_snap <= a.L;
            float Ss[D::nT][NS], zprev[NS];                       // ... of MY chunk
#pragma unroll
            for (int j = 0; j < kNlTile; ++j) {
                if (j < nt) {
                    if (j == q) {
#pragma unroll
                        for (int cc = 0; cc < D::nT; ++cc)
#pragma unroll
                            for (int s = 0; s < NS; ++s) Ss[cc][s] = carry[cc][s];
#pragma unroll
                        for (int s = 0; s < NS; ++s) zprev[s] = zcarry[s];
                    }
#pragma unroll
                    for (int cc = 0; cc < D::nT; ++cc) {
                        float sn[NS];
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            float sv = recs[hq][j][D::oS + cc * NS + s][lane];
#pragma unroll
                            for (int jj = 0; jj < NS; ++jj) sv = fmaf(recs[hq][j][D::oPsi + jj * NS + s][lane], carry[cc][jj], sv);
                            sn[s] = sv;
                        }
#pragma unroll
                        for (int s = 0; s < NS; ++s) carry[cc][s] = sn[s];
                    }
#pragma unroll
                    for (int s = 0; s < NS; ++s) zcarry[s] = recs[hq][j][D::oZe + s][lane];
                }
            }
            if (k < K) {
                if (k > 0) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const float m = fabsf(recs[hq][q][D::oZw + s][lane] - zprev[s]);
                        miss = fmaxf(miss, live ? m : 0.0f);
                        nbad += (live && !(m <= tol)) ? 1 : 0;
                    }
                }
                // what the chunk's start adds to its sums: H . S_start
#pragma unroll
                for (int cc = 0; cc < D::nT; ++cc) {
                    float d = 0.0f;
#pragma unroll
                    for (int s = 0; s < NS; ++s) d = fmaf(recs[hq][q][D::oH + s][lane], Ss[cc][s], d);
                    tot[D::gidx(cc)] += live ? (double)d : 0.0;
                }
                if (fix) {                                        // S0 + Psi S_start, in place
                    float* sp = a.snap + (((size_t)par * K + (k + 1)) * D::nSnap) * B + b;
#pragma unroll
                    for (int cc = 0; cc < D::nT; ++cc)
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            float sv = s0c[cc][s];
#pragma unroll
                            for (int j = 0; j < NS; ++j) sv = fmaf(psic[j][s], Ss[cc][j], sv);
                            sp[(size_t)(D::sS + cc * NS + s) * B] = sv;
                        }
                }
            }
        }
        }
    }
    // ---- the workgroup's verdict and sums -> wave 0
    {
        const int wb = wave_sum_dpp(nbad);
        const float wm = wave_max_dpp(miss);
#pragma unroll
        for (int i = 0; i <= D::nG; ++i) {
            const double s = wave_sum_dpp(tot[i]);
            if (lane == 0) red[wv][i] = s;
        }
        if (lane == 0) { rbad[wv] = wb; rmiss[wv] = wm; }
    }
    __syncthreads();
    if (wv != 0) return;
    int nbad_all = 0;
    float miss_all = 0.0f;
#pragma unroll
    for (int j = 0; j < kWaves; ++j) { nbad_all += rbad[j]; miss_all = fmaxf(miss_all, rmiss[j]); }
#pragma unroll
    for (int i = 0; i <= D::nG; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < kWaves; ++j) s += red[j][i];
        tot[i] = s;                                               // (the same value in every lane)
    }
    const int wbad = nbad_all;
    const float wmiss = miss_all;
    if (wbad == 0) {
        // the chunks' own sums of this group (lanes over chunks), then one value per accumulator in every lane
#pragma unroll
        for (int i = 0; i <= D::nG; ++i) {
            double s = gpre[i];
            for (int k = lane + 64; k < K; k += 64) s += a.gpart[((size_t)k * a.groups + grp) * (D::nG + 1) + i];
            tot[i] += wave_sum_dpp(s);
        }
    } else {
        // a boundary missed: this group again, sequentially -- exact (states, tangents, sums, the snapshots)
        C c;
        c.load(a.coef);
        SSDiode dp = {};
        nl_load_diode(a, dp);
#pragma unroll
        for (int i = 0; i <= D::nG; ++i) tot[i] = 0.0;
        for (int h = 0; h < WD; ++h) {                            // (one wave: a miss is rare, and the registers of the walk
                                                                  //  above would not survive a second copy of this loop's)
            const int64_t b_raw = (grp * 64 + lane) * WD + h;
            const bool live = b_raw < B;
            const int64_t b = live ? b_raw : B - 1;
            const float gs = live ? a.gscale : 0.0f, lv = live ? 1.0f : 0.0f;
            NlU<NS, NI, float, false> u;
#pragma unroll
            for (int s = 0; s < NS; ++s) u.z[s] = 0.0f;
            u.start();
            float G[D::nG], Hd[NS] = {}, sse = 0.0f;
#pragma unroll
            for (int i = 0; i < D::nG; ++i) G[i] = 0.0f;
            int since = 0;
            for (int64_t ts = 0; ts < a.T; ts += kLinBlk) {
                const int n = a.T - ts < kLinBlk ? (int)(a.T - ts) : kLinBlk;
                float xs[kLinBlk][NI], tg[kLinBlk];
                lin_load_x<NI, float>(a.x, B, b, ts, n, xs);
#pragma unroll
                for (int i = 0; i < kLinBlk; ++i) tg[i] = (i < n) ? a.target[(ts + i) * B + b] : 0.0f;
                const int64_t kk = ts / a.L;                       // the chunk this block lies in
                if (kk + 1 < K && ts == (kk + 1) * a.L - w_snap && live && w_snap <= a.L) {
                    float* sp = a.snap + (((size_t)par * K + (kk + 1)) * D::nSnap) * B + b;
#pragma unroll
                    for (int s = 0; s < NS; ++s) sp[(size_t)(D::sZ + s) * B] = u.z[s];
#pragma unroll
                    for (int cc = 0; cc < D::nT; ++cc)
#pragma unroll
                        for (int s = 0; s < NS; ++s) sp[(size_t)(D::sS + cc * NS + s) * B] = u.S[cc][s];
                }
#pragma unroll
                for (int i = 0; i < kLinBlk; ++i) {
                    if (i >= n) break;
                    const float yv = nl_step<NS, NI, SYM, float, false>(c, dp, xs[i], tg[i], gs, lv, u, G, Hd, sse);
                    if (live) a.y[(ts + i) * B + b] = yv;
                }
                if (++since == 4) {
                    since = 0;
#pragma unroll
                    for (int i = 0; i < D::nG; ++i) { tot[i] += (double)G[i]; G[i] = 0.0f; }
                    tot[D::nG] += (double)sse;
                    sse = 0.0f;
                }
            }
#pragma unroll
            for (int i = 0; i < D::nG; ++i) tot[i] += (double)G[i];
            tot[D::nG] += (double)sse;
        }
#pragma unroll
        for (int i = 0; i <= D::nG; ++i) tot[i] = wave_sum_dpp(tot[i]);
    }
    // ---- this group's partial {sums.., bad boundaries, largest miss}; the last wave adds them up in a fixed order
    constexpr int kPart = D::nG + 3;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i <= D::nG; ++i)
            __hip_atomic_store(a.part + (size_t)grp * kPart + i, tot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.part + (size_t)grp * kPart + D::nG + 1, (double)wbad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.part + (size_t)grp * kPart + D::nG + 2, (double)wmiss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned old = 0;
    if (lane == 0) old = atomicAdd(a.ticket, 1u);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != (unsigned)(a.groups - 1)) return;
    if (lane == 0) *a.ticket = 0u;
    double sum[D::nG + 1], nbd = 0.0, ggd = 0.0;
    float mm = 0.0f;
#pragma unroll
    for (int i = 0; i <= D::nG; ++i) sum[i] = 0.0;
    for (int64_t w = lane; w < a.groups; w += 64) {
        double v[kPart];
#pragma unroll
        for (int i = 0; i < kPart; ++i) v[i] = __hip_atomic_load(a.part + (size_t)w * kPart + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i <= D::nG; ++i) sum[i] += v[i];
        nbd += v[D::nG + 1];
        ggd += v[D::nG + 1] > 0.0 ? 1.0 : 0.0;
        mm = fmaxf(mm, (float)v[D::nG + 2]);
    }
#pragma unroll
    for (int i = 0; i <= D::nG; ++i) sum[i] = wave_sum_dpp(sum[i]);
    const int nb = (int)wave_sum_dpp(nbd), gg = (int)wave_sum_dpp(ggd);
    mm = wave_max_dpp(mm);
    // the root's own values (ss_grad_reduce_kernel's formulas): L = log(Rp Is / V)
    const double Is = pre.Is, Vv = pre.V, Rp = pre.Rp;
    const double sL = sum[C::kN], sV = sum[C::kN + 1];
    const double gRp = sL / Rp;
    if (lane < a.n_tree) {
        double gp = gRp * pre.jac[C::kN];
#pragma unroll
        for (int i = 0; i < C::kN; ++i) gp += sum[i] * pre.jac[i];
        a.out[1 + lane] = (float)gp;
    }
    if (lane == 0) {
        a.out[0] = (float)sum[D::nG];
        if (a.loss) *a.loss = (float)(0.5 * (double)a.gscale * sum[D::nG]);
        a.out[1 + a.n_tree] = (float)(sL / Is);
        a.out[2 + a.n_tree] = (float)(sV - sL / Vv);
        // what the snapshots of this call were taken with
#pragma unroll
        for (int i = 0; i < C::kN; ++i) a.coef_prev[i] = pre.coef[i];
        a.coef_prev[C::kN] = logf((float)Rp * (float)Is / (float)Vv);
        a.coef_prev[C::kN + 1] = (float)Vv;
        // the warm-up of the call after the next (the next one's is where this call's snapshots lie)
        NlStepCtl* ctl = a.ctl;
        ctl->n_bad = nb;
        ctl->max_miss = mm;
        ctl->gated_groups = gg;
        const int w_used = c0.w_cur, ws = c0.w_snap;
        int w = ws, cool = c0.cool;
        if (c0.have_snap) {                                      // (a cold call says nothing about the predicted starts)
            const int base = w_used > ws ? w_used : ws;
            if (nb) { w = base * 2; cool = c0.cool_miss; }
            else if (mm > c0.grow_at * tol) { w = base + 16; if (cool < 2) cool = 2; }
            else if (cool > 0) --cool;
            else if (mm <= c0.shrink_at * tol && w_used <= ws) w = ws - 8;
        }
        const int wmax = c0.w_max < (int)a.L ? c0.w_max : (int)a.L;
        if (w > wmax) w = wmax;
        if (w < c0.w_min) w = c0.w_min;
        ctl->w_used = w_used;
        ctl->w_cur = ws;
        ctl->w_snap = w;
        ctl->cool = cool;
        ctl->have_snap = (ws <= a.L) ? 1 : 0;
        ctl->parity = par ^ 1;
        ctl->call = c0.call + 1;
        ctl->total_gated = c0.total_gated + gg;
    }
}

}  // namespace wdf
