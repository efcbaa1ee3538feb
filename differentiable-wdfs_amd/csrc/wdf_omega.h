// wdf_omega.h -- Wright omega on the real axis and the diode-pair root, gfx950 device code.
//
// What it computes follows modules/toms917/toms917.cpp:134-375 restricted to real
// arguments (regions 3/4/7 start values :240-296, FSC iteration :347-352, conditional
// second iteration :356-364) and wdf_py/diode_clipper/diode_pretraining.py:39-60 (eqn 45;
// the N_up = N_down = 1 case is Toms917DiodePair.h:51-59).  How it computes it is fp32
// VALU code for CDNA4: raw v_exp_f32 / v_log_f32 / v_rcp_f32, branch-free region
// selection, and everything rare behind ONE wavefront ballot so a wave skips it unless one
// of its lanes (= training sequences) needs it.  All functions are written over a value type
// V (wdf_vec.h): float = one sequence per lane, v2f = two sequences per lane with packed
// v_pk_* arithmetic.
#pragma once

#include <hip/hip_runtime.h>

#include "wdf_vec.h"

namespace wdf {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kRegion4Hi = 4.141592653589793f;   // 1 + pi   (toms917.cpp:253-254)

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * kLn2; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// One Fritsch-Shafer-Crowley step (toms917.cpp:347-352):  returns w*(1+e); r is the residual
// BEFORE the step.
template <typename V>
__device__ __forceinline__ V fsc_step(V x, V w, V& r)
{
    r = vfma(vlog2(w), -kLn2, x - w);                          // x - w - log(w)
    const V wp1 = w + 1.0f;
    const V tmr = vfma(wp1 + wp1, vfma(r, 2.0f / 3.0f, wp1), -r);   // t - r,  t = 2 wp1 (wp1 + 2/3 r)
    const V e = (r * tmr) * vrcp(wp1 * (tmr - r));             // r (t - r) / (wp1 (t - 2r))
    return vfma(w, e, w);
}

// Below this argument the 5-term region-3 series is already exact to < 0.5 ulp in fp32
// (first dropped term, relative: 54/5 exp(x)^5 < 2^-25 for x < -4) while an FSC step would
// ADD error there: its residual r = x - w - log(w) carries ulp(|x|)/2 from log(w) ~ x.
// So the step is applied only above it.
constexpr float kSeriesOnlyBelow = -4.0f;

// The reference runs a second FSC iteration when |(2w^2-8w-1) r^4| >= eps 72 |w+1|^6
// (toms917.cpp:356-364).  (2w^2-8w-1)/(w+1)^6 is at most 1 in magnitude for w >= 0, so
// |r| < (72 eps)^(1/4) = 0.0541 (eps = FLT_EPSILON) implies the test is false.  With the start
// values below, max |r| over the whole real axis is 0.0487 (at x -> -2+), so in fp32 the
// second iteration is provably never needed; it is kept behind a wavefront ballot on the
// conservative test |r| >= 0.05 -- one compare per wave-instruction, never taken.
constexpr float kSecondIterResidual = 0.05f;

// The region-3 series: p (1 - p + 3/2 p^2 - 8/3 p^3 + 125/24 p^4), p = exp(x)  (:240-248)
template <typename V>
__device__ __forceinline__ V omega_series_of_exp(V p)
{
    return p * vfma(p, vfma(p, vfma(p, vfma(p, 125.0f / 24.0f, -8.0f / 3.0f), 1.5f), -1.0f), 1.0f);
}

template <typename V>
__device__ __forceinline__ V omega_series(V x)
{
    return omega_series_of_exp(vexp2(vmin_c(x, -2.0f) * kLog2e));   // clamp: never overflows
}

// The same series WITHOUT the clamp, for omega_start: above its region the exponential overflows to inf and the series to
// inf / NaN -- in lanes whose value the region select then discards (a select passes nothing of the operand it does not
// pick; there are no floating-point traps).  Two instructions per evaluation less.
template <typename V>
__device__ __forceinline__ V omega_series_unclamped(V x)
{
    return omega_series_of_exp(vexp2(x * kLog2e));
}

// Start value, branch-free: all three regional series are evaluated (outside its own region a series may
// overflow or go NaN: the select discards it) and selected per lane.  A wave holds 64+
// different sequences, so all regions are normally live in a wave anyway, and straight-line
// code lets the two omega evaluations of a diode pair interleave in the VALU.
template <typename V>
__device__ __forceinline__ V omega_start(V x)
{
    V wA = omega_series_unclamped(x);                          // region 3 (x <= -2); garbage above it, discarded below
    // region 4 (-2 < x <= 1+pi): series about x = 1                   (toms917.cpp:253-261)
    const V q = x - 1.0f;
    const V sB = vfma(q, vfma(q, vfma(q, 13.0f / 61440.0f, -1.0f / 3072.0f), -1.0f / 192.0f), 1.0f / 16.0f);
    V wB = vfma(sB, q * q, vfma(x, 0.5f, 0.5f));
    // region 7 (x > 1+pi): the leading terms of the series about +infinity, x - l + l/x, l = log x.
    // toms917.cpp:290-296 carries two more orders (l (l/2 - 1)/x^2 + l (l^2/3 - 3l/2 + 1)/x^3); they
    // are at most 0.033 in absolute value (at x = 1+pi, where omega = 2.96), i.e. a start value
    // within 1.1e-2 relative, and the FSC step that always follows is fourth order: what is left
    // is below 2e-8 relative, under fp32 rounding -- while the two orders cost 7 of the step's 68
    // VALU instructions.  |r| stays below kSecondIterResidual (0.044 at the boundary).
    // (no clamp of x into the region: log of a non-positive x is NaN / -inf, in lanes the select discards)
    const V lg = vlog2(x);
    const V l = lg * kLn2;
    V wC = vfma(l, vrcp(x), vfma(lg, -kLn2, x));
    // Pin the three values as computed: without this LLVM turns the selects back into
    // exec-masked branches around the transcendentals, which serialises the two omega
    // evaluations of a step and costs more in exec-mask bookkeeping than it saves.
    vpin(wA); vpin(wB); vpin(wC);
    return vsel(vle_c(x, -2.0f), wA, vsel(vle_c(x, kRegion4Hi), wB, wC));
}

// omega(x): start value + one FSC step (:347-352).  For x <= -4 the step's result is
// discarded (the series is exact there, and w0 may have underflowed to 0, making it NaN).
// `again` reports whether this lane asks for the reference's second iteration (:356-364).
template <typename V>
__device__ __forceinline__ V omega_one_step(V x, typename VT<V>::mask& again)
{
    const V w0 = omega_start(x);
    V r;
    V w1 = fsc_step(x, w0, r);
    vpin(w1);                                                  // keep the step unconditional
    const auto refine = vgt_c(x, kSeriesOnlyBelow);
    again = mand(refine, vge_c(vabs(r), kSecondIterResidual));
    return vsel(refine, w1, w0);
}

// The second FSC iteration, for the lanes that asked for it.
template <typename V>
__device__ __forceinline__ V omega_second_step(V x, V w, typename VT<V>::mask again)
{
    V r2;
    const V w2 = fsc_step(x, vsel(again, w, vsplat<V>(1.0f)), r2);
    return vsel(again, w2, w);
}

// omega(x), general argument.  `iters` (optional) reports 0/1/2 FSC iterations for tests.
template <bool COUNT_ITERS = false>
__device__ __forceinline__ float wright_omega(float x, int* iters = nullptr)
{
    bool again;
    float w = omega_one_step<float>(x, again);
    if (__builtin_amdgcn_ballot_w64(again)) w = omega_second_step<float>(x, w, again);   // per wave
    if constexpr (COUNT_ITERS) { if (iters) *iters = (x > kSeriesOnlyBelow ? 1 : 0) + (again ? 1 : 0); }
    return w;
}

// omega(x) for the LEAN root tier (diode_pair, TIER = 2): the region-3 start is the series' first two terms, p (1 - p)
// (relative error 3/2 p^2 <= 0.0275 at x = -2, i.e. |r| < 0.03: inside the one FSC step's basin like the other regions), and the
// step is applied on the WHOLE axis -- no "series only below -4" select.  Below -4 the step's residual carries the rounding of
// log(w) ~ x (half an ulp of |x|: <= 7e-7 relative on omega), which is why the general path skips it there; here the caller has
// checked that omega_0 matters only through 2 nVt omega_0 with omega_0 <= 0.018 in that region: 2e-9 V.  4 packed instructions,
// 2 compares and 2 selects per pair of sequences less than omega_one_step.  x must stay above the exponential's underflow
// (x > -87): the caller's tier test bounds it.
template <typename V>
__device__ __forceinline__ V omega_lean(V x)
{
    const V p = vexp2(x * kLog2e);
    V wA = vfma(-p, p, p);                                     // region 3 (x <= -2); inf / NaN above it, discarded below
    const V q = x - 1.0f;
    const V sB = vfma(q, vfma(q, vfma(q, 13.0f / 61440.0f, -1.0f / 3072.0f), -1.0f / 192.0f), 1.0f / 16.0f);
    V wB = vfma(sB, q * q, vfma(x, 0.5f, 0.5f));               // region 4, as omega_start
    const V lg = vlog2(x);
    const V l = lg * kLn2;
    V wC = vfma(l, vrcp(x), vfma(lg, -kLn2, x));               // region 7, as omega_start
    vpin(wA); vpin(wB); vpin(wC);
    const V w0 = vsel(vle_c(x, -2.0f), wA, vsel(vle_c(x, kRegion4Hi), wB, wC));
    V r;
    return fsc_step(x, w0, r);
}

// Root tiers of the clipper kernels (one wave-uniform test per kernel, wdf_clipper.h root_tier):
//   0  general: per-step ballot for omega_1's general evaluation and the second FSC iteration;
//   1  FAST:    omega_1 provably in its series-only region (L - log N <= -4): no ballot, log2(e) folded into its exponent;
//   2  LEAN:    (symmetric pair) L - log N <= -7.5, so omega_1 <= 5.6e-4: omega_1 = p (1 - p) and omega_1 / (1 + omega_1) =
//               omega_1 (1 - omega_1) to 2e-10 absolute, and omega_0 by omega_lean.  Every practical diode: 1N4148 behind
//               any pot value at 48 kHz sits at L < -8.5.
constexpr int kRootGeneral = 0, kRootFast = 1, kRootLean = 2;

// ---- diode pair ---------------------------------------------------------------------
// Per-sign constants that do not depend on the port resistance.
struct DiodeStatic {
    float i_up, i_dn;     // 1/(N_up nVt), 1/(N_down nVt)          (diode_pretraining.py:53-54)
    float l_up, l_dn;     // log N_up, log N_down
    float m_up, m_dn;     // N_up, N_down as float                 (:46-47)
    float two_v;          // 2 nVt                                 (:56)
};

__device__ __forceinline__ DiodeStatic make_diode_static(float nVt, int n_up, int n_down)
{
    DiodeStatic c;
    c.m_up = (float)n_up;
    c.m_dn = (float)n_down;
    c.i_up = 1.0f / (c.m_up * nVt);
    c.i_dn = 1.0f / (c.m_dn * nVt);
    c.l_up = logf(c.m_up);
    c.l_dn = logf(c.m_dn);
    c.two_v = 2.0f * nVt;
    return c;
}

template <typename V>
struct DiodeOutT {
    V b;        // reflected wave
    V w0, w1;   // the two omega values
    V lam;      // sign(a)
    V m0, m1;   // mu0, mu1 used
    V dw;       // w0 - w1 as b used it (LEAN: exactly 0 at a = 0)
};
using DiodeOut = DiodeOutT<float>;

// Reflected wave of the diode pair (diode_pretraining.py:46-59).  L = log(Rp Is / nVt) (a V:
// it varies per lane when a per-sample resistance is streamed).
// SYM: N_up == N_down (mu0 = mu1, no per-sign select).
// FAST: the caller has checked, once per kernel, that L - min(log N_up, log N_down) <=
// kSeriesOnlyBelow (static port resistance: a wave-uniform fact), i.e. that omega_1's argument
// can never leave the series-only region; the per-step wavefront ballot and its bookkeeping go
// (the second-iteration request it also guards is provably never raised in fp32, see
// kSecondIterResidual).  With SYM, lam (w0 - w1) becomes copysign(w0 - w1, a): w0 >= w1 since
// omega is increasing, and at a = 0 both are the same series of the same argument, so the
// difference is exactly 0 as lam = sign(0) = 0 requires.
template <bool SYM, typename V, int TIER = 0>
__device__ __forceinline__ DiodeOutT<V> diode_pair(V a, V L, const DiodeStatic& c)
{
    constexpr bool FAST = TIER >= kRootFast;
    constexpr bool LEAN = TIER == kRootLean && SYM;
    DiodeOutT<V> o;
    o.lam = vsign(a);                                          // np.sign (:52)
    const V aa = vabs(a);                                      // lam * a
    V u0, u1, e1 = vsplat<V>(0.0f);
    if constexpr (SYM) {
        o.m0 = o.m1 = vsplat<V>(c.m_dn);
        const V l0 = L - c.l_dn;
        u0 = vfma(aa, c.i_dn, l0);                             // (:57)
        u1 = vfma(aa, -c.i_dn, l0);                            // (:58)
        e1 = vfma(aa, -c.i_dn * kLog2e, l0 * kLog2e);          // u1 log2(e), FAST only
    } else {
        const auto pos = vge_c(a, 0.0f);                       // mu0 = N_down if a >= 0 (:46-47)
        o.m0 = vsel(pos, c.m_dn, c.m_up);
        o.m1 = vsel(pos, c.m_up, c.m_dn);
        const V i0 = vsel(pos, c.i_dn, c.i_up), i1 = vsel(pos, c.i_up, c.i_dn);
        const V l0 = L - vsel(pos, c.l_dn, c.l_up), l1 = L - vsel(pos, c.l_up, c.l_dn);
        u0 = vfma(aa, i0, l0);
        u1 = l1 - aa * i1;
    }
    // The hot path is one basic block so the two evaluations interleave in the VALU:
    //   w0 = omega(u0): general start value + one FSC step;
    //   w1 = omega(u1): u1 <= log(Rp Is/(mu1 nVt)), which for any practical diode (Rp Is << nVt)
    //        is below kSeriesOnlyBelow, where the region-3 series alone is exact.
    // Everything else -- a lane whose u1 needs the general evaluation, or a lane asking for the
    // second FSC iteration -- is handled after ONE wavefront ballot, so a wave skips it unless
    // one of its sequences needs it.
    typename VT<V>::mask again0{};
    if constexpr (LEAN) o.w0 = omega_lean<V>(u0);
    else o.w0 = omega_one_step<V>(u0, again0);
    if constexpr (LEAN) {
        const V p1 = vexp2(e1);
        o.w1 = vfma(-p1, p1, p1);                              // omega_1 <= 5.6e-4: p (1 - p), next term 3/2 p^3 < 3e-10
        vpin(o.w1);
    } else if constexpr (FAST && SYM) {
        // u1 <= kSeriesOnlyBelow on every lane: the same series as omega_series, with log2(e) folded
        // into the argument's FMA and without the overflow clamp.  At a = 0 the exponent is
        // l0 * log2(e), the very product omega_series(u0 = l0) forms, so w1 == w0 bit for bit there
        // (needed for lam = 0, see above).
        o.w1 = omega_series_of_exp<V>(vexp2(e1));
        vpin(o.w1);      // as a rounded value: contracted into w0 - w1 as an FMA it would not cancel at a = 0
    } else {
        o.w1 = omega_series<V>(u1);
    }
    if constexpr (!FAST) {
        const auto general1 = vgt_c(u1, kSeriesOnlyBelow);
        if (__builtin_amdgcn_ballot_w64(many(mor(again0, general1)))) {
            o.w0 = omega_second_step<V>(u0, o.w0, again0);
            typename VT<V>::mask again1;
            V w1g = omega_one_step<V>(u1, again1);
            w1g = omega_second_step<V>(u1, w1g, again1);
            o.w1 = vsel(general1, w1g, o.w1);
        }
    }
    o.dw = o.w0 - o.w1;
    // LEAN: omega_0 went through the FSC step and omega_1 did not, so at a = 0 the two differ in their last bits; lam =
    // sign(0) = 0 (zero input gives exactly zero output, as in the reference) is restored by the select
    if constexpr (LEAN) o.dw = vsel_nz(a, o.dw);
    if constexpr (SYM && FAST) o.b = a - (c.two_v * c.m_dn) * vcopysign(o.dw, a);
    else if constexpr (SYM) o.b = a - (c.two_v * c.m_dn) * (o.lam * (o.w0 - o.w1));   // (:56-59)
    else o.b = a - c.two_v * (o.lam * (o.m0 * o.w0 - o.m1 * o.w1));
    return o;
}

}  // namespace wdf
