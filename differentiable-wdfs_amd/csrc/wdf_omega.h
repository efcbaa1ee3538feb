// wdf_omega.h -- Wright omega on the real axis and the diode-pair root, gfx950 device code.
//
// What it computes follows modules/toms917/toms917.cpp:134-375 restricted to real
// arguments (regions 3/4/7 start values :240-296, FSC iteration :347-352, conditional
// second iteration :356-364) and wdf_py/diode_clipper/diode_pretraining.py:39-60 (eqn 45;
// the N_up = N_down = 1 case is Toms917DiodePair.h:51-59).  How it computes it is fp32
// VALU code for CDNA4: raw v_exp_f32 / v_log_f32 / v_rcp_f32, branch-light region
// selection, and the second FSC iteration behind a wavefront ballot so a wave skips it
// unless one of its 64 lanes (= 64 training sequences) needs it.
#pragma once

#include <hip/hip_runtime.h>

namespace wdf {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kRegion4Hi = 4.141592653589793f;   // 1 + pi   (toms917.cpp:253-254)
constexpr float kFltEps = 1.1920928955078125e-07f;  // TWOITERTOL for float (toms917.cpp:16)

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * kLn2; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// One Fritsch-Shafer-Crowley step (toms917.cpp:347-352):  returns w*(1+e), r is the residual
// BEFORE the step (needed by the two-iteration test).
__device__ __forceinline__ float fsc_step(float x, float w, float& r, float& wp1)
{
    r = x - w - fast_log(w);
    wp1 = w + 1.0f;
    const float t = 2.0f * wp1 * (wp1 + (2.0f / 3.0f) * r);
    const float e = (r * (t - r)) * fast_rcp(wp1 * (t - 2.0f * r));
    return fmaf(w, e, w);
}

// Below this argument the 5-term region-3 series is already exact to < 0.5 ulp in fp32
// (first dropped term, relative: 54/5 exp(x)^5 < 2^-25 for x < -4) while an FSC step would
// ADD error there: its residual r = x - w - log(w) carries ulp(|x|)/2 from log(w) ~ x.
// So the step is applied only above it.
constexpr float kSeriesOnlyBelow = -4.0f;

// omega(x).  `iters` (optional) reports 0/1/2 FSC iterations for tests.
template <bool COUNT_ITERS = false>
__device__ __forceinline__ float wright_omega(float x, int* iters = nullptr)
{
    float w;
    // ---- start value by region -------------------------------------------------------
    if (x <= -2.0f) {                                  // region 3: series in exp(x)  (:240-248)
        const float p = fast_exp(x);
        w = p * fmaf(p, fmaf(p, fmaf(p, fmaf(p, 125.0f / 24.0f, -8.0f / 3.0f), 1.5f), -1.0f), 1.0f);
    } else if (x <= kRegion4Hi) {                      // region 4: series about 1    (:253-261)
        const float q = x - 1.0f;
        const float s = fmaf(q, fmaf(q, fmaf(q, 13.0f / 61440.0f, -1.0f / 3072.0f), -1.0f / 192.0f), 1.0f / 16.0f);
        w = fmaf(s, q * q, fmaf(0.5f, x, 0.5f));
    } else {                                           // region 7: series about +inf (:290-296)
        const float l = fast_log(x);
        const float ix = fast_rcp(x);
        // ((1 + (-3/2 + l/3) l) l + ((-1 + l/2) l + (l + (-l + x) x) x) x) / x^3, Horner in 1/x
        const float c3 = l * fmaf(l, fmaf(l, 1.0f / 3.0f, -1.5f), 1.0f);
        const float c2 = l * fmaf(l, 0.5f, -1.0f);
        w = (x - l) + ix * fmaf(ix, fmaf(ix, c3, c2), l);
    }
    int n = 0;
    // ---- FSC iteration one (:347-352), skipped where the series alone is exact --------
    const bool refine = x > kSeriesOnlyBelow;
    float r = 0.0f, wp1 = 1.0f;
    if (__builtin_amdgcn_ballot_w64(refine)) {
        const float w1 = fsc_step(x, refine ? w : 1.0f, r, wp1);
        if (refine) { w = w1; n = 1; }
    }
    // ---- conditional iteration two (:356-364), decided per wave by ballot --------------
    // |(2w^2 - 8w - 1) r^4| >= eps * 72 * |w+1|^6
    {
        const float r2 = r * r;
        const float p2 = wp1 * wp1;
        const bool again = refine &&
            fabsf(fmaf(w, fmaf(2.0f, w, -8.0f), -1.0f)) * (r2 * r2) >= (kFltEps * 72.0f) * (p2 * p2 * p2);
        if (__builtin_amdgcn_ballot_w64(again)) {
            float r_, wp1_;
            const float w2 = fsc_step(x, again ? w : 1.0f, r_, wp1_);
            if (again) { w = w2; n = 2; }
        }
    }
    if constexpr (COUNT_ITERS) { if (iters) *iters = n; }
    return w;
}

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

struct DiodeOut {
    float b;        // reflected wave
    float w0, w1;   // the two omega values
    float lam;      // sign(a)
    float m0, m1;   // mu0, mu1 used
};

// Reflected wave of the diode pair (diode_pretraining.py:46-59).  L = log(Rp Is / nVt).
// SYM: N_up == N_down (mu0 = mu1, no per-sign select).
template <bool SYM>
__device__ __forceinline__ DiodeOut diode_pair(float a, float L, const DiodeStatic& c)
{
    DiodeOut o;
    o.lam = (a > 0.0f) ? 1.0f : ((a < 0.0f) ? -1.0f : 0.0f);   // np.sign (:52)
    const float aa = fabsf(a);                                 // lam * a
    float i0, i1, l0, l1;
    if constexpr (SYM) {
        o.m0 = o.m1 = c.m_dn;
        i0 = i1 = c.i_dn;
        l0 = l1 = L - c.l_dn;
    } else {
        const bool pos = a >= 0.0f;                            // mu0 = N_down if a >= 0 (:46-47)
        o.m0 = pos ? c.m_dn : c.m_up;
        o.m1 = pos ? c.m_up : c.m_dn;
        i0 = pos ? c.i_dn : c.i_up;
        i1 = pos ? c.i_up : c.i_dn;
        l0 = L - (pos ? c.l_dn : c.l_up);
        l1 = L - (pos ? c.l_up : c.l_dn);
    }
    o.w0 = wright_omega(fmaf(aa, i0, l0));                     // (:57)
    o.w1 = wright_omega(fmaf(-aa, i1, l1));                    // (:58)
    o.b = a - c.two_v * o.lam * (o.m0 * o.w0 - o.m1 * o.w1);   // (:56-59)
    return o;
}

}  // namespace wdf
