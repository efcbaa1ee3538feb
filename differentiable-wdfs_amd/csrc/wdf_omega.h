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

// One Fritsch-Shafer-Crowley step (toms917.cpp:347-352):  returns w*(1+e); r is the residual
// BEFORE the step.
__device__ __forceinline__ float fsc_step(float x, float w, float& r)
{
    r = fmaf(-kLn2, __builtin_amdgcn_logf(w), x - w);          // x - w - log(w)
    const float wp1 = w + 1.0f;
    const float t = (2.0f * wp1) * fmaf(2.0f / 3.0f, r, wp1);  // 2 wp1 (wp1 + 2/3 r)
    const float e = (r * (t - r)) * fast_rcp(wp1 * fmaf(-2.0f, r, t));
    return fmaf(w, e, w);
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

// Start value, branch-free: all three regional series are evaluated (their arguments clamped
// into their own regions so nothing overflows) and selected per lane.  A wave holds 64
// different sequences, so all regions are normally live in a wave anyway, and straight-line
// code lets the two omega evaluations of a diode pair interleave in the VALU.
__device__ __forceinline__ float omega_start(float x)
{
    // region 3 (x <= -2): series in p = exp(x)                       (toms917.cpp:240-248)
    const float p = __builtin_amdgcn_exp2f(fminf(x, -2.0f) * kLog2e);
    const float wA = p * fmaf(p, fmaf(p, fmaf(p, fmaf(p, 125.0f / 24.0f, -8.0f / 3.0f), 1.5f), -1.0f), 1.0f);
    // region 4 (-2 < x <= 1+pi): series about x = 1                   (toms917.cpp:253-261)
    const float q = x - 1.0f;
    const float sB = fmaf(q, fmaf(q, fmaf(q, 13.0f / 61440.0f, -1.0f / 3072.0f), -1.0f / 192.0f), 1.0f / 16.0f);
    const float wB = fmaf(sB, q * q, fmaf(0.5f, x, 0.5f));
    // region 7 (x > 1+pi): series about +infinity                     (toms917.cpp:290-296)
    //   ((1 + (-3/2 + l/3) l) l + ((-1 + l/2) l + (l + (x - l) x) x) x) / x^3, Horner in 1/x
    const float xc = fmaxf(x, kRegion4Hi);
    const float l = __builtin_amdgcn_logf(xc) * kLn2;
    const float ix = fast_rcp(xc);
    const float c3 = l * fmaf(l, fmaf(l, 1.0f / 3.0f, -1.5f), 1.0f);
    const float c2 = l * fmaf(l, 0.5f, -1.0f);
    float wC = (xc - l) + ix * fmaf(ix, fmaf(ix, c3, c2), l);
    // Pin the three values as computed: without this LLVM turns the selects back into
    // exec-masked branches around the transcendentals, which serialises the two omega
    // evaluations of a step and costs more in exec-mask bookkeeping than it saves.
    float wa = wA, wb = wB;
    asm volatile("" : "+v"(wa), "+v"(wb), "+v"(wC));
    return (x <= -2.0f) ? wa : ((x <= kRegion4Hi) ? wb : wC);
}

// omega(x): start value + one FSC step (:347-352).  For x <= -4 the step's result is
// discarded (the series is exact there, and w0 may have underflowed to 0, making it NaN).
// `again` reports whether this lane asks for the reference's second iteration (:356-364).
__device__ __forceinline__ float omega_one_step(float x, bool& again)
{
    const float w0 = omega_start(x);
    float r;
    float w1 = fsc_step(x, w0, r);
    asm volatile("" : "+v"(w1));                       // keep the step unconditional (see omega_start)
    const bool refine = x > kSeriesOnlyBelow;
    again = refine && (fabsf(r) >= kSecondIterResidual);
    return refine ? w1 : w0;
}

// The second FSC iteration, for the lanes that asked for it.
__device__ __forceinline__ float omega_second_step(float x, float w, bool again)
{
    float r2;
    const float w2 = fsc_step(x, again ? w : 1.0f, r2);
    return again ? w2 : w;
}

// omega(x), general argument.  `iters` (optional) reports 0/1/2 FSC iterations for tests.
template <bool COUNT_ITERS = false>
__device__ __forceinline__ float wright_omega(float x, int* iters = nullptr)
{
    bool again;
    float w = omega_one_step(x, again);
    if (__builtin_amdgcn_ballot_w64(again)) w = omega_second_step(x, w, again);   // per wave
    if constexpr (COUNT_ITERS) { if (iters) *iters = (x > kSeriesOnlyBelow ? 1 : 0) + (again ? 1 : 0); }
    return w;
}

// The region-3 series alone: exact in fp32 for x <= kSeriesOnlyBelow.
__device__ __forceinline__ float wright_omega_series(float x)
{
    const float p = __builtin_amdgcn_exp2f(fminf(x, -2.0f) * kLog2e);
    return p * fmaf(p, fmaf(p, fmaf(p, fmaf(p, 125.0f / 24.0f, -8.0f / 3.0f), 1.5f), -1.0f), 1.0f);
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
    // (:57-58) The hot path is one basic block so the two evaluations interleave in the VALU:
    //   w0 = omega(u0): general start value + one FSC step;
    //   w1 = omega(u1): u1 <= log(Rp Is/(mu1 nVt)), which for any practical diode (Rp Is << nVt)
    //        is below kSeriesOnlyBelow, where the region-3 series alone is exact.
    // Everything else -- a lane whose u1 needs the general evaluation, or a lane asking for the
    // second FSC iteration -- is handled after ONE wavefront ballot, so a wave skips it unless
    // one of its 64 sequences needs it.
    const float u0 = fmaf(aa, i0, l0);
    const float u1 = fmaf(-aa, i1, l1);
    bool again0;
    o.w0 = omega_one_step(u0, again0);
    o.w1 = wright_omega_series(u1);
    const bool general1 = u1 > kSeriesOnlyBelow;
    if (__builtin_amdgcn_ballot_w64(again0 || general1)) {
        o.w0 = omega_second_step(u0, o.w0, again0);
        const float w1g = wright_omega(u1);
        o.w1 = general1 ? w1g : o.w1;
    }
    o.b = a - c.two_v * o.lam * (o.m0 * o.w0 - o.m1 * o.w1);   // (:56-59)
    return o;
}

}  // namespace wdf
