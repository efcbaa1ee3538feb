// wdf_capi_ss.hip -- C ABI part 3 of 4: the generic state-space tree kernels and the
// two-different-diode (asymmetric) root.  Argument checking, template dispatch and launches.
// 
#include "wdf_capi_common.h"
#include "wdf_statespace.h"
#include "wdf_ss_step.h"
#include "wdf_asym.h"
using namespace wdfcapi;

namespace {

// ---- state-space dispatch ------------------------------------------------------------------
template <int NS, int NI, int ROOT, bool V4>
void ss_launch_fwd(const float* x, const float* coef, const float* rootp, int n_up, int n_down, float* y,
                   float* zstash, const float* z0, float* zT, int64_t B, int64_t T, hipStream_t s)
{
    const unsigned grid = (unsigned)((B + 63) / 64);
    hipLaunchKernelGGL((wdf::ss_fwd_kernel<NS, NI, ROOT, false, V4>), dim3(grid), dim3(64), 0, s, x, coef, rootp,
                       n_up, n_down, y, zstash, z0, zT, B, T);
}

template <int NS, int NI, int ROOT, bool V4>
void ss_launch_bwd(const float* x, const float* coef, const float* rootp, int n_up, int n_down, const float* zstash,
                   const float* gy, double* ws, float* gz0, int64_t B, int64_t T, hipStream_t s)
{
    const unsigned grid = (unsigned)((B + 63) / 64);
    hipLaunchKernelGGL((wdf::ss_bwd_kernel<NS, NI, ROOT, false, V4>), dim3(grid), dim3(64), 0, s, x, coef, rootp,
                       n_up, n_down, zstash, gy, ws, gz0, B, T);
}

#define WDF_SS_CASE(FN, NS_, NI_, ...)                                                           \
    if (ns == NS_ && ni == NI_) {                                                                \
        if (root == wdf::kRootNone) {                                                            \
            if (v4) FN<NS_, NI_, wdf::kRootNone, true>(__VA_ARGS__);                             \
            else FN<NS_, NI_, wdf::kRootNone, false>(__VA_ARGS__);                               \
        } else {                                                                                 \
            if (v4) FN<NS_, NI_, wdf::kRootDiode, true>(__VA_ARGS__);                            \
            else FN<NS_, NI_, wdf::kRootDiode, false>(__VA_ARGS__);                              \
        }                                                                                        \
    }
#define WDF_SS_DISPATCH(FN, ...)                                                                 \
    do {                                                                                         \
        WDF_SS_CASE(FN, 0, 1, __VA_ARGS__) WDF_SS_CASE(FN, 1, 1, __VA_ARGS__)                    \
        WDF_SS_CASE(FN, 2, 1, __VA_ARGS__) WDF_SS_CASE(FN, 3, 1, __VA_ARGS__)                    \
        WDF_SS_CASE(FN, 4, 1, __VA_ARGS__)                                                       \
        WDF_SS_CASE(FN, 0, 2, __VA_ARGS__) WDF_SS_CASE(FN, 1, 2, __VA_ARGS__)                    \
        WDF_SS_CASE(FN, 2, 2, __VA_ARGS__) WDF_SS_CASE(FN, 3, 2, __VA_ARGS__)                    \
        WDF_SS_CASE(FN, 4, 2, __VA_ARGS__)                                                       \
    } while (0)

int ss_check(const float* x, const float* coef, const float* rootp, int ns, int ni, int root, int n_up, int n_down,
             int64_t B, int64_t T, int flags)
{
    if (!x || !coef) return fail(WDF_EINVAL, "null x/coef");
    if (ns < 0 || ns > 4 || ni < 1 || ni > 2) return fail(WDF_EUNSUPPORTED, "state-space kernels cover ns in [0,4], ni in [1,2] (got ns=%d ni=%d)", ns, ni);
    if (root != wdf::kRootNone && root != wdf::kRootDiode) return fail(WDF_EINVAL, "unknown root kind %d", root);
    if (root == wdf::kRootDiode && !rootp) return fail(WDF_EINVAL, "diode root needs rootp = {Is, nVt, R_port}");
    if (root == wdf::kRootDiode && (n_up < 1 || n_down < 1 || n_up > 16 || n_down > 16)) return fail(WDF_EINVAL, "n_up/n_down must be in [1,16]");
    if (B <= 0 || T <= 0) return fail(WDF_EINVAL, "B and T must be positive");
    if (flags != 0) return fail(WDF_EINVAL, "state-space kernels take flags = 0");
    return WDF_OK;
}

}  // namespace

extern "C" {

int wdf_ss_ncoef(int ns, int ni) { return ns * ns + ns * ni + ns + ns + ni + ns + ni + 1; }

size_t wdf_ss_bwd_ws_bytes(int ns, int ni, int64_t B)
{
    return B > 0 ? (size_t)((B + 63) / 64) * (size_t)(wdf_ss_ncoef(ns, ni) + 2) * sizeof(double) : 0;
}

int wdf_ss_fwd(const float* x, const float* coef, const float* rootp, int ns, int ni, int root, int n_up, int n_down,
               float* y, float* zstash, const float* z0, float* zT, int64_t B, int64_t T, int flags, void* stream)
{
    int rc = ss_check(x, coef, rootp, ns, ni, root, n_up, n_down, B, T, flags);
    if (rc) return rc;
    if (!y) return fail(WDF_EINVAL, "null y");
    const bool v4 = ((T * ni) % 4 == 0) && aligned16(x);
    WDF_SS_DISPATCH(ss_launch_fwd, x, coef, rootp, n_up, n_down, y, zstash, z0, zT, B, T, (hipStream_t)stream);
    return check_launch("wdf_ss_fwd");
}

size_t wdf_ss_fwd_lin_tp_ws_bytes(int ns, int64_t B, int n_chunks)
{
    return (ns > 0 && B > 0 && n_chunks > 0) ? (size_t)2 * (size_t)n_chunks * (size_t)ns * (size_t)B * sizeof(float) : 0;
}

int wdf_ss_fwd_lin_tp(const float* x, const float* coef, int ns, int ni, float* y, float* zstash, const float* z0, float* zT,
                      int64_t B, int64_t T, int n_chunks, void* ws, void* stream)
{
    int rc = ss_check(x, coef, nullptr, ns, ni, wdf::kRootNone, 1, 1, B, T, 0);
    if (rc) return rc;
    if (!y || !ws) return fail(WDF_EINVAL, "null y/ws");
    if (ns < 1) return fail(WDF_EINVAL, "a tree without states has nothing to scan: use wdf_ss_fwd");
    if (n_chunks < 1) return fail(WDF_EINVAL, "n_chunks >= 1");
    int64_t L = (T + n_chunks - 1) / n_chunks;
    L = (L + wdf::kLinBurst - 1) / wdf::kLinBurst * wdf::kLinBurst;   // chunks in whole bursts of the row loads
    const int K = (int)((T + L - 1) / L);
    float* zend0 = (float*)ws;
    float* zstart = zend0 + (size_t)K * (size_t)ns * (size_t)B;
    const dim3 grid((unsigned)((B + 63) / 64), (unsigned)K), one((unsigned)((B + 63) / 64));
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = ((T * ni) % 4 == 0) && aligned16(x);
    if (K > 1 && hipMemsetAsync(zend0 + (size_t)(K - 1) * (size_t)ns * (size_t)B, 0, (size_t)ns * (size_t)B * sizeof(float), s) != hipSuccess)
        return fail(WDF_ELAUNCH, "wdf_ss_fwd_lin_tp: memset failed");     // (the last chunk's zero-state end is never used nor written)
#define WDF_LIN_V(NS_, NI_, V4_)                                                                                 \
    {                                                                                                            \
        if (K > 1) hipLaunchKernelGGL((wdf::ss_lin_zero_state_kernel<NS_, NI_, V4_>), grid, dim3(64), 0, s, x, coef, zend0, B, T, L);  \
        hipLaunchKernelGGL((wdf::ss_lin_starts_kernel<NS_>), one, dim3(64), 0, s, coef, zend0, z0, zstart, B, (int64_t)K, L); \
        EventBracket bracket(s);                                                                                 \
        hipLaunchKernelGGL((wdf::ss_lin_chunk_kernel<NS_, NI_, V4_>), grid, dim3(64), 0, s, x, coef, zstart, y, zstash, zT, B, T, L); \
    }
#define WDF_LIN(NS_, NI_)                                                                                        \
    if (ns == NS_ && ni == NI_) {                                                                                \
        if (v4) WDF_LIN_V(NS_, NI_, true) else WDF_LIN_V(NS_, NI_, false)                                        \
    }
    WDF_LIN(1, 1) WDF_LIN(2, 1) WDF_LIN(3, 1) WDF_LIN(4, 1) WDF_LIN(1, 2) WDF_LIN(2, 2) WDF_LIN(3, 2) WDF_LIN(4, 2)
#undef WDF_LIN
#undef WDF_LIN_V
    return check_launch("wdf_ss_fwd_lin_tp");
}

int wdf_ss_bwd(const float* x, const float* coef, const float* rootp, int ns, int ni, int root, int n_up, int n_down,
               const float* zstash, const float* gy, void* ws, float* gcoef, float* groot, float* gz0, int64_t B,
               int64_t T, int flags, void* stream)
{
    int rc = ss_check(x, coef, rootp, ns, ni, root, n_up, n_down, B, T, flags);
    if (rc) return rc;
    if (!gy || !ws || !gcoef) return fail(WDF_EINVAL, "null gy/ws/gcoef");
    if (ns > 0 && !zstash) return fail(WDF_EINVAL, "null zstash");
    if (root == wdf::kRootDiode && !groot) return fail(WDF_EINVAL, "null groot");
    const bool v4 = ((T * ni) % 4 == 0) && aligned16(x);
    WDF_SS_DISPATCH(ss_launch_bwd, x, coef, rootp, n_up, n_down, zstash, gy, (double*)ws, gz0, B, T,
                    (hipStream_t)stream);
    rc = check_launch("wdf_ss_bwd");
    if (rc) return rc;
    const int ncoef = wdf_ss_ncoef(ns, ni);
    hipLaunchKernelGGL(wdf::ss_grad_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)ws,
                       (int)((B + 63) / 64), ncoef + 2, ncoef, root == wdf::kRootDiode ? rootp : nullptr, gcoef,
                       root == wdf::kRootDiode ? groot : nullptr);
    return check_launch("wdf_ss_grad_reduce");
}

// ---- time-parallel state-space kernels (wdf_statespace.h, second half) --------------------------------------
static void ss_tp_geom(int64_t T, int n_chunks, int64_t& L, int& K)
{
    if (n_chunks < 1) n_chunks = 1;
    L = (T + n_chunks - 1) / n_chunks;
    L = (L + 7) / 8 * 8;
    K = (int)((T + L - 1) / L);
}

int wdf_ss_tp_chunks(int64_t T, int n_chunks)
{
    int64_t L; int K;
    if (T <= 0) return 0;
    ss_tp_geom(T, n_chunks, L, K);
    return K;
}

size_t wdf_ss_fwd_tp_ws_bytes(int ns, int64_t B, int n_chunks)
{
    if (ns < 1 || B <= 0 || n_chunks <= 0) return 0;
    return (size_t)2 * (size_t)n_chunks * (size_t)ns * (size_t)B * sizeof(float) + (size_t)((B + 63) / 64) * sizeof(unsigned);
}

int wdf_ss_tp_starts(int64_t T, int n_chunks, int warmup, int64_t* starts)
{
    if (T <= 0 || n_chunks < 1 || warmup < 0 || !starts) return fail(WDF_EINVAL, "T > 0, n_chunks >= 1, warmup >= 0, starts != NULL");
    int64_t L; int K;
    ss_tp_geom(T, n_chunks, L, K);
    if (K != n_chunks) return fail(WDF_EINVAL, "n_chunks = %d does not tile T = %lld in 8-step units: use wdf_ss_tp_chunks (%d)", n_chunks, (long long)T, K);
    const int64_t W = ((int64_t)warmup + 7) / 8 * 8;
    for (int k = 0; k < K; ++k) starts[k] = (k * L > W) ? k * L - W : 0;
    return WDF_OK;
}

int wdf_ss_fwd_tp(const float* x, const float* coef, const float* rootp, int ns, int ni, int n_up, int n_down, float* y,
                  float* zstash, const float* z0, float* zT, int64_t B, int64_t T, int n_chunks, int warmup, float tol,
                  const float* zinit, void* ws, void* status, void* stream)
{
    int rc = ss_check(x, coef, rootp, ns, ni, wdf::kRootDiode, n_up, n_down, B, T, 0);
    if (rc) return rc;
    if (!y || !ws || !status) return fail(WDF_EINVAL, "null y/ws/status");
    if (ns < 1) return fail(WDF_EINVAL, "a tree without states has nothing to speculate about: use wdf_ss_fwd");
    if (n_chunks < 1 || warmup < 0 || !(tol >= 0.0f)) return fail(WDF_EINVAL, "n_chunks >= 1, warmup >= 0, tol >= 0");
    int64_t L; int K;
    ss_tp_geom(T, n_chunks, L, K);
    if (K != n_chunks) return fail(WDF_EINVAL, "n_chunks = %d does not tile T = %lld in 8-step units: use wdf_ss_tp_chunks (%d)", n_chunks, (long long)T, K);
    float* zwarm = (float*)ws;
    float* zend = zwarm + (size_t)K * (size_t)ns * (size_t)B;
    unsigned* gate = (unsigned*)(zend + (size_t)K * (size_t)ns * (size_t)B);
    const dim3 grid((unsigned)((B + 63) / 64), (unsigned)K);
    hipStream_t s = (hipStream_t)stream;
    const bool sym = n_up == n_down;
    const bool v4 = ((T * ni) % 4 == 0) && aligned16(x);
    const int64_t W = ((int64_t)warmup + 7) / 8 * 8;
#define WDF_SS_TP(NS_, NI_)                                                                                                  \
    if (ns == NS_ && ni == NI_) {                                                                                            \
        {                                                                                                                    \
            EventBracket bracket(s);                                                                                         \
            if (sym && v4) hipLaunchKernelGGL((wdf::ss_fwd_tp_kernel<NS_, NI_, true, true>), grid, dim3(64), 0, s, x, coef, rootp, n_up, n_down, y, \
                                              zstash, z0, zT, zwarm, zend, (wdf::SsTpStatus*)status, B, T, L, W, zinit);             \
            else if (sym) hipLaunchKernelGGL((wdf::ss_fwd_tp_kernel<NS_, NI_, true, false>), grid, dim3(64), 0, s, x, coef, rootp, n_up, n_down, y, \
                                             zstash, z0, zT, zwarm, zend, (wdf::SsTpStatus*)status, B, T, L, W, zinit);              \
            else if (v4) hipLaunchKernelGGL((wdf::ss_fwd_tp_kernel<NS_, NI_, false, true>), grid, dim3(64), 0, s, x, coef, rootp, n_up, n_down, y,  \
                                            zstash, z0, zT, zwarm, zend, (wdf::SsTpStatus*)status, B, T, L, W, zinit);               \
            else hipLaunchKernelGGL((wdf::ss_fwd_tp_kernel<NS_, NI_, false, false>), grid, dim3(64), 0, s, x, coef, rootp, n_up, n_down, y,        \
                                    zstash, z0, zT, zwarm, zend, (wdf::SsTpStatus*)status, B, T, L, W, zinit);                       \
        }                                                                                                                    \
        if (K > 1) {                                                                                                         \
            hipLaunchKernelGGL(wdf::ss_tp_verify_kernel, dim3(grid.x), dim3(64), 0, s, (const float*)zwarm, (const float*)zend, ns, B,  \
                               (int64_t)K, tol, gate, (wdf::SsTpStatus*)status);                                             \
            if (v4) hipLaunchKernelGGL((wdf::ss_fwd_kernel<NS_, NI_, wdf::kRootDiode, false, true>), dim3(grid.x), dim3(64), 0, s, x,   \
                                       coef, rootp, n_up, n_down, y, zstash, z0, zT, B, T, (const unsigned*)gate);           \
            else hipLaunchKernelGGL((wdf::ss_fwd_kernel<NS_, NI_, wdf::kRootDiode, false, false>), dim3(grid.x), dim3(64), 0, s, x,    \
                                    coef, rootp, n_up, n_down, y, zstash, z0, zT, B, T, (const unsigned*)gate);              \
        }                                                                                                                    \
    }
    WDF_SS_TP(1, 1) WDF_SS_TP(2, 1) WDF_SS_TP(3, 1) WDF_SS_TP(4, 1) WDF_SS_TP(1, 2) WDF_SS_TP(2, 2) WDF_SS_TP(3, 2) WDF_SS_TP(4, 2)
#undef WDF_SS_TP
    return check_launch("wdf_ss_fwd_tp");
}

static int ss_tp_rec(int ns, int ni) { const int nacc = wdf_ss_ncoef(ns, ni) + 2; return ns * ns + ns + nacc * (ns + 1); }

size_t wdf_ss_bwd_tp_ws_bytes(int ns, int ni, int64_t B, int n_chunks)
{
    if (ns < 1 || B <= 0 || n_chunks <= 0) return 0;
    return wdf_ss_bwd_ws_bytes(ns, ni, B) + (size_t)n_chunks * (size_t)ss_tp_rec(ns, ni) * (size_t)B * sizeof(float);
}

int wdf_ss_bwd_tp(const float* x, const float* coef, const float* rootp, int ns, int ni, int root, int n_up, int n_down,
                  const float* zstash, const float* gy, void* ws, float* gcoef, float* groot, float* gz0, int64_t B, int64_t T,
                  int n_chunks, void* stream)
{
    int rc = ss_check(x, coef, rootp, ns, ni, root, n_up, n_down, B, T, 0);
    if (rc) return rc;
    if (!gy || !ws || !gcoef || !zstash) return fail(WDF_EINVAL, "null gy/ws/gcoef/zstash");
    if (ns < 1) return fail(WDF_EINVAL, "a tree without states has no adjoint to scan: use wdf_ss_bwd");
    if (root == wdf::kRootDiode && !groot) return fail(WDF_EINVAL, "null groot");
    int64_t L; int K;
    ss_tp_geom(T, n_chunks, L, K);
    if (K != n_chunks) return fail(WDF_EINVAL, "n_chunks = %d does not tile T = %lld in 8-step units: use wdf_ss_tp_chunks (%d)", n_chunks, (long long)T, K);
    double* part = (double*)ws;
    float* rec = (float*)((char*)ws + wdf_ss_bwd_ws_bytes(ns, ni, B));
    const dim3 grid((unsigned)((B + 63) / 64), (unsigned)K);
    hipStream_t s = (hipStream_t)stream;
    const bool sym = n_up == n_down;
    const bool v4 = ((T * ni) % 4 == 0) && aligned16(x);
#define WDF_SS_BTP(NS_, NI_)                                                                                                 \
    if (ns == NS_ && ni == NI_) {                                                                                            \
        {                                                                                                                    \
            EventBracket bracket(s);                                                                                         \
            if (root == wdf::kRootNone && v4)                                                                                \
                hipLaunchKernelGGL((wdf::ss_bwd_tp_kernel<NS_, NI_, wdf::kRootNone, true, true>), grid, dim3(64), 0, s, x, coef, rootp, n_up,  \
                                   n_down, zstash, gy, rec, B, T, L);                                                        \
            else if (root == wdf::kRootNone)                                                                                 \
                hipLaunchKernelGGL((wdf::ss_bwd_tp_kernel<NS_, NI_, wdf::kRootNone, true, false>), grid, dim3(64), 0, s, x, coef, rootp, n_up, \
                                   n_down, zstash, gy, rec, B, T, L);                                                        \
            else if (sym && v4)                                                                                              \
                hipLaunchKernelGGL((wdf::ss_bwd_tp_kernel<NS_, NI_, wdf::kRootDiode, true, true>), grid, dim3(64), 0, s, x, coef, rootp, n_up, \
                                   n_down, zstash, gy, rec, B, T, L);                                                        \
            else if (sym)                                                                                                    \
                hipLaunchKernelGGL((wdf::ss_bwd_tp_kernel<NS_, NI_, wdf::kRootDiode, true, false>), grid, dim3(64), 0, s, x, coef, rootp, n_up, \
                                   n_down, zstash, gy, rec, B, T, L);                                                        \
            else if (v4)                                                                                                     \
                hipLaunchKernelGGL((wdf::ss_bwd_tp_kernel<NS_, NI_, wdf::kRootDiode, false, true>), grid, dim3(64), 0, s, x, coef, rootp, n_up, \
                                   n_down, zstash, gy, rec, B, T, L);                                                        \
            else                                                                                                             \
                hipLaunchKernelGGL((wdf::ss_bwd_tp_kernel<NS_, NI_, wdf::kRootDiode, false, false>), grid, dim3(64), 0, s, x, coef, rootp, n_up, \
                                   n_down, zstash, gy, rec, B, T, L);                                                        \
        }                                                                                                                    \
        hipLaunchKernelGGL((wdf::ss_bwd_tp_combine_kernel<NS_, NI_>), dim3(grid.x), dim3(64), 0, s, (const float*)rec, part, gz0, B,   \
                           (int64_t)K);                                                                                      \
    }
    WDF_SS_BTP(1, 1) WDF_SS_BTP(2, 1) WDF_SS_BTP(3, 1) WDF_SS_BTP(4, 1) WDF_SS_BTP(1, 2) WDF_SS_BTP(2, 2) WDF_SS_BTP(3, 2) WDF_SS_BTP(4, 2)
#undef WDF_SS_BTP
    rc = check_launch("wdf_ss_bwd_tp");
    if (rc) return rc;
    const int ncoef = wdf_ss_ncoef(ns, ni);
    hipLaunchKernelGGL(wdf::ss_grad_reduce_kernel, dim3(1), dim3(64), 0, s, (const double*)part, (int)((B + 63) / 64), ncoef + 2, ncoef,
                       root == wdf::kRootDiode ? rootp : nullptr, gcoef, root == wdf::kRootDiode ? groot : nullptr);
    return check_launch("wdf_ss_grad_reduce");
}

int wdf_clipper_asym_fwd(const float* x, const float* theta6, float fs, int mode, double tol, int max_iter, float* y,
                         float* zstash, const float* z0, float* zT, long long* iters, int64_t B, int64_t T, void* stream)
{
    if (!x || !theta6 || !y) return fail(WDF_EINVAL, "null x/theta6/y");
    if (B <= 0 || T <= 0 || !(fs > 0.0f)) return fail(WDF_EINVAL, "B, T, fs must be positive");
    if (mode != WDF_ASYM_OMEGA_F32 && mode != WDF_ASYM_NEWTON_F64) return fail(WDF_EINVAL, "unknown mode %d", mode);
    if (mode == WDF_ASYM_NEWTON_F64 && (!(tol > 0.0) || max_iter < 1)) return fail(WDF_EINVAL, "tol > 0, max_iter >= 1");
    const unsigned grid = (unsigned)((B + 63) / 64);
    const bool v4 = (T % 4 == 0) && aligned16(x);
#define WDF_ASYM(NEWTON_, V4_)                                                                                \
    hipLaunchKernelGGL((wdf::clipper_asym_fwd_kernel<NEWTON_, V4_>), dim3(grid), dim3(64), 0, (hipStream_t)stream, x, \
                       theta6, fs, y, zstash, z0, zT, tol, max_iter, iters, B, T)
    if (mode == WDF_ASYM_NEWTON_F64) { if (v4) WDF_ASYM(true, true); else WDF_ASYM(true, false); }
    else { if (v4) WDF_ASYM(false, true); else WDF_ASYM(false, false); }
#undef WDF_ASYM
    return check_launch("wdf_clipper_asym_fwd");
}

size_t wdf_clipper_asym_fwd_tp_ws_bytes(int64_t B, int n_chunks)
{
    if (B <= 0 || n_chunks <= 0) return 0;
    return (size_t)2 * (size_t)n_chunks * (size_t)B * sizeof(float) + (size_t)((B + 63) / 64) * sizeof(unsigned);
}

int wdf_clipper_asym_fwd_tp(const float* x, const float* theta6, float fs, int mode, double tol, int max_iter, float* y,
                            float* zstash, const float* z0, float* zT, int64_t B, int64_t T, int n_chunks, int warmup,
                            float verify_tol, void* ws, void* status, void* stream)
{
    if (!x || !theta6 || !y || !ws || !status) return fail(WDF_EINVAL, "null x/theta6/y/ws/status");
    if (B <= 0 || T <= 0 || !(fs > 0.0f)) return fail(WDF_EINVAL, "B, T, fs must be positive");
    if (mode != WDF_ASYM_OMEGA_F32 && mode != WDF_ASYM_NEWTON_F64) return fail(WDF_EINVAL, "unknown mode %d", mode);
    if (mode == WDF_ASYM_NEWTON_F64 && (!(tol > 0.0) || max_iter < 1)) return fail(WDF_EINVAL, "tol > 0, max_iter >= 1");
    if (n_chunks < 1 || warmup < 0 || !(verify_tol >= 0.0f)) return fail(WDF_EINVAL, "n_chunks >= 1, warmup >= 0, verify_tol >= 0");
    int64_t L = (T + n_chunks - 1) / n_chunks;
    L = (L + 7) / 8 * 8;
    const int K = (int)((T + L - 1) / L);
    if (K != n_chunks) return fail(WDF_EINVAL, "n_chunks = %d does not tile T = %lld in 8-step units (%d does)", n_chunks, (long long)T, K);
    const int64_t W = ((int64_t)warmup + 7) / 8 * 8;
    float* zwarm = (float*)ws;
    float* zend = zwarm + (size_t)K * (size_t)B;
    unsigned* gate = (unsigned*)(zend + (size_t)K * (size_t)B);
    const dim3 grid((unsigned)((B + 63) / 64), (unsigned)K);
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = (T % 4 == 0) && aligned16(x);
    if (mode == WDF_ASYM_NEWTON_F64)
        hipLaunchKernelGGL((wdf::clipper_asym_fwd_tp_kernel<true>), grid, dim3(64), 0, s, x, theta6, fs, y, zstash, z0, zT, zwarm, zend,
                           tol, max_iter, (wdf::AsymTpStatus*)status, B, T, L, W);
    else
        hipLaunchKernelGGL((wdf::clipper_asym_fwd_tp_kernel<false>), grid, dim3(64), 0, s, x, theta6, fs, y, zstash, z0, zT, zwarm, zend,
                           tol, max_iter, (wdf::AsymTpStatus*)status, B, T, L, W);
    if (K > 1) {
        hipLaunchKernelGGL(wdf::asym_tp_verify_kernel, dim3(grid.x), dim3(64), 0, s, zwarm, zend, B, (int64_t)K, verify_tol, gate,
                           (wdf::AsymTpStatus*)status);
#define WDF_ASYM_GATED(NEWTON_, V4_)                                                                                       \
        hipLaunchKernelGGL((wdf::clipper_asym_fwd_kernel<NEWTON_, V4_>), dim3(grid.x), dim3(64), 0, s, x, theta6, fs, y, zstash, z0, \
                           zT, tol, max_iter, (long long*)nullptr, B, T, (const unsigned*)gate)
        if (mode == WDF_ASYM_NEWTON_F64) { if (v4) WDF_ASYM_GATED(true, true); else WDF_ASYM_GATED(true, false); }
        else { if (v4) WDF_ASYM_GATED(false, true); else WDF_ASYM_GATED(false, false); }
#undef WDF_ASYM_GATED
    }
    return check_launch("wdf_clipper_asym_fwd_tp");
}

size_t wdf_clipper_asym_bwd_ws_bytes(int64_t B) { return B > 0 ? (size_t)((B + 63) / 64) * 8 * sizeof(double) : 0; }

int wdf_clipper_asym_bwd(const float* x, const float* theta6, float fs, double tol, int max_iter, const float* zstash,
                         const float* gy, void* ws, float* gtheta6, int64_t B, int64_t T, void* stream)
{
    if (!x || !theta6 || !zstash || !gy || !ws || !gtheta6) return fail(WDF_EINVAL, "null x/theta6/zstash/gy/ws/gtheta6");
    if (B <= 0 || T <= 0 || !(fs > 0.0f)) return fail(WDF_EINVAL, "B, T, fs must be positive");
    if (!(tol > 0.0) || max_iter < 1) return fail(WDF_EINVAL, "tol > 0, max_iter >= 1");
    const unsigned grid = (unsigned)((B + 63) / 64);
    hipLaunchKernelGGL(wdf::clipper_asym_bwd_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, x, theta6, fs, zstash, gy, tol,
                       max_iter, (double*)ws, B, T);
    hipLaunchKernelGGL(wdf::clipper_asym_grad_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws,
                       (int)grid, theta6, fs, gtheta6);
    return check_launch("wdf_clipper_asym_bwd");
}

size_t wdf_clipper_asym_bwd_tp_ws_bytes(int64_t B, int n_chunks)
{
    if (B <= 0 || n_chunks <= 0) return 0;
    return (size_t)n_chunks * (size_t)wdf::kAsymRec * (size_t)B * sizeof(double) + (size_t)((B + 63) / 64) * 8 * sizeof(double);
}

int wdf_clipper_asym_bwd_tp(const float* x, const float* theta6, float fs, int mode, const float* zstash, const float* zT,
                            const float* gy, const float* gzT, void* ws, float* gtheta6, float* gz0, int64_t B, int64_t T,
                            int n_chunks, void* stream)
{
    if (!x || !theta6 || !zstash || !zT || !gy || !ws || !gtheta6) return fail(WDF_EINVAL, "null x/theta6/zstash/zT/gy/ws/gtheta6");
    if (B <= 0 || T <= 0 || !(fs > 0.0f)) return fail(WDF_EINVAL, "B, T, fs must be positive");
    if (mode != WDF_ASYM_OMEGA_F32 && mode != WDF_ASYM_NEWTON_F64) return fail(WDF_EINVAL, "unknown mode %d", mode);
    if (n_chunks < 1) return fail(WDF_EINVAL, "n_chunks >= 1");
    int64_t L = (T + n_chunks - 1) / n_chunks;
    L = (L + 7) / 8 * 8;
    const int K = (int)((T + L - 1) / L);
    if (K != n_chunks) return fail(WDF_EINVAL, "n_chunks = %d does not tile T = %lld in 8-step units (%d does)", n_chunks, (long long)T, K);
    double* rec = (double*)ws;
    double* part = rec + (size_t)K * (size_t)wdf::kAsymRec * (size_t)B;
    const dim3 grid((unsigned)((B + 63) / 64), (unsigned)K);
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = (T % 4 == 0) && aligned16(x);
#define WDF_ASYM_BWD(NEWTON_, V4_) \
    hipLaunchKernelGGL((wdf::clipper_asym_bwd_tp_kernel<NEWTON_, V4_>), grid, dim3(64), 0, s, x, theta6, fs, zstash, zT, gy, rec, B, T, L)
    if (mode == WDF_ASYM_NEWTON_F64) { if (v4) WDF_ASYM_BWD(true, true); else WDF_ASYM_BWD(true, false); }
    else { if (v4) WDF_ASYM_BWD(false, true); else WDF_ASYM_BWD(false, false); }
#undef WDF_ASYM_BWD
    hipLaunchKernelGGL(wdf::clipper_asym_bwd_combine_kernel, dim3(grid.x), dim3(64), 0, s, (const double*)rec, gzT, part, gz0, B, (int64_t)K);
    hipLaunchKernelGGL(wdf::clipper_asym_grad_reduce_kernel, dim3(1), dim3(256), 0, s, (const double*)part, (int)grid.x, theta6, fs, gtheta6);
    return check_launch("wdf_clipper_asym_bwd_tp");
}

int wdf_asym_root(const float* a, const float* theta6, float fs, int mode, double tol, int max_iter, double* b, int64_t n,
                  void* stream)
{
    if (!a || !theta6 || !b || n <= 0) return fail(WDF_EINVAL, "wdf_asym_root: bad arguments");
    if (mode != WDF_ASYM_OMEGA_F32 && mode != WDF_ASYM_NEWTON_F64) return fail(WDF_EINVAL, "unknown mode %d", mode);
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (mode == WDF_ASYM_NEWTON_F64)
        hipLaunchKernelGGL((wdf::asym_root_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, theta6, fs, b,
                           tol, max_iter, n);
    else
        hipLaunchKernelGGL((wdf::asym_root_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, theta6, fs, b,
                           tol, max_iter, n);
    return check_launch("wdf_asym_root");
}

// ---- the one-pass MSE step of linear trees (wdf_ss_step.h) -------------------------------------------------------------
static int probe_launch(const wdf_adam_job* jobs, int n_jobs, const int32_t* tape, int n_ops, const double* consts, const float* params,
                        int n_params, const int32_t* outs, int n_out, float* coef, double* coef64, double* jac, void* stream, const char* what)
{
    if (!tape || !consts || !params || !outs || !coef || !coef64 || !jac) return fail(WDF_EINVAL, "null argument");
    if (n_ops < 1 || n_ops > wdf::kProbeMaxOps) return fail(WDF_EUNSUPPORTED, "%s: 1..%d operations (got %d)", what, wdf::kProbeMaxOps, n_ops);
    if (n_params < 1 || n_params > wdf::kProbeMaxParams) return fail(WDF_EUNSUPPORTED, "%s: 1..%d parameters (got %d)", what, wdf::kProbeMaxParams, n_params);
    if (n_out < 1) return fail(WDF_EINVAL, "n_out >= 1");
    if (n_jobs < 0 || n_jobs > WDF_ADAM_MULTI_MAX || (n_jobs > 0 && !jobs)) return fail(WDF_EINVAL, "%s: 0..%d jobs", what, WDF_ADAM_MULTI_MAX);
    wdf::AdamJobs a{};
    for (int i = 0; i < n_jobs; ++i) {
        const wdf_adam_job& j = jobs[i];
        if (!j.theta || !j.grad || !j.m || !j.v || !j.step || !j.lr) return fail(WDF_EINVAL, "job %d: null theta/grad/m/v/step/lr", i);
        if (j.n <= 0 || j.n > 1024) return fail(WDF_EINVAL, "job %d: n must be in 1..1024 (got %d)", i, j.n);
        a.j[i] = wdf::AdamJob{j.theta, j.grad, j.m, j.v, j.step, j.lr, j.lo, j.hi, j.beta1, j.beta2, j.eps, j.n};
    }
    hipLaunchKernelGGL(wdf::ss_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tape, n_ops, consts, params, n_params, outs,
                       n_out, coef, coef64, jac, a, n_jobs);
    return check_launch(what);
}

int wdf_ss_probe(const int32_t* tape, int n_ops, const double* consts, const float* params, int n_params, const int32_t* outs,
                 int n_out, float* coef, double* coef64, double* jac, void* stream)
{
    return probe_launch(nullptr, 0, tape, n_ops, consts, params, n_params, outs, n_out, coef, coef64, jac, stream, "wdf_ss_probe");
}

// wdf_adam_step_multi(jobs) followed by wdf_ss_probe, in ONE launch: the jobs update (slices of) `params`.
int wdf_ss_probe_adam(const wdf_adam_job* jobs, int n_jobs, const int32_t* tape, int n_ops, const double* consts, const float* params,
                      int n_params, const int32_t* outs, int n_out, float* coef, double* coef64, double* jac, void* stream)
{
    return probe_launch(jobs, n_jobs, tape, n_ops, consts, params, n_params, outs, n_out, coef, coef64, jac, stream, "wdf_ss_probe_adam");
}

static bool lin_step_ok(int ns, int ni) { return ns >= 0 && ns <= 2 && ni >= 1 && ni <= 2; }
static int lin_step_d(int ns, int ni) { return ns * (1 + ns * ns + ns * ni); }
static int lin_step_g(int ns, int ni) { return ns * ns + ns * ni + ns + ni; }
static void lin_step_geom(int64_t T, int n_chunks, int64_t& L, int& K)
{
    if (n_chunks < 1) n_chunks = 1;
    L = (T + n_chunks - 1) / n_chunks;
    L = (L + 31) / 32 * 32;
    K = (int)((T + L - 1) / L);
}

size_t wdf_ss_lin_step_ws_bytes(int ns, int ni, int64_t B, int64_t T, int n_chunks)
{
    if (!lin_step_ok(ns, ni) || B <= 0 || T <= 0 || n_chunks < 1) return 0;
    int64_t L; int K;
    lin_step_geom(T, n_chunks, L, K);
    const size_t waves = (size_t)((B + 63) / 64) * (size_t)K;
    return 256 + (size_t)K * (size_t)lin_step_d(ns, ni) * (size_t)B * sizeof(float) + 256 +
           waves * (size_t)(lin_step_g(ns, ni) + 1) * sizeof(double);
}

// x: [T][ni][B] time-major (the resident training set); coef / jac: wdf_ss_probe's outputs; target, y: [T][B].
// ws: wdf_ss_lin_step_ws_bytes() bytes, ZERO before the first call (the step leaves it clean).
// out: float [1 + n_params] = {sum of squared errors, d(gscale/2 x that sum)/d component value}; loss_out: (or NULL) <- gscale/2 x
// that sum; gcoef_out: float
// [ns^2 + ns ni + ns + ni] (dLoss/d{A, Bx, cy, dy}) or NULL.
// z0: float [ns][B] the capacitor states the call starts from, or NULL (zero); zT: float [ns][B] <- the states it ends in, or NULL
// (lpf.py:30-49 never resets C1: epoch n starts from epoch n - 1's final state).  z0 is a constant of the call (no gradient
// flows into the previous call, as in the reference, whose stored state belongs to the previous tape); zT must not alias z0.
int wdf_ss_lin_step_mse(const float* x, const float* coef, const double* jac, int n_params, int ns, int ni, const float* target,
                        float gscale, float* y, void* ws, float* out, float* loss_out, float* gcoef_out, int64_t B, int64_t T, int n_chunks,
                        const float* z0, float* zT, void* stream)
{
    if (z0 && z0 == zT) return fail(WDF_EINVAL, "wdf_ss_lin_step_mse: zT must not alias z0 (every chunk reads z0)");
    if (!x || !coef || !jac || !target || !y || !ws || !out) return fail(WDF_EINVAL, "null argument");
    if (!lin_step_ok(ns, ni)) return fail(WDF_EUNSUPPORTED, "wdf_ss_lin_step_mse: ns in 0..2, ni in 1..2 (got %d, %d)", ns, ni);
    if (B <= 0 || T <= 0 || n_chunks < 1 || n_params < 1 || n_params > wdf::kProbeMaxParams) return fail(WDF_EINVAL, "B, T, n_chunks >= 1, 1..%d parameters", wdf::kProbeMaxParams);
    int64_t L; int K;
    lin_step_geom(T, n_chunks, L, K);
    const int D = lin_step_d(ns, ni);
    unsigned* ticket = (unsigned*)ws;
    float* uend0 = (float*)((char*)ws + 256);
    double* part = (double*)(((uintptr_t)(uend0 + (size_t)K * (size_t)D * (size_t)B) + 255) & ~(uintptr_t)255);
    // two sequences per lane (8-byte loads and stores) when the rows of x, target, y and the workspace allow it
    const bool pair = (B % 2 == 0) && ((((uintptr_t)x | (uintptr_t)target | (uintptr_t)y | (uintptr_t)ws | (uintptr_t)z0 | (uintptr_t)zT) & 7u) == 0);
    const int64_t per_wave = pair ? 128 : 64;
    const dim3 grid((unsigned)((B + per_wave - 1) / per_wave), (unsigned)K);
    hipStream_t s = (hipStream_t)stream;
#define WDF_LIN_STEP_V(NS_, NI_, V_)                                                                               \
    {                                                                                                              \
        if (NS_ > 0 && K > 1)                                                                                      \
            hipLaunchKernelGGL((wdf::ss_lin_step_zero_kernel<NS_, NI_, V_>), grid, dim3(64), 0, s, x, coef, uend0, B, T, L);   \
        EventBracket bracket(s);                                                                                   \
        hipLaunchKernelGGL((wdf::ss_lin_step_kernel<NS_, NI_, V_>), grid, dim3(64), 0, s, x, coef, (const float*)uend0, \
                           target, gscale, y, part, ticket, jac, n_params, out, loss_out, gcoef_out, B, T, L, z0, zT);       \
    }
#define WDF_LIN_STEP(NS_, NI_)                                                                                     \
    if (ns == NS_ && ni == NI_) {                                                                                  \
        if (pair) WDF_LIN_STEP_V(NS_, NI_, wdf::v2f) else WDF_LIN_STEP_V(NS_, NI_, float)                          \
    }
    WDF_LIN_STEP(0, 1) WDF_LIN_STEP(0, 2) WDF_LIN_STEP(1, 1) WDF_LIN_STEP(1, 2) WDF_LIN_STEP(2, 1) WDF_LIN_STEP(2, 2)
#undef WDF_LIN_STEP
#undef WDF_LIN_STEP_V
    return check_launch("wdf_ss_lin_step_mse");
}

}  // extern "C"
