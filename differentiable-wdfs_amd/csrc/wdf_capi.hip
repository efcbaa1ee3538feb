// wdf_capi.hip -- the C ABI of libwdf_hip.so (include/wdf_hip.h).  gfx950 only.
//
// Argument checking, template dispatch and launches; no algorithm lives here.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/wdf_hip.h"
#include "wdf_clipper.h"
#include "wdf_asym.h"
#include "wdf_mlp.h"
#include "wdf_mlp_row.h"
#include "wdf_statespace.h"
#include "wdf_optim.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(WDF_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return WDF_OK;
}

// wdf_event_bracket_next(): events to record immediately before / after the next RECURRENCE
// kernel launched from this thread (the forward or reverse sweep itself, not the verify /
// combine / reduce helpers that share its C call), so a harness can time exactly the kernel
// rocprofv3 reports.  One-shot.
thread_local hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;

struct EventBracket {
    hipStream_t s;
    hipEvent_t e1;
    explicit EventBracket(hipStream_t stream) : s(stream), e1(g_ev1)
    {
        if (g_ev0) (void)hipEventRecord(g_ev0, s);
        g_ev0 = g_ev1 = nullptr;
    }
    ~EventBracket()
    {
        if (e1) (void)hipEventRecord(e1, s);
    }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <bool DYN_R, bool SYM, bool TM, bool V4>
void launch_fwd(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down, float* y,
                float* zstash, const float* z0, float* zT, int64_t B, int64_t T, int general, hipStream_t s)
{
    const unsigned grid = (unsigned)((B + 63) / 64);
    EventBracket bracket(s);
    if (zstash)
        hipLaunchKernelGGL((wdf::clipper_fwd_kernel<DYN_R, SYM, TM, V4, true>), dim3(grid), dim3(64), 0, s, x, r, theta,
                           fs, n_up, n_down, y, zstash, z0, zT, B, T, general);
    else
        hipLaunchKernelGGL((wdf::clipper_fwd_kernel<DYN_R, SYM, TM, V4, false>), dim3(grid), dim3(64), 0, s, x, r,
                           theta, fs, n_up, n_down, y, zstash, z0, zT, B, T, general);
}

template <bool DYN_R, bool SYM, bool TM, bool V4>
void launch_bwd(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                const float* zstash, const float* gy, double* ws, float* gz0, int64_t B, int64_t T, hipStream_t s)
{
    const unsigned grid = (unsigned)((B + 63) / 64);
    EventBracket bracket(s);
    hipLaunchKernelGGL((wdf::clipper_bwd_kernel<DYN_R, SYM, TM, V4>), dim3(grid), dim3(64), 0, s, x, r, theta, fs,
                       n_up, n_down, zstash, gy, ws, gz0, B, T);
}

// expands the 4 boolean template parameters from runtime flags
#define WDF_DISPATCH4(FN, dyn, sym, tm, v4, ...)                                                      \
    do {                                                                                              \
        const int key = ((dyn) ? 8 : 0) | ((sym) ? 4 : 0) | ((tm) ? 2 : 0) | ((v4) ? 1 : 0);          \
        switch (key) {                                                                                \
        case 0: FN<false, false, false, false>(__VA_ARGS__); break;                                   \
        case 1: FN<false, false, false, true>(__VA_ARGS__); break;                                    \
        case 2: FN<false, false, true, false>(__VA_ARGS__); break;                                    \
        case 4: FN<false, true, false, false>(__VA_ARGS__); break;                                    \
        case 5: FN<false, true, false, true>(__VA_ARGS__); break;                                     \
        case 6: FN<false, true, true, false>(__VA_ARGS__); break;                                     \
        case 8: FN<true, false, false, false>(__VA_ARGS__); break;                                    \
        case 9: FN<true, false, false, true>(__VA_ARGS__); break;                                     \
        case 10: FN<true, false, true, false>(__VA_ARGS__); break;                                    \
        case 12: FN<true, true, false, false>(__VA_ARGS__); break;                                    \
        case 13: FN<true, true, false, true>(__VA_ARGS__); break;                                     \
        case 14: FN<true, true, true, false>(__VA_ARGS__); break;                                     \
        default: FN<false, false, false, false>(__VA_ARGS__); break;                                  \
        }                                                                                             \
    } while (0)

int check_common(const float* x, const float* theta, int n_up, int n_down, int64_t B, int64_t T, int flags)
{
    if (!x || !theta) return fail(WDF_EINVAL, "null x/theta");
    if (B <= 0 || T <= 0) return fail(WDF_EINVAL, "B and T must be positive (got B=%lld T=%lld)", (long long)B, (long long)T);
    if (B > (int64_t)64 * 0x7fffffff) return fail(WDF_EINVAL, "B too large");
    if (n_up < 1 || n_down < 1 || n_up > 16 || n_down > 16) return fail(WDF_EINVAL, "n_up/n_down must be in [1,16]");
    if (flags & ~(WDF_X_TIME_MAJOR | WDF_PREC_F64 | WDF_TP_PACK2 | WDF_GENERAL_ROOT)) return fail(WDF_EINVAL, "unknown flag bits 0x%x", flags);
    if (flags & WDF_PREC_F64) return fail(WDF_EUNSUPPORTED, "WDF_PREC_F64 is not available for the Wright-omega clipper");
    return WDF_OK;
}

// ---- time-parallel clipper dispatch -----------------------------------------------------------
#define WDF_DISPATCH3(FN, dyn, sym, v4, ...)                                                     \
    do {                                                                                         \
        const int key3 = ((dyn) ? 4 : 0) | ((sym) ? 2 : 0) | ((v4) ? 1 : 0);                     \
        switch (key3) {                                                                          \
        case 0: FN<false, false, false>(__VA_ARGS__); break;                                     \
        case 1: FN<false, false, true>(__VA_ARGS__); break;                                      \
        case 2: FN<false, true, false>(__VA_ARGS__); break;                                      \
        case 3: FN<false, true, true>(__VA_ARGS__); break;                                       \
        case 4: FN<true, false, false>(__VA_ARGS__); break;                                      \
        case 5: FN<true, false, true>(__VA_ARGS__); break;                                       \
        case 6: FN<true, true, false>(__VA_ARGS__); break;                                       \
        default: FN<true, true, true>(__VA_ARGS__); break;                                       \
        }                                                                                        \
    } while (0)

struct TpGeom { int64_t L; int K; };

TpGeom tp_geom(int64_t T, int n_chunks)
{
    if (n_chunks < 1) n_chunks = 1;
    int64_t L = (T + n_chunks - 1) / n_chunks;
    L = (L + wdf::kTile - 1) / wdf::kTile * wdf::kTile;
    return {L, (int)((T + L - 1) / L)};
}

template <bool DYN_R, bool SYM, bool TM, bool V4>
void launch_fwd_tp(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down, float* y,
                   float* zstash, const float* z0, float* zT, float* zwarm, float* zend, wdf::TpStatus* status,
                   float tol, int64_t B, int64_t T, TpGeom g, int64_t W, bool pack, int general, hipStream_t s)
{
    const unsigned gseq = (unsigned)((B + 63) / 64);
    const int64_t Bh = pack ? (B + 1) / 2 : B;
    const dim3 grid((unsigned)((Bh + 63) / 64), (unsigned)g.K);
#define WDF_FWD_TP(STASH_, V_)                                                                             \
    hipLaunchKernelGGL((wdf::clipper_fwd_tp_kernel<DYN_R, SYM, TM, V4, STASH_, V_>), grid, dim3(64), 0, s, x, r, theta, \
                       fs, n_up, n_down, y, zstash, z0, zT, zwarm, zend, status, B, Bh, T, g.L, W, general)
    {
        EventBracket bracket(s);
        if (pack) {
            if (zstash) WDF_FWD_TP(true, wdf::v2f); else WDF_FWD_TP(false, wdf::v2f);
        } else {
            if (zstash) WDF_FWD_TP(true, float); else WDF_FWD_TP(false, float);
        }
    }
#undef WDF_FWD_TP
    if (g.K > 1) {
        // the repair path reads x row-wise; give it a batch-major view only if x is batch-major
        if (zstash)
            hipLaunchKernelGGL((wdf::clipper_tp_verify_fix_kernel<DYN_R, SYM, TM, true>), dim3(gseq), dim3(64), 0, s,
                               x, r, theta, fs, n_up, n_down, y, zstash, z0, zT, zwarm, zend, B, T, (int64_t)g.K, tol,
                               status, general);
        else
            hipLaunchKernelGGL((wdf::clipper_tp_verify_fix_kernel<DYN_R, SYM, TM, false>), dim3(gseq), dim3(64), 0, s,
                               x, r, theta, fs, n_up, n_down, y, zstash, z0, zT, zwarm, zend, B, T, (int64_t)g.K, tol,
                               status, general);
    }
}

template <bool DYN_R, bool SYM, bool TM, bool V4>
void launch_bwd_tp(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                   const float* zstash, const float* gy, const float* target, const float* zT, float gscale,
                   float* part, double* ws, float* gz0, int64_t B, int64_t T, TpGeom g, bool pack, const float* gcoef,
                   int64_t skip, unsigned* ticket, float* gtheta, int accumulate, float* sse_out, wdf::AdamTail adam,
                   hipStream_t s)
{
    const int64_t Bh = pack ? (B + 1) / 2 : B;
    const dim3 grid((unsigned)((Bh + 63) / 64), (unsigned)g.K);
#define WDF_BWD_TP(MSE_, V_)                                                                               \
    hipLaunchKernelGGL((wdf::clipper_bwd_tp_kernel<DYN_R, SYM, TM, V4, MSE_, V_>), grid, dim3(64), 0, s, x, r, theta, \
                       fs, n_up, n_down, zstash, gy, target, zT, gscale, part, B, Bh, T, g.L, gcoef, skip, ticket)
    {
        EventBracket bracket(s);
        if (gcoef) {
            WDF_BWD_TP(2, float);                            // MSE + ESR (one sequence per lane only)
        } else if (pack) {
            if (target) WDF_BWD_TP(1, wdf::v2f); else WDF_BWD_TP(0, wdf::v2f);
        } else {
            if (target) WDF_BWD_TP(1, float); else WDF_BWD_TP(0, float);
        }
    }
#undef WDF_BWD_TP
    hipLaunchKernelGGL(wdf::clipper_bwd_tp_combine_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, part, B,
                       (int64_t)g.K, ws, gz0, ticket, theta, fs, DYN_R ? 1 : 0, gtheta, accumulate, sse_out, adam);
}

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
        WDF_SS_CASE(FN, 0, 2, __VA_ARGS__) WDF_SS_CASE(FN, 1, 2, __VA_ARGS__)                    \
        WDF_SS_CASE(FN, 2, 2, __VA_ARGS__) WDF_SS_CASE(FN, 3, 2, __VA_ARGS__)                    \
    } while (0)

int ss_check(const float* x, const float* coef, const float* rootp, int ns, int ni, int root, int n_up, int n_down,
             int64_t B, int64_t T, int flags)
{
    if (!x || !coef) return fail(WDF_EINVAL, "null x/coef");
    if (ns < 0 || ns > 3 || ni < 1 || ni > 2) return fail(WDF_EUNSUPPORTED, "state-space kernels cover ns in [0,3], ni in [1,2] (got ns=%d ni=%d)", ns, ni);
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

int wdf_abi_version(void) { return WDF_HIP_ABI_VERSION; }

const char* wdf_last_error(void) { return g_err; }

int wdf_device_info(int device, char* name, int cap)
{
    hipDeviceProp_t p;
    const hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) return fail(WDF_ELAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (name && cap > 0) {
        strncpy(name, p.gcnArchName, (size_t)cap - 1);
        name[cap - 1] = 0;
    }
    return p.multiProcessorCount;
}

int wdf_clipper_fwd(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down, float* y,
                    float* zstash, const float* z0, float* zT, int64_t B, int64_t T, int flags, void* stream)
{
    int rc = check_common(x, theta, n_up, n_down, B, T, flags);
    if (rc) return rc;
    if (!y) return fail(WDF_EINVAL, "null y");
    if (!(fs > 0.0f)) return fail(WDF_EINVAL, "fs must be positive");
    const bool tm = flags & WDF_X_TIME_MAJOR;
    const bool v4 = !tm && (T % 4 == 0) && aligned16(x) && (!r || aligned16(r));
    WDF_DISPATCH4(launch_fwd, r != nullptr, n_up == n_down, tm, v4, x, r, theta, fs, n_up, n_down, y, zstash, z0, zT,
                  B, T, (flags & WDF_GENERAL_ROOT) ? 1 : 0, (hipStream_t)stream);
    return check_launch("wdf_clipper_fwd");
}

size_t wdf_clipper_bwd_ws_bytes(int64_t B) { return B > 0 ? (size_t)((B + 63) / 64) * 4 * sizeof(double) : 0; }

int wdf_clipper_bwd(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                    const float* zstash, const float* gy, void* ws, float* gtheta, float* gz0, int accumulate,
                    int64_t B, int64_t T, int flags, void* stream)
{
    int rc = check_common(x, theta, n_up, n_down, B, T, flags);
    if (rc) return rc;
    if (!zstash || !gy || !ws || !gtheta) return fail(WDF_EINVAL, "null zstash/gy/ws/gtheta");
    if (!(fs > 0.0f)) return fail(WDF_EINVAL, "fs must be positive");
    const bool tm = flags & WDF_X_TIME_MAJOR;
    const bool v4 = !tm && (T % 4 == 0) && aligned16(x) && (!r || aligned16(r));
    WDF_DISPATCH4(launch_bwd, r != nullptr, n_up == n_down, tm, v4, x, r, theta, fs, n_up, n_down, zstash, gy,
                  (double*)ws, gz0, B, T, (hipStream_t)stream);
    rc = check_launch("wdf_clipper_bwd");
    if (rc) return rc;
    const int nparts = (int)((B + 63) / 64);
    hipLaunchKernelGGL(wdf::clipper_grad_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       (const double*)ws, nparts, theta, fs, r != nullptr ? 1 : 0, gtheta, accumulate,
                       (float*)nullptr);
    return check_launch("wdf_clipper_grad_reduce");
}

int wdf_clipper_tp_chunks(int64_t T, int n_chunks) { return T > 0 ? tp_geom(T, n_chunks).K : 0; }

size_t wdf_clipper_fwd_tp_ws_bytes(int64_t B, int n_chunks)
{
    return (B > 0 && n_chunks > 0) ? (size_t)2 * (size_t)n_chunks * (size_t)B * sizeof(float) : 0;
}

int wdf_clipper_fwd_tp(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down, float* y,
                       float* zstash, const float* z0, float* zT, int64_t B, int64_t T, int n_chunks, int warmup,
                       float tol, void* ws, void* status, int flags, void* stream)
{
    int rc = check_common(x, theta, n_up, n_down, B, T, flags);
    if (rc) return rc;
    if (!y || !ws || !status) return fail(WDF_EINVAL, "null y/ws/status");
    if (!(fs > 0.0f)) return fail(WDF_EINVAL, "fs must be positive");
    if (n_chunks < 1 || warmup < 0 || !(tol >= 0.0f)) return fail(WDF_EINVAL, "n_chunks >= 1, warmup >= 0, tol >= 0");
    const TpGeom g = tp_geom(T, n_chunks);
    const int64_t W = ((int64_t)warmup + wdf::kTile - 1) / wdf::kTile * wdf::kTile;
    float* zwarm = (float*)ws;
    float* zend = zwarm + (size_t)g.K * (size_t)B;
    const bool tm = (flags & WDF_X_TIME_MAJOR) != 0;
    const bool v4 = !tm && (T % 4 == 0) && aligned16(x) && (!r || aligned16(r));
    WDF_DISPATCH4(launch_fwd_tp, r != nullptr, n_up == n_down, tm, v4, x, r, theta, fs, n_up, n_down, y, zstash, z0, zT,
                  zwarm, zend, (wdf::TpStatus*)status, tol, B, T, g, W, (flags & WDF_TP_PACK2) != 0 && B >= 2,
                  (flags & WDF_GENERAL_ROOT) ? 1 : 0,
                  (hipStream_t)stream);
    return check_launch("wdf_clipper_fwd_tp");
}

static int bwd_tp_common(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                         const float* zstash, const float* gy, const float* target, const float* zT, float gscale,
                         void* ws, float* gtheta, float* sse, float* gz0, int accumulate, int64_t B, int64_t T,
                         int n_chunks, int flags, void* stream, const float* gcoef = nullptr, int64_t skip = 0,
                         wdf::AdamTail adam = wdf::AdamTail{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr});

size_t wdf_clipper_bwd_tp_ws_bytes(int64_t B, int n_chunks)
{
    if (B <= 0 || n_chunks <= 0) return 0;
    return (size_t)n_chunks * wdf::kTpOut * (size_t)B * sizeof(float) + wdf_clipper_bwd_ws_bytes(B) + 16;   // + ticket
}

int wdf_clipper_bwd_tp(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                       const float* zstash, const float* gy, void* ws, float* gtheta, float* gz0, int accumulate,
                       int64_t B, int64_t T, int n_chunks, int flags, void* stream)
{
    if (!gy) return fail(WDF_EINVAL, "null gy");
    return bwd_tp_common(x, r, theta, fs, n_up, n_down, zstash, gy, nullptr, nullptr, 0.0f, ws, gtheta, nullptr, gz0,
                         accumulate, B, T, n_chunks, flags, stream);
}

int wdf_clipper_bwd_mse_tp(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                           const float* zstash, const float* zT, const float* target, float gscale, void* ws,
                           float* gtheta, float* sse, float* gz0, int accumulate, int64_t B, int64_t T, int n_chunks,
                           int flags, void* stream)
{
    if (!zT || !target) return fail(WDF_EINVAL, "null zT/target");
    return bwd_tp_common(x, r, theta, fs, n_up, n_down, zstash, nullptr, target, zT, gscale, ws, gtheta, sse, gz0,
                         accumulate, B, T, n_chunks, flags, stream);
}

static int bwd_tp_common(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                         const float* zstash, const float* gy, const float* target, const float* zT, float gscale,
                         void* ws, float* gtheta, float* sse, float* gz0, int accumulate, int64_t B, int64_t T,
                         int n_chunks, int flags, void* stream, const float* gcoef, int64_t skip, wdf::AdamTail adam)
{
    int rc = check_common(x, theta, n_up, n_down, B, T, flags);
    if (rc) return rc;
    if (!zstash || !ws || !gtheta) return fail(WDF_EINVAL, "null zstash/ws/gtheta");
    if (!(fs > 0.0f)) return fail(WDF_EINVAL, "fs must be positive");
    if (n_chunks < 1) return fail(WDF_EINVAL, "n_chunks >= 1");
    const TpGeom g = tp_geom(T, n_chunks);
    double* wsd = (double*)ws;                                       // [nparts][4] doubles first (8-byte aligned)
    float* part = (float*)((char*)ws + wdf_clipper_bwd_ws_bytes(B)); // then [K][9][B] floats
    unsigned* ticket = (unsigned*)((char*)ws + wdf_clipper_bwd_tp_ws_bytes(B, n_chunks) - 16);   // then the block ticket
    const bool tm = (flags & WDF_X_TIME_MAJOR) != 0;
    const bool v4 = !tm && (T % 4 == 0) && aligned16(x) && (!r || aligned16(r));
    // sweep, then combine -- whose last block also reduces, applies the chain rule and (optionally) Adam
    WDF_DISPATCH4(launch_bwd_tp, r != nullptr, n_up == n_down, tm, v4, x, r, theta, fs, n_up, n_down, zstash, gy, target,
                  zT, gscale, part, wsd, gz0, B, T, g, (flags & WDF_TP_PACK2) != 0 && B >= 2 && !gcoef, gcoef, skip,
                  ticket, gtheta, accumulate, target ? sse : nullptr, adam, (hipStream_t)stream);
    return check_launch("wdf_clipper_bwd_tp");
}

int wdf_clipper_bwd_mse_tp_adam(const float* x, const float* r, float* theta, float fs, int n_up, int n_down,
                                const float* zstash, const float* zT, const float* target, float gscale, void* ws,
                                float* gtheta, float* sse, int64_t B, int64_t T, int n_chunks, int flags, float* m,
                                float* v, int32_t* step, const float* lr, float beta1, float beta2, float eps,
                                const float* lo, const float* hi, void* stream)
{
    if (!zT || !target) return fail(WDF_EINVAL, "null zT/target");
    if (!m || !v || !step || !lr) return fail(WDF_EINVAL, "null m/v/step/lr");
    const wdf::AdamTail adam{theta, m, v, step, lr, beta1, beta2, eps, lo, hi};
    return bwd_tp_common(x, r, theta, fs, n_up, n_down, zstash, nullptr, target, zT, gscale, ws, gtheta, sse, nullptr, 0, B,
                         T, n_chunks, flags, stream, nullptr, 0, adam);
}

int wdf_clipper_bwd_esr_tp(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                           const float* zstash, const float* zT, const float* target, const float* gcoef, int64_t skip,
                           void* ws, float* gtheta, float* sse, float* gz0, int accumulate, int64_t B, int64_t T,
                           int n_chunks, int flags, void* stream)
{
    if (!zT || !target || !gcoef) return fail(WDF_EINVAL, "null zT/target/gcoef");
    if (skip < 0 || skip > T) return fail(WDF_EINVAL, "skip must be in 0..T");
    return bwd_tp_common(x, r, theta, fs, n_up, n_down, zstash, nullptr, target, zT, 0.0f, ws, gtheta, sse, gz0,
                         accumulate, B, T, n_chunks, flags, stream, gcoef, skip);
}

static unsigned loss_blocks(int64_t n)
{
    const int64_t want = (n + 256 * 8 - 1) / (256 * 8);
    return (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
}

int64_t wdf_loss_sums_ws_bytes(void) { return 2048 * 2 * (int64_t)sizeof(double); }

int wdf_loss_sums(const float* y, const float* target, int64_t B, int64_t T, int64_t skip, void* ws, double* sums,
                  void* stream)
{
    if (!y || !target || !ws || !sums) return fail(WDF_EINVAL, "null y/target/ws/sums");
    if (B <= 0 || T <= 0 || skip < 0 || skip >= T) return fail(WDF_EINVAL, "need B, T > 0 and 0 <= skip < T");
    const int64_t n0 = skip * B, n1 = T * B;
    const unsigned nblk = loss_blocks(n1 - n0);
    hipLaunchKernelGGL(wdf::loss_sums_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, y, target, n0, n1, (double*)ws);
    hipLaunchKernelGGL(wdf::loss_sums_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws,
                       (int)nblk, sums);
    return check_launch("wdf_loss_sums");
}

int wdf_esr_coef(const double* sums, double n, double eps, float* gcoef, float* loss, void* stream)
{
    if (!sums || !gcoef || !loss) return fail(WDF_EINVAL, "null sums/gcoef/loss");
    if (!(n > 0.0)) return fail(WDF_EINVAL, "n must be positive");
    hipLaunchKernelGGL(wdf::esr_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sums, n, eps, gcoef, loss);
    return check_launch("wdf_esr_coef");
}

int wdf_clipper_asym_fwd(const float* x, const float* theta6, float fs, int mode, double tol, int max_iter, float* y,
                         const float* z0, float* zT, long long* iters, int64_t B, int64_t T, void* stream)
{
    if (!x || !theta6 || !y) return fail(WDF_EINVAL, "null x/theta6/y");
    if (B <= 0 || T <= 0 || !(fs > 0.0f)) return fail(WDF_EINVAL, "B, T, fs must be positive");
    if (mode != WDF_ASYM_OMEGA_F32 && mode != WDF_ASYM_NEWTON_F64) return fail(WDF_EINVAL, "unknown mode %d", mode);
    if (mode == WDF_ASYM_NEWTON_F64 && (!(tol > 0.0) || max_iter < 1)) return fail(WDF_EINVAL, "tol > 0, max_iter >= 1");
    const unsigned grid = (unsigned)((B + 63) / 64);
    const bool v4 = (T % 4 == 0) && aligned16(x);
#define WDF_ASYM(NEWTON_, V4_)                                                                                \
    hipLaunchKernelGGL((wdf::clipper_asym_fwd_kernel<NEWTON_, V4_>), dim3(grid), dim3(64), 0, (hipStream_t)stream, x, \
                       theta6, fs, y, z0, zT, tol, max_iter, iters, B, T)
    if (mode == WDF_ASYM_NEWTON_F64) { if (v4) WDF_ASYM(true, true); else WDF_ASYM(true, false); }
    else { if (v4) WDF_ASYM(false, true); else WDF_ASYM(false, false); }
#undef WDF_ASYM
    return check_launch("wdf_clipper_asym_fwd");
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

// the architectures the MLP kernels are instantiated for (every one among the reference's model files)
static bool mlp_arch_ok(int hidden, int n_tanh_layers)
{
    return ((hidden == 4 || hidden == 8 || hidden == 16) && n_tanh_layers == 3) ||
           ((hidden == 4 || hidden == 8) && (n_tanh_layers == 4 || n_tanh_layers == 5));
}

int wdf_mlp_weight_count(int hidden, int n_tanh_layers)
{
    if (hidden < 1 || n_tanh_layers < 1) return 0;
    return 2 * hidden + hidden + (n_tanh_layers - 1) * (hidden * hidden + hidden) + hidden + 1;
}

#define WDF_MLP_CASE(H_, NL_, DYN_, KERNEL, ...)                                                              \
    if (hidden == H_ && n_tanh_layers == NL_ && dyn == DYN_)                                                  \
        hipLaunchKernelGGL((wdf::KERNEL<H_, NL_, DYN_>), dim3(grid), dim3(64), 0, (hipStream_t)stream, __VA_ARGS__);
#define WDF_MLP_DISPATCH(KERNEL, ...)                                                                         \
    WDF_MLP_CASE(4, 3, false, KERNEL, __VA_ARGS__) WDF_MLP_CASE(4, 3, true, KERNEL, __VA_ARGS__)              \
    WDF_MLP_CASE(8, 3, false, KERNEL, __VA_ARGS__) WDF_MLP_CASE(8, 3, true, KERNEL, __VA_ARGS__)              \
    WDF_MLP_CASE(16, 3, false, KERNEL, __VA_ARGS__) WDF_MLP_CASE(16, 3, true, KERNEL, __VA_ARGS__)            \
    WDF_MLP_CASE(4, 5, false, KERNEL, __VA_ARGS__) WDF_MLP_CASE(4, 5, true, KERNEL, __VA_ARGS__)              \
    WDF_MLP_CASE(8, 5, false, KERNEL, __VA_ARGS__) WDF_MLP_CASE(8, 5, true, KERNEL, __VA_ARGS__)              \
    WDF_MLP_CASE(4, 4, false, KERNEL, __VA_ARGS__) WDF_MLP_CASE(4, 4, true, KERNEL, __VA_ARGS__)              \
    WDF_MLP_CASE(8, 4, false, KERNEL, __VA_ARGS__) WDF_MLP_CASE(8, 4, true, KERNEL, __VA_ARGS__)

static int mlp_check(const float* x, const float* theta2, const float* w, int hidden, int n_tanh_layers, float fs,
                     int64_t B, int64_t T, int flags)
{
    if (!x || !theta2 || !w) return fail(WDF_EINVAL, "null x/theta2/w");
    if (B <= 0 || T <= 0) return fail(WDF_EINVAL, "B and T must be positive");
    if (!(fs > 0.0f)) return fail(WDF_EINVAL, "fs must be positive");
    if (flags & ~WDF_MLP_LANE_PER_SEQUENCE) return fail(WDF_EINVAL, "MLP-root kernels take flags = 0 or WDF_MLP_LANE_PER_SEQUENCE");
    if (!mlp_arch_ok(hidden, n_tanh_layers))
        return fail(WDF_EUNSUPPORTED,
                    "MLP root: hidden in {4,8,16} with 3 tanh layers or {4,8} with 4 or 5 (got width %d, %d tanh layers)",
                    hidden, n_tanh_layers);
    return WDF_OK;
}

int wdf_clipper_mlp_fwd(const float* x, const float* r, const float* theta2, const float* w, int hidden,
                        int n_tanh_layers, float fs, float* y, float* zstash, const float* z0, float* zT, int64_t B,
                        int64_t T, int flags, void* stream)
{
    int rc = mlp_check(x, theta2, w, hidden, n_tanh_layers, fs, B, T, flags);
    if (rc) return rc;
    if (!y) return fail(WDF_EINVAL, "null y");
    const unsigned grid = (unsigned)((B + 63) / 64);
    const bool dyn = r != nullptr;
    if (flags & WDF_MLP_LANE_PER_SEQUENCE) {
        WDF_MLP_DISPATCH(clipper_mlp_fwd_kernel, x, r, theta2, w, fs, y, zstash, z0, zT, B, T)
    } else {                                      // one 16-lane row per sequence (wdf_mlp_row.h)
        const unsigned grow = (unsigned)((B + 3) / 4);
#define WDF_ROW_FWD(NL_)                                                                                       \
    if (n_tanh_layers == NL_) {                                                                                \
        if (dyn) hipLaunchKernelGGL((wdf::clipper_mlp_row_fwd_kernel<NL_, true>), dim3(grow), dim3(64), 0,        \
                                    (hipStream_t)stream, x, r, theta2, w, hidden, fs, y, zstash, z0, zT, B, T);   \
        else hipLaunchKernelGGL((wdf::clipper_mlp_row_fwd_kernel<NL_, false>), dim3(grow), dim3(64), 0,           \
                                (hipStream_t)stream, x, r, theta2, w, hidden, fs, y, zstash, z0, zT, B, T);       \
    }
        WDF_ROW_FWD(3) WDF_ROW_FWD(4) WDF_ROW_FWD(5)
#undef WDF_ROW_FWD
    }
    return check_launch("wdf_clipper_mlp_fwd");
}

int64_t wdf_clipper_mlp_bwd_w_ws_bytes(int hidden, int n_tanh_layers, int64_t B)
{
    if (B <= 0 || !mlp_arch_ok(hidden, n_tanh_layers)) return 0;
    const int64_t nblk = (B + 3) / 4;
    return nblk * 4 * (int64_t)sizeof(double) + nblk * wdf_mlp_weight_count(hidden, n_tanh_layers) * (int64_t)sizeof(float);
}

int wdf_clipper_mlp_bwd_w(const float* x, const float* r, const float* theta2, const float* w, int hidden,
                          int n_tanh_layers, float fs, const float* zstash, const float* gy, void* ws, float* gtheta2,
                          float* gw, int64_t B, int64_t T, int flags, void* stream)
{
    int rc = mlp_check(x, theta2, w, hidden, n_tanh_layers, fs, B, T, 0);
    if (rc) return rc;
    if (flags != 0) return fail(WDF_EINVAL, "wdf_clipper_mlp_bwd_w takes flags = 0");
    if (!zstash || !gy || !ws || !gtheta2 || !gw) return fail(WDF_EINVAL, "null zstash/gy/ws/gtheta2/gw");
    const unsigned grid = (unsigned)((B + 3) / 4);
    const bool dyn = r != nullptr;
    double* wsd = (double*)ws;
    float* wsw = (float*)((char*)ws + (size_t)grid * 4 * sizeof(double));
#define WDF_ROW_BWD_W(NL_)                                                                                     \
    if (n_tanh_layers == NL_) {                                                                                \
        if (dyn) hipLaunchKernelGGL((wdf::clipper_mlp_row_bwd_w_kernel<NL_, true>), dim3(grid), dim3(64), 0,      \
                                    (hipStream_t)stream, x, r, theta2, w, hidden, fs, zstash, gy, wsw, wsd, B, T);  \
        else hipLaunchKernelGGL((wdf::clipper_mlp_row_bwd_w_kernel<NL_, false>), dim3(grid), dim3(64), 0,         \
                                (hipStream_t)stream, x, r, theta2, w, hidden, fs, zstash, gy, wsw, wsd, B, T);    \
    }
    WDF_ROW_BWD_W(3) WDF_ROW_BWD_W(4) WDF_ROW_BWD_W(5)
#undef WDF_ROW_BWD_W
    rc = check_launch("wdf_clipper_mlp_bwd_w");
    if (rc) return rc;
    hipLaunchKernelGGL(wdf::clipper_mlp_grad_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       (const double*)wsd, (int)grid, theta2, fs, dyn ? 1 : 0, gtheta2);
    const int count = wdf_mlp_weight_count(hidden, n_tanh_layers);
    hipLaunchKernelGGL(wdf::mlp_wgrad_reduce_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)wsw, (int)grid, count, gw);
    return check_launch("wdf_clipper_mlp_bwd_w reduce");
}

size_t wdf_clipper_mlp_bwd_ws_bytes(int64_t B) { return B > 0 ? (size_t)((B + 3) / 4) * 4 * sizeof(double) : 0; }

int wdf_clipper_mlp_bwd(const float* x, const float* r, const float* theta2, const float* w, int hidden,
                        int n_tanh_layers, float fs, const float* zstash, const float* gy, float* gb, float* ain,
                        float* lrin, void* ws, float* gtheta2, int64_t B, int64_t T, int flags, void* stream)
{
    int rc = mlp_check(x, theta2, w, hidden, n_tanh_layers, fs, B, T, flags);
    if (rc) return rc;
    if (!zstash || !gy || !gb || !ain || !ws || !gtheta2) return fail(WDF_EINVAL, "null zstash/gy/gb/ain/ws/gtheta2");
    if (r && !lrin) return fail(WDF_EINVAL, "per-sample resistance needs lrin");
    unsigned grid = (unsigned)((B + 63) / 64);
    const bool dyn = r != nullptr;
    if (flags & WDF_MLP_LANE_PER_SEQUENCE) {
        WDF_MLP_DISPATCH(clipper_mlp_bwd_kernel, x, r, theta2, w, fs, zstash, gy, gb, ain, lrin, (double*)ws, B, T)
    } else {                                      // one 16-lane row per sequence (wdf_mlp_row.h)
        grid = (unsigned)((B + 3) / 4);
#define WDF_ROW_BWD(NL_)                                                                                       \
    if (n_tanh_layers == NL_) {                                                                                \
        if (dyn) hipLaunchKernelGGL((wdf::clipper_mlp_row_bwd_kernel<NL_, true>), dim3(grid), dim3(64), 0,        \
                                    (hipStream_t)stream, x, r, theta2, w, hidden, fs, zstash, gy, gb, ain, lrin,  \
                                    (double*)ws, B, T);                                                        \
        else hipLaunchKernelGGL((wdf::clipper_mlp_row_bwd_kernel<NL_, false>), dim3(grid), dim3(64), 0,           \
                                (hipStream_t)stream, x, r, theta2, w, hidden, fs, zstash, gy, gb, ain, lrin,      \
                                (double*)ws, B, T);                                                            \
    }
        WDF_ROW_BWD(3) WDF_ROW_BWD(4) WDF_ROW_BWD(5)
#undef WDF_ROW_BWD
    }
    rc = check_launch("wdf_clipper_mlp_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(wdf::clipper_mlp_grad_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       (const double*)ws, (int)grid, theta2, fs, dyn ? 1 : 0, gtheta2);
    return check_launch("wdf_clipper_mlp_grad_reduce");
}

static unsigned mlp_wgrad_blocks(int64_t S)
{
    const int64_t want = (S + 63) / 64;
    return (unsigned)(want < 2048 ? want : 2048);
}

int64_t wdf_clipper_mlp_wgrad_ws_bytes(int hidden, int n_tanh_layers, int64_t S)
{
    const int64_t count = wdf_mlp_weight_count(hidden, n_tanh_layers);
    if (count <= 0 || S <= 0 || !mlp_arch_ok(hidden, n_tanh_layers)) return 0;
    return (int64_t)mlp_wgrad_blocks(S) * count * (int64_t)sizeof(float);
}

#define WDF_WGRAD_CASE(H_, NL_)                                                                               \
    if (hidden == H_ && n_tanh_layers == NL_)                                                                 \
        hipLaunchKernelGGL((wdf::mlp_wgrad_kernel<H_, NL_>), dim3(nblk, wdf::Mlp<H_, NL_>::kParts), dim3(64), 0, (hipStream_t)stream, ain,  \
                           lrin, gb, theta2, w, fs, (float*)ws, S);

int wdf_clipper_mlp_wgrad(const float* ain, const float* lrin, const float* gb, const float* theta2, const float* w,
                          int hidden, int n_tanh_layers, float fs, void* ws, float* gw, int64_t S, void* stream)
{
    if (!ain || !gb || !w || !ws || !gw) return fail(WDF_EINVAL, "null ain/gb/w/ws/gw");
    if (!lrin && !theta2) return fail(WDF_EINVAL, "theta2 is needed when lrin is NULL");
    if (S <= 0) return fail(WDF_EINVAL, "S must be positive");
    const int count = wdf_mlp_weight_count(hidden, n_tanh_layers);
    if (!mlp_arch_ok(hidden, n_tanh_layers))
        return fail(WDF_EUNSUPPORTED, "MLP root: unsupported network (width %d, %d tanh layers)", hidden, n_tanh_layers);
    const unsigned nblk = mlp_wgrad_blocks(S);
    WDF_WGRAD_CASE(4, 3) WDF_WGRAD_CASE(8, 3) WDF_WGRAD_CASE(16, 3) WDF_WGRAD_CASE(4, 4) WDF_WGRAD_CASE(8, 4)
    WDF_WGRAD_CASE(4, 5) WDF_WGRAD_CASE(8, 5)
    int rc = check_launch("wdf_clipper_mlp_wgrad");
    if (rc) return rc;
    hipLaunchKernelGGL(wdf::mlp_wgrad_reduce_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)ws, (int)nblk, count, gw);
    return check_launch("wdf_clipper_mlp_wgrad_reduce");
}

#define WDF_EVAL_CASE(H_, NL_)                                                                                \
    if (hidden == H_ && n_tanh_layers == NL_)                                                                 \
        hipLaunchKernelGGL((wdf::mlp_eval_kernel<H_, NL_>), dim3(nblk), dim3(64), 0, (hipStream_t)stream, ain, lrin, w, \
                           out, S);

int wdf_mlp_eval(const float* ain, const float* lrin, const float* w, int hidden, int n_tanh_layers, float* out,
                 int64_t S, void* stream)
{
    if (!ain || !lrin || !w || !out) return fail(WDF_EINVAL, "null ain/lrin/w/out");
    if (S <= 0) return fail(WDF_EINVAL, "S must be positive");
    if (!mlp_arch_ok(hidden, n_tanh_layers))
        return fail(WDF_EUNSUPPORTED, "MLP root: unsupported network (width %d, %d tanh layers)", hidden, n_tanh_layers);
    const unsigned nblk = mlp_wgrad_blocks(S);
    WDF_EVAL_CASE(4, 3) WDF_EVAL_CASE(8, 3) WDF_EVAL_CASE(16, 3) WDF_EVAL_CASE(4, 4) WDF_EVAL_CASE(8, 4)
    WDF_EVAL_CASE(4, 5) WDF_EVAL_CASE(8, 5)
    return check_launch("wdf_mlp_eval");
}

#define WDF_FIT_CASE(H_, NL_)                                                                                 \
    if (hidden == H_ && n_tanh_layers == NL_)                                                                 \
        hipLaunchKernelGGL((wdf::mlp_fit_epoch_kernel<H_, NL_>), dim3(1), dim3(64 * wdf::Mlp<H_, NL_>::kParts), 0,   \
                           (hipStream_t)stream, xa, xl, ys, S, batch, w, m, v, step, lr, beta1, beta2, eps, esr_n,   \
                           eps_energy, loss_sum);

int wdf_mlp_fit_epoch(const float* xa, const float* xl, const float* ys, int64_t S, int batch, float* w, float* m,
                      float* v, int32_t* step, float lr, float beta1, float beta2, float eps, float esr_n,
                      float eps_energy, double* loss_sum, int hidden, int n_tanh_layers, void* stream)
{
    if (!xa || !xl || !ys || !w || !m || !v || !step || !loss_sum) return fail(WDF_EINVAL, "null argument");
    if (S <= 0) return fail(WDF_EINVAL, "S must be positive");
    if (batch < 1 || batch > 64) return fail(WDF_EUNSUPPORTED, "wdf_mlp_fit_epoch: batch must be in 1..64 (got %d)", batch);
    if (!(esr_n > 0.0f)) return fail(WDF_EINVAL, "esr_n must be positive");
    if (!mlp_arch_ok(hidden, n_tanh_layers))
        return fail(WDF_EUNSUPPORTED, "MLP root: unsupported network (width %d, %d tanh layers)", hidden, n_tanh_layers);
    WDF_FIT_CASE(4, 3) WDF_FIT_CASE(8, 3) WDF_FIT_CASE(16, 3) WDF_FIT_CASE(4, 4) WDF_FIT_CASE(8, 4)
    WDF_FIT_CASE(4, 5) WDF_FIT_CASE(8, 5)
    return check_launch("wdf_mlp_fit_epoch");
}

int wdf_omega_f32(const float* x, float* w, int32_t* iters, int64_t n, void* stream)
{
    if (!x || !w || n <= 0) return fail(WDF_EINVAL, "wdf_omega_f32: bad arguments");
    hipLaunchKernelGGL(wdf::omega_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w,
                       iters, n);
    return check_launch("wdf_omega_f32");
}

int wdf_diode_pair_f32(const float* a, const float* R_port, float Is, float nVt, int n_up, int n_down, float* b,
                       int64_t n, void* stream)
{
    if (!a || !R_port || !b || n <= 0) return fail(WDF_EINVAL, "wdf_diode_pair_f32: bad arguments");
    if (n_up < 1 || n_down < 1) return fail(WDF_EINVAL, "n_up/n_down must be >= 1");
    hipLaunchKernelGGL(wdf::diode_pair_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a,
                       R_port, Is, nVt, n_up, n_down, b, n);
    return check_launch("wdf_diode_pair_f32");
}

int wdf_adam_step(float* theta, const float* grad, float* m, float* v, int32_t* step, const float* lr, float beta1,
                  float beta2, float eps, const float* lo, const float* hi, int n, void* stream)
{
    if (!theta || !grad || !m || !v || !step || !lr) return fail(WDF_EINVAL, "null theta/grad/m/v/step/lr");
    if (n <= 0 || n > 1024) return fail(WDF_EINVAL, "wdf_adam_step: n must be in 1..1024 (got %d)", n);
    hipLaunchKernelGGL(wdf::adam_clip_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, theta, grad, m, v, step, lr,
                       beta1, beta2, eps, lo, hi, n);
    return check_launch("wdf_adam_step");
}

void* wdf_event_create(void)
{
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return (void*)e;
}

int wdf_event_record(void* ev, void* stream)
{
    const hipError_t e = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
    return e == hipSuccess ? WDF_OK : fail(WDF_ELAUNCH, "hipEventRecord: %s", hipGetErrorString(e));
}

int wdf_event_elapsed_ms(void* start, void* stop, float* ms)
{
    hipError_t e = hipEventSynchronize((hipEvent_t)stop);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    return e == hipSuccess ? WDF_OK : fail(WDF_ELAUNCH, "hipEventElapsedTime: %s", hipGetErrorString(e));
}

void wdf_event_bracket_next(void* start, void* stop)
{
    g_ev0 = (hipEvent_t)start;
    g_ev1 = (hipEvent_t)stop;
}

void wdf_event_destroy(void* ev)
{
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}

}  // extern "C"
