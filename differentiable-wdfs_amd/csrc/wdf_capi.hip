// wdf_capi.hip -- the C ABI of libwdf_hip.so (include/wdf_hip.h).  gfx950 only.
//
// Argument checking, template dispatch and launches; no algorithm lives here.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/wdf_hip.h"
#include "wdf_clipper.h"

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

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <bool DYN_R, bool SYM, bool TM, bool V4>
void launch_fwd(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down, float* y,
                float* zstash, const float* z0, float* zT, int64_t B, int64_t T, hipStream_t s)
{
    const unsigned grid = (unsigned)((B + 63) / 64);
    if (zstash)
        hipLaunchKernelGGL((wdf::clipper_fwd_kernel<DYN_R, SYM, TM, V4, true>), dim3(grid), dim3(64), 0, s, x, r, theta,
                           fs, n_up, n_down, y, zstash, z0, zT, B, T);
    else
        hipLaunchKernelGGL((wdf::clipper_fwd_kernel<DYN_R, SYM, TM, V4, false>), dim3(grid), dim3(64), 0, s, x, r,
                           theta, fs, n_up, n_down, y, zstash, z0, zT, B, T);
}

template <bool DYN_R, bool SYM, bool TM, bool V4>
void launch_bwd(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                const float* zstash, const float* gy, double* ws, float* gz0, int64_t B, int64_t T, hipStream_t s)
{
    const unsigned grid = (unsigned)((B + 63) / 64);
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
    if (flags & ~(WDF_X_TIME_MAJOR | WDF_PREC_F64)) return fail(WDF_EINVAL, "unknown flag bits 0x%x", flags);
    if (flags & WDF_PREC_F64) return fail(WDF_EUNSUPPORTED, "WDF_PREC_F64 is not available for the Wright-omega clipper");
    return WDF_OK;
}

}  // namespace

extern "C" {

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
                  B, T, (hipStream_t)stream);
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
                       (const double*)ws, nparts, theta, fs, r != nullptr ? 1 : 0, gtheta, accumulate);
    return check_launch("wdf_clipper_grad_reduce");
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

void wdf_event_destroy(void* ev)
{
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}

}  // extern "C"
