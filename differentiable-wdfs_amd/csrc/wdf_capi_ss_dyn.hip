// wdf_capi_ss_dyn.hip -- C ABI part 7 of 7: the state-space recursion with per-sample coefficient rows and the MLP root on
// any small tree (csrc/wdf_ss_dyn.h): argument checking, template dispatch, launches.
#include "wdf_capi_common.h"
#include "wdf_ss_dyn.h"
using namespace wdfcapi;

namespace {

bool dyn_ok(int ns, int ni) { return ns >= 0 && ns <= wdf::kDynMaxS && ni >= 1 && ni <= wdf::kDynMaxI; }

bool dyn_mlp_ok(int h, int nl) { return (nl == 3 && (h == 4 || h == 8 || h == 16)) || (nl == 5 && (h == 4 || h == 8)); }

int dyn_check(const char* who, const float* x, const float* crow, int ns, int ni, int root, const float* rootp, const float* w,
              int hidden, int n_tanh_layers, int n_up, int n_down, int64_t B, int64_t T, int per_sample)
{
    if (!x || !crow) return fail(WDF_EINVAL, "%s: null x / rows", who);
    if (B <= 0 || T <= 0) return fail(WDF_EINVAL, "%s: B and T must be positive", who);
    if (!dyn_ok(ns, ni)) return fail(WDF_EUNSUPPORTED, "%s: ns <= %d, 1 <= ni <= %d (got %d, %d)", who, wdf::kDynMaxS, wdf::kDynMaxI, ns, ni);
    if (per_sample != 0 && per_sample != 1) return fail(WDF_EINVAL, "%s: per_sample is 0 or 1", who);
    if (root == WDF_ROOT_DIODE_PAIR) {
        if (!rootp) return fail(WDF_EINVAL, "%s: diode root needs rootp = {Is, nVt}", who);
        if (n_up < 1 || n_down < 1) return fail(WDF_EINVAL, "%s: n_up, n_down >= 1", who);
    } else if (root == WDF_ROOT_MLP) {
        if (!w) return fail(WDF_EINVAL, "%s: MLP root needs the flat weights", who);
        if (!dyn_mlp_ok(hidden, n_tanh_layers))
            return fail(WDF_EUNSUPPORTED, "%s: MLP root on a generic tree: width 4 / 8 / 16 with 3 tanh layers, 4 / 8 with 5 (got %d, %d)",
                        who, hidden, n_tanh_layers);
    } else if (root != WDF_ROOT_NONE) {
        return fail(WDF_EINVAL, "%s: unknown root kind %d", who, root);
    }
    return WDF_OK;
}

}  // namespace

extern "C" {

int wdf_ss_dyn_row_len(int ns, int ni) { return dyn_ok(ns, ni) ? wdf::DynLayout(ns, ni).n : 0; }

#define WDF_DYN_DISPATCH(KERNEL, ...)                                                                                          \
    do {                                                                                                                     \
        if (root == WDF_ROOT_NONE) hipLaunchKernelGGL((wdf::KERNEL<wdf::kDynRootNone, true, 4, 3>), grid, dim3(64), 0, s, __VA_ARGS__); \
        else if (root == WDF_ROOT_DIODE_PAIR && n_up == n_down)                                                              \
            hipLaunchKernelGGL((wdf::KERNEL<wdf::kDynRootDiode, true, 4, 3>), grid, dim3(64), 0, s, __VA_ARGS__);             \
        else if (root == WDF_ROOT_DIODE_PAIR)                                                                                \
            hipLaunchKernelGGL((wdf::KERNEL<wdf::kDynRootDiode, false, 4, 3>), grid, dim3(64), 0, s, __VA_ARGS__);            \
        else if (hidden == 4 && n_tanh_layers == 3) hipLaunchKernelGGL((wdf::KERNEL<wdf::kDynRootMlp, true, 4, 3>), grid, dim3(64), 0, s, __VA_ARGS__); \
        else if (hidden == 8 && n_tanh_layers == 3) hipLaunchKernelGGL((wdf::KERNEL<wdf::kDynRootMlp, true, 8, 3>), grid, dim3(64), 0, s, __VA_ARGS__); \
        else if (hidden == 16 && n_tanh_layers == 3) hipLaunchKernelGGL((wdf::KERNEL<wdf::kDynRootMlp, true, 16, 3>), grid, dim3(64), 0, s, __VA_ARGS__); \
        else if (hidden == 4 && n_tanh_layers == 5) hipLaunchKernelGGL((wdf::KERNEL<wdf::kDynRootMlp, true, 4, 5>), grid, dim3(64), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((wdf::KERNEL<wdf::kDynRootMlp, true, 8, 5>), grid, dim3(64), 0, s, __VA_ARGS__);               \
    } while (0)

int wdf_ss_dyn_fwd(const float* x, const float* rows, int per_sample, int ns, int ni, int root, const float* rootp, const float* w,
                   int hidden, int n_tanh_layers, int n_up, int n_down, float* y, float* zstash, const float* z0, float* zT,
                   int64_t B, int64_t T, void* stream)
{
    int rc = dyn_check("wdf_ss_dyn_fwd", x, rows, ns, ni, root, rootp, w, hidden, n_tanh_layers, n_up, n_down, B, T, per_sample);
    if (rc) return rc;
    if (!y) return fail(WDF_EINVAL, "wdf_ss_dyn_fwd: null y");
    const int64_t n = wdf::DynLayout(ns, ni).n;
    const int64_t cs = per_sample ? B : 1, ts = per_sample ? n * B : 0, bs = per_sample ? 1 : 0;
    const dim3 grid((unsigned)((B + 63) / 64));
    hipStream_t s = (hipStream_t)stream;
    EventBracket bracket(s);
    WDF_DYN_DISPATCH(ss_dyn_fwd_kernel, x, rows, cs, ts, bs, rootp, w, n_up, n_down, y, zstash, z0, zT, ns, ni, B, T);
    return check_launch("wdf_ss_dyn_fwd");
}

size_t wdf_ss_dyn_bwd_ws_bytes(int64_t B) { return B > 0 ? (size_t)((B + 63) / 64) * 2 * sizeof(double) : 0; }

int wdf_ss_dyn_bwd(const float* x, const float* rows, int per_sample, int ns, int ni, int root, const float* rootp, const float* w,
                   int hidden, int n_tanh_layers, int n_up, int n_down, const float* zstash, const float* gy, float* grows,
                   void* ws, float* gb, float* ain, float* lrin, float* gz0, int64_t B, int64_t T, void* stream)
{
    int rc = dyn_check("wdf_ss_dyn_bwd", x, rows, ns, ni, root, rootp, w, hidden, n_tanh_layers, n_up, n_down, B, T, per_sample);
    if (rc) return rc;
    if (!gy || !grows || !ws || (ns > 0 && !zstash)) return fail(WDF_EINVAL, "wdf_ss_dyn_bwd: null gy / grows / ws / zstash");
    if (root == WDF_ROOT_MLP && (!gb || !ain || !lrin)) return fail(WDF_EINVAL, "wdf_ss_dyn_bwd: the MLP root needs gb / ain / lrin [T][B]");
    const int64_t n = wdf::DynLayout(ns, ni).n;
    const int64_t cs = per_sample ? B : 1, ts = per_sample ? n * B : 0, bs = per_sample ? 1 : 0;
    const dim3 grid((unsigned)((B + 63) / 64));
    hipStream_t s = (hipStream_t)stream;
    EventBracket bracket(s);
    WDF_DYN_DISPATCH(ss_dyn_bwd_kernel, x, rows, cs, ts, bs, rootp, w, n_up, n_down, zstash, gy, grows, (double*)ws, gb, ain, lrin,
                     gz0, ns, ni, B, T);
    return check_launch("wdf_ss_dyn_bwd");
}

}  // extern "C"
