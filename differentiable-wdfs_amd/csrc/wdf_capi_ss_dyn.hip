// wdf_capi_ss_dyn.hip -- C ABI part 7 of 7: the state-space recursion with per-sample coefficient rows and the MLP root on
// any small tree (csrc/wdf_ss_dyn.h): argument checking, template dispatch, launches.
#include "wdf_capi_common.h"
#include "wdf_ss_dyn.h"
#include "wdf_ss_dyn_rows.h"
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
    if (per_sample < 0 || per_sample > 2) return fail(WDF_EINVAL, "%s: per_sample is 0 (one static row), 1 (a row per sample) or 2 (a row per sequence)", who);
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

// grid.x: one lane per sequence, or -- the network root, evaluated in 16-lane rows (wdf_ss_dyn.h DynLanes) -- four sequences per wave
#define WDF_DYN_GRIDX(ROW_) ((unsigned)((ROW_) ? (B + 3) / 4 : (B + 63) / 64))

// MS_: state slots the kernel is compiled for (4: trees of up to four capacitors; 8: five to eight)
#define WDF_DYN_FWD_MS(MS_, GY_, ...)                                                                                         \
    do {                                                                                                                     \
        const dim3 grid(WDF_DYN_GRIDX(root == WDF_ROOT_MLP), (unsigned)(GY_));                                               \
        if (root == WDF_ROOT_NONE) hipLaunchKernelGGL((wdf::ss_dyn_fwd_kernel<wdf::kDynRootNone, true, 16, 3, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden); \
        else if (root == WDF_ROOT_DIODE_PAIR && n_up == n_down)                                                              \
            hipLaunchKernelGGL((wdf::ss_dyn_fwd_kernel<wdf::kDynRootDiode, true, 16, 3, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden);   \
        else if (root == WDF_ROOT_DIODE_PAIR)                                                                                \
            hipLaunchKernelGGL((wdf::ss_dyn_fwd_kernel<wdf::kDynRootDiode, false, 16, 3, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden);  \
        else if (n_tanh_layers == 3) hipLaunchKernelGGL((wdf::ss_dyn_fwd_kernel<wdf::kDynRootMlp, true, 16, 3, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden); \
        else hipLaunchKernelGGL((wdf::ss_dyn_fwd_kernel<wdf::kDynRootMlp, true, 16, 5, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden);     \
    } while (0)
#define WDF_DYN_FWD(GY_, ...) do { if (ns > 4) WDF_DYN_FWD_MS(8, GY_, __VA_ARGS__); else WDF_DYN_FWD_MS(4, GY_, __VA_ARGS__); } while (0)

// the reverse sweep's kernel in one of its modes (0: sequential, 1: chunk maps + root partials; 2 -- no root code in it -- below)
#define WDF_DYN_BWD_MS(MS_, MODE_, GY_, ...)                                                                                  \
    do {                                                                                                                     \
        const dim3 grid(WDF_DYN_GRIDX(root == WDF_ROOT_MLP), (unsigned)(GY_));                                               \
        if (root == WDF_ROOT_NONE) hipLaunchKernelGGL((wdf::ss_dyn_bwd_kernel<wdf::kDynRootNone, true, 16, 3, MODE_, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden, acc); \
        else if (root == WDF_ROOT_DIODE_PAIR && n_up == n_down)                                                              \
            hipLaunchKernelGGL((wdf::ss_dyn_bwd_kernel<wdf::kDynRootDiode, true, 16, 3, MODE_, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden, acc);   \
        else if (root == WDF_ROOT_DIODE_PAIR)                                                                                \
            hipLaunchKernelGGL((wdf::ss_dyn_bwd_kernel<wdf::kDynRootDiode, false, 16, 3, MODE_, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden, acc);  \
        else if (n_tanh_layers == 3) hipLaunchKernelGGL((wdf::ss_dyn_bwd_kernel<wdf::kDynRootMlp, true, 16, 3, MODE_, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden, acc); \
        else hipLaunchKernelGGL((wdf::ss_dyn_bwd_kernel<wdf::kDynRootMlp, true, 16, 5, MODE_, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden, acc);     \
    } while (0)
#define WDF_DYN_BWD(MODE_, GY_, ...) do { if (ns > 4) WDF_DYN_BWD_MS(8, MODE_, GY_, __VA_ARGS__); else WDF_DYN_BWD_MS(4, MODE_, GY_, __VA_ARGS__); } while (0)

#define WDF_DYN_BWD_EMIT_MS(MS_, GY_, ...)                                                                                    \
    do {                                                                                                                     \
        const dim3 grid(WDF_DYN_GRIDX(false), (unsigned)(GY_));                                                              \
        if (root == WDF_ROOT_NONE) hipLaunchKernelGGL((wdf::ss_dyn_bwd_kernel<wdf::kDynRootNone, true, 16, 3, 2, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden, acc); \
        else if (root == WDF_ROOT_DIODE_PAIR) hipLaunchKernelGGL((wdf::ss_dyn_bwd_kernel<wdf::kDynRootDiode, true, 16, 3, 2, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden, acc); \
        else hipLaunchKernelGGL((wdf::ss_dyn_bwd_kernel<wdf::kDynRootMlp, true, 16, 3, 2, MS_>), grid, dim3(64), 0, s, __VA_ARGS__, hidden, acc);          \
    } while (0)
#define WDF_DYN_BWD_EMIT(GY_, ...) do { if (ns > 4) WDF_DYN_BWD_EMIT_MS(8, GY_, __VA_ARGS__); else WDF_DYN_BWD_EMIT_MS(4, GY_, __VA_ARGS__); } while (0)

namespace {
// chunk length (a multiple of 8) and count for n_chunks requested; false when the count does not tile T that way
bool dyn_geom(int64_t T, int n_chunks, int64_t& L, int& K)
{
    if (n_chunks < 1) return false;
    L = (T + n_chunks - 1) / n_chunks;
    L = (L + 7) / 8 * 8;
    K = (int)((T + L - 1) / L);
    return K == n_chunks;
}
}  // namespace

int wdf_ss_dyn_fwd(const float* x, const float* rows, int per_sample, int ns, int ni, int root, const float* rootp, const float* w,
                   int hidden, int n_tanh_layers, int n_up, int n_down, float* y, float* zstash, const float* z0, float* zT,
                   int64_t B, int64_t T, void* stream)
{
    int rc = dyn_check("wdf_ss_dyn_fwd", x, rows, ns, ni, root, rootp, w, hidden, n_tanh_layers, n_up, n_down, B, T, per_sample);
    if (rc) return rc;
    if (!y) return fail(WDF_EINVAL, "wdf_ss_dyn_fwd: null y");
    const int64_t n = wdf::DynLayout(ns, ni).n;
    const int64_t cs = per_sample ? B : 1, ts = per_sample == 1 ? n * B : 0, bs = per_sample ? 1 : 0;   // (2: rows [n][B], the same row at every step)
    hipStream_t s = (hipStream_t)stream;
    EventBracket bracket(s);
    WDF_DYN_FWD(1, x, rows, cs, ts, bs, rootp, w, n_up, n_down, y, zstash, z0, zT, ns, ni, B, T, T, (int64_t)0,
                (float*)nullptr, (float*)nullptr, (const unsigned*)nullptr, (const float*)nullptr);
    return check_launch("wdf_ss_dyn_fwd");
}

size_t wdf_ss_dyn_fwd_tp_ws_bytes(int ns, int64_t B, int n_chunks)
{
    if (ns < 0 || B <= 0 || n_chunks <= 0) return 0;
    return (size_t)2 * (size_t)n_chunks * (size_t)(ns > 0 ? ns : 1) * (size_t)B * sizeof(float) + (size_t)((B + 63) / 64) * sizeof(unsigned);
}

int wdf_ss_dyn_fwd_tp(const float* x, const float* rows, int per_sample, int ns, int ni, int root, const float* rootp, const float* w,
                      int hidden, int n_tanh_layers, int n_up, int n_down, float* y, float* zstash, const float* z0, float* zT,
                      int64_t B, int64_t T, int n_chunks, int warmup, float tol, const float* zinit, void* ws, void* status, void* stream)
{
    int rc = dyn_check("wdf_ss_dyn_fwd_tp", x, rows, ns, ni, root, rootp, w, hidden, n_tanh_layers, n_up, n_down, B, T, per_sample);
    if (rc) return rc;
    if (!y || !ws || !status) return fail(WDF_EINVAL, "wdf_ss_dyn_fwd_tp: null y / ws / status");
    if (ns < 1) return fail(WDF_EINVAL, "wdf_ss_dyn_fwd_tp: a tree without states has no chunks to verify: use wdf_ss_dyn_fwd");
    if (warmup < 0 || !(tol >= 0.0f)) return fail(WDF_EINVAL, "wdf_ss_dyn_fwd_tp: warmup >= 0, tol >= 0");
    int64_t Lc;
    int K;
    if (!dyn_geom(T, n_chunks, Lc, K)) return fail(WDF_EINVAL, "wdf_ss_dyn_fwd_tp: n_chunks = %d does not tile T = %lld in 8-step units (%d does)", n_chunks, (long long)T, K);
    const int64_t n = wdf::DynLayout(ns, ni).n;
    const int64_t cs = per_sample ? B : 1, ts = per_sample == 1 ? n * B : 0, bs = per_sample ? 1 : 0;   // (2: rows [n][B], the same row at every step)
    float* zwarm = (float*)ws;
    float* zend = zwarm + (size_t)K * (size_t)ns * (size_t)B;
    unsigned* gate = (unsigned*)(zend + (size_t)K * (size_t)ns * (size_t)B);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(status, 0, sizeof(wdf::SsTpStatus), s) != hipSuccess) return fail(WDF_ELAUNCH, "wdf_ss_dyn_fwd_tp: memset failed");
    {
        EventBracket bracket(s);
        WDF_DYN_FWD(K, x, rows, cs, ts, bs, rootp, w, n_up, n_down, y, zstash, z0, zT, ns, ni, B, T, Lc,
                    (int64_t)warmup, zwarm, zend, (const unsigned*)nullptr, zinit);
    }
    if (K > 1) {
        hipLaunchKernelGGL(wdf::ss_tp_verify_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, (const float*)zwarm, (const float*)zend, ns,
                           B, (int64_t)K, tol, gate, (wdf::SsTpStatus*)status);
        // the 64-sequence groups with a miss again, sequentially (the gate lets the others leave at once)
        WDF_DYN_FWD(1, x, rows, cs, ts, bs, rootp, w, n_up, n_down, y, zstash, z0, zT, ns, ni, B, T, T, (int64_t)0,
                    (float*)nullptr, (float*)nullptr, (const unsigned*)gate, (const float*)nullptr);
    }
    return check_launch("wdf_ss_dyn_fwd_tp");
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
    const int64_t cs = per_sample ? B : 1, ts = per_sample == 1 ? n * B : 0, bs = per_sample ? 1 : 0;   // (2: rows [n][B], the same row at every step)
    hipStream_t s = (hipStream_t)stream;
    const int acc = per_sample != 1;       // rows that do not change in time: dL/d(row) summed over the steps -> grows [1][n][B]
    EventBracket bracket(s);
    WDF_DYN_BWD(0, 1, x, rows, cs, ts, bs, rootp, w, n_up, n_down, zstash, gy, grows, (double*)ws, gb, ain, lrin, gz0, ns, ni, B, T, T,
                (float*)nullptr, (float*)nullptr, (const float*)nullptr);
    return check_launch("wdf_ss_dyn_bwd");
}

// ws: double [K waves][2] partial sums | rec float [K][(ns + 1) ns][B] | lam_in float [K][ns][B] | rpart float [T][5][B]
size_t wdf_ss_dyn_bwd_tp_ws_bytes(int ns, int64_t B, int64_t T, int n_chunks)
{
    if (ns < 0 || B <= 0 || T <= 0 || n_chunks <= 0) return 0;
    const size_t nsa = ns > 0 ? ns : 1, waves = (size_t)((B + 63) / 64);
    return (size_t)n_chunks * waves * 2 * sizeof(double) +
           ((size_t)n_chunks * (nsa + 1) * nsa + (size_t)n_chunks * nsa + (size_t)T * wdf::kDynRpart) * (size_t)B * sizeof(float);
}

int wdf_ss_dyn_bwd_tp(const float* x, const float* rows, int per_sample, int ns, int ni, int root, const float* rootp, const float* w,
                      int hidden, int n_tanh_layers, int n_up, int n_down, const float* zstash, const float* gy, float* grows,
                      void* ws, float* gb, float* ain, float* lrin, float* gz0, int64_t B, int64_t T, int n_chunks, void* stream)
{
    int rc = dyn_check("wdf_ss_dyn_bwd_tp", x, rows, ns, ni, root, rootp, w, hidden, n_tanh_layers, n_up, n_down, B, T, per_sample);
    if (rc) return rc;
    if (!gy || !grows || !ws || !zstash) return fail(WDF_EINVAL, "wdf_ss_dyn_bwd_tp: null gy / grows / ws / zstash");
    if (ns < 1) return fail(WDF_EINVAL, "wdf_ss_dyn_bwd_tp: a tree without states has no adjoint to chunk: use wdf_ss_dyn_bwd");
    if (root == WDF_ROOT_MLP && (!gb || !ain || !lrin)) return fail(WDF_EINVAL, "wdf_ss_dyn_bwd_tp: the MLP root needs gb / ain / lrin [T][B]");
    int64_t Lc;
    int K;
    if (!dyn_geom(T, n_chunks, Lc, K)) return fail(WDF_EINVAL, "wdf_ss_dyn_bwd_tp: n_chunks = %d does not tile T = %lld in 8-step units (%d does)", n_chunks, (long long)T, K);
    const int64_t n = wdf::DynLayout(ns, ni).n;
    const int64_t cs = per_sample ? B : 1, ts = per_sample == 1 ? n * B : 0, bs = per_sample ? 1 : 0;   // (2: rows [n][B], the same row at every step)
    const size_t waves = (size_t)((B + 63) / 64);
    double* part = (double*)ws;
    float* rec = (float*)(part + (size_t)K * waves * 2);
    float* lam_in = rec + (size_t)K * (size_t)(ns + 1) * (size_t)ns * (size_t)B;
    float* rpart = lam_in + (size_t)K * (size_t)ns * (size_t)B;
    hipStream_t s = (hipStream_t)stream;
    const int acc = per_sample != 1;       // (grows [K][n][B] then: one partial per chunk, added up by the caller)
    {
        EventBracket bracket(s);
        WDF_DYN_BWD(1, K, x, rows, cs, ts, bs, rootp, w, n_up, n_down, zstash, gy, grows, part, gb, ain, lrin, gz0, ns, ni, B, T, Lc, rpart, rec,
                    (const float*)nullptr);
    }
    hipLaunchKernelGGL(wdf::ss_dyn_bwd_combine_kernel, dim3((unsigned)waves), dim3(64), 0, s, (const float*)rec, lam_in, ns, B, (int64_t)K);
    WDF_DYN_BWD_EMIT(K, x, rows, cs, ts, bs, rootp, w, n_up, n_down, zstash, gy, grows, part, gb, ain, lrin, gz0, ns, ni, B, T, Lc, rpart, rec,
                     (const float*)lam_in);
    return check_launch("wdf_ss_dyn_bwd_tp");
}


/* ---- the coefficient rows from the probed step's tape (wdf_ss_dyn_rows.h) ---- */
namespace {
// checks the tape (host arrays) and packs it for the kernels' argument block
int rows_pack(const char* who, const int32_t* ops, int n_ops, const double* consts, int n_consts, const int32_t* outs, int n_out,
              int n_params, int chan, wdf::RowsTape& tp)
{
    if (!ops || !outs || (n_consts > 0 && !consts)) return fail(WDF_EINVAL, "%s: null tape", who);
    if (n_ops < 1 || n_ops > wdf::kRowsMaxOps || n_consts < 0 || n_consts > wdf::kRowsMaxConsts || n_out < 1 || n_out > wdf::kRowsMaxOut ||
        n_params < 0 || n_params > wdf::kRowsMaxParams)
        return fail(WDF_EUNSUPPORTED, "%s: the device evaluates tapes of <= %d operations, %d constants, %d row entries, %d component values "
                                      "(got %d, %d, %d, %d)", who, wdf::kRowsMaxOps, wdf::kRowsMaxConsts, wdf::kRowsMaxOut, wdf::kRowsMaxParams,
                    n_ops, n_consts, n_out, n_params);
    if (chan < -1 || chan >= n_params) return fail(WDF_EINVAL, "%s: chan = %d with %d component values", who, chan, n_params);
    for (int i = 0; i < n_ops; ++i) {
        const int op = ops[3 * i], a = ops[3 * i + 1], b = ops[3 * i + 2];
        bool ok;
        switch (op) {
        case wdf::kOpConst: ok = a >= 0 && a < n_consts; break;
        case wdf::kOpParam: ok = a >= 0 && a < n_params; break;
        case wdf::kOpAdd: case wdf::kOpSub: case wdf::kOpMul: case wdf::kOpDiv: ok = a >= 0 && a < i && b >= 0 && b < i; break;
        case wdf::kOpNeg: case wdf::kOpRecip: ok = a >= 0 && a < i; break;
        default: ok = false;
        }
        if (!ok) return fail(WDF_EINVAL, "%s: operation %d of the tape (%d, %d, %d) is not one the probe records", who, i, op, a, b);
        tp.code[i] = (uint32_t)op | ((uint32_t)a << 8) | ((uint32_t)(op >= wdf::kOpAdd && op <= wdf::kOpDiv ? b : 0) << 16);
    }
    for (int k = 0; k < n_out; ++k) {
        if (outs[k] < 0 || outs[k] >= n_ops) return fail(WDF_EINVAL, "%s: row entry %d names node %d of %d", who, k, outs[k], n_ops);
        tp.outs[k] = (uint8_t)outs[k];
    }
    for (int j = 0; j < n_consts; ++j) tp.consts[j] = consts[j];
    tp.n_ops = n_ops; tp.n_out = n_out; tp.n_params = n_params; tp.chan = chan;
    return WDF_OK;
}

int64_t rows_blocks_t(int64_t T) { return (T + wdf::kRowsSteps - 1) / wdf::kRowsSteps; }

// a launch with more than the default 64 KB of dynamic LDS asks for it first (per function and device: asked every such call)
int rows_lds_limit(const char* who, const void* kernel, size_t lds)
{
    if (lds <= 64 * 1024) return WDF_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return fail(WDF_ELAUNCH, "%s: %zu bytes of LDS for this tape could not be asked for", who, lds);
    return WDF_OK;
}

int rows_args(const char* who, const double* params, int n_params, int chan, const float* r, int64_t B, int64_t T)
{
    if (B <= 0 || T <= 0) return fail(WDF_EINVAL, "%s: B and T must be positive", who);
    if (rows_blocks_t(T) > 65535) return fail(WDF_EUNSUPPORTED, "%s: T <= %d", who, 65535 * wdf::kRowsSteps);
    if (n_params > 0 && !params) return fail(WDF_EINVAL, "%s: null params", who);
    if ((chan >= 0) != (r != nullptr)) return fail(WDF_EINVAL, "%s: a channel parameter (chan >= 0) and its values r [T][B] come together", who);
    return WDF_OK;
}
}  // namespace

int wdf_ss_dyn_rows(const int32_t* tape_ops, int n_ops, const double* consts, int n_consts, const int32_t* outs, int n_out,
                    const double* params, int n_params, int chan, const float* r, float* rows, int64_t B, int64_t T, void* stream)
{
    wdf::RowsTape tp{};
    int rc = rows_pack("wdf_ss_dyn_rows", tape_ops, n_ops, consts, n_consts, outs, n_out, n_params, chan, tp);
    if (rc) return rc;
    if ((rc = rows_args("wdf_ss_dyn_rows", params, n_params, chan, r, B, T))) return rc;
    if (!rows) return fail(WDF_EINVAL, "wdf_ss_dyn_rows: null rows");
    const dim3 grid((unsigned)((B + 63) / 64), (unsigned)rows_blocks_t(T));
    const size_t lds = (size_t)n_ops * 64 * sizeof(double);
    if ((rc = rows_lds_limit("wdf_ss_dyn_rows", (const void*)wdf::ss_dyn_rows_kernel, lds))) return rc;
    hipLaunchKernelGGL(wdf::ss_dyn_rows_kernel, grid, dim3(64), lds, (hipStream_t)stream, tp, params, r, rows, B, T);
    return check_launch("wdf_ss_dyn_rows");
}

size_t wdf_ss_dyn_rows_bwd_ws_bytes(int n_params, int64_t B, int64_t T)
{
    if (B <= 0 || T <= 0 || n_params < 1 || n_params > wdf::kRowsMaxParams) return 0;
    return (size_t)((B + 63) / 64) * (size_t)rows_blocks_t(T) * (size_t)n_params * sizeof(double);
}

int wdf_ss_dyn_rows_bwd(const int32_t* tape_ops, int n_ops, const double* consts, int n_consts, const int32_t* outs, int n_out,
                        const double* params, int n_params, int chan, const float* r, const float* grows, void* ws, double* gparams,
                        int64_t B, int64_t T, void* stream)
{
    wdf::RowsTape tp{};
    int rc = rows_pack("wdf_ss_dyn_rows_bwd", tape_ops, n_ops, consts, n_consts, outs, n_out, n_params, chan, tp);
    if (rc) return rc;
    if ((rc = rows_args("wdf_ss_dyn_rows_bwd", params, n_params, chan, r, B, T))) return rc;
    if (n_params < 1) return fail(WDF_EINVAL, "wdf_ss_dyn_rows_bwd: no component values to differentiate");
    if (!grows || !ws || !gparams) return fail(WDF_EINVAL, "wdf_ss_dyn_rows_bwd: null grows / ws / gparams");
    const dim3 grid((unsigned)((B + 63) / 64), (unsigned)rows_blocks_t(T));
    const size_t lds = (size_t)n_ops * 64 * (sizeof(double) + sizeof(float)) + (size_t)n_params * 64 * sizeof(double);
    if ((rc = rows_lds_limit("wdf_ss_dyn_rows_bwd", (const void*)wdf::ss_dyn_rows_bwd_kernel, lds))) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wdf::ss_dyn_rows_bwd_kernel, grid, dim3(64), lds, s, tp, params, r, grows, (double*)ws, B, T);
    rc = check_launch("wdf_ss_dyn_rows_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(wdf::ss_dyn_rows_reduce_kernel, dim3((unsigned)n_params), dim3(256), 0, s, (const double*)ws,
                       (int64_t)grid.x * (int64_t)grid.y, n_params, gparams);
    return check_launch("wdf_ss_dyn_rows_reduce");
}

}  // extern "C"
