// wdf_capi_mlp_step.hip -- C ABI part 5 of 6: the resident training step of the MLP-root pot clipper
// (csrc/wdf_mlp_step.h): state layout, plan upload, template dispatch and the five launches of a step.
#include <cstdlib>
#include <vector>

#include "wdf_capi_common.h"
#include "wdf_mlp_step.h"
using namespace wdfcapi;

namespace {

struct StepLayout {
    size_t ctl, items, cols, wcol, cool, wpeak, ticket, ticket2, nflag, colseq, flag, hwid, colmiss, zwarm, zend, zpre, lossblk, colsum,
        maps, snap, wsw, total;
    int n_cols, kw;
    int64_t lw;
};

size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

bool step_arch_ok(int hidden, int n_layers, int activation)
{
    return (hidden == 4 || hidden == 8 || hidden == 16) && n_layers >= 3 && n_layers <= 5 && (activation == 0 || activation == 1);
}

int step_weight_count(int H, int NL) { return 3 * H + (NL - 1) * (H * H + H) + H + 1; }

StepLayout step_layout(int hidden, int n_layers, int64_t B, int64_t T, int n_items, int wgrad_chunks)
{
    StepLayout L{};
    L.n_cols = (int)((B + 15) / 16);
    int64_t lw = (T + wgrad_chunks - 1) / wgrad_chunks;
    lw = (lw + 15) / 16 * 16;
    L.lw = lw;
    L.kw = (int)((T + lw - 1) / lw);
    const size_t nc = (size_t)L.n_cols, ni = (size_t)n_items, nb = (size_t)(T / 16);
    size_t o = 0;
    auto take = [&o](size_t bytes) { const size_t at = o; o = up(o + bytes, 256); return at; };
    L.ctl = take(sizeof(wdf::MlpStepCtl));
    L.items = take(ni * sizeof(wdf::MlpStepItem));
    L.cols = take(nc * sizeof(wdf::MlpStepCol));
    L.wcol = take(nc * 4); L.cool = take(nc * 4); L.wpeak = take(nc * 4); L.ticket = take(nc * 4); L.ticket2 = take(nc * 4);
    L.nflag = take(nc * 4); L.colseq = take(nc * 4); L.flag = take(ni * 4); L.hwid = take(ni * 8); L.colmiss = take(nc * 16);
    L.zwarm = take(ni * 16 * 4); L.zend = take(ni * 16 * 4);
    L.zpre = take(ni * wdf::kStepPre * 16 * 4);
    L.lossblk = take(nb * nc * 2 * 4); L.colsum = take(nc * 2 * 8);
    L.maps = take(nb * 3 * (size_t)B * 4);
    L.snap = take(2 * nb * (size_t)B * 4);
    L.wsw = take(nc * (size_t)L.kw * (size_t)step_weight_count(hidden, n_layers) * 4);
    L.total = o;
    return L;
}

int step_check_shape(int hidden, int n_layers, int activation, int64_t B, int64_t T, int n_items, int wgrad_chunks)
{
    if (!step_arch_ok(hidden, n_layers, activation))
        return fail(WDF_EUNSUPPORTED, "MLP-root training step: width in {4,8,16}, 3..5 hidden layers, activation 0 (tanh) or 1 (relu); "
                                      "got width %d, %d layers, activation %d", hidden, n_layers, activation);
    if (B <= 0 || T <= 0 || (T & 15)) return fail(WDF_EUNSUPPORTED, "MLP-root training step: T must be a positive multiple of 16 (got %lld)", (long long)T);
    if (n_items < (B + 15) / 16 || wgrad_chunks < 1) return fail(WDF_EINVAL, "n_items >= ceil(B/16), wgrad_chunks >= 1");
    return WDF_OK;
}

}  // namespace

extern "C" {

size_t wdf_clipper_mlp_step_state_bytes(int hidden, int n_layers, int64_t B, int64_t T, int n_items, int wgrad_chunks)
{
    if (step_check_shape(hidden, n_layers, 0, B, T, n_items, wgrad_chunks)) return 0;
    return step_layout(hidden, n_layers, B, T, n_items, wgrad_chunks).total;
}

// items: HOST int32[n_items][4] = {column, chunk index within the column, t0, t1}, sorted by column then time; every
// column's chunks tile [0, T) in multiples of 16.  Uploads the plan and leaves flags / tickets clean.  reset != 0 also
// resets the controller (warm-up units, snapshots, call count) -- the first call; reset == 0 re-plans a live state
// (the snapshots are by absolute time: they stay valid).
int wdf_clipper_mlp_step_plan(void* state, int hidden, int n_layers, int64_t B, int64_t T, int n_items, int wgrad_chunks,
                              const int32_t* items, int reset, int warm16, int cold16, int w_min, int w_max, float tol,
                              void* stream)
{
    int rc = step_check_shape(hidden, n_layers, 0, B, T, n_items, wgrad_chunks);
    if (rc) return rc;
    if (!state || !items) return fail(WDF_EINVAL, "null state/items");
    if (!(tol > 0.0f) || warm16 < 0 || cold16 < 0 || w_min < 0 || w_max < w_min) return fail(WDF_EINVAL, "tol > 0, warm-ups >= 0, w_min <= w_max");
    const StepLayout L = step_layout(hidden, n_layers, B, T, n_items, wgrad_chunks);
    // validate the plan and build the per-column index
    std::vector<wdf::MlpStepCol> cols((size_t)L.n_cols, wdf::MlpStepCol{-1, 0});
    int64_t expect_t = 0;
    int prev_col = -1;
    for (int i = 0; i < n_items; ++i) {
        const int col = items[4 * i], k = items[4 * i + 1], t0 = items[4 * i + 2], t1 = items[4 * i + 3];
        if (col < 0 || col >= L.n_cols || (t0 & 15) || (t1 & 15) || t1 <= t0 || t1 > T) return fail(WDF_EINVAL, "plan item %d is malformed", i);
        if (col != prev_col) {
            if (prev_col >= 0 && expect_t != T) return fail(WDF_EINVAL, "plan: column %d does not end at T", prev_col);
            if (col != prev_col + 1 || k != 0 || t0 != 0) return fail(WDF_EINVAL, "plan: columns must come in order, each from t = 0");
            cols[(size_t)col].first = i;
            prev_col = col;
        } else if (k != cols[(size_t)col].K || t0 != expect_t) {
            return fail(WDF_EINVAL, "plan item %d does not continue its column", i);
        }
        cols[(size_t)col].K += 1;
        expect_t = t1;
    }
    if (prev_col != L.n_cols - 1 || expect_t != T) return fail(WDF_EINVAL, "plan does not cover every column up to T");
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)state;
    hipError_t e = hipSuccess;
    auto ok = [&e](hipError_t r) { if (e == hipSuccess) e = r; };
    ok(hipMemcpyAsync(base + L.items, items, (size_t)n_items * sizeof(wdf::MlpStepItem), hipMemcpyHostToDevice, s));
    ok(hipMemcpyAsync(base + L.cols, cols.data(), cols.size() * sizeof(wdf::MlpStepCol), hipMemcpyHostToDevice, s));
    ok(hipMemsetAsync(base + L.wpeak, 0, L.flag + (size_t)n_items * 4 - L.wpeak, s));        // warm-up peaks, tickets, flags, gates
    if (reset) {
        wdf::MlpStepCtl c{};
        c.cold16 = cold16; c.w_min = w_min; c.w_max = w_max; c.slack = 2; c.cool_miss = 8; c.cool_shrink = 2;
        c.tol = tol; c.grow_at = 0.6f; c.shrink_at = 0.6f; c.repair_at = 1.0f;
        ok(hipMemcpyAsync(base + L.ctl, &c, sizeof(c), hipMemcpyHostToDevice, s));
        std::vector<int32_t> w0((size_t)L.n_cols, warm16 > w_max ? w_max : warm16);
        ok(hipMemcpyAsync(base + L.wcol, w0.data(), w0.size() * 4, hipMemcpyHostToDevice, s));
        ok(hipMemsetAsync(base + L.cool, 0, (size_t)L.n_cols * 4, s));
        ok(hipMemsetAsync(base + L.snap, 0, 2 * (size_t)(T / 16) * (size_t)B * 4, s));
    }
    ok(hipStreamSynchronize(s));                                   // (the host arrays above go out of scope)
    if (e != hipSuccess) return fail(WDF_ELAUNCH, "wdf_clipper_mlp_step_plan: %s", hipGetErrorString(e));
    return WDF_OK;
}

// Host copies of the controller: ctl_out int32[32] (MlpStepCtl), wcol_out int32[ceil(B/16)], wpeak_out int32[ceil(B/16)] (the
// largest warm-up each column has run with since the plan was installed), hwid_out int32[n_items][2]
// (HW_ID and XCC_ID of the wave that ran each forward item in the last call: placement diagnostics), colmiss_out
// float[ceil(B/16)][4] (per column: the last verification's arrival miss and the misses 16, 32, 48 steps before arrival);
// any may be NULL.
// Synchronises the stream.
int wdf_clipper_mlp_step_read(const void* state, int hidden, int n_layers, int64_t B, int64_t T, int n_items, int wgrad_chunks,
                              int32_t* ctl_out, int32_t* wcol_out, int32_t* wpeak_out, int32_t* hwid_out, float* colmiss_out, void* stream)
{
    int rc = step_check_shape(hidden, n_layers, 0, B, T, n_items, wgrad_chunks);
    if (rc) return rc;
    if (!state) return fail(WDF_EINVAL, "null state");
    const StepLayout L = step_layout(hidden, n_layers, B, T, n_items, wgrad_chunks);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (ctl_out) e = hipMemcpyAsync(ctl_out, (const char*)state + L.ctl, sizeof(wdf::MlpStepCtl), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && wcol_out) e = hipMemcpyAsync(wcol_out, (const char*)state + L.wcol, (size_t)L.n_cols * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && wpeak_out) e = hipMemcpyAsync(wpeak_out, (const char*)state + L.wpeak, (size_t)L.n_cols * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && hwid_out) e = hipMemcpyAsync(hwid_out, (const char*)state + L.hwid, (size_t)n_items * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && colmiss_out) e = hipMemcpyAsync(colmiss_out, (const char*)state + L.colmiss, (size_t)L.n_cols * 16, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(WDF_ELAUNCH, "wdf_clipper_mlp_step_read: %s", hipGetErrorString(e));
    return WDF_OK;
}

// Controller knobs of a live state (device-side writes on the stream, no synchronisation): field = index into
// MlpStepCtl as int32 words (6 slack, 7 cool_miss, 8 cool_shrink, 9 tol, 10 grow_at, 11 shrink_at, 12 freeze, 13 repair_at).
int wdf_clipper_mlp_step_set(void* state, int field, int32_t bits, void* stream)
{
    if (!state || field < 3 || field > 13) return fail(WDF_EINVAL, "wdf_clipper_mlp_step_set: field 3..13");
    const hipError_t e = hipMemcpyAsync((char*)state + 4 * (size_t)field, &bits, 4, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return fail(WDF_ELAUNCH, "wdf_clipper_mlp_step_set: %s", hipGetErrorString(e));
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? WDF_OK : fail(WDF_ELAUNCH, "wdf_clipper_mlp_step_set: sync");
}

int wdf_clipper_mlp_step_set_wcol(void* state, int hidden, int n_layers, int64_t B, int64_t T, int n_items, int wgrad_chunks,
                                  const int32_t* wcol, void* stream)
{
    int rc = step_check_shape(hidden, n_layers, 0, B, T, n_items, wgrad_chunks);
    if (rc) return rc;
    if (!state || !wcol) return fail(WDF_EINVAL, "null state/wcol");
    const StepLayout L = step_layout(hidden, n_layers, B, T, n_items, wgrad_chunks);
    hipError_t e = hipMemcpyAsync((char*)state + L.wcol, wcol, (size_t)L.n_cols * 4, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(WDF_ELAUNCH, "wdf_clipper_mlp_step_set_wcol: %s", hipGetErrorString(e));
    return WDF_OK;
}

int wdf_clipper_mlp_step_prepare(const float* r, const float* theta2, float fs, int64_t B, int64_t T, float* p, float* lr,
                                 void* stream)
{
    if (!r || !theta2 || !p || !lr) return fail(WDF_EINVAL, "null r/theta2/p/lr");
    if (B <= 0 || T <= 0 || !(fs > 0.0f)) return fail(WDF_EINVAL, "B, T, fs must be positive");
    const int64_t n = B * T;
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(wdf::mlp_step_prepare_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, r, theta2, fs, n, p, lr);
    return check_launch("wdf_clipper_mlp_step_prepare");
}

#define WDF_STEP_FWD_(NL_, DYN_, ACT_)                                                                            \
    {                                                                                                             \
        if (phase & WDF_MLP_STEP_FWD) {                                                                           \
            {                                                                                                     \
                EventBracket bracket(s);                                                                          \
                hipLaunchKernelGGL((wdf::mlp_step_fwd_kernel<NL_, DYN_, ACT_, 0>), dim3((unsigned)((n_items + 3) / 4)), dim3(256), 0, s, A); \
            }                                                                                                     \
            hipLaunchKernelGGL((wdf::mlp_step_fwd_kernel<NL_, DYN_, ACT_, 1>), dim3((unsigned)((n_items + 3) / 4)), dim3(256), 0, s, A);     \
            hipLaunchKernelGGL((wdf::mlp_step_fwd_kernel<NL_, DYN_, ACT_, 2>), dim3((unsigned)((L.n_cols + 3) / 4)), dim3(256), 0, s, A);    \
        }                                                                                                         \
        if (phase & WDF_MLP_STEP_SUMS)                                                                            \
            hipLaunchKernelGGL(wdf::mlp_step_sums_kernel, dim3(1), dim3(64), 0, s, (const double*)A.colsum, L.n_cols, sums);    \
        if (phase & WDF_MLP_STEP_BWD) {                                                                           \
            EventBracket bracket(s);                                                                              \
            if (wgrad_bs == 4)                                                                                    \
                hipLaunchKernelGGL((wdf::mlp_step_wgrad_kernel<NL_, DYN_, ACT_, 4>), dim3((unsigned)((L.n_cols * L.kw + 3) / 4)), dim3(256), 0, s, A, \
                                   gsums, n_global, eps_energy, L.lw, L.kw, wsw, gcoef, adam_m ? adam_step : (int32_t*)nullptr);      \
            else if (wgrad_bs == 16)                                                                              \
                hipLaunchKernelGGL((wdf::mlp_step_wgrad_kernel<NL_, DYN_, ACT_, 16>), dim3((unsigned)((L.n_cols * L.kw + 3) / 4)), dim3(256), 0, s, A, \
                                   gsums, n_global, eps_energy, L.lw, L.kw, wsw, gcoef, adam_m ? adam_step : (int32_t*)nullptr);      \
            else                                                                                                  \
                hipLaunchKernelGGL((wdf::mlp_step_wgrad_kernel<NL_, DYN_, ACT_, 8>), dim3((unsigned)((L.n_cols * L.kw + 3) / 4)), dim3(256), 0, s, A, \
                                   gsums, n_global, eps_energy, L.lw, L.kw, wsw, gcoef, adam_m ? adam_step : (int32_t*)nullptr);      \
        }                                                                                                         \
    }
#define WDF_STEP_FWD(NL_)                                                                                         \
    if (n_layers == NL_) {                                                                                        \
        if (dyn) { if (activation == 1) WDF_STEP_FWD_(NL_, true, 1) else WDF_STEP_FWD_(NL_, true, 0) }            \
        else { if (activation == 1) WDF_STEP_FWD_(NL_, false, 1) else WDF_STEP_FWD_(NL_, false, 0) }              \
    }

// One training step (or one phase of it) on the resident set.  x [B][T]; p, lr [B][T] from wdf_clipper_mlp_step_prepare
// (both NULL: static R from theta2); target, y, zstash, kappa [T][B]; w: the flat weights, UPDATED IN PLACE when
// adam_m != NULL (single rank: phase = FWD | BWD).  sums: double[2] device: written by WDF_MLP_STEP_SUMS; read by
// WDF_MLP_STEP_BWD when WDF_MLP_STEP_GLOBAL_SUMS is set (multi-rank: all-reduce it in between).  gw: float[count]
// (always written by BWD).  loss3: float[3] = {mse, esr, mse + esr} (BWD).  gcoef: float[2] or NULL.
int wdf_clipper_mlp_step(const float* x, const float* p, const float* lr, const float* theta2, float* w, int hidden, int n_layers,
                         int activation, float fs, const float* target, int64_t skip, double n_global, double eps_energy, float* y,
                         float* zstash, float* kappa, void* state, int64_t B, int64_t T, int n_items, int wgrad_chunks, int phase,
                         double* sums, float* gw, float* loss3, float* gcoef, float* adam_m, float* adam_v, int32_t* adam_step,
                         const float* adam_lr, float beta1, float beta2, float eps, void* stream)
{
    int rc = step_check_shape(hidden, n_layers, activation, B, T, n_items, wgrad_chunks);
    if (rc) return rc;
    if (!x || !w || !target || !y || !zstash || !kappa || !state) return fail(WDF_EINVAL, "null x/w/target/y/zstash/kappa/state");
    if ((p == nullptr) != (lr == nullptr)) return fail(WDF_EINVAL, "p and lr go together");
    if (!p && !theta2) return fail(WDF_EINVAL, "static R needs theta2");
    if (!(fs > 0.0f) || skip < 0 || !(n_global > 0.0)) return fail(WDF_EINVAL, "fs > 0, skip >= 0, n_global > 0");
    if (!aligned16(x) || (p && (!aligned16(p) || !aligned16(lr)))) return fail(WDF_EINVAL, "x, p, lr must be 16-byte aligned");
    if (!(phase & (WDF_MLP_STEP_FWD | WDF_MLP_STEP_SUMS | WDF_MLP_STEP_BWD))) return fail(WDF_EINVAL, "empty phase");
    if ((phase & WDF_MLP_STEP_SUMS) && !sums) return fail(WDF_EINVAL, "WDF_MLP_STEP_SUMS needs sums");
    if ((phase & WDF_MLP_STEP_GLOBAL_SUMS) && !sums) return fail(WDF_EINVAL, "WDF_MLP_STEP_GLOBAL_SUMS needs sums");
    if ((phase & WDF_MLP_STEP_BWD) && !gw) return fail(WDF_EINVAL, "WDF_MLP_STEP_BWD needs gw");
    if (adam_m && (!adam_v || !adam_step || !adam_lr)) return fail(WDF_EINVAL, "Adam needs m, v, step, lr");
    const StepLayout L = step_layout(hidden, n_layers, B, T, n_items, wgrad_chunks);
    char* base = (char*)state;
    wdf::MlpStepArgs A{};
    A.x = x; A.p = p; A.lr = lr; A.theta2 = theta2; A.w = w; A.target = target;
    A.y = y; A.zstash = zstash; A.kappa = kappa;
    A.maps = (float*)(base + L.maps); A.snap = (float*)(base + L.snap);
    A.zwarm = (float*)(base + L.zwarm); A.zend = (float*)(base + L.zend);
    A.zpre = (float*)(base + L.zpre);
    A.lossblk = (float*)(base + L.lossblk); A.colsum = (double*)(base + L.colsum);
    A.items = (const wdf::MlpStepItem*)(base + L.items); A.cols = (const wdf::MlpStepCol*)(base + L.cols);
    A.wcol = (int*)(base + L.wcol); A.cool = (int*)(base + L.cool); A.wpeak = (int*)(base + L.wpeak);
    A.ticket = (unsigned*)(base + L.ticket); A.ticket2 = (unsigned*)(base + L.ticket2);
    A.hwid = (unsigned*)(base + L.hwid); A.colmiss = (float*)(base + L.colmiss);
    A.flag = (unsigned*)(base + L.flag); A.nflag = (unsigned*)(base + L.nflag); A.colseq = (unsigned*)(base + L.colseq);
    A.ctl = (wdf::MlpStepCtl*)(base + L.ctl);
    A.B = B; A.T = T; A.skip = skip; A.H = hidden; A.n_items = n_items; A.n_cols = L.n_cols; A.fs = fs;
    float* wsw = (float*)(base + L.wsw);
    const double* gsums = (phase & WDF_MLP_STEP_GLOBAL_SUMS) ? sums : nullptr;
    const bool dyn = p != nullptr;
    hipStream_t s = (hipStream_t)stream;
    // steps staged per block of loads in the reverse sweep (registers against occupancy: 8 -> 3 waves per SIMD, 16 -> 2).
    // Measured at 1340 x 2048 (same box, alternating): three hidden layers (2x16): 4: 0.2006 ms, 8: 0.1694-0.1707, 16: 0.1649-0.1654
    // -> 16; five hidden layers (4x8): 8: 0.33, 16: 0.39 (its registers no longer fit two waves) -> 8.
    // WDF_MLP_STEP_BS = 4 / 8 / 16: A/B timing.
    static const int wgrad_env = [] { const char* e = getenv("WDF_MLP_STEP_BS"); return e ? atoi(e) : 0; }();
    const int wgrad_bs = wgrad_env ? wgrad_env : (n_layers <= 3 ? 16 : 8);
    WDF_STEP_FWD(3) WDF_STEP_FWD(4) WDF_STEP_FWD(5)
    rc = check_launch("wdf_clipper_mlp_step");
    if (rc) return rc;
    if (phase & WDF_MLP_STEP_BWD) {
        const int count = step_weight_count(hidden, n_layers);
        hipLaunchKernelGGL(wdf::mlp_step_reduce_adam_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64, 16), 0, s,
                           (const float*)wsw, (L.n_cols * L.kw + 3) / 4, count, gw, w, adam_m, adam_v, (const int32_t*)adam_step, adam_lr,
                           beta1, beta2, eps, (const double*)A.colsum, L.n_cols, gsums, n_global, eps_energy, loss3, A.ctl);
        rc = check_launch("wdf_clipper_mlp_step reduce");
    }
    return rc;
}


/* ---- weight gradient over a list of samples (wdf_mlp_step.h: mlp_wgrad_list_kernel) ---- */
namespace {
bool list_arch_ok(int hidden, int n_tanh_layers)                 // the MLP-root family's architectures (wdf_capi_mlp.hip)
{
    return ((hidden == 4 || hidden == 8 || hidden == 16) && n_tanh_layers == 3) ||
           ((hidden == 4 || hidden == 8) && (n_tanh_layers == 4 || n_tanh_layers == 5));
}
// samples per wave: a multiple of 128, about one wave per SIMD, at most 4096 (fp32 accumulators: more waves instead)
int64_t list_per_wave(int64_t S)
{
    int64_t p = ((S + 1023) / 1024 + 127) / 128 * 128;
    return p < 128 ? 128 : (p > 4096 ? 4096 : p);
}
int64_t list_workgroups(int64_t S)
{
    const int64_t p = list_per_wave(S);
    return ((S + p - 1) / p + 3) / 4;
}
}  // namespace

int64_t wdf_clipper_mlp_wgrad_ws_bytes(int hidden, int n_tanh_layers, int64_t S)
{
    if (S <= 0 || !list_arch_ok(hidden, n_tanh_layers)) return 0;
    return list_workgroups(S) * (int64_t)step_weight_count(hidden, n_tanh_layers) * (int64_t)sizeof(float);
}

int wdf_clipper_mlp_wgrad(const float* ain, const float* lrin, const float* gb, const float* theta2, const float* w,
                          int hidden, int n_tanh_layers, float fs, void* ws, float* gw, int64_t S, void* stream)
{
    if (!ain || !gb || !w || !ws || !gw) return fail(WDF_EINVAL, "null ain/gb/w/ws/gw");
    if (!lrin && !theta2) return fail(WDF_EINVAL, "theta2 is needed when lrin is NULL");
    if (S <= 0) return fail(WDF_EINVAL, "S must be positive");
    if (!list_arch_ok(hidden, n_tanh_layers))
        return fail(WDF_EUNSUPPORTED, "MLP root: unsupported network (width %d, %d tanh layers)", hidden, n_tanh_layers);
    const int count = step_weight_count(hidden, n_tanh_layers);
    const int64_t per_wave = list_per_wave(S), nwg = list_workgroups(S);
    hipStream_t s = (hipStream_t)stream;
#define WDF_LIST_CASE(NL_)                                                                                                     \
    if (n_tanh_layers == NL_)                                                                                                  \
        hipLaunchKernelGGL((wdf::mlp_wgrad_list_kernel<NL_>), dim3((unsigned)nwg), dim3(256), 0, s, ain, lrin, gb, theta2, fs, w, hidden, S,  \
                           per_wave, (float*)ws);
    WDF_LIST_CASE(3) WDF_LIST_CASE(4) WDF_LIST_CASE(5)
#undef WDF_LIST_CASE
    int rc = check_launch("wdf_clipper_mlp_wgrad");
    if (rc) return rc;
    hipLaunchKernelGGL(wdf::mlp_wgrad_reduce_wide_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64, 16), 0, s, (const float*)ws, (int)nwg,
                       count, gw);
    return check_launch("wdf_clipper_mlp_wgrad_reduce");
}

}  // extern "C"
