// wdf_capi_mlp.hip -- C ABI part 4 of 4: the tanh-MLP root (layers.py DenseRootModel) kernels.
// Argument checking, template dispatch and launches.
// 
#include "wdf_capi_common.h"
#include "wdf_mlp.h"
#include "wdf_mlp_row.h"
#include "wdf_mlp_tp.h"
#include "wdf_mlp_mfma.h"
using namespace wdfcapi;

extern "C" {

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

// The weight-gradient pass of wdf_clipper_mlp_bwd_w_tp(_kappa) runs on the matrix cores (wdf_mlp_mfma.h, 16 sequences per wave)
// when that fills at least half the chip at two waves per SIMD: the chunk count it then uses, else 0.  WDF_MLP_WGRAD_MFMA = a
// chunk count forces it, 0 switches it off.  Exported so that a harness pricing that kernel asks the library instead of
// restating the rule.
extern "C" int wdf_clipper_mlp_wgrad_matrix_core_chunks(int64_t B, int64_t T)
{
    if (B <= 0 || T <= 0) return 0;
    int wm_chunks = 0;
    const int64_t w16 = (B + 15) / 16, kmax = T / 64 > 1 ? T / 64 : 1;
    int64_t kw = 2048 / w16;
    kw = kw > kmax ? kmax : (kw < 1 ? 1 : kw);
    if (w16 * kw >= 512) wm_chunks = (int)kw;
    if (const char* e = getenv("WDF_MLP_WGRAD_MFMA")) wm_chunks = atoi(e);
    return wm_chunks;
}

// Which forward kernel: the row kernel (4 sequences per wave, DPP; the shorter dependent chain per step: 0.31 us against
// 0.55 us) while the batch leaves SIMDs idle, the matrix-core kernel (16 per wave, wdf_mlp_mfma.h; 1.5x the row
// kernel's throughput) once the row kernel would stack three waves on a SIMD.  WDF_MLP_FWD_ROW = 1 / 0 forces one.
// n_chunks > 1 (the time-parallel entry points), width-16 nets with three tanh layers: also when the matrix-core waves,
// one per SIMD, cover a chunk count the row kernel could only reach with more than two waves per SIMD -- the
// reference's 1340 sequences in 12 chunks are 1008 waves of 16 against 4020 of 4 (bench.py --root mlp2x16, warm-started
// training loop: forward call 0.336 vs 0.386 ms at 6 row chunks).  Narrower nets are zero-padded to 16 on the matrix
// cores and deeper ones lengthen the MFMA chain: 2x8 0.83 vs 0.75 ms per step, 4x8 1.66 vs 1.29 -- those stay on the rows.
static bool mlp_fwd_on_matrix_cores(int64_t B, int n_chunks = 1, int hidden = 0, int n_tanh_layers = 0)
{
    const char* e = getenv("WDF_MLP_FWD_ROW");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '0';
    if (B >= 12288) return true;
    constexpr int64_t kSimd = 1024;                            // MI355X: 256 CUs x 4
    return n_chunks > 1 && hidden == 16 && n_tanh_layers == 3 && (B + 15) / 16 * n_chunks <= kSimd &&
           (B + 3) / 4 * n_chunks > 2 * kSimd;
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
    } else if (mlp_fwd_on_matrix_cores(B)) {      // 16 sequences per wave, one chunk (wdf_mlp_mfma.h)
        const dim3 gm((unsigned)((B + 15) / 16), 1u);
        const int64_t Lp = (T + 15) / 16 * 16;
#define WDF_MFMA_FWD(NL_)                                                                                      \
    if (n_tanh_layers == NL_) {                                                                                \
        if (dyn) hipLaunchKernelGGL((wdf::clipper_mlp_mfma_fwd_tp_kernel<NL_, true, false>), gm, dim3(64), 0,     \
                                    (hipStream_t)stream, x, r, theta2, w, hidden, fs, y, zstash, z0, zT, (float*)nullptr, \
                                    (float*)nullptr, (const int*)nullptr, (wdf::MlpTpStatus*)nullptr, B, T, Lp,     \
                                    (int64_t)0, Lp, (float*)nullptr, (const float*)nullptr, (const unsigned*)nullptr);                   \
        else hipLaunchKernelGGL((wdf::clipper_mlp_mfma_fwd_tp_kernel<NL_, false, false>), gm, dim3(64), 0,        \
                                (hipStream_t)stream, x, r, theta2, w, hidden, fs, y, zstash, z0, zT, (float*)nullptr, \
                                (float*)nullptr, (const int*)nullptr, (wdf::MlpTpStatus*)nullptr, B, T, Lp,         \
                                (int64_t)0, Lp, (float*)nullptr, (const float*)nullptr, (const unsigned*)nullptr);                       \
    }
        WDF_MFMA_FWD(3) WDF_MFMA_FWD(4) WDF_MFMA_FWD(5)
#undef WDF_MFMA_FWD
    } else {                                      // one 16-lane row per sequence (wdf_mlp_row.h)
        const unsigned grow = (unsigned)((B + 3) / 4);
#define WDF_ROW_FWD(NL_)                                                                                       \
    if (n_tanh_layers == NL_) {                                                                                \
        if (dyn) hipLaunchKernelGGL((wdf::clipper_mlp_row_fwd_kernel<NL_, true>), dim3(grow), dim3(64), 0,        \
                                    (hipStream_t)stream, x, r, theta2, w, hidden, fs, y, zstash, z0, zT, B, T, (const unsigned*)nullptr);   \
        else hipLaunchKernelGGL((wdf::clipper_mlp_row_fwd_kernel<NL_, false>), dim3(grow), dim3(64), 0,           \
                                (hipStream_t)stream, x, r, theta2, w, hidden, fs, y, zstash, z0, zT, B, T, (const unsigned*)nullptr);       \
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

// ---- time-parallel MLP-root kernels (wdf_mlp_tp.h) ------------------------------------------------
struct MlpTpGeom { int64_t L; int K; };
static MlpTpGeom mlp_tp_geom(int64_t T, int n_chunks)
{
    if (n_chunks < 1) n_chunks = 1;
    int64_t L = (T + n_chunks - 1) / n_chunks;
    L = (L + 15) / 16 * 16;
    return {L, (int)((T + L - 1) / L)};
}

// Chunk 0 needs no warm-up: left equal, its waves finish after L steps while every other wave runs L + W.  Balance
// them by COST: a warm-up step costs rho owned steps (rho = 1 for the plain forward; 0.7 when the owned steps also
// evaluate the input Jacobian for kappa -- tools/mlp_chunk_probe.py: 0.76 / 1.02 us on the matrix cores, 0.49 / 0.72 us
// on the row kernel), so L0 = L + rho W with L0 + (K-1) L = T: every wave then takes about as long as chunk 0's
// (one warm-up value for the batch only; with per-wave warm-ups the chunks stay equal).  -> L0; g.L = the others' length.
static int64_t mlp_tp_balance(int64_t T, int64_t W, bool one_warmup, bool with_kappa, MlpTpGeom& g)
{
    int64_t L0 = g.L;
    if (g.K >= 3 && one_warmup && W > 0) {
        const int64_t Wc = with_kappa ? (7 * W + 9) / 10 : W;            // the warm-up in owned-step units
        int64_t l0 = ((T + (int64_t)(g.K - 1) * Wc) / g.K + 15) / 16 * 16;
        const int64_t lmax = (T - 16 * (int64_t)(g.K - 1)) / 16 * 16;   // (chunk starts stay multiples of 16)
        if (l0 > lmax) l0 = lmax;
        if (l0 > g.L) {
            const int64_t rest = ((T - l0 + (g.K - 1) - 1) / (g.K - 1) + 15) / 16 * 16;
            if (rest >= 16 && l0 + (int64_t)(g.K - 2) * rest < T) { L0 = l0; g.L = rest; }   // (every chunk non-empty)
        }
    }
    return L0;
}

// starts[k] = the sample chunk k's wave begins at (its warm-up included), k < wdf_clipper_mlp_tp_chunks(T, n_chunks),
// for ONE warm-up value: where a warm-started call (zinit) wants the previous call's states from.  Host only.
int wdf_clipper_mlp_tp_starts(int64_t T, int n_chunks, int warmup, int64_t* starts)
{   // (the geometry of wdf_clipper_mlp_fwd_tp_kappa: the only entry point that takes zinit)
    if (T <= 0 || n_chunks < 1 || warmup < 0 || !starts) return fail(WDF_EINVAL, "T > 0, n_chunks >= 1, warmup >= 0, starts");
    MlpTpGeom g = mlp_tp_geom(T, n_chunks);
    const int64_t W = ((int64_t)warmup + 15) / 16 * 16;
    const int64_t L0 = mlp_tp_balance(T, W, true, true, g);
    for (int k = 0; k < g.K; ++k) {
        const int64_t t0 = k == 0 ? 0 : L0 + (int64_t)(k - 1) * g.L;
        starts[k] = t0 > W ? t0 - W : 0;
    }
    return WDF_OK;
}

int wdf_clipper_mlp_tp_chunks(int64_t T, int n_chunks) { return T > 0 ? mlp_tp_geom(T, n_chunks).K : 0; }

size_t wdf_clipper_mlp_fwd_tp_ws_bytes(int64_t B, int n_chunks)
{
    if (B <= 0 || n_chunks <= 0) return 0;
    // arrival and end states of the first pass and of the chunk-local repair pass, the two gates
    return (size_t)4 * (size_t)n_chunks * (size_t)B * sizeof(float) + (size_t)2 * (size_t)((B + 3) / 4) * sizeof(unsigned);
}

// kappa != nullptr: the forward also leaves kappa[T][B] (the adjoint recurrence's coefficient) for
// wdf_clipper_mlp_bwd_w_tp_kappa; the waves the sequential kernel re-runs get theirs from the stash afterwards.
static int mlp_fwd_tp_common(const float* x, const float* r, const float* theta2, const float* w, int hidden,
                             int n_tanh_layers, float fs, float* y, float* zstash, const float* z0, float* zT, int64_t B,
                             int64_t T, int n_chunks, int warmup, const int32_t* warmup_per_wave, float tol, void* ws,
                             void* status, float* kappa, bool want_kappa, const float* zinit, void* stream)
{
    int rc = mlp_check(x, theta2, w, hidden, n_tanh_layers, fs, B, T, 0);
    if (rc) return rc;
    if (!y || !ws || !status) return fail(WDF_EINVAL, "null y/ws/status");
    if (want_kappa && (!kappa || !zstash)) return fail(WDF_EINVAL, "null kappa/zstash");
    if (n_chunks < 1 || warmup < 0 || !(tol >= 0.0f)) return fail(WDF_EINVAL, "n_chunks >= 1, warmup >= 0, tol >= 0");
    MlpTpGeom g = mlp_tp_geom(T, n_chunks);
    const MlpTpGeom gu = g;                                    // equal chunks: the gated kappa pass's grid
    const int64_t W = ((int64_t)warmup + 15) / 16 * 16;
    const int64_t L0 = mlp_tp_balance(T, W, warmup_per_wave == nullptr, want_kappa, g);
    float* zwarm = (float*)ws;
    float* zend = zwarm + (size_t)g.K * (size_t)B;
    float* zwarm2 = zend + (size_t)g.K * (size_t)B;
    float* zend2 = zwarm2 + (size_t)g.K * (size_t)B;
    unsigned* gate = (unsigned*)(zend2 + (size_t)g.K * (size_t)B);
    unsigned* gate2 = gate + (size_t)((B + 3) / 4);
    const float* zprev = zend - B;                             // zprev[k][b] = zend[k-1][b]: where chunk k's predecessor ended
    // The chunks run on the row kernel or on the matrix cores (mlp_fwd_on_matrix_cores: by batch size; shapes that
    // want chunks at all are small, so normally the row kernel).  The verify / sequential / kappa launches keep the row grid.
    const bool use_row = !mlp_fwd_on_matrix_cores(B, g.K, hidden, n_tanh_layers);
    const unsigned grid_row = (unsigned)((B + 3) / 4);
    const dim3 grid(use_row ? grid_row : (unsigned)((B + 15) / 16), (unsigned)g.K);
    const bool dyn = r != nullptr;
    hipStream_t s = (hipStream_t)stream;
#define WDF_ROW_FWD_TP_LAUNCH_(NL_, DYN_, KAP_, ZW_, ZE_, WROW_, ST_, WW_, ZI_, GATE_)                                \
    if (use_row)                                                                                                 \
        hipLaunchKernelGGL((wdf::clipper_mlp_row_fwd_tp_kernel<NL_, DYN_, KAP_>), grid, dim3(64), 0, s, x, r, theta2, w, \
                           hidden, fs, y, zstash, z0, zT, ZW_, ZE_, WROW_, (wdf::MlpTpStatus*)(ST_), B, T, g.L,      \
                           (int64_t)(WW_), L0, kappa, ZI_, (const unsigned*)(GATE_));                             \
    else                                                                                                         \
        hipLaunchKernelGGL((wdf::clipper_mlp_mfma_fwd_tp_kernel<NL_, DYN_, KAP_>), grid, dim3(64), 0, s, x, r, theta2,   \
                           w, hidden, fs, y, zstash, z0, zT, ZW_, ZE_, WROW_, (wdf::MlpTpStatus*)(ST_), B, T, g.L,   \
                           (int64_t)(WW_), L0, kappa, ZI_, (const unsigned*)(GATE_))
#define WDF_ROW_FWD_TP_LAUNCH(NL_, DYN_, KAP_)                                                                     \
    WDF_ROW_FWD_TP_LAUNCH_(NL_, DYN_, KAP_, zwarm, zend, warmup_per_wave, status, W, zinit, nullptr)
// the chunk-local repair: the flagged waves again, every chunk from the state its predecessor ended in, no warm-up
#define WDF_ROW_FWD_TP_REPAIR(NL_, DYN_, KAP_)                                                                     \
    WDF_ROW_FWD_TP_LAUNCH_(NL_, DYN_, KAP_, zwarm2, zend2, (const int32_t*)nullptr, nullptr, 0, zprev, gate)
#define WDF_ROW_FWD_TP(NL_)                                                                                      \
    if (n_tanh_layers == NL_) {                                                                                  \
        {                                                                                                        \
            EventBracket bracket(s);                                                                             \
            if (want_kappa) {                                                                                    \
                if (dyn) { WDF_ROW_FWD_TP_LAUNCH(NL_, true, true); }                                             \
                else { WDF_ROW_FWD_TP_LAUNCH(NL_, false, true); }                                                \
            } else {                                                                                             \
                if (dyn) { WDF_ROW_FWD_TP_LAUNCH(NL_, true, false); }                                            \
                else { WDF_ROW_FWD_TP_LAUNCH(NL_, false, false); }                                               \
            }                                                                                                    \
        }                                                                                                        \
        if (g.K > 1) {                                                                                           \
            hipLaunchKernelGGL(wdf::mlp_tp_verify_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, zwarm, zend, B, \
                               (int64_t)g.K, tol, gate, (wdf::MlpTpStatus*)status, (const unsigned*)nullptr);     \
            if (want_kappa) {                                                                                    \
                if (dyn) { WDF_ROW_FWD_TP_REPAIR(NL_, true, true); }                                             \
                else { WDF_ROW_FWD_TP_REPAIR(NL_, false, true); }                                                \
            } else {                                                                                             \
                if (dyn) { WDF_ROW_FWD_TP_REPAIR(NL_, true, false); }                                            \
                else { WDF_ROW_FWD_TP_REPAIR(NL_, false, false); }                                               \
            }                                                                                                    \
            hipLaunchKernelGGL(wdf::mlp_tp_verify_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s,            \
                               (const float*)zwarm2, (const float*)zend2, B, (int64_t)g.K, tol, gate2,           \
                               (wdf::MlpTpStatus*)status, (const unsigned*)gate);                                \
            if (dyn) hipLaunchKernelGGL((wdf::clipper_mlp_row_fwd_kernel<NL_, true>), dim3(grid_row), dim3(64), 0, s, x, r, \
                                        theta2, w, hidden, fs, y, zstash, z0, zT, B, T, (const unsigned*)gate2);  \
            else hipLaunchKernelGGL((wdf::clipper_mlp_row_fwd_kernel<NL_, false>), dim3(grid_row), dim3(64), 0, s, x, r,    \
                                    theta2, w, hidden, fs, y, zstash, z0, zT, B, T, (const unsigned*)gate2);      \
            if (want_kappa) {                                                                                    \
                const dim3 kgrid(grid_row, (unsigned)gu.K);                                                      \
                if (dyn) hipLaunchKernelGGL((wdf::clipper_mlp_row_kappa_kernel<NL_, true>), kgrid, dim3(64), 0, s, x, r,   \
                                            theta2, w, hidden, fs, (const float*)zstash, kappa, B, T, gu.L,       \
                                            (const unsigned*)gate2);                                             \
                else hipLaunchKernelGGL((wdf::clipper_mlp_row_kappa_kernel<NL_, false>), kgrid, dim3(64), 0, s, x, r,      \
                                        theta2, w, hidden, fs, (const float*)zstash, kappa, B, T, gu.L,           \
                                        (const unsigned*)gate2);                                                 \
            }                                                                                                    \
        }                                                                                                        \
    }
    WDF_ROW_FWD_TP(3) WDF_ROW_FWD_TP(4) WDF_ROW_FWD_TP(5)
#undef WDF_ROW_FWD_TP_LAUNCH
#undef WDF_ROW_FWD_TP_REPAIR
#undef WDF_ROW_FWD_TP_LAUNCH_
#undef WDF_ROW_FWD_TP
    return check_launch("wdf_clipper_mlp_fwd_tp");
}

int wdf_clipper_mlp_fwd_tp(const float* x, const float* r, const float* theta2, const float* w, int hidden,
                           int n_tanh_layers, float fs, float* y, float* zstash, const float* z0, float* zT, int64_t B,
                           int64_t T, int n_chunks, int warmup, const int32_t* warmup_per_wave, float tol, void* ws,
                           void* status, void* stream)
{
    return mlp_fwd_tp_common(x, r, theta2, w, hidden, n_tanh_layers, fs, y, zstash, z0, zT, B, T, n_chunks, warmup,
                             warmup_per_wave, tol, ws, status, nullptr, false, nullptr, stream);
}

int wdf_clipper_mlp_fwd_tp_kappa(const float* x, const float* r, const float* theta2, const float* w, int hidden,
                                 int n_tanh_layers, float fs, float* y, float* zstash, float* kappa, const float* z0,
                                 float* zT, const float* zinit, int64_t B, int64_t T, int n_chunks, int warmup,
                                 const int32_t* warmup_per_wave, float tol, void* ws, void* status, void* stream)
{
    if (zinit && warmup_per_wave) return fail(WDF_EINVAL, "zinit goes with one warm-up value (wdf_clipper_mlp_tp_starts)");
    return mlp_fwd_tp_common(x, r, theta2, w, hidden, n_tanh_layers, fs, y, zstash, z0, zT, B, T, n_chunks, warmup,
                             warmup_per_wave, tol, ws, status, kappa, true, zinit, stream);
}

int64_t wdf_clipper_mlp_bwd_w_tp_ws_bytes(int hidden, int n_tanh_layers, int64_t B, int64_t T, int n_chunks)
{
    if (B <= 0 || T <= 0 || n_chunks <= 0 || !mlp_arch_ok(hidden, n_tanh_layers)) return 0;
    const int64_t nparts = (B + 3) / 4 * mlp_tp_geom(T, n_chunks).K;
    const int64_t scan_chunks = (T + wdf::kScanChunk - 1) / wdf::kScanChunk;
    return T * B * (int64_t)sizeof(float) + nparts * 4 * (int64_t)sizeof(double) +
           nparts * wdf_mlp_weight_count(hidden, n_tanh_layers) * (int64_t)sizeof(float) +
           scan_chunks * B * (int64_t)sizeof(float2) + 8;
}

// kappa_in != nullptr: pass (A) is skipped, the scan reads the forward's kappa (wdf_clipper_mlp_fwd_tp_kappa).
static int mlp_bwd_w_tp_common(const float* x, const float* r, const float* theta2, const float* w, int hidden,
                               int n_tanh_layers, float fs, const float* zstash, const float* kappa_in, bool have_kappa,
                               const float* gy, void* ws, float* gtheta2, float* gw, int64_t B, int64_t T, int n_chunks,
                               void* stream)
{
    int rc = mlp_check(x, theta2, w, hidden, n_tanh_layers, fs, B, T, 0);
    if (rc) return rc;
    if (!zstash || !gy || !ws || !gtheta2 || !gw) return fail(WDF_EINVAL, "null zstash/gy/ws/gtheta2/gw");
    if (have_kappa && !kappa_in) return fail(WDF_EINVAL, "null kappa");
    if (n_chunks < 1) return fail(WDF_EINVAL, "n_chunks >= 1");
    const MlpTpGeom g = mlp_tp_geom(T, n_chunks);
    const dim3 grid((unsigned)((B + 3) / 4), (unsigned)g.K);
    const int nparts = (int)(grid.x * grid.y);
    double* wsd = (double*)ws;                                             // [nparts][4] doubles (8-byte aligned)
    float* kap = (float*)((char*)ws + (size_t)nparts * 4 * sizeof(double));  // kappa, then g_b2n in place [T][B]
    float* wsw = kap + (size_t)T * (size_t)B;                              // [nparts][count]
    const int count_w = wdf_mlp_weight_count(hidden, n_tanh_layers);
    const unsigned scan_chunks = (unsigned)((T + wdf::kScanChunk - 1) / wdf::kScanChunk);
    float2* smap = (float2*)(((uintptr_t)(wsw + (size_t)nparts * (size_t)count_w) + 7) & ~(uintptr_t)7);   // [scan_chunks][B]
    const dim3 sgrid((unsigned)((B + 63) / 64), scan_chunks);
    // (C) on the matrix cores (wdf_mlp_mfma.h, 16 sequences per wave) when that fills at least half the chip, two waves
    // per SIMD -- bench.py --root mlp2x16 / 2x8 / 4x8: step 0.655 -> 0.578, 0.663 -> 0.596, 1.170 -> 0.978 ms.
    // WDF_MLP_WGRAD_MFMA = a chunk count forces it, 0 switches it off.
    int wm_chunks = wdf_clipper_mlp_wgrad_matrix_core_chunks(B, T);
    const MlpTpGeom gm = mlp_tp_geom(T, wm_chunks > 0 ? wm_chunks : 1);
    const dim3 wmgrid((unsigned)((B + 15) / 16), (unsigned)gm.K);
    if (wm_chunks > 0 && (int64_t)wmgrid.x * wmgrid.y > (int64_t)nparts) wm_chunks = 0;     // (the partial buffers are sized for the row grid)
    const int rparts = wm_chunks > 0 ? (int)(wmgrid.x * wmgrid.y) : nparts;
    const bool dyn = r != nullptr;
    hipStream_t s = (hipStream_t)stream;
#define WDF_ROW_BWD_TP(NL_)                                                                                      \
    if (n_tanh_layers == NL_) {                                                                                  \
        if (!have_kappa) {                                                                                       \
            if (dyn) hipLaunchKernelGGL((wdf::clipper_mlp_row_kappa_kernel<NL_, true>), grid, dim3(64), 0, s, x, r, theta2,  \
                                        w, hidden, fs, zstash, kap, B, T, g.L, (const unsigned*)nullptr);        \
            else hipLaunchKernelGGL((wdf::clipper_mlp_row_kappa_kernel<NL_, false>), grid, dim3(64), 0, s, x, r, theta2,     \
                                    w, hidden, fs, zstash, kap, B, T, g.L, (const unsigned*)nullptr);            \
        }                                                                                                        \
        if (scan_chunks > 1) {                                                                                   \
            hipLaunchKernelGGL(wdf::mlp_adjoint_chunk_map_kernel, sgrid, dim3(64), 0, s,                         \
                               have_kappa ? kappa_in : (const float*)kap, gy, smap, B, T);                       \
            hipLaunchKernelGGL(wdf::mlp_adjoint_scan_chunked_kernel, sgrid, dim3(64), 0, s,                      \
                               have_kappa ? kappa_in : (const float*)kap, gy, (const float2*)smap, kap, B, T);   \
        } else {                                                                                                 \
            hipLaunchKernelGGL(wdf::mlp_adjoint_scan_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s,         \
                               have_kappa ? kappa_in : (const float*)kap, gy, kap, B, T);                        \
        }                                                                                                        \
        {                                                                                                        \
            EventBracket bracket(s);                                                                             \
            if (wm_chunks > 0) {                                                                                 \
                if (dyn) hipLaunchKernelGGL((wdf::clipper_mlp_mfma_wgrad_tp_kernel<NL_, true>), wmgrid, dim3(64), 0, s, x, r,  \
                                            theta2, w, hidden, fs, zstash, kap, wsw, wsd, B, T, gm.L);            \
                else hipLaunchKernelGGL((wdf::clipper_mlp_mfma_wgrad_tp_kernel<NL_, false>), wmgrid, dim3(64), 0, s, x, r,     \
                                        theta2, w, hidden, fs, zstash, kap, wsw, wsd, B, T, gm.L);                \
            } else {                                                                                             \
                if (dyn) hipLaunchKernelGGL((wdf::clipper_mlp_row_wgrad_tp_kernel<NL_, true>), grid, dim3(64), 0, s, x, r,     \
                                            theta2, w, hidden, fs, zstash, kap, wsw, wsd, B, T, g.L);             \
                else hipLaunchKernelGGL((wdf::clipper_mlp_row_wgrad_tp_kernel<NL_, false>), grid, dim3(64), 0, s, x, r,        \
                                        theta2, w, hidden, fs, zstash, kap, wsw, wsd, B, T, g.L);                 \
            }                                                                                                    \
        }                                                                                                        \
    }
    WDF_ROW_BWD_TP(3) WDF_ROW_BWD_TP(4) WDF_ROW_BWD_TP(5)
#undef WDF_ROW_BWD_TP
    rc = check_launch("wdf_clipper_mlp_bwd_w_tp");
    if (rc) return rc;
    hipLaunchKernelGGL(wdf::clipper_mlp_grad_reduce_kernel, dim3(1), dim3(256), 0, s, (const double*)wsd, rparts, theta2, fs,
                       dyn ? 1 : 0, gtheta2);
    const int count = wdf_mlp_weight_count(hidden, n_tanh_layers);
    hipLaunchKernelGGL(wdf::mlp_wgrad_reduce_wide_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64, 16), 0, s,
                       (const float*)wsw, rparts, count, gw);
    return check_launch("wdf_clipper_mlp_bwd_w_tp reduce");
}

int wdf_clipper_mlp_bwd_w_tp(const float* x, const float* r, const float* theta2, const float* w, int hidden,
                             int n_tanh_layers, float fs, const float* zstash, const float* gy, void* ws, float* gtheta2,
                             float* gw, int64_t B, int64_t T, int n_chunks, void* stream)
{
    return mlp_bwd_w_tp_common(x, r, theta2, w, hidden, n_tanh_layers, fs, zstash, nullptr, false, gy, ws, gtheta2, gw,
                               B, T, n_chunks, stream);
}

int wdf_clipper_mlp_bwd_w_tp_kappa(const float* x, const float* r, const float* theta2, const float* w, int hidden,
                                   int n_tanh_layers, float fs, const float* zstash, const float* kappa, const float* gy,
                                   void* ws, float* gtheta2, float* gw, int64_t B, int64_t T, int n_chunks, void* stream)
{
    return mlp_bwd_w_tp_common(x, r, theta2, w, hidden, n_tanh_layers, fs, zstash, kappa, true, gy, ws, gtheta2, gw, B,
                               T, n_chunks, stream);
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

// (wdf_clipper_mlp_wgrad: wdf_capi_mlp_step.hip, next to the matrix-core accumulators it shares with the resident step)

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

}  // extern "C"
