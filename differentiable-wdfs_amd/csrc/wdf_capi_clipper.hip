// wdf_capi_clipper.hip -- C ABI part 2 of 4: the diode-clipper sequence kernels (sequential and
// time-parallel forward / reverse sweep).  Argument checking, template dispatch and launches.
// 
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <cstddef>
#include "wdf_capi_common.h"
#include "wdf_clipper.h"
#include "wdf_clipper_fused.h"
#include "wdf_omega64.h"
using namespace wdfcapi;

namespace {

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
                const float* zstash, const float* gy, double* ws, float* gz0, const float* gzT, int64_t B, int64_t T,
                hipStream_t s)
{
    const unsigned grid = (unsigned)((B + 63) / 64);
    EventBracket bracket(s);
    hipLaunchKernelGGL((wdf::clipper_bwd_kernel<DYN_R, SYM, TM, V4>), dim3(grid), dim3(64), 0, s, x, r, theta, fs,
                       n_up, n_down, zstash, gy, ws, gz0, gzT, B, T);
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
    if (flags & ~(WDF_X_TIME_MAJOR | WDF_PREC_F64 | WDF_GENERAL_ROOT)) return fail(WDF_EINVAL, "unknown flag bits 0x%x", flags);
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

// state: nullptr (stateless, cold every call) or the caller's persistent warm-start buffer
// [TpCtl][tile tickets + accumulators][snapshot ring kTpRing x J x K x B floats]
struct TpWarm { wdf::TpCtl* ctl; float* snap; int J; };

// per-tile tickets, 4 accumulator words, per-tile repair flags
inline size_t tp_ticket_bytes(int64_t B) { return ((2 * (size_t)((B + 63) / 64) + 4) * sizeof(unsigned) + 63) / 64 * 64; }

template <bool DYN_R, bool SYM, bool TM, bool V4>
void launch_fwd_tp(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down, float* y,
                   float* zstash, const float* z0, float* zT, float* zwarm, float* zend, wdf::TpStatus* status,
                   float tol, int64_t B, int64_t T, TpGeom g, int64_t W, TpWarm warm, unsigned* tickets, int general,
                   hipStream_t s)
{
    const dim3 grid((unsigned)((B + 63) / 64), (unsigned)g.K);
    // a stateless call (no warm-start block to steer): the chunk boundaries are verified by the launch behind the forward
    const int later = (warm.ctl == nullptr && g.K > 1) ? 1 : 0;
#define WDF_FWD_TP(STASH_)                                                                                   \
    hipLaunchKernelGGL((wdf::clipper_fwd_tp_kernel<DYN_R, SYM, TM, V4, STASH_, float>), grid, dim3(64), 0, s, x, r, theta, \
                       fs, n_up, n_down, y, zstash, z0, zT, zwarm, zend, status, warm.ctl, warm.snap, warm.J, tickets, tol, \
                       B, B, T, g.L, W, general, later)
    {
        EventBracket bracket(s);
        if (zstash) WDF_FWD_TP(true); else WDF_FWD_TP(false);
    }
#undef WDF_FWD_TP
    if (g.K > 1) {                              // blocks of unflagged tiles (normally all of them) leave at once
#define WDF_REPAIR(STASH_)                                                                                   \
    hipLaunchKernelGGL((wdf::clipper_tp_repair_kernel<DYN_R, SYM, TM, STASH_>), dim3(grid.x), dim3(64), 0, s, x, r, theta, \
                       fs, n_up, n_down, y, zstash, zT, zwarm, zend, B, T, (int64_t)g.K, g.L, tol, status, warm.ctl,       \
                       warm.snap, warm.J, tickets, general, later)
        if (zstash) WDF_REPAIR(true); else WDF_REPAIR(false);
#undef WDF_REPAIR
    }
}

template <bool DYN_R, bool SYM, bool TM, bool V4>
void launch_bwd_tp(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                   const float* zstash, const float* gy, const float* target, const float* zT, float gscale,
                   float* part, double* ws, float* gz0, int64_t B, int64_t T, TpGeom g, const float* gcoef,
                   int64_t skip, unsigned* tickets, float* gtheta, int accumulate, float* sse_out, wdf::AdamTail adam,
                   int general, hipStream_t s)
{
    const dim3 grid((unsigned)((B + 63) / 64), (unsigned)g.K);
    // ONE launch: the sweep; the last chunk wave of every tile combines the tile's chunk records, the last
    // tile reduces, applies the chain rule and (optionally) Adam
#define WDF_BWD_TP(MSE_)                                                                                     \
    hipLaunchKernelGGL((wdf::clipper_bwd_tp_kernel<DYN_R, SYM, TM, V4, MSE_, float>), grid, dim3(64), 0, s, x, r, theta, \
                       fs, n_up, n_down, zstash, gy, target, zT, gscale, part, B, B, T, g.L, gcoef, skip, tickets, ws,   \
                       gz0, gtheta, accumulate, sse_out, adam, general)
    EventBracket bracket(s);
    if (gcoef) WDF_BWD_TP(2);                                // MSE + ESR
    else if (target) WDF_BWD_TP(1);
    else WDF_BWD_TP(0);
#undef WDF_BWD_TP
}

inline size_t bwd_ticket_bytes(int64_t B) { return (((size_t)((B + 63) / 64) + 4) * sizeof(unsigned) + 63) / 64 * 64; }


// ---- the one-pass training step (wdf_clipper_fused.h) --------------------------------------------
struct FusedWs { double* part; double* wpart; float* zwarm; float* zend; float* rec; unsigned* tickets; unsigned* gticket; };

// ONE layout for both losses (sized for the larger, MSE + ESR; a workspace may serve either from call to call):
// [tiles][8] doubles (the tiles' sums; MSE uses 4 of them), [tiles][K][slots][8] doubles (the chunk waves' own sums, one set
// per sequence slot of a lane), zwarm / zend [K][B], the record granules, then the ticket words.
inline size_t fused_tiles(int64_t B) { return (size_t)((B + 63) / 64) + 1; }      // >= tiles x slots for either lane form
inline size_t fused_part_bytes(int64_t B) { return fused_tiles(B) * 8 * sizeof(double); }
inline size_t fused_wpart_bytes(int64_t B, int K) { return fused_tiles(B) * (size_t)K * wdf::kFsPartEsr * sizeof(double); }

inline size_t fused_body_bytes(int64_t B, int K)
{
    // zwarm / zend [K][B] floats; records: 16-byte granules [K][3][lanes], lanes <= 64 x ceil(B / 64)
    const size_t body = fused_part_bytes(B) + fused_wpart_bytes(B, K) + (size_t)2 * (size_t)K * (size_t)B * sizeof(float) +
                        (size_t)K * wdf::kFsQuads * (size_t)((B + 63) / 64 * 64) * 16;
    return (body + 63) / 64 * 64;
}

inline FusedWs fused_ws(void* ws, int64_t B, int K)
{
    FusedWs w;
    w.part = (double*)ws;
    w.wpart = (double*)((char*)ws + fused_part_bytes(B));
    w.zwarm = (float*)((char*)ws + fused_part_bytes(B) + fused_wpart_bytes(B, K));
    w.zend = w.zwarm + (size_t)K * (size_t)B;
    w.rec = w.zend + (size_t)K * (size_t)B;
    w.tickets = (unsigned*)((char*)ws + fused_body_bytes(B, K));
    w.gticket = (unsigned*)((char*)w.tickets + tp_ticket_bytes(B));
    return w;
}

// Skewed chunk spans (wdf_clipper_fused.h, chunk_span): only when the launch puts about two chunk waves on every SIMD --
// that is the situation the skew answers -- the chunk count is even, the ragged last chunk keeps more than `skew` steps and
// the shorter chunks still hold the warm-start snapshots.  WDF_FUSED_SKEW=0 switches it off, =force applies it whenever
// the geometry allows (tests).
inline int64_t fused_skew(int64_t n_waves, int K, int64_t L, int64_t T, int max_warm_tiles)
{
    static const int mode = []() { const char* e = getenv("WDF_FUSED_SKEW"); return !e ? 1 : (strcmp(e, "force") == 0 ? 2 : atoi(e) != 0); }();
    if (mode == 0 || K < 2 || (K & 1)) return 0;
    if (mode == 1) {
        static const int n_simd = []() { int dev = 0, cus = 256; (void)hipGetDevice(&dev);
                                         (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); return 4 * cus; }();
        if (2 * n_waves <= 3 * (int64_t)n_simd || 2 * n_waves > 5 * (int64_t)n_simd) return 0;
    }
    static const int64_t steps = []() { const char* e = getenv("WDF_FUSED_SKEW_STEPS"); return e ? (int64_t)atoi(e) : (int64_t)-1; }();   // (A/B runs)
    const int64_t skew = (steps >= 0 ? steps : L / 4) / wdf::kTile * wdf::kTile;
    if (skew <= 0 || skew >= L) return 0;
    if (T - (int64_t)(K - 1) * L <= skew) return 0;                                   // the last chunk would be empty
    if ((int64_t)max_warm_tiles * wdf::kWarmStep > L - skew) return 0;                      // snapshots reach further back than the short chunks
    return skew;
}

template <int DYN_R, bool SYM, bool TM, bool V4>
void launch_fused(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down, const float* target,
                  float hgs, int64_t skip, float* y, const float* z0, float* zT, FusedWs w, wdf::TpStatus* status, float tol,
                  int64_t B, int64_t T, TpGeom g, int64_t W, TpWarm warm, int general, bool pairs, bool esr, wdf::FusedOut out,
                  int64_t skew, hipStream_t s)
{
    // pairs: two adjacent sequences per lane, packed fp32 arithmetic (wdf_clipper_fused.h); a tile is then 128 sequences
    const int per_tile = pairs ? 128 : 64;
    const dim3 grid((unsigned)((B + per_tile - 1) / per_tile), (unsigned)g.K);
    // Round 6: the step's tail (verification, walk over the chunk records, reduction, chain rule, warm-start steering, Adam) and the
    // repair of missed tiles are ONE launch behind the chunk kernel, several waves per tile (clipper_fused_finish_kernel).
    // WDF_FUSED_FINISH=inkernel restores the round-5 form (the tile's last chunk wave does the tail, an idle repair launch follows).
    static const bool finish_inkernel = []() { const char* e = getenv("WDF_FUSED_FINISH"); return e && strcmp(e, "inkernel") == 0; }();
    const int later = (g.K > 1 && !finish_inkernel) ? 1 : 0;
    const int fin_waves = (int)std::min<int64_t>(wdf::kFinMaxWaves, (g.K + wdf::kFinSeg - 1) / wdf::kFinSeg);
#define WDF_FUSED(V_, LOSS_)                                                                                                     \
    hipLaunchKernelGGL((wdf::clipper_fused_tp_kernel<DYN_R, SYM, TM, V4, V_, LOSS_>), grid, dim3(64), 0, s, x, r, theta, fs, n_up, \
                       n_down, target, hgs, skip, y, z0, zT, w.zwarm, w.zend, w.rec, status, warm.ctl, warm.snap, warm.J,        \
                       w.tickets, w.gticket, tol, B, T, g.L, W, general, w.part, out, skew, w.wpart, later)
#define WDF_FUSED_FINISH(N_, LOSS_)                                                                                              \
    hipLaunchKernelGGL((wdf::clipper_fused_finish_kernel<DYN_R, SYM, TM, N_, LOSS_>), dim3(grid.x), dim3(64 * fin_waves), 0, s, x, r, \
                       theta, fs, n_up, n_down, target, hgs, skip, y, zT, w.zwarm, w.zend, w.rec, B, T, (int64_t)g.K, g.L, tol,   \
                       status, warm.ctl, warm.snap, warm.J, w.tickets, w.gticket, general, w.part, out, skew, w.wpart, W)
#define WDF_FUSED_REPAIR(N_, LOSS_)                                                                                              \
    hipLaunchKernelGGL((wdf::clipper_fused_repair_kernel<DYN_R, SYM, TM, N_, LOSS_>), dim3(grid.x), dim3(64), 0, s, x, r, theta, fs, \
                       n_up, n_down, target, hgs, skip, y, zT, w.zwarm, w.zend, w.rec, B, T, (int64_t)g.K, g.L, tol, status,     \
                       warm.ctl, warm.snap, warm.J, w.tickets, w.gticket, general, w.part, out, skew, w.wpart, W)
    {
        EventBracket bracket(s);
        if (esr) { if (pairs) WDF_FUSED(wdf::v2f, 2); else WDF_FUSED(float, 2); }
        else { if (pairs) WDF_FUSED(wdf::v2f, 1); else WDF_FUSED(float, 1); }
    }
    if (later) {
        if (esr) { if (pairs) WDF_FUSED_FINISH(2, 2); else WDF_FUSED_FINISH(1, 2); }
        else { if (pairs) WDF_FUSED_FINISH(2, 1); else WDF_FUSED_FINISH(1, 1); }
    } else if (g.K > 1) {                       // blocks of unflagged tiles (normally all of them) leave at once
        if (esr) { if (pairs) WDF_FUSED_REPAIR(2, 2); else WDF_FUSED_REPAIR(1, 2); }
        else { if (pairs) WDF_FUSED_REPAIR(2, 1); else WDF_FUSED_REPAIR(1, 1); }
    }
#undef WDF_FUSED
#undef WDF_FUSED_FINISH
#undef WDF_FUSED_REPAIR
}

// the finish of the MSE + ESR step after the ranks' sums10 have been all-reduced (esr_tile_partial_and_finish's last lines)
__global__ void esr_finish_kernel(const float* __restrict__ sums10, double n, double eps, float* __restrict__ gtheta,
                                  float* __restrict__ loss3)
{
    const double S = sums10[0], E = (double)sums10[1] + eps;
    const double mse = S / n, esr = sqrt(S / E / n);
    const double ga = 2.0 / n + (esr > 0.0 ? 1.0 / (esr * E * n) : 0.0), gb = -esr / E;
    for (int k = 0; k < 4; ++k) gtheta[k] = (float)(ga * (double)sums10[2 + k] + gb * (double)sums10[6 + k]);
    if (loss3) { loss3[0] = (float)mse; loss3[1] = (float)esr; loss3[2] = (float)(mse + esr); }
}

}  // namespace

extern "C" {
#ifdef WDF_DBG_TIMES
__attribute__((visibility("default"))) int wdf_debug_set_times(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(wdf::g_dbg_times), &p, sizeof(p)); }
#endif

int wdf_clipper_fwd(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down, float* y,
                    float* zstash, const float* z0, float* zT, int64_t B, int64_t T, int flags, void* stream)
{
    int rc = check_common(x, theta, n_up, n_down, B, T, flags);
    if (rc) return rc;
    if (!y) return fail(WDF_EINVAL, "null y");
    if (!(fs > 0.0f)) return fail(WDF_EINVAL, "fs must be positive");
    const bool tm = flags & WDF_X_TIME_MAJOR;
    if (flags & WDF_PREC_F64) {                  // tree and root in fp64 (csrc/wdf_omega64.h): the on-device accuracy reference
        const unsigned grid = (unsigned)((B + 63) / 64);
        hipStream_t s = (hipStream_t)stream;
        if (r) { if (tm) hipLaunchKernelGGL((wdf::clipper_fwd_f64_kernel<true, true>), dim3(grid), dim3(64), 0, s, x, r, theta, fs, n_up, n_down, y, zstash, z0, zT, B, T);
                 else hipLaunchKernelGGL((wdf::clipper_fwd_f64_kernel<true, false>), dim3(grid), dim3(64), 0, s, x, r, theta, fs, n_up, n_down, y, zstash, z0, zT, B, T); }
        else   { if (tm) hipLaunchKernelGGL((wdf::clipper_fwd_f64_kernel<false, true>), dim3(grid), dim3(64), 0, s, x, r, theta, fs, n_up, n_down, y, zstash, z0, zT, B, T);
                 else hipLaunchKernelGGL((wdf::clipper_fwd_f64_kernel<false, false>), dim3(grid), dim3(64), 0, s, x, r, theta, fs, n_up, n_down, y, zstash, z0, zT, B, T); }
        return check_launch("wdf_clipper_fwd (fp64)");
    }
    const bool v4 = !tm && (T % 4 == 0) && aligned16(x) && (!r || aligned16(r));
    WDF_DISPATCH4(launch_fwd, r != nullptr, n_up == n_down, tm, v4, x, r, theta, fs, n_up, n_down, y, zstash, z0, zT,
                  B, T, (flags & WDF_GENERAL_ROOT) ? 1 : 0, (hipStream_t)stream);
    return check_launch("wdf_clipper_fwd");
}

size_t wdf_clipper_bwd_ws_bytes(int64_t B) { return B > 0 ? (size_t)((B + 63) / 64) * 4 * sizeof(double) : 0; }

int wdf_clipper_bwd(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                    const float* zstash, const float* gy, void* ws, float* gtheta, float* gz0, const float* gzT,
                    int accumulate, int64_t B, int64_t T, int flags, void* stream)
{
    int rc = check_common(x, theta, n_up, n_down, B, T, flags);
    if (rc) return rc;
    if (!zstash || !gy || !ws || !gtheta) return fail(WDF_EINVAL, "null zstash/gy/ws/gtheta");
    if (!(fs > 0.0f)) return fail(WDF_EINVAL, "fs must be positive");
    const bool tm = flags & WDF_X_TIME_MAJOR;
    if (flags & WDF_PREC_F64) {                  // the adjoint in fp64 (csrc/wdf_omega64.h): the on-device accuracy reference
        const unsigned grid = (unsigned)((B + 63) / 64);
        hipStream_t s = (hipStream_t)stream;
        if (r) { if (tm) hipLaunchKernelGGL((wdf::clipper_bwd_f64_kernel<true, true>), dim3(grid), dim3(64), 0, s, x, r, theta, fs, n_up, n_down, zstash, gy, (double*)ws, gz0, gzT, B, T);
                 else hipLaunchKernelGGL((wdf::clipper_bwd_f64_kernel<true, false>), dim3(grid), dim3(64), 0, s, x, r, theta, fs, n_up, n_down, zstash, gy, (double*)ws, gz0, gzT, B, T); }
        else   { if (tm) hipLaunchKernelGGL((wdf::clipper_bwd_f64_kernel<false, true>), dim3(grid), dim3(64), 0, s, x, r, theta, fs, n_up, n_down, zstash, gy, (double*)ws, gz0, gzT, B, T);
                 else hipLaunchKernelGGL((wdf::clipper_bwd_f64_kernel<false, false>), dim3(grid), dim3(64), 0, s, x, r, theta, fs, n_up, n_down, zstash, gy, (double*)ws, gz0, gzT, B, T); }
    } else {
        const bool v4 = !tm && (T % 4 == 0) && aligned16(x) && (!r || aligned16(r));
        WDF_DISPATCH4(launch_bwd, r != nullptr, n_up == n_down, tm, v4, x, r, theta, fs, n_up, n_down, zstash, gy,
                      (double*)ws, gz0, gzT, B, T, (hipStream_t)stream);
    }
    rc = check_launch("wdf_clipper_bwd");
    if (rc) return rc;
    const int nparts = (int)((B + 63) / 64);
    hipLaunchKernelGGL(wdf::clipper_grad_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                       (const double*)ws, nparts, theta, fs, r != nullptr ? 1 : 0, gtheta, accumulate,
                       (float*)nullptr);
    return check_launch("wdf_clipper_grad_reduce");
}

int wdf_clipper_tp_chunks(int64_t T, int n_chunks) { return T > 0 ? tp_geom(T, n_chunks).K : 0; }
int wdf_clipper_tp_warm_unit(void) { return wdf::kWarmStep; }

size_t wdf_clipper_fwd_tp_ws_bytes(int64_t B, int n_chunks)
{
    // zwarm [K][B], zend [K][B], then (stateless calls) the tile tickets
    return (B > 0 && n_chunks > 0) ? (size_t)2 * (size_t)n_chunks * (size_t)B * sizeof(float) + tp_ticket_bytes(B) : 0;
}

static int fwd_tp_common(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down, float* y,
                         float* zstash, const float* z0, float* zT, int64_t B, int64_t T, int n_chunks, int warmup,
                         float tol, void* ws, void* status, void* state, int max_warm_tiles, int flags, void* stream)
{
    int rc = check_common(x, theta, n_up, n_down, B, T, flags);
    if (rc) return rc;
    if (!y || !ws || !status) return fail(WDF_EINVAL, "null y/ws/status");
    if (!(fs > 0.0f)) return fail(WDF_EINVAL, "fs must be positive");
    if (n_chunks < 1 || warmup < 0 || !(tol >= 0.0f)) return fail(WDF_EINVAL, "n_chunks >= 1, warmup >= 0, tol >= 0");
    if (B >= ((int64_t)1 << 24)) return fail(WDF_EINVAL, "the time-parallel forward addresses a 16-row tile with 32-bit offsets: B < 2^24");
    if (flags & WDF_PREC_F64) return fail(WDF_EUNSUPPORTED, "WDF_PREC_F64 applies to wdf_clipper_fwd and wdf_omega_f64 only");
    const TpGeom g = tp_geom(T, n_chunks);
    const int64_t W = ((int64_t)warmup + wdf::kTile - 1) / wdf::kTile * wdf::kTile;
    float* zwarm = (float*)ws;
    float* zend = zwarm + (size_t)g.K * (size_t)B;
    TpWarm warm{nullptr, nullptr, 1};
    unsigned* tickets;
    if (state) {
        if (max_warm_tiles < 1 || max_warm_tiles > wdf::kTpMaxWarmTiles)
            return fail(WDF_EINVAL, "max_warm_tiles must be in 1..%d", wdf::kTpMaxWarmTiles);
        if (g.K >= (1 << 20)) return fail(WDF_EINVAL, "too many chunks for a warm-start state");
        tickets = (unsigned*)((char*)state + sizeof(wdf::TpCtl));            // zeroed by the reset, left clean by every launch
        warm = TpWarm{(wdf::TpCtl*)state, (float*)((char*)tickets + tp_ticket_bytes(B)), max_warm_tiles + 1};
    } else {                                // (stateless: verified by the launch behind the forward, no tickets; one chunk: no boundaries --
        tickets = (unsigned*)(zend + (size_t)g.K * (size_t)B);                //  the last tile's ticket still counts the tiles)
        if (g.K <= 1) {
            const hipError_t e = hipMemsetAsync(tickets, 0, tp_ticket_bytes(B), (hipStream_t)stream);
            if (e != hipSuccess) return fail(WDF_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
        }
    }
    const bool tm = (flags & WDF_X_TIME_MAJOR) != 0;
    const bool v4 = !tm && (T % 4 == 0) && T < (1 << 23) && aligned16(x) && (!r || aligned16(r));
    WDF_DISPATCH4(launch_fwd_tp, r != nullptr, n_up == n_down, tm, v4, x, r, theta, fs, n_up, n_down, y, zstash, z0, zT,
                  zwarm, zend, (wdf::TpStatus*)status, tol, B, T, g, W, warm, tickets, (flags & WDF_GENERAL_ROOT) ? 1 : 0,
                  (hipStream_t)stream);
    return check_launch("wdf_clipper_fwd_tp");
}

int wdf_clipper_fwd_tp(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down, float* y,
                       float* zstash, const float* z0, float* zT, int64_t B, int64_t T, int n_chunks, int warmup,
                       float tol, void* ws, void* status, int flags, void* stream)
{
    return fwd_tp_common(x, r, theta, fs, n_up, n_down, y, zstash, z0, zT, B, T, n_chunks, warmup, tol, ws, status,
                         nullptr, 0, flags, stream);
}

size_t wdf_clipper_fwd_tp_state_bytes(int64_t B, int n_chunks, int max_warm_tiles)
{
    if (B <= 0 || n_chunks <= 0 || max_warm_tiles < 1 || max_warm_tiles > wdf::kTpMaxWarmTiles) return 0;
    return sizeof(wdf::TpCtl) + tp_ticket_bytes(B) +
           (size_t)wdf::kTpRing * (size_t)(max_warm_tiles + 1) * (size_t)n_chunks * (size_t)B * sizeof(float);
}

int wdf_clipper_fwd_tp_state_reset(void* state, int64_t B, int min_warm_tiles, void* stream)
{
    if (!state || B <= 0) return fail(WDF_EINVAL, "null state / bad B");
    if (min_warm_tiles < 0 || min_warm_tiles > wdf::kTpMaxWarmTiles) return fail(WDF_EINVAL, "min_warm_tiles must be in 0..%d", wdf::kTpMaxWarmTiles);
    hipError_t e = hipMemsetAsync(state, 0, sizeof(wdf::TpCtl) + tp_ticket_bytes(B), (hipStream_t)stream);
    if (e == hipSuccess && min_warm_tiles > 0)                                // TpCtl::j_floor is its last 32-bit word
        e = hipMemsetD32Async((hipDeviceptr_t)((char*)state + offsetof(wdf::TpCtl, j_floor)), min_warm_tiles, 1, (hipStream_t)stream);
    return e == hipSuccess ? WDF_OK : fail(WDF_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
}

int wdf_clipper_fwd_tp_warm(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down, float* y,
                            float* zstash, const float* z0, float* zT, int64_t B, int64_t T, int n_chunks, int warmup,
                            float tol, void* ws, void* status, void* state, int max_warm_tiles, int flags, void* stream)
{
    if (!state) return fail(WDF_EINVAL, "null state");
    const TpGeom g = tp_geom(T, n_chunks > 0 ? n_chunks : 1);
    if ((int64_t)max_warm_tiles * wdf::kWarmStep > g.L)
        return fail(WDF_EINVAL, "max_warm_tiles * 16 must not exceed the chunk length (%lld)", (long long)g.L);
    return fwd_tp_common(x, r, theta, fs, n_up, n_down, y, zstash, z0, zT, B, T, n_chunks, warmup, tol, ws, status, state,
                         max_warm_tiles, flags, stream);
}

static int bwd_tp_common(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                         const float* zstash, const float* gy, const float* target, const float* zT, float gscale,
                         void* ws, float* gtheta, float* sse, float* gz0, int accumulate, int64_t B, int64_t T,
                         int n_chunks, int flags, void* stream, const float* gcoef = nullptr, int64_t skip = 0,
                         wdf::AdamTail adam = wdf::AdamTail{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr});

size_t wdf_clipper_bwd_tp_ws_bytes(int64_t B, int n_chunks)
{
    if (B <= 0 || n_chunks <= 0) return 0;
    // [tiles][4] doubles, [K][9][B] floats, then the tickets (tiles done + one per tile)
    const size_t body = ((size_t)n_chunks * wdf::kTpOut * (size_t)B * sizeof(float) + wdf_clipper_bwd_ws_bytes(B) + 63) / 64 * 64;
    return body + bwd_ticket_bytes(B);
}

int wdf_clipper_bwd_tp_ws_init(void* ws, int64_t B, int n_chunks, void* stream)
{
    const size_t total = wdf_clipper_bwd_tp_ws_bytes(B, n_chunks);
    if (!ws || total == 0) return fail(WDF_EINVAL, "null ws / bad B, n_chunks");
    const hipError_t e = hipMemsetAsync((char*)ws + total - bwd_ticket_bytes(B), 0, bwd_ticket_bytes(B), (hipStream_t)stream);
    return e == hipSuccess ? WDF_OK : fail(WDF_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
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
    if (B >= ((int64_t)1 << 30)) return fail(WDF_EINVAL, "time-parallel kernels address a [B] row with 32-bit byte offsets: B < 2^30");
    if (flags & WDF_PREC_F64) return fail(WDF_EUNSUPPORTED, "WDF_PREC_F64 applies to wdf_clipper_fwd and wdf_omega_f64 only");
    const TpGeom g = tp_geom(T, n_chunks);
    double* wsd = (double*)ws;                                       // [nparts][4] doubles first (8-byte aligned)
    float* part = (float*)((char*)ws + wdf_clipper_bwd_ws_bytes(B)); // then [K][9][B] floats
    unsigned* ticket = (unsigned*)((char*)ws + wdf_clipper_bwd_tp_ws_bytes(B, n_chunks) - bwd_ticket_bytes(B));   // then the tickets
    const bool tm = (flags & WDF_X_TIME_MAJOR) != 0;
    const bool v4 = !tm && (T % 4 == 0) && aligned16(x) && (!r || aligned16(r));
    WDF_DISPATCH4(launch_bwd_tp, r != nullptr, n_up == n_down, tm, v4, x, r, theta, fs, n_up, n_down, zstash, gy, target,
                  zT, gscale, part, wsd, gz0, B, T, g, gcoef, skip,
                  ticket, gtheta, accumulate, target ? sse : nullptr, adam, (flags & WDF_GENERAL_ROOT) ? 1 : 0, (hipStream_t)stream);
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

size_t wdf_clipper_step_mse_tp_ws_bytes(int64_t B, int n_chunks)
{
    if (B <= 0 || n_chunks <= 0) return 0;
    return fused_body_bytes(B, n_chunks) + tp_ticket_bytes(B) + 64;      // (one layout for either loss)
}

int wdf_clipper_step_mse_tp_ws_init(void* ws, int64_t B, int n_chunks, void* stream)
{
    if (!ws || B <= 0 || n_chunks <= 0) return fail(WDF_EINVAL, "null ws / bad B, n_chunks");
    const size_t from = fused_body_bytes(B, n_chunks), total = wdf_clipper_step_mse_tp_ws_bytes(B, n_chunks);
    const hipError_t e = hipMemsetAsync((char*)ws + from, 0, total - from, (hipStream_t)stream);
    return e == hipSuccess ? WDF_OK : fail(WDF_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
}

static int step_tp_common(const float* x, const float* r, float* theta, float fs, int n_up, int n_down, const float* target,
                          float hgs, int64_t skip, float* y, const float* z0, float* zT, int64_t B, int64_t T, int n_chunks,
                          int warmup, float tol, void* ws, void* status, void* state, int max_warm_tiles, bool esr,
                          wdf::FusedOut out, int flags, void* stream, const char* what)
{
    int rc = check_common(x, theta, n_up, n_down, B, T, flags & ~(WDF_ONE_SEQUENCE_PER_LANE | WDF_R_PER_SEQUENCE));
    if (rc) return rc;
    if (!target || !y || !ws || !status) return fail(WDF_EINVAL, "null target/y/ws/status");
    if (!(fs > 0.0f)) return fail(WDF_EINVAL, "fs must be positive");
    if (n_chunks < 1 || warmup < 0 || !(tol >= 0.0f)) return fail(WDF_EINVAL, "n_chunks >= 1, warmup >= 0, tol >= 0");
    if (skip < 0 || skip > T) return fail(WDF_EINVAL, "skip must be in 0..T");
    if (B >= ((int64_t)1 << 24)) return fail(WDF_EINVAL, "the one-pass step addresses a 32-row tile with 32-bit offsets: B < 2^24");
    if (flags & WDF_PREC_F64) return fail(WDF_EUNSUPPORTED, "WDF_PREC_F64 applies to wdf_clipper_fwd and wdf_omega_f64 only");
    const TpGeom g = tp_geom(T, n_chunks);
    if (g.K != n_chunks) return fail(WDF_EINVAL, "n_chunks = %d does not tile T = %lld in 32-step units: use wdf_clipper_tp_chunks (%d)",
                                     n_chunks, (long long)T, g.K);
    const int64_t W = ((int64_t)warmup + wdf::kTile - 1) / wdf::kTile * wdf::kTile;
    TpWarm warm{nullptr, nullptr, 1};
    if (state) {
        if (max_warm_tiles < 1 || max_warm_tiles > wdf::kTpMaxWarmTiles)
            return fail(WDF_EINVAL, "max_warm_tiles must be in 1..%d", wdf::kTpMaxWarmTiles);
        if ((int64_t)max_warm_tiles * wdf::kWarmStep > g.L)
            return fail(WDF_EINVAL, "max_warm_tiles * 16 must not exceed the chunk length (%lld)", (long long)g.L);
        if (g.K >= (1 << 20)) return fail(WDF_EINVAL, "too many chunks for a warm-start state");
        // same layout as wdf_clipper_fwd_tp_warm's state: [TpCtl][its ticket area, unused here][snapshot ring]
        warm = TpWarm{(wdf::TpCtl*)state, (float*)((char*)state + sizeof(wdf::TpCtl) + tp_ticket_bytes(B)), max_warm_tiles + 1};
    }
    const bool tm = (flags & WDF_X_TIME_MAJOR) != 0;
    const bool v4 = !tm && (T % 4 == 0) && T < (1 << 23) && aligned16(x) && (!r || aligned16(r));
    // two adjacent sequences per lane (8-byte row accesses, packed arithmetic) whenever the rows allow it
    const bool pairs = !(flags & WDF_ONE_SEQUENCE_PER_LANE) && (B % 2 == 0) && aligned8(x) && aligned8(target) && aligned8(y) &&
                       (!r || aligned8(r));
    const int64_t skew = fused_skew((B + (pairs ? 127 : 63)) / (pairs ? 128 : 64) * (int64_t)g.K, g.K, g.L, T, state ? max_warm_tiles : 0);
    if (r != nullptr && (flags & WDF_R_PER_SEQUENCE)) {
        // one pot value per sequence (the caller vouches for it): calc_impedance once per chunk, the channel not streamed
#define WDF_FUSED_SEQ(SYM_, TM_, V4_)                                                                                             \
        launch_fused<2, SYM_, TM_, V4_>(x, r, theta, fs, n_up, n_down, target, hgs, skip, y, z0, zT, fused_ws(ws, B, g.K),          \
                                        (wdf::TpStatus*)status, tol, B, T, g, W, warm, (flags & WDF_GENERAL_ROOT) ? 1 : 0, pairs, esr, out, \
                                        skew, (hipStream_t)stream)
        const bool sym = n_up == n_down;
        if (tm) { if (sym) WDF_FUSED_SEQ(true, true, false); else WDF_FUSED_SEQ(false, true, false); }
        else if (v4) { if (sym) WDF_FUSED_SEQ(true, false, true); else WDF_FUSED_SEQ(false, false, true); }
        else { if (sym) WDF_FUSED_SEQ(true, false, false); else WDF_FUSED_SEQ(false, false, false); }
#undef WDF_FUSED_SEQ
        return check_launch(what);
    }
    WDF_DISPATCH4(launch_fused, r != nullptr, n_up == n_down, tm, v4, x, r, theta, fs, n_up, n_down, target, hgs, skip, y, z0, zT,
                  fused_ws(ws, B, g.K), (wdf::TpStatus*)status, tol, B, T, g, W, warm,
                  (flags & WDF_GENERAL_ROOT) ? 1 : 0, pairs, esr, out, skew, (hipStream_t)stream);
    return check_launch(what);
}

int wdf_clipper_step_mse_tp(const float* x, const float* r, float* theta, float fs, int n_up, int n_down,
                            const float* target, float gscale, int64_t skip, float* y, const float* z0, float* zT,
                            int64_t B, int64_t T, int n_chunks, int warmup, float tol, void* ws, void* status, void* state,
                            int max_warm_tiles, float* gtheta, float* sse, int accumulate, float* m, float* v, int32_t* step,
                            const float* lr, float beta1, float beta2, float eps, const float* lo, const float* hi, int flags,
                            void* stream)
{
    if (!gtheta || !sse) return fail(WDF_EINVAL, "null gtheta/sse");
    if (m && (!v || !step || !lr)) return fail(WDF_EINVAL, "Adam update asked for (m) but v/step/lr missing");
    const wdf::AdamTail adam{m ? theta : nullptr, m, v, step, lr, beta1, beta2, eps, lo, hi};
    const wdf::FusedOut out{gtheta, accumulate, sse, adam, 0.0, 0.0, nullptr, nullptr, wdf::TpFinishCtx{nullptr, nullptr, 0, nullptr, 0.0f, 0, 0, 0, false}};
    return step_tp_common(x, r, theta, fs, n_up, n_down, target, 0.5f * gscale, skip, y, z0, zT, B, T, n_chunks, warmup, tol, ws,
                          status, state, max_warm_tiles, false, out, flags, stream, "wdf_clipper_step_mse_tp");
}

int wdf_clipper_step_esr_tp(const float* x, const float* r, float* theta, float fs, int n_up, int n_down,
                            const float* target, double n_global, double eps_energy, int64_t skip, float* y, const float* z0,
                            float* zT, int64_t B, int64_t T, int n_chunks, int warmup, float tol, void* ws, void* status,
                            void* state, int max_warm_tiles, float* sums10, float* gtheta, float* loss3, float* m, float* v,
                            int32_t* step, const float* lr, float beta1, float beta2, float eps, const float* lo,
                            const float* hi, int flags, void* stream)
{
    if (!sums10) return fail(WDF_EINVAL, "null sums10");
    if (!(n_global > 0.0)) return fail(WDF_EINVAL, "n_global must be positive");
    if (m && (!gtheta || !v || !step || !lr)) return fail(WDF_EINVAL, "Adam update asked for (m) but gtheta/v/step/lr missing");
    const wdf::AdamTail adam{m ? theta : nullptr, m, v, step, lr, beta1, beta2, eps, lo, hi};
    const wdf::FusedOut out{gtheta, 0, nullptr, adam, n_global, eps_energy, sums10, loss3, wdf::TpFinishCtx{nullptr, nullptr, 0, nullptr, 0.0f, 0, 0, 0, false}};
    return step_tp_common(x, r, theta, fs, n_up, n_down, target, 0.5f, skip, y, z0, zT, B, T, n_chunks, warmup, tol, ws, status,
                          state, max_warm_tiles, true, out, flags, stream, "wdf_clipper_step_esr_tp");
}

int wdf_esr_finish(const float* sums10, double n_global, double eps_energy, float* gtheta, float* loss3, void* stream)
{
    if (!sums10 || !gtheta || !(n_global > 0.0)) return fail(WDF_EINVAL, "wdf_esr_finish: null sums10/gtheta or n_global <= 0");
    hipLaunchKernelGGL(esr_finish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sums10, n_global, eps_energy, gtheta, loss3);
    return check_launch("wdf_esr_finish");
}

int wdf_omega_f64(const double* x, double* w, int32_t* iters, int64_t n, void* stream)
{
    if (!x || !w || n <= 0) return fail(WDF_EINVAL, "wdf_omega_f64: bad arguments");
    hipLaunchKernelGGL(wdf::omega64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, iters, n);
    return check_launch("wdf_omega_f64");
}

}  // extern "C"
