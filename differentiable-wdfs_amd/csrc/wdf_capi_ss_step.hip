// wdf_capi_ss_step.hip -- C ABI part 6 of 6: the one-pass MSE step of small state-space trees with a diode-pair root
// (csrc/wdf_ss_nl_step.h): workspace layout, plan, template dispatch and the two launches of a step.
#include "wdf_capi_common.h"
#include "wdf_ss_nl_step.h"
using namespace wdfcapi;

namespace {

bool nl_ok(int ns, int ni) { return ns >= 1 && ns <= 2 && ni >= 1 && ni <= 2; }
int nl_kn(int ns, int ni) { return ns * ns + ns * ni + ns + ns + ni + ns + ni + 1; }
int nl_nt(int ns, int ni) { return ns * ns + ns * ni + ns + ns + ni + 2; }
int nl_nrec(int ns, int ni) { return 2 * ns + ns * ns + nl_nt(ns, ni) * ns + ns; }
int nl_nsnap(int ns, int ni) { return ns + nl_nt(ns, ni) * ns + ns * ns; }

void nl_geom(int64_t T, int n_chunks, int64_t& L, int& K)
{
    if (n_chunks < 1) n_chunks = 1;
    L = (T + n_chunks - 1) / n_chunks;
    L = (L + 31) / 32 * 32;
    K = (int)((T + L - 1) / L);
}

size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct NlLayout {
    size_t ctl, ticket, coef_prev, rec, snap, gpart, part, total;
    int K, groups_max;
    int64_t L;
};

// (sized for one sequence per lane: a call that pairs them up uses half the groups)
NlLayout nl_layout(int ns, int ni, int64_t B, int64_t T, int n_chunks)
{
    NlLayout l{};
    nl_geom(T, n_chunks, l.L, l.K);
    l.groups_max = (int)((B + 63) / 64);
    const size_t ng1 = (size_t)nl_kn(ns, ni) + 3;
    l.ctl = 0;
    l.ticket = 128;
    l.coef_prev = 256;
    l.rec = 512;
    l.snap = up(l.rec + (size_t)l.K * nl_nrec(ns, ni) * (size_t)B * sizeof(float), 256);
    l.gpart = up(l.snap + (size_t)2 * l.K * nl_nsnap(ns, ni) * (size_t)B * sizeof(float), 256);
    l.part = up(l.gpart + (size_t)l.K * l.groups_max * ng1 * sizeof(double), 256);
    l.total = up(l.part + (size_t)l.groups_max * (ng1 + 2) * sizeof(double), 256);   // (+ bad boundaries, largest miss)
    return l;
}

__global__ void nl_plan_kernel(wdf::NlStepCtl* ctl, unsigned* ticket, int cold, int warm, int w_min, int w_max, float tol)
{
    wdf::NlStepCtl c = {};
    c.w_cur = cold;
    c.w_snap = warm;
    c.w_min = w_min;
    c.w_max = w_max;
    c.tol = tol;
    c.grow_at = 0.5f;
    c.shrink_at = 0.125f;
    c.cool_miss = 32;
    *ctl = c;
    *ticket = 0u;
}

__global__ void nl_set_kernel(wdf::NlStepCtl* ctl, int field, double v)
{
    switch (field) {
    case 3: ctl->w_cur = (int)v; break;
    case 4: ctl->w_snap = (int)v; break;
    case 5: ctl->w_min = (int)v; break;
    case 6: ctl->w_max = (int)v; break;
    case 7: ctl->cool = (int)v; break;
    case 8: ctl->tol = (float)v; break;
    case 9: ctl->grow_at = (float)v; break;
    case 10: ctl->shrink_at = (float)v; break;
    case 11: ctl->cool_miss = (int)v; break;
    case 2: ctl->have_snap = (int)v; break;
    default: break;
    }
}

}  // namespace

extern "C" {

size_t wdf_ss_nl_step_ws_bytes(int ns, int ni, int64_t B, int64_t T, int n_chunks)
{
    if (!nl_ok(ns, ni) || B <= 0 || T <= 0 || n_chunks < 1) return 0;
    return nl_layout(ns, ni, B, T, n_chunks).total;
}

int wdf_ss_nl_step_chunk_len(int64_t T, int n_chunks)
{
    if (T <= 0 || n_chunks < 1) return 0;
    int64_t L; int K;
    nl_geom(T, n_chunks, L, K);
    return (int)L;
}

// cold_warmup: the warm-up of the first call (chunks start from z = 0); warm_warmup: where that call takes the snapshots the
// second one starts from; afterwards the device steers it inside [w_min, min(w_max, chunk length)].  All multiples of 8.
int wdf_ss_nl_step_plan(void* ws, int ns, int ni, int64_t B, int64_t T, int n_chunks, int cold_warmup, int warm_warmup, int w_min,
                        int w_max, float tol, void* stream)
{
    if (!ws) return fail(WDF_EINVAL, "null argument");
    if (!nl_ok(ns, ni)) return fail(WDF_EUNSUPPORTED, "wdf_ss_nl_step_plan: ns in 1..2, ni in 1..2 (got %d, %d)", ns, ni);
    if (B <= 0 || T <= 0 || n_chunks < 1) return fail(WDF_EINVAL, "B, T, n_chunks >= 1");
    const NlLayout l = nl_layout(ns, ni, B, T, n_chunks);
    if (cold_warmup < 0 || cold_warmup % 8 || warm_warmup < 8 || warm_warmup % 8 || w_min < 8 || w_min % 8 || w_max < w_min ||
        w_max % 8 || warm_warmup > l.L || !(tol > 0.0f))
        return fail(WDF_EINVAL, "wdf_ss_nl_step_plan: warm-ups are multiples of 8, 8 <= w_min <= w_max, warm_warmup <= the chunk length %lld, tol > 0",
                    (long long)l.L);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, l.rec, s) != hipSuccess) return fail(WDF_ELAUNCH, "wdf_ss_nl_step_plan: memset failed");
    hipLaunchKernelGGL(nl_plan_kernel, dim3(1), dim3(1), 0, s, (wdf::NlStepCtl*)ws, (unsigned*)((char*)ws + l.ticket), cold_warmup,
                       warm_warmup, w_min, w_max, tol);
    return check_launch("wdf_ss_nl_step_plan");
}

int wdf_ss_nl_step_set(void* ws, int field, double value, void* stream)
{
    if (!ws) return fail(WDF_EINVAL, "null argument");
    if (field < 2 || field > 11) return fail(WDF_EINVAL, "wdf_ss_nl_step_set: field 2..11");
    hipLaunchKernelGGL(nl_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (wdf::NlStepCtl*)ws, field, value);
    return check_launch("wdf_ss_nl_step_set");
}

// ctl_out: 32 words (NlStepCtl, csrc/wdf_ss_nl_step.h).  Synchronises the stream.
int wdf_ss_nl_step_read(const void* ws, int32_t* ctl_out, void* stream)
{
    if (!ws || !ctl_out) return fail(WDF_EINVAL, "null argument");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(ctl_out, ws, sizeof(wdf::NlStepCtl), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return fail(WDF_ELAUNCH, "wdf_ss_nl_step_read: copy failed");
    return WDF_OK;
}

// x: [T][ni][B] time-major; coef: the probe's outputs (SSCoef order, then the port resistance the root sees); params: the
// component values on the device, the root's Is and nVt at [n_tree], [n_tree + 1]; jac: double [ncoef + 1][n_tree];
// target, y: [T][B]; ws: planned by wdf_ss_nl_step_plan.
// out: float [1 + n_tree + 2] = {sum of squared errors, d(gscale / 2 x that sum) / d{component values, Is, nVt}}; loss_out (or NULL)
// <- gscale / 2 x that sum.
int wdf_ss_nl_step_mse(const float* x, const float* coef, const float* params, const double* jac, int n_tree, int ns, int ni, int n_up,
                       int n_down, const float* target, float gscale, float* y, void* ws, float* out, float* loss_out, int64_t B, int64_t T,
                       int n_chunks, void* stream)
{
    if (!x || !coef || !params || !jac || !target || !y || !ws || !out) return fail(WDF_EINVAL, "null argument");
    if (!nl_ok(ns, ni)) return fail(WDF_EUNSUPPORTED, "wdf_ss_nl_step_mse: ns in 1..2, ni in 1..2 (got %d, %d)", ns, ni);
    if (B <= 0 || T <= 0 || n_chunks < 1 || n_tree < 1 || n_tree > wdf::kProbeMaxParams || n_up < 1 || n_down < 1)
        return fail(WDF_EINVAL, "B, T, n_chunks, n_up, n_down >= 1, 1..7 component values");
    const NlLayout l = nl_layout(ns, ni, B, T, n_chunks);
    const bool pair = (B % 2 == 0) && aligned8(x) && aligned8(target) && aligned8(y) && aligned8(ws);
    const bool sym = n_up == n_down;
    wdf::NlStepArgs a{};
    a.x = x; a.coef = coef; a.pIs = params + n_tree; a.pV = params + n_tree + 1; a.pRp = coef + nl_kn(ns, ni);
    a.target = target; a.y = y;
    a.ctl = (wdf::NlStepCtl*)ws;
    a.ticket = (unsigned*)((char*)ws + l.ticket);
    a.coef_prev = (float*)((char*)ws + l.coef_prev);
    a.rec = (float*)((char*)ws + l.rec);
    a.snap = (float*)((char*)ws + l.snap);
    a.gpart = (double*)((char*)ws + l.gpart);
    a.part = (double*)((char*)ws + l.part);
    a.jac = jac; a.out = out; a.loss = loss_out;
    a.B = B; a.T = T; a.L = l.L; a.K = l.K;
    a.groups = (int)((B + (pair ? 127 : 63)) / (pair ? 128 : 64));
    a.n_tree = n_tree; a.n_up = n_up; a.n_down = n_down; a.gscale = gscale;
    const int64_t units = (int64_t)a.groups * a.K;
    const dim3 grid((unsigned)((units + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
#define WDF_NL_V(NS_, NI_, SYM_, V_, WD_)                                                                          \
    {                                                                                                              \
        {                                                                                                          \
            EventBracket bracket(s);                                                                               \
            hipLaunchKernelGGL((wdf::ss_nl_step_kernel<NS_, NI_, SYM_, V_>), grid, dim3(256), 0, s, a);            \
        }                                                                                                          \
        hipLaunchKernelGGL((wdf::ss_nl_step_finish_kernel<NS_, NI_, SYM_, WD_>), dim3(a.groups), dim3(64 * WD_ * wdf::NlTile<NS_, WD_>::n), 0, s, a); \
    }
#define WDF_NL(NS_, NI_)                                                                                           \
    if (ns == NS_ && ni == NI_) {                                                                                  \
        if (sym) { if (pair) WDF_NL_V(NS_, NI_, true, wdf::v2f, 2) else WDF_NL_V(NS_, NI_, true, float, 1) }       \
        else { if (pair) WDF_NL_V(NS_, NI_, false, wdf::v2f, 2) else WDF_NL_V(NS_, NI_, false, float, 1) }         \
    }
    WDF_NL(1, 1) WDF_NL(1, 2) WDF_NL(2, 1) WDF_NL(2, 2)
#undef WDF_NL
#undef WDF_NL_V
    return check_launch("wdf_ss_nl_step_mse");
}

}  // extern "C"
