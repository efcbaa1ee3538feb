// wdf_capi.hip -- the C ABI of libwdf_hip.so (include/wdf_hip.h), part 1 of 4: library state,
// element-wise building blocks, loss sums, optimizer, events.  gfx950 only.
//
// Argument checking, template dispatch and launches; no algorithm lives here.
// (clipper kernels: wdf_capi_clipper.hip; state-space and asym root: wdf_capi_ss.hip; MLP root: wdf_capi_mlp.hip)
#include "wdf_capi_common.h"
#include "wdf_elementwise.h"
#include "wdf_optim.h"
using namespace wdfcapi;

namespace wdfcapi {
thread_local char g_err[512] = "";
std::atomic<EventPair*> g_bracket{nullptr};
}

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

int wdf_loss_esr_grad(const float* y, const float* target, const float* gcoef, int64_t B, int64_t T, int64_t skip,
                      float* gy, void* stream)
{
    if (!y || !target || !gcoef || !gy) return fail(WDF_EINVAL, "null y/target/gcoef/gy");
    if (B <= 0 || T <= 0 || skip < 0 || skip >= T) return fail(WDF_EINVAL, "need B, T > 0 and 0 <= skip < T");
    hipLaunchKernelGGL(wdf::loss_esr_grad_kernel, dim3(loss_blocks(T * B)), dim3(256), 0, (hipStream_t)stream, y, target,
                       gcoef, skip * B, T * B, gy);
    return check_launch("wdf_loss_esr_grad");
}

int wdf_esr_coef(const double* sums, double n, double eps, float* gcoef, float* loss, void* stream)
{
    if (!sums || !gcoef || !loss) return fail(WDF_EINVAL, "null sums/gcoef/loss");
    if (!(n > 0.0)) return fail(WDF_EINVAL, "n must be positive");
    hipLaunchKernelGGL(wdf::esr_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sums, n, eps, gcoef, loss);
    return check_launch("wdf_esr_coef");
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

int wdf_adam_step_multi(const wdf_adam_job* jobs, int n_jobs, void* stream)
{
    if (!jobs) return fail(WDF_EINVAL, "null jobs");
    if (n_jobs < 1 || n_jobs > WDF_ADAM_MULTI_MAX) return fail(WDF_EINVAL, "wdf_adam_step_multi: 1..%d jobs (got %d)", WDF_ADAM_MULTI_MAX, n_jobs);
    wdf::AdamJobs a{};
    for (int i = 0; i < n_jobs; ++i) {
        const wdf_adam_job& j = jobs[i];
        if (!j.theta || !j.grad || !j.m || !j.v || !j.step || !j.lr) return fail(WDF_EINVAL, "job %d: null theta/grad/m/v/step/lr", i);
        if (j.n <= 0 || j.n > 1024) return fail(WDF_EINVAL, "job %d: n must be in 1..1024 (got %d)", i, j.n);
        a.j[i] = wdf::AdamJob{j.theta, j.grad, j.m, j.v, j.step, j.lr, j.lo, j.hi, j.beta1, j.beta2, j.eps, j.n};
    }
    hipLaunchKernelGGL(wdf::adam_clip_multi_kernel, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("wdf_adam_step_multi");
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
    delete g_bracket.exchange(new EventPair{(hipEvent_t)start, (hipEvent_t)stop});   // (re-arming drops a pair nobody consumed)
}

// out[0] = the shader clock counter (s_memtime), out[1] = the constant-rate counter (s_memrealtime, 100 MHz), read by a
// one-lane kernel on `stream`: two stamps around a stretch of work give the clock the chip ran at while doing it.
static __global__ void clock_stamp_kernel(unsigned long long* out)
{
    out[0] = (unsigned long long)clock64();
    out[1] = (unsigned long long)wall_clock64();
}

int wdf_clock_stamp(uint64_t* out, void* stream)
{
    if (!out) return fail(WDF_EINVAL, "null out");
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)out);
    return check_launch("wdf_clock_stamp");
}

void wdf_event_destroy(void* ev)
{
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}

}  // extern "C"
