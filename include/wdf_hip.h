/*
 * wdf_hip.h -- C ABI of libwdf_hip.so: the MI355X (gfx950) engine for the wdf_py hot path.
 *
 * The reference has no FFI for this path: its boundary is the Python module surface of
 * wdf_py/lib (tf_wdf.py, layers.py) plus the per-sample loops the scripts own
 * (lpf.py:30-49, clipper_pot.py:103-127).  The drop-in keeps that Python surface
 * (differentiable-wdfs_amd/lib/tf_wdf.py, layers.py) and lowers ONE WHOLE SEQUENCE LOOP
 * per call to the entry points below.  Each entry point cites the reference code it
 * replaces.  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (torch allocates it in the Python host layer); the library allocates nothing
 *     persistent and keeps no state besides the per-thread error string;
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream), so the caller's stream / RCCL ordering applies unchanged;
 *   - return 0 on success, a negative WDF_E* code on error (never throws, never exits);
 *     wdf_last_error() describes the last failure on the calling thread;
 *   - sequences are independent: lane b of the grid owns sequence b.  Inputs are batch-major
 *     [B][T] as the reference scripts index them (input[:, i]) -- or time-major [T][B] with
 *     WDF_X_TIME_MAJOR -- outputs are time-major [T][B] as TensorArray.stack() returns them
 *     (lpf.py:48, clipper_pot.py:126).
 */
#ifndef WDF_HIP_H
#define WDF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: exactly the functions declared between this push and the pop at the end of
 * the header are exported (tests/test_cabi_cpu.py compares `nm -D` with this header both ways). */
#pragma GCC visibility push(default)

#define WDF_HIP_ABI_VERSION 6   /* 6: wdf_ss_lin_step_mse takes z0 / zT (round 6); exports limited to this header.  5: + wdf_clipper_asym_bwd_tp, wdf_ss_dyn_* (round 5).  3: + wdf_clipper_mlp_step_* (round 4); TpCtl is 128 bytes, warm-up units are 16 steps.  4: + wdf_ss_nl_step_*; the linear step's workspace shrank */

enum {
    WDF_OK = 0,
    WDF_EINVAL = -1,   /* bad argument (null pointer, size <= 0, unknown enum)          */
    WDF_ELAUNCH = -2,  /* HIP launch / runtime error (message has hipGetErrorString)     */
    WDF_EUNSUPPORTED = -3
};

/* flags */
enum {
    WDF_X_TIME_MAJOR = 1 << 0, /* x (and r) are [T][B] instead of [B][T]                 */
    WDF_PREC_F64     = 1 << 1, /* wdf_clipper_fwd and wdf_clipper_bwd (the sequential pair; static or per-sample R): tree,
                                  root (Wright omega with toms917's second-iteration test, csrc/wdf_omega64.h) and
                                  adjoint in fp64, I/O stays f32 -- the on-device accuracy reference of config C5,
                                  not a fast path; every other entry point answers WDF_EUNSUPPORTED */
    WDF_MLP_LANE_PER_SEQUENCE = 1 << 4, /* MLP-root kernels: the one-lane-per-sequence variant (csrc/wdf_mlp.h)
                                  instead of the default 16-lane row per sequence (csrc/wdf_mlp_row.h) */
    WDF_GENERAL_ROOT = 1 << 3, /* always take the general per-step root evaluation (the kernels otherwise
                                  switch, once per launch, to a shorter step when the static port
                                  resistance keeps omega_1 in its series-only region); for parity tests
                                  and A/B timing -- results agree to fp32 rounding */
    WDF_R_PER_SEQUENCE = 1 << 6, /* wdf_clipper_step_mse_tp / _esr_tp with a resistance channel r (round 6): the caller vouches that r is
                                  constant along every sequence (the reference's recordings: one pot value per file,
                                  dataimport.py:96) -- calc_impedance is then evaluated once per chunk from each sequence's
                                  first sample and the channel is not streamed; results as with the per-sample evaluation */
    WDF_ONE_SEQUENCE_PER_LANE = 1 << 5 /* wdf_clipper_step_mse_tp: one sequence per lane even when the batch is even
                                  (by default a lane then runs two adjacent sequences with packed fp32
                                  arithmetic); for parity tests and A/B timing */
};

/* ------------------------------------------------------------------------------------
 * Diode clipper:  P1 = Parallel(ResistiveVoltageSource(R), Capacitor(C, fs)), root =
 * diode pair  b = a - 2 nVt lam (mu0 w(log(Rp Is/(mu0 nVt)) + lam a/(mu0 nVt))
 *                               - mu1 w(log(Rp Is/(mu1 nVt)) - lam a/(mu1 nVt)))
 *
 * Replaces, per call, the whole loop  clipper_pot.py:103-127  run with the analytic root
 * (DiodeClipperWDF.cpp:24-28 order: dp.incident(P1.reflected()); P1.incident(dp.reflected())):
 *     Capacitor.calc_impedance  tf_wdf.py:114-115     Parallel.calc_impedance  tf_wdf.py:168-177
 *     Parallel.reflected        tf_wdf.py:185-192     Parallel.incident        tf_wdf.py:179-183
 *     ResistiveVoltageSource.*  tf_wdf.py:31-59       Capacitor.incident/reflected :120-126
 *     voltage(C)                tf_wdf.py:8-10        diode pair  diode_pretraining.py:39-60,
 *     Wright omega              toms917.cpp:134-375   Toms917DiodePair.h:28-59
 *
 * theta   device float[4] = {Is, nVt, R, C}  (read on the device: no host sync, so an
 *         optimizer can update it in place between calls)
 * x       [B][T] input voltage (clipper_pot.py:114); r: optional [B][T] per-sample source
 *         resistance (clipper_pot.py:116, overrides theta[2]); NULL = scalar R
 * y       [T][B] capacitor voltage (clipper_pot.py:123-124)
 * zstash  optional [T][B]: capacitor state BEFORE each step, kept for wdf_clipper_bwd
 * z0      optional [B] initial capacitor state (NULL = reset(), clipper_pot.py:110-111)
 * zT      optional [B] final capacitor state
 * ---------------------------------------------------------------------------------- */
int wdf_clipper_fwd(const float* x, const float* r, const float* theta,
                    float fs, int n_up, int n_down,
                    float* y, float* zstash, const float* z0, float* zT,
                    int64_t B, int64_t T, int flags, void* stream);

/* Reverse sweep of the same loop: what tape.gradient(loss, [Is, nVt, R, C]) computes in the
 * reference's execution model (clipper_pot.py:246-268), for a given dL/dy.
 * gy      [T][B] = dL/dy
 * ws      workspace of wdf_clipper_bwd_ws_bytes(B) bytes
 * gtheta  device float[4] = dL/d{Is, nVt, R, C}; entry 2 is 0 when r != NULL.
 *         Overwritten (not accumulated) unless accumulate != 0.
 * gz0     optional [B]: dL/d z0 (adjoint of the initial state)
 * gzT     optional [B]: dL/d zT, for a loss that also reads the final state wdf_clipper_fwd
 *         returned (a state carried into the next call under one tape); NULL = 0            */
int wdf_clipper_bwd(const float* x, const float* r, const float* theta,
                    float fs, int n_up, int n_down,
                    const float* zstash, const float* gy,
                    void* ws, float* gtheta, float* gz0, const float* gzT, int accumulate,
                    int64_t B, int64_t T, int flags, void* stream);

size_t wdf_clipper_bwd_ws_bytes(int64_t B);

/* ------------------------------------------------------------------------------------
 * Time-parallel variants of the two calls above (same results, more waves in flight; see
 * csrc/wdf_clipper.h "Time-parallel variants").  The time axis is cut into n_chunks chunks
 * (chunk length and warm-up rounded up to multiples of 32; wdf_clipper_tp_chunks() returns the number
 * actually used).  x may be batch-major [B][T] or, with WDF_X_TIME_MAJOR, [T][B] (an engine
 * that keeps its training inputs resident in that layout gets fully coalesced loads).
 *
 * wdf_clipper_bwd_tp: EXACT -- the adjoint recurrence is linear; the chunk results of a 64-sequence
 *   tile are combined by the last of its chunk waves to finish, the tiles by the last tile (one
 *   launch); only the (fixed) summation order differs from wdf_clipper_bwd.
 * wdf_clipper_fwd_tp: every chunk starts `warmup` steps early from z = 0; the state it
 *   arrives with is checked on the device against the previous chunk's final state
 *   (|diff| <= tol for every sequence and chunk) by the last wave of each 64-sequence tile to
 *   finish (no second launch), which, where a check fails, re-runs that chunk for its 64
 *   sequences from the correct state on the spot.
 *   Either way the outputs are within tol of wdf_clipper_fwd's, with no host round trip.
 *   status: device int32[4] = {n_bad pairs, max |miss| (float bits), chunk re-runs, 0},
 *   written by the call's last wave; the caller may read it later to report / adapt `warmup`.
 * wdf_clipper_fwd_tp_warm: the same call with a persistent `state` buffer
 *   (wdf_clipper_fwd_tp_state_bytes; wdf_clipper_fwd_tp_state_reset before the first use and
 *   whenever x, B, T or the chunking change; min_warm_tiles: a floor for the device controller,
 *   min = max pins the warm-up).  The reference's training loop runs the same
 *   train_X through the circuit every epoch with slowly moving parameters
 *   (clipper_pot.py:245-269): each call leaves snapshots of every chunk's state near its end,
 *   the next call starts its chunks from them (extrapolated along the parameter path from the
 *   last two calls) and runs only the few warm-up tiles the measured boundary miss asks
 *   for -- steered on the device, between 0 and max_warm_tiles units of wdf_clipper_tp_warm_unit()
 *   steps (16; max_warm_tiles <= 32; the first call after a reset is a cold one with `warmup`).
 *   Same verification, same guarantee.
 * ---------------------------------------------------------------------------------- */
int wdf_clipper_tp_chunks(int64_t T, int n_chunks);
int wdf_clipper_tp_warm_unit(void);    /* steps per warm-start unit ("warm tile") */
size_t wdf_clipper_fwd_tp_ws_bytes(int64_t B, int n_chunks);
int wdf_clipper_fwd_tp(const float* x, const float* r, const float* theta,
                       float fs, int n_up, int n_down,
                       float* y, float* zstash, const float* z0, float* zT,
                       int64_t B, int64_t T, int n_chunks, int warmup, float tol,
                       void* ws, void* status, int flags, void* stream);
size_t wdf_clipper_fwd_tp_state_bytes(int64_t B, int n_chunks, int max_warm_tiles);
int wdf_clipper_fwd_tp_state_reset(void* state, int64_t B, int min_warm_tiles, void* stream);
int wdf_clipper_fwd_tp_warm(const float* x, const float* r, const float* theta,
                            float fs, int n_up, int n_down,
                            float* y, float* zstash, const float* z0, float* zT,
                            int64_t B, int64_t T, int n_chunks, int warmup, float tol,
                            void* ws, void* status, void* state, int max_warm_tiles,
                            int flags, void* stream);
size_t wdf_clipper_bwd_tp_ws_bytes(int64_t B, int n_chunks);
/* The reverse-sweep workspace ends in a few ticket words that the kernels count in and leave at zero:
 * call this once after allocating a workspace (and after a launch that was aborted) -- one small
 * memset on `stream` -- not per step. */
int wdf_clipper_bwd_tp_ws_init(void* ws, int64_t B, int n_chunks, void* stream);
/* MSE-fused reverse sweep (lpf.py:78 / clipper_pot.py:176 loss): `target` [T][B] is the
 * training target and zT [B] the forward's final state; the kernel rebuilds y[n] =
 * (z[n+1] + z[n])/2 from the state stash, forms dL/dy = gscale (y - target) itself
 * (gscale = 2/N for a mean over N samples, N the GLOBAL sample count under data
 * parallelism) and *sse (device float) receives sum (y - target)^2 over this call's batch.
 * It reads exactly x, the stash and the target: 12 B/sample. */
int wdf_clipper_bwd_mse_tp(const float* x, const float* r, const float* theta,
                           float fs, int n_up, int n_down,
                           const float* zstash, const float* zT, const float* target, float gscale,
                           void* ws, float* gtheta, float* sse, float* gz0, int accumulate,
                           int64_t B, int64_t T, int n_chunks, int flags, void* stream);

/* wdf_clipper_bwd_mse_tp followed, in the same launches, by the Adam update of theta = {Is, nVt, R, C}
 * (wdf_adam_step's rule and arguments, n = 4): the whole tail of a single-GPU training step --
 * combine, reduce, chain rule, update -- rides in the sweep kernel's last waves.  With several ranks the
 * gradient all-reduce sits between sweep and update: use wdf_clipper_bwd_mse_tp + wdf_adam_step. */
int wdf_clipper_bwd_mse_tp_adam(const float* x, const float* r, float* theta, float fs, int n_up, int n_down,
                                const float* zstash, const float* zT, const float* target, float gscale,
                                void* ws, float* gtheta, float* sse, int64_t B, int64_t T, int n_chunks,
                                int flags, float* m, float* v, int32_t* step, const float* lr,
                                float beta1, float beta2, float eps, const float* lo, const float* hi,
                                void* stream);

/* The whole MSE training step in ONE pass over the data (csrc/wdf_clipper_fused.h): forward, loss and the
 * gradient d(mean squared error)/d{Is, nVt, R, C} -- what the reference obtains from the sequence loop under
 * tf.GradientTape + tape.gradient (lpf.py:30-49,87-90; clipper_pot.py:103-127,246-269) -- with x and `target`
 * [T][B] read once and y [T][B] written once: 12 B/sample, no state stash, one root solve per sample.  The
 * gradient is carried forward in time as the state's tangent (three statistics), chunks are time-parallel and
 * verified like wdf_clipper_fwd_tp_warm's (same `state` buffer and meaning of n_chunks / warmup / tol / status /
 * max_warm_tiles; state may be NULL: cold warm-up every call), the combine, reduction, chain rule and -- when
 * m is not NULL -- the Adam update of theta (wdf_adam_step's rule) ride in the kernel's last waves.
 * dL/dy = gscale (y - target) on steps >= skip (gscale = 2/N, N the GLOBAL number of samples past skip);
 * *sse receives sum (y - target)^2 over those steps of this call's batch, gtheta[4] the gradient.
 * n_chunks must be a value wdf_clipper_tp_chunks returns.  ws: wdf_clipper_step_mse_tp_ws_bytes bytes,
 * initialised ONCE with wdf_clipper_step_mse_tp_ws_init (ticket words the kernels count in and leave at zero). */
size_t wdf_clipper_step_mse_tp_ws_bytes(int64_t B, int n_chunks);
int wdf_clipper_step_mse_tp_ws_init(void* ws, int64_t B, int n_chunks, void* stream);
int wdf_clipper_step_mse_tp(const float* x, const float* r, float* theta, float fs, int n_up, int n_down,
                            const float* target, float gscale, int64_t skip, float* y, const float* z0, float* zT,
                            int64_t B, int64_t T, int n_chunks, int warmup, float tol, void* ws, void* status, void* state,
                            int max_warm_tiles, float* gtheta, float* sse, int accumulate, float* m, float* v, int32_t* step,
                            const float* lr, float beta1, float beta2, float eps, const float* lo, const float* hi, int flags,
                            void* stream);

/* The same one-pass step for the scripts' training loss, MSE + ESR on the samples past `skip` (loss_func = MSE + esr_loss with
 * the scripts' argument order, clipper_pot.py:146-156,177,232,248):  loss = S/n + sqrt(S / (E + eps_energy) / n),
 * S = sum (y - target)^2, E = sum y^2, n = n_global.  dLoss/dy = ga (y - target) + gb y needs the GLOBAL S and E, so the pass
 * carries both tangent-weighted sums and hands back sums10 = {S, E, gP[4], gQ[4]} of this call's batch (gP = d(S/2)/dtheta,
 * gQ = d(E/2)/dtheta): with several ranks all-reduce the ten floats and call wdf_esr_finish (-> gtheta = ga gP + gb gQ and
 * loss3 = {mse, esr, mse + esr}), then wdf_adam_step.  With gtheta != NULL the kernel finishes the step itself as a single
 * rank (same formulas; loss3 optional; Adam when m != NULL).  Workspace and state as wdf_clipper_step_mse_tp (same sizes). */
int wdf_clipper_step_esr_tp(const float* x, const float* r, float* theta, float fs, int n_up, int n_down,
                            const float* target, double n_global, double eps_energy, int64_t skip, float* y, const float* z0,
                            float* zT, int64_t B, int64_t T, int n_chunks, int warmup, float tol, void* ws, void* status,
                            void* state, int max_warm_tiles, float* sums10, float* gtheta, float* loss3, float* m, float* v,
                            int32_t* step, const float* lr, float beta1, float beta2, float eps, const float* lo,
                            const float* hi, int flags, void* stream);
int wdf_esr_finish(const float* sums10, double n_global, double eps_energy, float* gtheta, float* loss3, void* stream);

/* MSE + ESR, the training loss of clipper_pot.py (:146-156 esr_loss, :177 loss_func, :232,248
 * evaluated past skip_samples with (outs, target) passed as (target_y, predicted_y), so the energy
 * is the model output's):   loss = S/n + sqrt(S / (E + eps) / n),  S = sum (y-t)^2, E = sum y^2.
 *   wdf_loss_sums   S and E over this rank's y, target [T][B] rows skip..T-1 -> sums[2] (device
 *                   double; all-reduce them when the batch is sharded); ws: wdf_loss_sums_ws_bytes()
 *   wdf_esr_coef    sums, n (global sample count), eps -> gcoef[2] = {ga, gb} with
 *                   dL/dy = ga (y - t) + gb y, and loss[3] = {mse, esr, mse + esr}
 *   wdf_clipper_bwd_esr_tp   the reverse sweep of wdf_clipper_bwd_mse_tp with that dL/dy (rows
 *                   before `skip` contribute nothing); sse receives this rank's S.              */
int64_t wdf_loss_sums_ws_bytes(void);
int wdf_loss_sums(const float* y, const float* target, int64_t B, int64_t T, int64_t skip, void* ws,
                  double* sums, void* stream);
int wdf_esr_coef(const double* sums, double n, double eps, float* gcoef, float* loss, void* stream);
/* dL/dy [T][B] of that loss as an array, for a reverse sweep that takes an upstream gradient (wdf_clipper_mlp_bwd_w*):
 * rows before skip 0, then gcoef[0] (y - target) + gcoef[1] y; gcoef on the device (wdf_esr_coef's output). */
int wdf_loss_esr_grad(const float* y, const float* target, const float* gcoef, int64_t B, int64_t T, int64_t skip,
                      float* gy, void* stream);
int wdf_clipper_bwd_esr_tp(const float* x, const float* r, const float* theta, float fs, int n_up, int n_down,
                           const float* zstash, const float* zT, const float* target,
                           const float* gcoef, int64_t skip, void* ws, float* gtheta, float* sse,
                           float* gz0, int accumulate, int64_t B, int64_t T, int n_chunks,
                           int flags, void* stream);
int wdf_clipper_bwd_tp(const float* x, const float* r, const float* theta,
                       float fs, int n_up, int n_down,
                       const float* zstash, const float* gy,
                       void* ws, float* gtheta, float* gz0, int accumulate,
                       int64_t B, int64_t T, int n_chunks, int flags, void* stream);

/* ------------------------------------------------------------------------------------
 * Diode clipper with the tanh-MLP root of clipper_pot.py (csrc/wdf_mlp_row.h: one 16-lane row per
 * sequence, the default; csrc/wdf_mlp.h: one lane per sequence, flags = WDF_MLP_LANE_PER_SEQUENCE;
 * same results to fp32 rounding).  Replaces, per call,
 * ClipperModel.forward's loop (clipper_pot.py:103-127): P1 = Parallel(Vs, C), model_in =
 * (P1.reflected(), log P1.R) (:119), P1.incident(-model.reflected()) (:121) with
 * DenseRootModel / DenseLayer (layers.py:38-39,72-82).
 *
 * Network 2 -> hidden -> ... -> hidden -> 1 with n_tanh_layers tanh layers and a linear output;
 * hidden in {4, 8, 16} with n_tanh_layers = 3, or hidden in {4, 8} with 4 or 5 (every
 * architecture among the reference's committed model files).
 * w       device float[wdf_mlp_weight_count()]: per layer kernel[in][out] then bias[out]
 *         (the JSON "weights" order, layers.py:31-36)
 * theta2  device float[2] = {R, C}; r: optional [B][T] per-sample resistance (clipper_pot.py:116)
 * bwd:    gy [T][B] -> gtheta2[2] = dL/d{R, C} (R entry 0 when r != NULL), and for the
 *         weight-gradient pass: gb [T][B] = dL/d(root output b = -MLP), ain [T][B] = a,
 *         lrin [T][B] = log P1.R (written only when r != NULL).
 * wgrad:  gw[wdf_mlp_weight_count()] = dL/dw = -sum_n gb[n] dMLP(ain[n], lrin[n])/dw over the
 *         S = B*T samples bwd wrote (lrin NULL: log P1.R from theta2); what tape.gradient
 *         returns for model.trainable_variables (clipper_pot.py:181-184).  ws: scratch of
 *         wdf_clipper_mlp_wgrad_ws_bytes() bytes.  Deterministic (fixed-order reduction).
 * ---------------------------------------------------------------------------------- */
int wdf_mlp_weight_count(int hidden, int n_tanh_layers);
int wdf_clipper_mlp_fwd(const float* x, const float* r, const float* theta2, const float* w,
                        int hidden, int n_tanh_layers, float fs,
                        float* y, float* zstash, const float* z0, float* zT,
                        int64_t B, int64_t T, int flags, void* stream);
size_t wdf_clipper_mlp_bwd_ws_bytes(int64_t B);   /* size of wdf_clipper_mlp_bwd's ws */
/* The reverse sweep with the weight gradient folded in (csrc/wdf_mlp_row.h): gtheta2[2] and
 * gw[wdf_mlp_weight_count()] in one pass -- no gb / ain / lrin arrays, no wdf_clipper_mlp_wgrad.
 * What tape.gradient(loss, trainable_variables) returns at clipper_pot.py:181-184.  Deterministic. */
int64_t wdf_clipper_mlp_bwd_w_ws_bytes(int hidden, int n_tanh_layers, int64_t B);
int wdf_clipper_mlp_bwd_w(const float* x, const float* r, const float* theta2, const float* w,
                          int hidden, int n_tanh_layers, float fs,
                          const float* zstash, const float* gy, void* ws, float* gtheta2, float* gw,
                          int64_t B, int64_t T, int flags, void* stream);
int wdf_clipper_mlp_bwd(const float* x, const float* r, const float* theta2, const float* w,
                        int hidden, int n_tanh_layers, float fs,
                        const float* zstash, const float* gy,
                        float* gb, float* ain, float* lrin, void* ws, float* gtheta2,
                        int64_t B, int64_t T, int flags, void* stream);
/* out[n] = MLP(ain[n], lrin[n]), n < S: DenseRootModel on a table of (a, log R) points
 * (layers.py:76-82), the forward of the pre-training fit (diode_pretraining.py:113-126,159-160);
 * its weight gradient is wdf_clipper_mlp_wgrad with gb = -dL/dout (theta2 may be NULL there
 * when lrin is given).                                                                    */
int wdf_mlp_eval(const float* ain, const float* lrin, const float* w, int hidden, int n_tanh_layers,
                 float* out, int64_t S, void* stream);
/* One epoch of the pre-training fit in ONE launch (diode_pretraining.py:159-160: Keras fit =
 * Adam over shuffled mini-batches; loss = MSE + ESR, :136-155 with its N = esr_n): visits the S
 * table points (xa = a, xl = log R, ys = target) in the order given, `batch` (<= 64) at a time,
 * updating w and the Adam moments m, v and the iteration counter `step` in place; *loss_sum
 * receives the sum of the batch losses.  The host reshuffles between epochs.                  */
int wdf_mlp_fit_epoch(const float* xa, const float* xl, const float* ys, int64_t S, int batch,
                      float* w, float* m, float* v, int32_t* step, float lr, float beta1,
                      float beta2, float eps, float esr_n, float eps_energy, double* loss_sum,
                      int hidden, int n_tanh_layers, void* stream);
int64_t wdf_clipper_mlp_wgrad_ws_bytes(int hidden, int n_tanh_layers, int64_t S);
int wdf_clipper_mlp_wgrad(const float* ain, const float* lrin, const float* gb,
                          const float* theta2, const float* w, int hidden, int n_tanh_layers,
                          float fs, void* ws, float* gw, int64_t S, void* stream);

/* ------------------------------------------------------------------------------------
 * The one-pass MSE training step of LINEAR trees (csrc/wdf_ss_step.h): what one epoch of lpf.py:86-99 /
 * voltage_divider.py:69-93 does -- Model.forward over the batch (lpf.py:30-49), MeanSquaredError (:78), tape.gradient to
 * the component values through calc_impedance (:38,87-90) -- with the component values resident on the device.
 *   wdf_ss_probe          evaluates the probed step recorded once on the host (lib/wdf_hip/probe_tape.py: the elements'
 *                         own calc_impedance / reflected / incident arithmetic as a tape of + - * / neg recip, int32
 *                         [n_ops][3] = {op, a, b}) in float64 with forward-mode tangents: coef (the coefficient vector in
 *                         wdf_ss_fwd's layout, float32 and float64) and jac = d coef / d params, double [n_out][n_params].
 *   wdf_ss_lin_step_mse   forward, squared error and the gradient carried forward in time (no stash, no reverse sweep),
 *                         in EXACT time chunks (a pass from zero state, a walk over the chunk boundaries, the pass itself);
 *                         its last wave contracts dLoss/d coef with jac: out = {SSE, dLoss/d params}, *loss_out (optional) = gscale/2 SSE.  x is TIME-major
 *                         [T][ni][B].  ns <= 2, ni <= 2.  z0 / zT (ABI 6): float [ns][B] capacitor states the
 *                         call starts from (NULL: zero) / ends in (NULL: not wanted) -- lpf.py:30-49 never resets C1.      */
int wdf_ss_probe(const int32_t* tape, int n_ops, const double* consts, const float* params, int n_params,
                 const int32_t* outs, int n_out, float* coef, double* coef64, double* jac, void* stream);
/* wdf_adam_step_multi(jobs, n_jobs) (below: the optimizers' updates of `params`) and wdf_ss_probe in ONE launch. */
struct wdf_adam_job;
int wdf_ss_probe_adam(const struct wdf_adam_job* jobs, int n_jobs, const int32_t* tape, int n_ops, const double* consts,
                      const float* params, int n_params, const int32_t* outs, int n_out, float* coef, double* coef64,
                      double* jac, void* stream);
size_t wdf_ss_lin_step_ws_bytes(int ns, int ni, int64_t B, int64_t T, int n_chunks);
int wdf_ss_lin_step_mse(const float* x, const float* coef, const double* jac, int n_params, int ns, int ni,
                        const float* target, float gscale, float* y, void* ws, float* out, float* loss_out,
                        float* gcoef_out, int64_t B, int64_t T, int n_chunks, const float* z0, float* zT,
                        void* stream);

/* ------------------------------------------------------------------------------------
 * The one-pass MSE training step of small trees with a DIODE-PAIR root (csrc/wdf_ss_nl_step.h): the same epoch
 * (forward, MeanSquaredError, tape.gradient to the component values and to the root's Is / nVt -- tf_wdf.py:179-214,
 * HPFDiodeClipper.h:28-32's circuit) with the state's forward-mode tangents carried next to the state: no state
 * stash, no dLoss/dy array, no reverse sweep -- 12 bytes per sample.  Time chunks: the tangents are exact (linear
 * given the trajectory: chunk records + a walk); the state of chunk k starts w steps early from the PREVIOUS call's
 * state at that sample moved along its tangents by the change of the coefficients, every boundary is verified on the
 * device (tol) and a group of sequences that missed is re-run sequentially by its finishing wave; w is steered on the
 * device.  Two launches.  ns 1..2, ni 1..2, zero initial state, x TIME-major [T][ni][B].
 *   wdf_ss_nl_step_plan   zeroes the control part of ws and sets the warm-ups (multiples of 8): cold_warmup for the
 *                         first call (from z = 0), warm_warmup where it takes the snapshots for the second; afterwards
 *                         the device moves it inside [w_min, min(w_max, chunk length)].
 *   wdf_ss_nl_step_mse    coef: wdf_ss_probe's float32 outputs (coefficients, then the port resistance the root sees);
 *                         params: the component values on the device with Is, nVt at [n_tree], [n_tree + 1];
 *                         jac: double [ncoef + 1][n_tree].  out: float [1 + n_tree + 2] = {SSE, d(gscale/2 SSE)/d{component
 *                         values, Is, nVt}}.
 *   wdf_ss_nl_step_read   the 32-word control block {call, parity, have_snap, w_cur, w_snap, w_min, w_max, cool, tol,
 *                         grow_at, shrink_at, cool_miss, n_bad, max_miss, gated_groups, total_gated, ..., w_used at [19]}
 *                         (synchronises).  wdf_ss_nl_step_set: field 2..11 of it (tests, tuning).                  */
size_t wdf_ss_nl_step_ws_bytes(int ns, int ni, int64_t B, int64_t T, int n_chunks);
int wdf_ss_nl_step_chunk_len(int64_t T, int n_chunks);
int wdf_ss_nl_step_plan(void* ws, int ns, int ni, int64_t B, int64_t T, int n_chunks, int cold_warmup, int warm_warmup,
                        int w_min, int w_max, float tol, void* stream);
int wdf_ss_nl_step_set(void* ws, int field, double value, void* stream);
int wdf_ss_nl_step_read(const void* ws, int32_t* ctl_out, void* stream);
int wdf_ss_nl_step_mse(const float* x, const float* coef, const float* params, const double* jac, int n_tree, int ns,
                       int ni, int n_up, int n_down, const float* target, float gscale, float* y, void* ws, float* out,
                       float* loss_out, int64_t B, int64_t T, int n_chunks, void* stream);

/* ------------------------------------------------------------------------------------
 * The RESIDENT training step of the MLP-root pot clipper (csrc/wdf_mlp_step.h): what one epoch of
 * clipper_pot.py:245-269 does -- ClipperModel.forward (:103-127) over the whole training set, MSE + ESR past
 * skip_samples with the (outs, target) swap (:141-177,248), tape.gradient to the DenseRootModel weights
 * (layers.py:72-82), Adam (:180) -- as FIVE launches steered on the device (one HIP graph can replay it):
 * forward in verified time chunks with per-column chunk counts and warm-ups, two gated repair launches (idle unless
 * a chunk boundary missed), the exact reverse sweep with the adjoint recurrence run inside the weight-gradient
 * kernel, and the fixed-order reduction + Adam.  The training set is resident: x [B][T], target [T][B], and the
 * per-sample adaptor coefficients p, lr [B][T] of the pot channel (wdf_clipper_mlp_step_prepare: set_resistance +
 * calc_impedance of every step, clipper_pot.py:116-117, hoisted out of the loop -- they do not depend on the weights).
 * T must be a multiple of 16.  activation: 0 tanh, 1 relu (layers.py:63-65), hidden layers only.
 *
 *   state   caller-owned device buffer of wdf_clipper_mlp_step_state_bytes(): plan, per-column warm-up controller,
 *           snapshots of every 16th state of the last two calls (a chunk starts from the previous call's state at
 *           its sample), block maps of the adjoint recurrence, partial sums.
 *   plan    HOST int32[n_items][4] = {column (16 sequences), chunk index in the column, t0, t1}: every column's
 *           chunks tile [0, T) in multiples of 16; columns in order.  reset != 0: first use of the state.
 *   phase   WDF_MLP_STEP_FWD | WDF_MLP_STEP_BWD with the Adam buffers: the whole step, single rank.  Multi-rank:
 *           FWD | SUMS -> all-reduce sums[2] -> BWD | GLOBAL_SUMS without Adam -> all-reduce gw -> wdf_adam_step.
 *   read    host copies of the controller block (int32[32]: call, parity, have_snap, cold16, w_min, w_max, slack,
 *           cool_miss, cool_shrink, tol, grow_at, shrink_at, freeze, 3 pad, status[2][4] = {bad boundaries, max miss
 *           bits, columns flagged, columns sequential} by call parity, total flagged, total sequential) and of the
 *           per-column warm-ups (16-step units; wpeak_out: the largest each column ran with since the plan); hwid_out int32[n_items][2]: HW_ID and XCC_ID registers of the wave
 *           that ran each forward item of the last call (where the dispatcher placed it); colmiss_out float[cols][4]:
 *           per column the last verification's arrival miss and the misses 16, 32, 48 steps before arrival (what the
 *           controller steers the warm-up by).  Synchronises.                                          */
enum {
    WDF_MLP_STEP_FWD = 1, WDF_MLP_STEP_SUMS = 2, WDF_MLP_STEP_BWD = 4, WDF_MLP_STEP_GLOBAL_SUMS = 8
};
size_t wdf_clipper_mlp_step_state_bytes(int hidden, int n_layers, int64_t B, int64_t T, int n_items, int wgrad_chunks);
int wdf_clipper_mlp_step_plan(void* state, int hidden, int n_layers, int64_t B, int64_t T, int n_items, int wgrad_chunks,
                              const int32_t* items, int reset, int warm16, int cold16, int w_min, int w_max, float tol,
                              void* stream);
int wdf_clipper_mlp_step_read(const void* state, int hidden, int n_layers, int64_t B, int64_t T, int n_items,
                              int wgrad_chunks, int32_t* ctl_out, int32_t* wcol_out, int32_t* wpeak_out, int32_t* hwid_out,
                              float* colmiss_out,
                              void* stream);
int wdf_clipper_mlp_step_set(void* state, int field, int32_t bits, void* stream);
int wdf_clipper_mlp_step_set_wcol(void* state, int hidden, int n_layers, int64_t B, int64_t T, int n_items, int wgrad_chunks,
                                  const int32_t* wcol, void* stream);
int wdf_clipper_mlp_step_prepare(const float* r, const float* theta2, float fs, int64_t B, int64_t T, float* p, float* lr,
                                 void* stream);
int wdf_clipper_mlp_step(const float* x, const float* p, const float* lr, const float* theta2, float* w, int hidden,
                         int n_layers, int activation, float fs, const float* target, int64_t skip, double n_global,
                         double eps_energy, float* y, float* zstash, float* kappa, void* state, int64_t B, int64_t T,
                         int n_items, int wgrad_chunks, int phase, double* sums, float* gw, float* loss3, float* gcoef,
                         float* adam_m, float* adam_v, int32_t* adam_step, const float* adam_lr, float beta1, float beta2,
                         float eps, void* stream);

/* ------------------------------------------------------------------------------------
 * Generic tree + one root, as a state-space recursion (csrc/wdf_statespace.h):
 *     a = ca.z + da.x ;  b = root(a) ;  z' = A z + Bx x + E b ;  y = cy.z + dy.x + fy b
 * Replaces, per call, the whole per-sample loop of lpf.py:39-46 / voltage_divider.py:35-42 /
 * clipper_pot.py:113-124 for ANY tree built from tf_wdf.py's Resistor, Capacitor,
 * ResistiveVoltageSource, Series, Parallel, Inverter (:31-214) with static impedances
 * (calc_impedance once per forward, lpf.py:38).  The host derives the matrices from the
 * component values by running its tf_wdf-compatible elements on unit vectors.
 *
 * ns (0..4) capacitor states (4 since round 5), ni (1..2) input channels.
 * coef    device float[wdf_ss_ncoef(ns, ni)]:
 *         A[ns][ns] | Bx[ns][ni] | E[ns] | ca[ns] | da[ni] | cy[ns] | dy[ni] | fy
 * root_kind  WDF_ROOT_NONE (ideal source folded into the matrices; IdealVoltageSource
 *         tf_wdf.py:13-28) or WDF_ROOT_DIODE_PAIR with rootp = device float[3] {Is, nVt, R_port}
 * x       [B][T][ni] ; y [T][B] ; zstash [T][ns][B] ; z0, zT, gz0 [ns][B]
 * bwd:    gcoef[ncoef] = dL/dcoef, groot[3] = dL/d{Is, nVt, R_port}; ws of wdf_ss_bwd_ws_bytes.
 * ---------------------------------------------------------------------------------- */
enum { WDF_ROOT_NONE = 0, WDF_ROOT_DIODE_PAIR = 2, WDF_ROOT_MLP = 3 /* wdf_ss_dyn_* only */ };

int wdf_ss_ncoef(int ns, int ni);
int wdf_ss_fwd(const float* x, const float* coef, const float* rootp,
               int ns, int ni, int root_kind, int n_up, int n_down,
               float* y, float* zstash, const float* z0, float* zT,
               int64_t B, int64_t T, int flags, void* stream);

/* Exact time-parallel forward of a LINEAR tree (root kind NONE: lpf.py:30-49, voltage_divider.py:27-46 with the
 * ideal source folded into the matrices): chunk end states from a zero state, chunk start states by
 * z(t0 + L) = A^L z(t0) + end0, then every chunk from its exact start.  Same outputs as wdf_ss_fwd up to fp32
 * rounding; ws: wdf_ss_fwd_lin_tp_ws_bytes(ns, B, n_chunks). */
size_t wdf_ss_fwd_lin_tp_ws_bytes(int ns, int64_t B, int n_chunks);
int wdf_ss_fwd_lin_tp(const float* x, const float* coef, int ns, int ni, float* y, float* zstash,
                      const float* z0, float* zT, int64_t B, int64_t T, int n_chunks, void* ws, void* stream);
int wdf_ss_bwd(const float* x, const float* coef, const float* rootp,
               int ns, int ni, int root_kind, int n_up, int n_down,
               const float* zstash, const float* gy,
               void* ws, float* gcoef, float* groot, float* gz0,
               int64_t B, int64_t T, int flags, void* stream);

/* Time-parallel forms of the two calls above (csrc/wdf_statespace.h, second half) for trees with >= 1 state; n_chunks
 * must be a value wdf_ss_tp_chunks returns (chunks of a multiple of 8 steps).
 * wdf_ss_fwd_tp (diode root): chunk k starts `warmup` steps early from z = 0 -- or, with zinit [n_chunks][ns][B], from
 *   the states the caller supplies for the samples wdf_ss_tp_starts names (a training loop that re-visits its batch hands
 *   in the previous call's stash rows: a fraction of the cold warm-up then closes the gap); chunk boundaries are verified
 *   on the device (|arrival - predecessor's end| <= tol per state) and the sequential kernel, launched behind it, re-runs
 *   exactly the 64-sequence waves that missed.  status: device int32[4] = {n_bad, max |miss| (float bits), gated waves, 0}.
 *   ws: wdf_ss_fwd_tp_ws_bytes(ns, B, n_chunks) bytes.  Same outputs as wdf_ss_fwd within tol (HPFDiodeClipper.h:28-32).
 * wdf_ss_bwd_tp (any root): EXACT -- the adjoint recurrence is linear in the adjoint entering a chunk, every chunk leaves
 *   an affine record, one walk per sequence composes them; same results as wdf_ss_bwd up to summation order.
 *   ws: wdf_ss_bwd_tp_ws_bytes(ns, ni, B, n_chunks) bytes.                                                            */
int wdf_ss_tp_chunks(int64_t T, int n_chunks);
size_t wdf_ss_fwd_tp_ws_bytes(int ns, int64_t B, int n_chunks);
int wdf_ss_tp_starts(int64_t T, int n_chunks, int warmup, int64_t* starts /* [n_chunks]: first sample each chunk's wave runs */);
int wdf_ss_fwd_tp(const float* x, const float* coef, const float* rootp, int ns, int ni, int n_up, int n_down, float* y,
                  float* zstash, const float* z0, float* zT, int64_t B, int64_t T, int n_chunks, int warmup, float tol,
                  const float* zinit, void* ws, void* status, void* stream);
size_t wdf_ss_bwd_tp_ws_bytes(int ns, int ni, int64_t B, int n_chunks);
int wdf_ss_bwd_tp(const float* x, const float* coef, const float* rootp, int ns, int ni, int root, int n_up, int n_down,
                  const float* zstash, const float* gy, void* ws, float* gcoef, float* groot, float* gz0, int64_t B, int64_t T,
                  int n_chunks, void* stream);
size_t wdf_ss_bwd_ws_bytes(int ns, int ni, int64_t B);

/* ------------------------------------------------------------------------------------
 * State-space recursion with PER-SAMPLE coefficient rows, and the MLP root on any small tree (round 5; csrc/wdf_ss_dyn.h).
 * Replaces, for any tree of <= 8 capacitors (round 6; 4 until round 5) and <= 2 sources:
 *   - a per-sample impedance: set_resistance on any ResistiveVoltageSource / Resistor (tf_wdf.py:51-52,80-81) followed by
 *     calc_impedance every step (clipper_pot.py:116-117) -- the host evaluates the probed step's coefficients over the whole
 *     resistance channel and hands one ROW per (sample, sequence);
 *   - layers.DenseRootModel terminating the tree (layers.py:72-82): b = -MLP(a, log R_port) (clipper_pot.py:119-121).
 * rows       per_sample = 1: device float [T][n][B]; per_sample = 0: ONE row, float [n]; per_sample = 2 (round 6): one row per
 *            SEQUENCE, float [n][B] -- a pot that keeps its value over a recording (dataimport.py:96 repeats the file's
 *            resistance down the channel): calc_impedance's result does not change along the sequence;
 *            n = wdf_ss_dyn_row_len(ns, ni) =
 *            wdf_ss_ncoef(ns, ni) + 1:  A | Bx | E | ca | da | cy | dy | fy | R_port   (the coef layout above, then the port resistance)
 * root       WDF_ROOT_NONE, WDF_ROOT_DIODE_PAIR (rootp = device float[2] {Is, nVt}; R_port comes from the row) or WDF_ROOT_MLP
 *            (w: flat weights, hidden in {4, 8, 16} with 3 tanh layers, {4, 8} with 5; tanh only)
 * x [B][T][ni]; y, gy [T][B]; zstash [T][ns][B]; z0 / zT / gz0 [ns][B] (optional).
 * wdf_ss_dyn_bwd: reverse sweep for dL/dy = gy.  per_sample = 1: grows [T][n][B] receives dL/d(row entry) of EVERY sample.
 *            per_sample = 0 / 2 (round 6: rows constant in time): the kernel sums over the steps itself (float64 accumulators):
 *            grows [K][n][B], K = 1 (wdf_ss_dyn_bwd) or the chunk count (wdf_ss_dyn_bwd_tp) -- the caller adds the K partials
 *            (per_sample = 0: and the B sequences); ws (wdf_ss_dyn_bwd_ws_bytes): double [(B+63)/64][2] = per-wave {sum gb D_L, sum gb D_V}
 *            of a diode root (dL/dIs = S_L / Is, dL/dnVt = S_V - S_L / nVt); MLP root: gb, ain, lrin [T][B] receive dL/db, a and
 *            log R_port of every step -- the operands of wdf_clipper_mlp_wgrad (gw = -sum gb dMLP/dw).
 * One lane per sequence, sequential in time: the general path (the clipper topology keeps its own kernels).
 * ---------------------------------------------------------------------------------- */
int wdf_ss_dyn_row_len(int ns, int ni);
int wdf_ss_dyn_fwd(const float* x, const float* rows, int per_sample, int ns, int ni, int root, const float* rootp, const float* w,
                   int hidden, int n_tanh_layers, int n_up, int n_down, float* y, float* zstash, const float* z0, float* zT,
                   int64_t B, int64_t T, void* stream);
/* The same in time chunks (n_chunks tiles T in 8-step units).  Forward: a chunk starts `warmup` steps early from z = 0, the device
 * compares what every chunk arrives with against what its predecessor ended with (tol) and re-runs, sequentially, exactly the
 * 64-sequence waves with a miss -- the result is within tol of wdf_ss_dyn_fwd's or IS it; status: int32 {n_bad, max-miss float
 * bits, gated waves, 0}; ws: wdf_ss_dyn_fwd_tp_ws_bytes; zinit (optional) float [n_chunks][ns][B]: the state chunk k starts from at
 * sample max(0, k L - warmup), L = the 8-step-rounded chunk length -- a training loop hands in the previous call's stash rows and
 * runs a fraction of the cold warm-up.  Reverse sweep: EXACT (the adjoint is linear in the adjoint entering a
 * chunk: chunk maps, one walk per sequence, then every chunk re-walked from its true entering adjoint with the root's partials
 * kept from the first walk); ws: wdf_ss_dyn_bwd_tp_ws_bytes; the per-(chunk, wave) diode sums sit at its head as
 * double [n_chunks (B+63)/64][2]. */
size_t wdf_ss_dyn_fwd_tp_ws_bytes(int ns, int64_t B, int n_chunks);
int wdf_ss_dyn_fwd_tp(const float* x, const float* rows, int per_sample, int ns, int ni, int root, const float* rootp, const float* w,
                      int hidden, int n_tanh_layers, int n_up, int n_down, float* y, float* zstash, const float* z0, float* zT,
                      int64_t B, int64_t T, int n_chunks, int warmup, float tol, const float* zinit, void* ws, void* status, void* stream);
size_t wdf_ss_dyn_bwd_tp_ws_bytes(int ns, int64_t B, int64_t T, int n_chunks);
int wdf_ss_dyn_bwd_tp(const float* x, const float* rows, int per_sample, int ns, int ni, int root, const float* rootp, const float* w,
                      int hidden, int n_tanh_layers, int n_up, int n_down, const float* zstash, const float* gy, float* grows,
                      void* ws, float* gb, float* ain, float* lrin, float* gz0, int64_t B, int64_t T, int n_chunks, void* stream);
size_t wdf_ss_dyn_bwd_ws_bytes(int64_t B);
int wdf_ss_dyn_bwd(const float* x, const float* rows, int per_sample, int ns, int ni, int root, const float* rootp, const float* w,
                   int hidden, int n_tanh_layers, int n_up, int n_down, const float* zstash, const float* gy, float* grows,
                   void* ws, float* gb, float* ain, float* lrin, float* gz0, int64_t B, int64_t T, void* stream);

/* The rows themselves, on the device (csrc/wdf_ss_dyn_rows.h): the probed step of the tree as a straight-line scalar program
 * ("tape") over its component values -- the elements' own calc_impedance / reflected / incident arithmetic (tf_wdf.py:31-214)
 * recorded once on the host -- run per (sample, sequence) in float64 with ONE component value taken from a resistance channel:
 * set_resistance + calc_impedance every step (tf_wdf.py:51-52,80-81; clipper_pot.py:116-117) for the whole batch in one launch,
 * and its chain rule (tape.gradient through calc_impedance, lpf.py:38,87-90) in two.
 * tape_ops   HOST int32 [n_ops][3] = {op, a, b}: 0 CONST (a: index into consts), 1 PARAM (a: component value), 2 ADD, 3 SUB, 4 MUL,
 *            5 DIV (a, b: earlier operations), 6 NEG, 7 RECIP (a); n_ops <= 192.  consts: HOST double [n_consts <= 32].
 * outs       HOST int32 [n_out]: the operation whose value is row entry k (n_out = wdf_ss_dyn_row_len(ns, ni) for the kernels above)
 * params     device double [n_params <= 15]: the component values; chan: the one that is the channel (-1: none)
 * r          device float [T][B]: the channel (NULL with chan = -1 -- B = T = 1 then gives the one static row)
 * rows       device float [T][n_out][B] (out)
 * _bwd:      grows [T][n_out][B] (what wdf_ss_dyn_bwd left) -> gparams device double [n_params] = dLoss/d(component value) summed
 *            over all samples in a fixed order (the channel's entry: 0); ws: wdf_ss_dyn_rows_bwd_ws_bytes.
 * The tape is checked on every call (operands must name earlier operations); WDF_EUNSUPPORTED beyond the sizes above. */
int wdf_ss_dyn_rows(const int32_t* tape_ops, int n_ops, const double* consts, int n_consts, const int32_t* outs, int n_out,
                    const double* params, int n_params, int chan, const float* r, float* rows, int64_t B, int64_t T, void* stream);
size_t wdf_ss_dyn_rows_bwd_ws_bytes(int n_params, int64_t B, int64_t T);
int wdf_ss_dyn_rows_bwd(const int32_t* tape_ops, int n_ops, const double* consts, int n_consts, const int32_t* outs, int n_out,
                        const double* params, int n_params, int chan, const float* r, const float* grows, void* ws, double* gparams,
                        int64_t B, int64_t T, void* stream);

/* ------------------------------------------------------------------------------------
 * Clipper with two DIFFERENT antiparallel diodes (BASELINE config 5; csrc/wdf_asym.h).  New
 * API, no reference counterpart (its pairs are copies of one diode, diode_pretraining.py:46-47).
 * theta6  device float[6] = {Is_up, nVt_up, Is_down, nVt_down, R, C}
 * mode    WDF_ASYM_OMEGA_F32: fp32 Wright-omega closed form (two-diode form of eqn 39);
 *         WDF_ASYM_NEWTON_F64: fp64 Newton on the exact Shockley pair, iterated per wave until
 *         every lane meets |dv| <= tol (|v| + nVt) (wavefront ballot) or max_iter; two iterations
 *         in fp32 from the closed form's value come first (counted in iters).
 * iters   optional device int64[(B+63)/64]: Newton iterations each wave ran (sum / (B T / 64
 *         * ...) gives the mean per sample).
 * zstash  optional [T][B]: state before each step, for wdf_clipper_asym_bwd.
 * wdf_clipper_asym_bwd: reverse sweep of the NEWTON-mode loop (the exact model) for a given dL/dy: the root
 *         is re-solved per step from the stash and differentiated implicitly (F(v; a, theta) = 0);
 *         gtheta6 = dL/d{Is_up, nVt_up, Is_down, nVt_down, R, C}; ws: wdf_clipper_asym_bwd_ws_bytes(B).
 * wdf_asym_root: b[i] = root(a[i]) (double out) for accuracy sweeps.
 * ---------------------------------------------------------------------------------- */
enum { WDF_ASYM_OMEGA_F32 = 0, WDF_ASYM_NEWTON_F64 = 1 };
int wdf_clipper_asym_fwd(const float* x, const float* theta6, float fs, int mode, double tol, int max_iter,
                         float* y, float* zstash, const float* z0, float* zT, long long* iters,
                         int64_t B, int64_t T, void* stream);

/* Time-parallel form: the time axis in n_chunks chunks (a count that tiles T in 8-step units), every chunk but the first
 * starting `warmup` steps early from z = 0; the device verifies every chunk boundary to verify_tol and re-runs, sequentially,
 * exactly the 64-sequence waves where one missed (status: int32 {n_bad, max-miss float bits, gated waves, 0}).  The result is
 * the sequential kernel's to verify_tol; 128 waves become 128 n_chunks.  ws: wdf_clipper_asym_fwd_tp_ws_bytes bytes. */
size_t wdf_clipper_asym_fwd_tp_ws_bytes(int64_t B, int n_chunks);
int wdf_clipper_asym_fwd_tp(const float* x, const float* theta6, float fs, int mode, double tol, int max_iter, float* y,
                            float* zstash, const float* z0, float* zT, int64_t B, int64_t T, int n_chunks, int warmup,
                            float verify_tol, void* ws, void* status, void* stream);
size_t wdf_clipper_asym_bwd_ws_bytes(int64_t B);
int wdf_clipper_asym_bwd(const float* x, const float* theta6, float fs, double tol, int max_iter,
                         const float* zstash, const float* gy, void* ws, float* gtheta6,
                         int64_t B, int64_t T, void* stream);
/* Time-parallel reverse sweep, BOTH modes (round 5).  The root is not re-solved: b = z[t+1] + p (z[t] - x[t]) from two
 * consecutive stash entries (zT = z[T], as wdf_clipper_asym_fwd(_tp) returns it); the adjoint recurrence is linear in the
 * adjoint entering a chunk, so the n_chunks chunks (a count that tiles T in 8-step units) run independently and one walk per
 * sequence composes them -- exact.  mode NEWTON differentiates the exact Shockley pair implicitly (wdf_clipper_asym_bwd's
 * formulas), mode OMEGA the fp32 closed form the OMEGA forward evaluates.  gzT (optional [B]): dL/dz[T]; gz0 (optional [B]):
 * receives dL/dz[0].  ws: wdf_clipper_asym_bwd_tp_ws_bytes(B, n_chunks). */
size_t wdf_clipper_asym_bwd_tp_ws_bytes(int64_t B, int n_chunks);
int wdf_clipper_asym_bwd_tp(const float* x, const float* theta6, float fs, int mode, const float* zstash, const float* zT,
                            const float* gy, const float* gzT, void* ws, float* gtheta6, float* gz0, int64_t B, int64_t T,
                            int n_chunks, void* stream);
int wdf_asym_root(const float* a, const float* theta6, float fs, int mode, double tol, int max_iter,
                  double* b, int64_t n, void* stream);

/* Element-wise diode-pair root and Wright omega on device arrays (n elements): the
 * building blocks above, exposed for parity tests against diode_pretraining.py:39-60 /
 * toms917.cpp.  R_port is the port resistance seen by the root (P1.R).
 * iters (optional int32[n]) receives the number of FSC iterations taken per element.     */
int wdf_omega_f32(const float* x, float* w, int32_t* iters, int64_t n, void* stream);
/* omega in fp64 with the reference's iteration count (iters: 0 = start value only, 1, 2: toms917.cpp:347-364) */
int wdf_omega_f64(const double* x, double* w, int32_t* iters, int64_t n, void* stream);
int wdf_diode_pair_f32(const float* a, const float* R_port, float Is, float nVt,
                       int n_up, int n_down, float* b, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * Optimizer step on the device (csrc/wdf_optim.h): tf.keras.optimizers.Adam.apply_gradients
 * as the scripts call it (lpf.py:79-80,93-94; clipper_pot.py:179,183-184) followed by the
 * Variable constraint clip (tf_wdf.py:74,104), for n <= 1024 parameters (component values or a
 * flat MLP weight vector).  lr: per-parameter learning rates (the reference keeps one optimizer
 * per component); lo / hi: optional clip bounds; step: device iteration counter, incremented.
 * ---------------------------------------------------------------------------------- */
int wdf_adam_step(float* theta, const float* grad, float* m, float* v, int32_t* step,
                  const float* lr, float beta1, float beta2, float eps,
                  const float* lo, const float* hi, int n, void* stream);

/* Up to WDF_ADAM_MULTI_MAX independent wdf_adam_step updates in ONE launch (one Adam per component, lpf.py:79-80,93-94: the
 * host layer queues the apply_gradients calls of a training step and sends them together before anything reads the values). */
#define WDF_ADAM_MULTI_MAX 8
typedef struct wdf_adam_job {
    float* theta; const float* grad; float* m; float* v; int32_t* step; const float* lr; const float* lo; const float* hi;
    float beta1, beta2, eps; int n;
} wdf_adam_job;
int wdf_adam_step_multi(const wdf_adam_job* jobs, int n_jobs, void* stream);

/* Time-parallel variants of the MLP-root calls (csrc/wdf_mlp_tp.h): the reference's 1340 x 2048 training set
 * (clipper_pot.py:58,232) is only 335 waves when every sequence runs its 2048 steps in one wave.
 *   wdf_clipper_mlp_fwd_tp   the time axis in n_chunks chunks (length rounded up to a multiple of 16;
 *       wdf_clipper_mlp_tp_chunks() = the number used); every chunk but the first starts `warmup` steps
 *       early from z = 0 -- or warmup_per_wave[ceil(B/4)] steps (device int32, multiples of 16: one value
 *       per 4 consecutive sequences, the pot resistance sets the circuit's memory) -- and a verify kernel
 *       compares the state each chunk arrives with against the state its predecessor ended in
 *       (|diff| <= tol); waves with a miss run once more, every chunk from the state its predecessor ended in and
 *       without warm-up (a chunk-local repair: one chunk's steps instead of T), are verified again, and only what
 *       still misses is re-run sequentially by a gated launch of the plain kernel.
 *       status: device int32[4] = {n_bad, max |miss| (float bits), waves with a miss at the first pass,
 *       waves that went through the sequential kernel}.
 *   wdf_clipper_mlp_bwd_w_tp  EXACT reverse sweep, parallel over all steps: kappa[n] = d z'/d z of every
 *       step from the stash (no recurrence), the scalar adjoint recurrence by one lane per sequence, then
 *       the weight-gradient and {R, C} sums of every step with the known adjoint.  Same outputs as
 *       wdf_clipper_mlp_bwd_w.                                                                  */
int wdf_clipper_mlp_tp_chunks(int64_t T, int n_chunks);
size_t wdf_clipper_mlp_fwd_tp_ws_bytes(int64_t B, int n_chunks);
int wdf_clipper_mlp_fwd_tp(const float* x, const float* r, const float* theta2, const float* w,
                           int hidden, int n_tanh_layers, float fs,
                           float* y, float* zstash, const float* z0, float* zT,
                           int64_t B, int64_t T, int n_chunks, int warmup, const int32_t* warmup_per_wave,
                           float tol, void* ws, void* status, void* stream);
int64_t wdf_clipper_mlp_bwd_w_tp_ws_bytes(int hidden, int n_tanh_layers, int64_t B, int64_t T, int n_chunks);
int wdf_clipper_mlp_bwd_w_tp(const float* x, const float* r, const float* theta2, const float* w,
                             int hidden, int n_tanh_layers, float fs,
                             const float* zstash, const float* gy, void* ws, float* gtheta2, float* gw,
                             int64_t B, int64_t T, int n_chunks, void* stream);
/* The same pair with the adjoint recurrence's coefficient handed across: the forward's owned steps also evaluate the
 * network's input Jacobian (the activations are in registers there) and store kappa[n] = d z[n+1]/d z[n] [T][B]
 * (waves the sequential kernel re-runs get theirs from the stash by a gated pass); the reverse sweep then starts at
 * the scan -- one network evaluation per step less.  Same y, stash, gradients as the pair above.  zstash is required. */
/* zinit [chunks][B] or NULL: the state every chunk but the first starts its warm-up from instead of 0 -- the stash of
 * the previous call on the same batch at the samples wdf_clipper_mlp_tp_starts() names (a training loop: the weights
 * moved a little, so a short warm-up closes the gap; the verification decides as before).  One warm-up value only
 * (warmup_per_wave must be NULL with zinit).  wdf_clipper_mlp_tp_starts is a host function (no GPU work). */
int wdf_clipper_mlp_tp_starts(int64_t T, int n_chunks, int warmup, int64_t* starts);
int wdf_clipper_mlp_fwd_tp_kappa(const float* x, const float* r, const float* theta2, const float* w,
                                 int hidden, int n_tanh_layers, float fs,
                                 float* y, float* zstash, float* kappa, const float* z0, float* zT, const float* zinit,
                                 int64_t B, int64_t T, int n_chunks, int warmup, const int32_t* warmup_per_wave,
                                 float tol, void* ws, void* status, void* stream);
int wdf_clipper_mlp_bwd_w_tp_kappa(const float* x, const float* r, const float* theta2, const float* w,
                                   int hidden, int n_tanh_layers, float fs,
                                   const float* zstash, const float* kappa, const float* gy, void* ws,
                                   float* gtheta2, float* gw, int64_t B, int64_t T, int n_chunks, void* stream);

/* library / device info */
/* chunk count the weight-gradient pass of wdf_clipper_mlp_bwd_w_tp(_kappa) uses on the matrix cores for this shape, 0 when it stays on
 * the row kernel (the library's own dispatch rule, for a harness that prices that kernel). */
int wdf_clipper_mlp_wgrad_matrix_core_chunks(int64_t B, int64_t T);
int wdf_abi_version(void);
const char* wdf_last_error(void);
/* fills name (<= cap bytes) with the gcnArchName of `device`, returns the CU count or <0 */
int wdf_device_info(int device, char* name, int cap);

/* Timing helper for bench.py: HIP events on the stream the kernels run on.
 * wdf_timer_* wrap hipEventCreate/Record/ElapsedTime so Python needs no HIP binding.     */
void* wdf_event_create(void);
int wdf_event_record(void* ev, void* stream);
int wdf_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */
/* One-shot, process-wide (the consuming call may run on another thread, e.g. an autograd
 * backward): record `start` / `stop` immediately around the next recurrence kernel the library
 * launches through wdf_clipper_fwd / _bwd / _fwd_tp / _bwd_tp / _bwd_mse_tp (the sweep itself,
 * not the verify / combine / reduce helpers of the same call): the duration rocprofv3 reports
 * for that kernel.                                                                        */
void wdf_event_bracket_next(void* start, void* stop);
void wdf_event_destroy(void* ev);
/* out (device uint64[2]) <- {shader clock counter (s_memtime), constant-rate counter (s_memrealtime, 100 MHz)} when the
 * stream reaches this point: two stamps around a stretch of work give the clock the chip sustained over it
 * (bench.py value_sustained).                                                                */
int wdf_clock_stamp(uint64_t* out, void* stream);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* WDF_HIP_H */
