// wdf_ss_dyn_rows.h -- the coefficient rows of wdf_ss_dyn.h's kernels, made on the device: set_resistance + calc_impedance
// of every sample (tf_wdf.py:51-52,80-81,114-115,139-145,168-177; clipper_pot.py:116-117) hoisted out of the time loop.
//
// The probed step of a tree is a straight-line scalar program over its component values (lib/wdf_hip/probe_tape.py: +, -,
// *, /, negate, reciprocal -- the elements' own calc_impedance / reflected / incident code run once on tracing scalars).
// One of the component values may be a CHANNEL: a resistance per (sample, sequence).  Here a lane takes one sequence and
// walks its samples; per sample it runs the whole tape in float64 (the adaptors' coefficients cancel: 1 - p, R1 - R2) with
// the node values in LDS [node][lane], and
//   forward:  writes the row  A | Bx | E | ca | da | cy | dy | fy | R_port  of that sample as float32, rows [T][n][B];
//   reverse:  runs the tape again, then backwards with the row's adjoint (what wdf_ss_dyn_bwd left in grows [T][n][B]):
//             dLoss/d(component value), summed over every sample in float64 -- tape.gradient through calc_impedance
//             (lpf.py:38,87-90) for all of a sequence's steps at once.  Adjoints of the nodes are float32 (they start from
//             float32 row adjoints); per-lane sums are float64, reduced in a fixed order: lanes (DPP), waves (partials +
//             one more launch).
// The tape travels in the kernel arguments and sits in three VGPRs (one dword per operation, lane = operation: v_readlane at a
// wave-uniform index; the switch is wave-uniform).
// No channel (r == null, B = T = 1): the one static row of a tree whose components only train.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wdf {

constexpr int kRowsMaxOps = 192;       // reverse: 12 bytes per node and lane + the parameter sums: 155 KB of the CU's 160 KB LDS
constexpr int kRowsMaxConsts = 32;
constexpr int kRowsMaxOut = 48;        // ns = 4, ni = 2: 42 coefficients + R_port
constexpr int kRowsMaxParams = 15;

// probe_tape.py's operation codes
enum : int { kOpConst = 0, kOpParam = 1, kOpAdd = 2, kOpSub = 3, kOpMul = 4, kOpDiv = 5, kOpNeg = 6, kOpRecip = 7 };

struct RowsTape {
    uint32_t code[kRowsMaxOps];        // op | a << 8 | b << 16   (operands: earlier nodes; CONST: a = index into consts; PARAM: a = parameter)
    uint8_t outs[kRowsMaxOut];         // the node of every row entry
    double consts[kRowsMaxConsts];
    int n_ops, n_out, n_params, chan;  // chan: the parameter that is the channel, or -1
};

// The tape, the constants and the component values spread over the wave's lanes (operation i: lane i & 63 of word i >> 6),
// read back with v_readlane at a wave-uniform index: no memory access in the interpreter's dependent chain but the nodes' LDS.
struct RowsRegs {
    uint32_t c0, c1, c2;
    double kv, pv;

    __device__ __forceinline__ void load(const RowsTape& tp, const double* __restrict__ params, int lane)
    {
        c0 = tp.code[lane]; c1 = tp.code[64 + lane]; c2 = tp.code[128 + lane];
        kv = tp.consts[lane & (kRowsMaxConsts - 1)];
        pv = lane < tp.n_params ? params[lane] : 0.0;
    }
    __device__ __forceinline__ uint32_t code(int i) const
    {
        // (three v_readlane and two scalar selects: selecting the VGPR first makes an indexed array of them, in scratch)
        const int l = i & 63;
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)c0, l), r1 = (uint32_t)__builtin_amdgcn_readlane((int)c1, l),
                       r2 = (uint32_t)__builtin_amdgcn_readlane((int)c2, l);
        return i < 64 ? r0 : (i < 128 ? r1 : r2);
    }
    static __device__ __forceinline__ double lane_f64(double v, int l)
    {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ double constant(int a) const { return lane_f64(kv, a); }
    __device__ __forceinline__ double param(int a) const { return lane_f64(pv, a); }
};

// +, -, *, negate as ONE branch-free form  v = c1 x + c2 y + c3 x y  with wave-uniform coefficients (the interpreter's time went
// into the taken branches of a per-operation switch, not into the arithmetic); / and reciprocal, constants and component values
// are the rare operations and branch.
// FINITE TAPES ONLY: the form multiplies the operand an operation does not use by 0 (NEG reads node 0 as y, MUL forms 0 x), so an
// infinite or NaN node would poison results that tape.evaluate_torch keeps finite (0 x inf).  The nodes of a probed step are
// resistances, conductances and their ratios of component values the element classes clip to finite positive ranges
// (tf_wdf.py:69-75,99-105: R in [180, 1e6], C in [1e-13, 1]) and of a resistance channel the caller owns: a channel holding 0, inf
// or NaN gives NaN rows here (and a division by zero in the reference's own calc_impedance).
struct RowsForm {
    double c1, c2, c3;
    __device__ __forceinline__ explicit RowsForm(int op)
        : c1(op == kOpMul ? 0.0 : (op == kOpNeg ? -1.0 : 1.0)), c2(op == kOpAdd ? 1.0 : (op == kOpSub ? -1.0 : 0.0)), c3(op == kOpMul ? 1.0 : 0.0) {}
    static __device__ __forceinline__ bool covers(int op) { return op == kOpAdd || op == kOpSub || op == kOpMul || op == kOpNeg; }
};

// the tape forward for this lane's sample: val[node * 64 + lane]
__device__ __forceinline__ void rows_forward(const RowsRegs& rg, int n_ops, int chan, double rv, double* __restrict__ val, int lane)
{
    for (int i = 0; i < n_ops; ++i) {
        const uint32_t c = rg.code(i);
        const int op = (int)(c & 0xffu), a = (int)((c >> 8) & 0xffu), b = (int)((c >> 16) & 0xffu);
        double v;
        if (__builtin_expect(RowsForm::covers(op), 1)) {
            const RowsForm f(op);
            const double x = val[a * 64 + lane], y = val[b * 64 + lane];      // (negate: b = 0, c2 = c3 = 0)
            v = fma(f.c3 * x, y, fma(f.c2, y, f.c1 * x));
        } else if (op == kOpDiv) {
            v = val[a * 64 + lane] / val[b * 64 + lane];
        } else if (op == kOpRecip) {
            v = 1.0 / val[a * 64 + lane];
        } else if (op == kOpParam) {
            v = (a == chan) ? rv : rg.param(a);
        } else {
            v = rg.constant(a);                                   // kOpConst
        }
        val[i * 64 + lane] = v;
    }
}

constexpr int kRowsSteps = 8;          // samples a wave walks: their channel values are asked for at once, before the first tape pass

// the channel values of the wave's steps (clamped reads past T; r null: zeros)
__device__ __forceinline__ void rows_channel(const float* __restrict__ r, int64_t t0, int64_t T, int64_t B, int64_t b, float (&rv)[kRowsSteps])
{
#pragma unroll
    for (int q = 0; q < kRowsSteps; ++q) {
        const int64_t t = t0 + q < T ? t0 + q : T - 1;
        rv[q] = r ? r[t * B + b] : 0.0f;
    }
}

// grid (ceil(B / 64), ceil(T / kRowsSteps)), block 64, dynamic LDS n_ops * 64 * 8 bytes
__global__ __launch_bounds__(64) void ss_dyn_rows_kernel(const RowsTape tp, const double* __restrict__ params, const float* __restrict__ r,
                                                         float* __restrict__ rows, int64_t B, int64_t T)
{
    extern __shared__ double rows_lds[];
    double* __restrict__ val = rows_lds;
    const int lane = threadIdx.x;
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + lane;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const int64_t t0 = (int64_t)blockIdx.y * kRowsSteps;
    const int n = tp.n_out;
    RowsRegs rg;
    rg.load(tp, params, lane);
    float rv[kRowsSteps];
    rows_channel(r, t0, T, B, b, rv);
#pragma unroll
    for (int q = 0; q < kRowsSteps; ++q) {
        const int64_t t = t0 + q;
        if (t >= T) break;
        rows_forward(rg, tp.n_ops, tp.chan, (double)rv[q], val, lane);
        if (live) {                                                // (stores only: nothing in the loop waits for memory)
            float* __restrict__ o = rows + t * n * B + b;
            for (int k = 0; k < n; ++k) o[(int64_t)k * B] = (float)val[(int)tp.outs[k] * 64 + lane];
        }
    }
}

// The reverse pass.  grid as above; dynamic LDS n_ops * 64 * 12 + n_params * 64 * 8 bytes.
// part: double [gridDim.y * gridDim.x][n_params] -- one partial per wave (the channel's entry is left 0).
__global__ __launch_bounds__(64) void ss_dyn_rows_bwd_kernel(const RowsTape tp, const double* __restrict__ params, const float* __restrict__ r,
                                                             const float* __restrict__ grows, double* __restrict__ part, int64_t B, int64_t T)
{
    extern __shared__ double rows_lds[];
    const int lane = threadIdx.x, n_ops = tp.n_ops, P = tp.n_params;
    double* __restrict__ val = rows_lds;                           // [n_ops][64]
    double* __restrict__ gp = val + n_ops * 64;                    // [P][64]
    float* __restrict__ adj = reinterpret_cast<float*>(gp + P * 64);   // [n_ops][64]
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + lane;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const int64_t t0 = (int64_t)blockIdx.y * kRowsSteps;
    const int n = tp.n_out;
    for (int p = 0; p < P; ++p) gp[p * 64 + lane] = 0.0;
    RowsRegs rg;
    rg.load(tp, params, lane);
    const int chan = tp.chan;
    float rv[kRowsSteps];
    rows_channel(r, t0, T, B, b, rv);
#pragma unroll
    for (int q = 0; q < kRowsSteps; ++q) {
        const int64_t t = t0 + q;
        if (t >= T) break;
        // the row's adjoint, four entries per round trip, asked for before the forward pass needs the LDS pipe
        const float* __restrict__ g = grows + t * n * B + b;
        for (int i = 0; i < n_ops; ++i) adj[i * 64 + lane] = 0.0f;
        for (int k0 = 0; k0 < n; k0 += 4) {
            float g4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) g4[j] = (live && k0 + j < n) ? g[(int64_t)(k0 + j < n ? k0 + j : n - 1) * B] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j < n) adj[(int)tp.outs[k0 + j] * 64 + lane] += g4[j];           // (an entry may name a node twice: in order)
        }
        rows_forward(rg, n_ops, chan, (double)rv[q], val, lane);
        for (int i = n_ops - 1; i >= 0; --i) {
            const uint32_t c = rg.code(i);
            const int op = (int)(c & 0xffu), a = (int)((c >> 8) & 0xffu), bb = (int)((c >> 16) & 0xffu);
            const double gi = (double)adj[i * 64 + lane];
            if (__builtin_expect(RowsForm::covers(op), 1)) {
                // v = c1 x + c2 y + c3 x y:  dx = g (c1 + c3 y), dy = g (c2 + c3 x)
                const RowsForm f(op);
                const double x = val[a * 64 + lane], y = val[bb * 64 + lane];
                const float ga = (float)(gi * fma(f.c3, y, f.c1)), gb = (float)(gi * fma(f.c3, x, f.c2));
                if (a == bb) adj[a * 64 + lane] += ga + gb;        // (x + x, x * x: one node, both parts)
                else {
                    const float aa = adj[a * 64 + lane], ab = adj[bb * 64 + lane];
                    adj[a * 64 + lane] = aa + ga;
                    adj[bb * 64 + lane] = ab + gb;
                }
            } else if (op == kOpDiv) {
                const double ib = 1.0 / val[bb * 64 + lane], q = val[i * 64 + lane];
                adj[a * 64 + lane] += (float)(gi * ib);
                adj[bb * 64 + lane] -= (float)(gi * q * ib);
            } else if (op == kOpRecip) {                           // d(1/u) = -(1/u)^2 du
                const double q = val[i * 64 + lane];
                adj[a * 64 + lane] -= (float)(gi * q * q);
            } else if (op == kOpParam) {
                if (a != chan) gp[a * 64 + lane] += gi;
            }
        }
    }
    const int64_t wave = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    for (int p = 0; p < P; ++p) {
        double s = gp[p * 64 + lane];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);      // (a butterfly: the same order on every run)
        if (lane == 0) part[wave * P + p] = s;
    }
}

// gparams[p] = sum over the waves' partials, fixed order: grid n_params, block 256
static __global__ __launch_bounds__(256) void ss_dyn_rows_reduce_kernel(const double* __restrict__ part, int64_t n_waves, int P,
                                                                        double* __restrict__ gparams)
{
    __shared__ double sh[256];
    const int p = blockIdx.x, j = threadIdx.x;
    double a = 0.0;
    for (int64_t w = j; w < n_waves; w += 256) a += part[w * P + p];
    sh[j] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (j < off) sh[j] += sh[j + off];
        __syncthreads();
    }
    if (j == 0) gparams[p] = sh[0];
}

}  // namespace wdf
