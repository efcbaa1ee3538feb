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
// The tape travels in the kernel arguments (one dword per operation: scalar loads, the switch is wave-uniform).
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

// the tape forward for this lane's sample: val[node * 64 + lane]
__device__ __forceinline__ void rows_forward(const RowsTape& tp, const double* __restrict__ params, double rv, double* __restrict__ val, int lane)
{
    for (int i = 0; i < tp.n_ops; ++i) {
        const uint32_t c = tp.code[i];
        const int op = (int)(c & 0xffu), a = (int)((c >> 8) & 0xffu), b = (int)((c >> 16) & 0xffu);
        double v;
        switch (op) {
        case kOpConst: v = tp.consts[a]; break;
        case kOpParam: v = (a == tp.chan) ? rv : params[a]; break;
        case kOpAdd: v = val[a * 64 + lane] + val[b * 64 + lane]; break;
        case kOpSub: v = val[a * 64 + lane] - val[b * 64 + lane]; break;
        case kOpMul: v = val[a * 64 + lane] * val[b * 64 + lane]; break;
        case kOpDiv: v = val[a * 64 + lane] / val[b * 64 + lane]; break;
        case kOpNeg: v = -val[a * 64 + lane]; break;
        default: v = 1.0 / val[a * 64 + lane]; break;             // kOpRecip
        }
        val[i * 64 + lane] = v;
    }
}

// grid (ceil(B / 64), ceil(T / tc)), block 64, dynamic LDS n_ops * 64 * 8 bytes
__global__ __launch_bounds__(64) void ss_dyn_rows_kernel(const RowsTape tp, const double* __restrict__ params, const float* __restrict__ r,
                                                         float* __restrict__ rows, int64_t B, int64_t T, int64_t tc)
{
    extern __shared__ double rows_lds[];
    double* __restrict__ val = rows_lds;
    const int lane = threadIdx.x;
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + lane;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const int64_t t0 = (int64_t)blockIdx.y * tc, t1 = t0 + tc < T ? t0 + tc : T;
    const int n = tp.n_out;
    for (int64_t t = t0; t < t1; ++t) {
        const double rv = r ? (double)r[t * B + b] : 0.0;
        rows_forward(tp, params, rv, val, lane);
        if (live) {
            float* __restrict__ o = rows + t * n * B + b;
            for (int k = 0; k < n; ++k) o[(int64_t)k * B] = (float)val[(int)tp.outs[k] * 64 + lane];
        }
    }
}

// The reverse pass.  grid as above; dynamic LDS n_ops * 64 * 12 + n_params * 64 * 8 bytes.
// part: double [gridDim.y * gridDim.x][n_params] -- one partial per wave (the channel's entry is left 0).
__global__ __launch_bounds__(64) void ss_dyn_rows_bwd_kernel(const RowsTape tp, const double* __restrict__ params, const float* __restrict__ r,
                                                             const float* __restrict__ grows, double* __restrict__ part, int64_t B, int64_t T,
                                                             int64_t tc)
{
    extern __shared__ double rows_lds[];
    const int lane = threadIdx.x, n_ops = tp.n_ops, P = tp.n_params;
    double* __restrict__ val = rows_lds;                           // [n_ops][64]
    double* __restrict__ gp = val + n_ops * 64;                    // [P][64]
    float* __restrict__ adj = reinterpret_cast<float*>(gp + P * 64);   // [n_ops][64]
    const int64_t b_raw = (int64_t)blockIdx.x * 64 + lane;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const int64_t t0 = (int64_t)blockIdx.y * tc, t1 = t0 + tc < T ? t0 + tc : T;
    const int n = tp.n_out;
    for (int p = 0; p < P; ++p) gp[p * 64 + lane] = 0.0;
    for (int64_t t = t0; t < t1; ++t) {
        const double rv = r ? (double)r[t * B + b] : 0.0;
        rows_forward(tp, params, rv, val, lane);
        for (int i = 0; i < n_ops; ++i) adj[i * 64 + lane] = 0.0f;
        const float* __restrict__ g = grows + t * n * B + b;
        for (int k = 0; k < n; ++k) adj[(int)tp.outs[k] * 64 + lane] += live ? g[(int64_t)k * B] : 0.0f;    // (an entry may name a node twice)
        for (int i = n_ops - 1; i >= 0; --i) {
            const uint32_t c = tp.code[i];
            const int op = (int)(c & 0xffu), a = (int)((c >> 8) & 0xffu), bb = (int)((c >> 16) & 0xffu);
            const double gi = (double)adj[i * 64 + lane];
            switch (op) {
            case kOpConst: break;
            case kOpParam: if (a != tp.chan) gp[a * 64 + lane] += gi; break;
            case kOpAdd: adj[a * 64 + lane] += (float)gi; adj[bb * 64 + lane] += (float)gi; break;
            case kOpSub: adj[a * 64 + lane] += (float)gi; adj[bb * 64 + lane] -= (float)gi; break;
            case kOpMul: {
                const double va = val[a * 64 + lane], vb = val[bb * 64 + lane];
                adj[a * 64 + lane] += (float)(gi * vb);
                adj[bb * 64 + lane] += (float)(gi * va);            // (a == bb: both land, 2 gi va)
            } break;
            case kOpDiv: {
                const double ib = 1.0 / val[bb * 64 + lane], q = val[i * 64 + lane];
                adj[a * 64 + lane] += (float)(gi * ib);
                adj[bb * 64 + lane] -= (float)(gi * q * ib);
            } break;
            case kOpNeg: adj[a * 64 + lane] -= (float)gi; break;
            default: {                                            // kOpRecip: d(1/u) = -(1/u)^2 du
                const double q = val[i * 64 + lane];
                adj[a * 64 + lane] -= (float)(gi * q * q);
            } break;
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
