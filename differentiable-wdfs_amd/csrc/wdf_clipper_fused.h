// wdf_clipper_fused.h -- the diode-clipper TRAINING STEP in one pass over the data (gfx950).
//
// What the two-kernel step (clipper_fwd_tp_kernel + clipper_bwd_tp_kernel, wdf_clipper.h) moves through
// HBM twice -- x, the state stash written by the forward and read back by the reverse sweep, the target --
// this kernel moves once: x and the target in, y out, 12 B/sample instead of 24, and the root is solved
// once per sample instead of twice.  It can do that because the loss is known while the forward runs
// (mean squared error against a resident target: lpf.py:78,87-90, clipper_pot.py:176) and the circuit has
// only three sufficient statistics for its four parameters {Is, nVt, R, C} (wdf_clipper.h,
// grad_chain_rule: dL/dL, dL/dV|_L, dL/dp): the parameter gradient is carried FORWARD in time as the
// tangent of the state,
//     s_i[n+1] = kappa_n s_i[n] + d_i[n]          kappa = dz'/dz,  d = (dz'/dL, dz'/dV, dz'/dp) at fixed z
//     dLoss/dtheta_i = sum_n g_n (s_i[n+1] + s_i[n]) / 2     (y = (z' + z)/2, g_n = dLoss/dy_n)
// which is the same sum the reverse sweep forms (tf.GradientTape's result, lpf.py:87-90), taken in the
// other order: bwd_step's partials Da, DL, DV, cP are used unchanged, and tests hold the two against each
// other and against the oracle.  No stash, no second root evaluation, ~12 extra VALU per step.
//
// Time-parallel like the other kernels: chunk k of a 64-sequence tile starts from a speculated state
// (warm start from the previous call's snapshots, or a cold warm-up) and from an UNKNOWN tangent sigma;
// everything it accumulates is affine in sigma (s = A sigma + c), so the chunk publishes the record
//     {A_end, c_end[3], GA, G[3], SSE}        (kFsOut floats per sequence)
// and the tile's last wave -- after verifying the state boundaries exactly as tp_finish does -- walks the K
// records in time order (sigma_{k+1} = A_end sigma_k + c_end) and adds up the tile's sums; the last tile
// reduces over tiles, applies the chain rule and (single rank) the Adam update.  A tile with a failed
// boundary is left to clipper_fused_repair_kernel, the next launch, which re-runs the failing chunks from
// the exact state (outputs, record) and then does that tile's combine; whichever tile arrives last --
// in either kernel -- finishes the step.
//
// TWO SEQUENCES PER LANE.  The kernel is bound by VALU issue (~88 instructions per sample-step against 12 bytes), and
// a CDNA4 SIMD issues a packed v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two fp32 operations per lane) in the slot of
// one plain instruction.  Written over the value type V (wdf_vec.h), with V = v2f a lane owns the ADJACENT sequences
// 2l and 2l + 1: every add / mul / fma of the step packs (69 of the 109 instructions two sequences take, no register
// shuffles: the pair lives in an aligned VGPR pair from its 8-byte load to its 8-byte store), transcendentals, compares
// and selects stay per component.  54 issue slots per sample-step instead of 88.  Needs an even B; V = float is the
// same code for odd batches and for A/B.
#pragma once

#include "wdf_clipper.h"

namespace wdf {

// What a chunk hands to its tile's combine.  Per (chunk, sequence) -- the part that depends on the tangent sigma entering
// the chunk, which is only known once the chunks before it have been walked: the record {A, c[3], GA} (MSE + ESR: + HA).
// Per (chunk, tile) -- everything that does not: {G[3], SSE} (MSE + ESR: + {H[3], SYY}) summed over the wave's sequences by
// the chunk wave itself, in double, one set per sequence slot of a lane (so that a repair can replace one slot's share).
// (Round 2 kept all nine / fourteen values per sequence: the walk of the tile's last wave, on the step's critical path,
// moved 9 K floats per sequence in four dependent round trips at K = 32; now 5 K in two.)
constexpr int kFsOut = 5;       // record floats per (chunk, sequence), MSE: {A, c[3], GA}
constexpr int kFsOutEsr = 6;    // MSE + ESR adds HA
constexpr int kFsPart = 4;      // per-wave sums, MSE: {G[3], SSE}
constexpr int kFsPartEsr = 8;   // MSE + ESR adds {H[3], SYY}
template <int LOSS> struct FusedRec {
    static constexpr int N = LOSS == 2 ? kFsOutEsr : kFsOut;
    static constexpr int NP = LOSS == 2 ? kFsPartEsr : kFsPart;
};
// The records live in 16-byte granules, one lane's sequences back to back: float4 [K][kFsQuads][lanes], lanes = 64 x tiles of
// the launch.  A lane stores / loads its chunk record with <= 3 instructions (two sequences x 5 or 6 floats = 10 or 12), and
// the walk of a tile's 32 chunks is 96 loads in flight in two batches -- a wave holds at most 63 outstanding memory
// instructions, which is what made the dword layout's 288 (round 2) and 160 loads four round trips.
constexpr int kFsQuads = 3;
template <int NSEQ, int LOSS> struct FusedQuads { static constexpr int NQ = (NSEQ * FusedRec<LOSS>::N + 3) / 4; };
__device__ __forceinline__ float* rec_chunk(float* rec, int64_t k) { return rec + (size_t)k * kFsQuads * ((size_t)gridDim.x * 64) * 4; }
constexpr int kFusedSchedGroup = 1;   // steps the instruction scheduler may interleave
// Build-time knobs of the A/B runs recorded in DESIGN.md (tools/ab_libs.sh builds variants with -D...)
#ifndef WDF_FUSED_ROWS
#define WDF_FUSED_ROWS 16           // rows per load burst with one sequence per lane and a static resistance
#endif
#ifndef WDF_FUSED_PREFETCH_AT
#define WDF_FUSED_PREFETCH_AT 0     // 0: the next tile's loads at the top of the tile, 1: in its middle
#endif
#ifndef WDF_FUSED_WAVES
#define WDF_FUSED_WAVES 2           // waves per SIMD the one-pass kernel's register allocation is held to
#endif
constexpr int kFlushSteps = 32;     // fp32 -> fp64 accumulation granularity of the loss / tangent sums (as the kernel pair's 32-step blocks)
constexpr int kFusedPrefetchAt = WDF_FUSED_PREFETCH_AT;
constexpr int kFusedRows = WDF_FUSED_ROWS;

// Steps per load burst: 16 with one sequence per lane and a static resistance, half of that with two sequences per
// lane, half again with the per-sample resistance channel (the tile buffers are the bulk of the VGPRs: x, target and r,
// current and next; a tile's loads + stores must also stay inside vmcnt's 6 bits).
template <typename V, int DYN_R> struct FusedTile { static constexpr int NR = kFusedRows / VT<V>::N / (DYN_R == 1 ? 2 : 1); };

// s_waitcnt vmcnt(N) (gfx9 encoding: vmcnt in bits 3:0 and 15:14; expcnt / lgkmcnt fields at "no wait").
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | 0x0F70);
}

// hgs on the steps that carry loss, 0 on the first n_masked steps of a tile: two SALU instructions, written out because
// the compiler otherwise forms the per-step mask on the VALU (v_cmp_lt_i64 + v_cndmask, 2 of the step's instructions).
__device__ __forceinline__ float step_loss_scale(int i, int n_masked, float hgs)
{
    float hm;
    asm("s_cmp_lt_i32 %1, %2\n\ts_cselect_b32 %0, 0, %3" : "=s"(hm) : "s"(i), "s"(n_masked), "s"(hgs) : "scc");
    return hm;
}

// The lane's sequences: N = VT<V>::N adjacent ones starting at b (B is a multiple of N: the C ABI picks V).
// Lanes past the end shadow the last group (same values to the same addresses).
template <typename V>
struct LaneOwn {
    static constexpr int N = VT<V>::N;
    int64_t b;
    uint32_t boff;           // b * 4: byte offset inside a [B] row (B < 2^24)
    bool live;
    __device__ __forceinline__ explicit LaneOwn(int64_t B)
    {
        const int64_t raw = ((int64_t)blockIdx.x * 64 + threadIdx.x) * N;
        live = raw < B;
        b = live ? raw : B - N;
        boff = (uint32_t)b * 4u;
    }
};

// Chunk spans.  skew = 0: K equal chunks of L steps.  skew > 0 (K even): the first K/2 chunks own L + skew steps, the other
// K/2 own L - skew.  Why: with two chunk waves per SIMD the hardware issues oldest-first, so the wave dispatched first (the
// lower chunk index: workgroups are dispatched in grid order) runs at ~3.7 steps/us, its younger partner at ~2.3, and after
// the older one has finished the younger runs alone at the single-wave rate for a third of the kernel (tools/dbg_times.py).
// Giving the older wave the longer chunk lets both finish together: -5 % (one sequence per lane, 16 chunks) / -3.5 % (two
// per lane, 32 chunks) on the same box.  The host applies it only when the launch is ~2 waves per SIMD (wdf_capi_clipper.hip).
__device__ __forceinline__ void chunk_span(int64_t k, int64_t K, int64_t L, int64_t skew, int64_t T, int64_t& t0, int64_t& t1)
{
    const int64_t half = K / 2;
    const bool first = k < half;
    t0 = first ? k * (L + skew) : half * (L + skew) + (k - half) * (L - skew);
    const int64_t len = skew == 0 ? L : (first ? L + skew : L - skew);
    t1 = (t0 + len < T) ? t0 + len : T;
}

// Rows of the time-major arrays through buffer descriptors: `buffer_load_dword[x2] v, voff, s[rsrc], soff offen` takes
// the row's byte offset from an SGPR and the lane's from one VGPR, so a tile of rows costs no VALU instruction and no
// SGPR pair per row (with global_load / global_store the compiler either adds 64-bit addresses on the VALU or keeps 32
// row pointers per stream in SGPRs, spills them to VGPR lanes and reads them back with v_readlane: ~7 of the step's
// ~105 VALU instructions).  x, target and y share the row offsets i * 4B.  The descriptor is rebuilt per tile from
// a 64-bit scalar base; offsets inside a tile stay below 2^32 for B < 2^24 (checked by the C ABI).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float* row0)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row0), 0, 0xffffffffu, 0x00020000);
}

typedef unsigned int v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void buf_load(float& v, __amdgpu_buffer_rsrc_t rs, uint32_t boff, uint32_t soff)
{
    v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, boff, soff, 0));
}
__device__ __forceinline__ void buf_load(v2f& v, __amdgpu_buffer_rsrc_t rs, uint32_t boff, uint32_t soff)
{
    v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rs, boff, soff, 0));
}
// aux 2: nt (streaming output, not read again by this kernel)
#ifndef WDF_FUSED_Y_AUX
#define WDF_FUSED_Y_AUX 2           // (A/B: 16 = sc1, write-through at agent scope; 18 = nt | sc1)
#endif
__device__ __forceinline__ void buf_store_nt(float v, __amdgpu_buffer_rsrc_t rs, uint32_t boff, uint32_t soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, boff, soff, WDF_FUSED_Y_AUX);
}
__device__ __forceinline__ void buf_store_nt(v2f v, __amdgpu_buffer_rsrc_t rs, uint32_t boff, uint32_t soff)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), rs, boff, soff, WDF_FUSED_Y_AUX);
}

// 16-byte write-through store / load at agent scope (`buffer_store_dwordx4 ... sc1` / `buffer_load_dwordx4 ... sc1`): how a
// wave hands data to another wave of the same launch (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 on both sides).
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void publish_quad(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff, const float (&f)[4])
{
    __builtin_amdgcn_raw_buffer_store_b128(v4u{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])},
                                           rs, voff, soff, 16);
}
__device__ __forceinline__ v4u load_published_quad(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16);
}

template <typename V, int NR>
__device__ __forceinline__ void load_rows(const float* row0, uint32_t boff, uint32_t rowb, V (&v)[NR])
{
    const __amdgpu_buffer_rsrc_t rs = row_rsrc(row0);
#pragma unroll
    for (int i = 0; i < NR; ++i) buf_load(v[i], rs, boff, i * rowb);
}

// x tile: time-major through the descriptor, batch-major as 16-byte loads of each sequence's own row
template <typename V, bool TM, bool VEC4, int NR>
__device__ __forceinline__ void load_x_tile(const float* __restrict__ x, const LaneOwn<V>& q, int64_t B, int64_t T, int64_t t0,
                                            uint32_t rowb, V (&v)[NR])
{
    if constexpr (TM) {
        load_rows<V, NR>(x + t0 * B, q.boff, rowb, v);
    } else {
#pragma unroll
        for (int j = 0; j < VT<V>::N; ++j) {
            float tmp[NR];
            load_row<NR, VEC4>(x, q.b + j, T, t0, tmp);
#pragma unroll
            for (int i = 0; i < NR; ++i) vset(v[i], j, tmp[i]);
        }
    }
}

// Batch-major x through LDS.  Read straight from its own rows (load_x_tile above) a lane's 16-byte load touches a 128-byte
// line of its own -- 64 lines per instruction, an eighth of each used -- and by the time the lane comes back for the next
// 16 bytes the line has left L2: measured (TCC_EA0_RDREQ_128B, 8192 x 4096), every visit was a 128-byte read from HBM, and
// with 16-step (64-byte) visits x was still fetched twice, 575 MB per step against the time-major step's 441.
// Here a wave takes each line of x ONCE, whole: eight adjacent lanes read the eight 16-byte granules of one row's 128
// bytes (32 steps), eight rows per instruction, and park them in LDS ([row][32 + 4 floats]: with the pad the eight lanes of
// a 128-bit access fall on distinct banks but for one pair), from where each lane reads its own rows' steps a tile at a
// time.  A whole 32-step window of the wave's 64 N rows would be 64 staging registers per lane; so the two HALVES of the
// rows run their windows 16 steps apart -- rows [0, 32 N) start a window where t = 0 mod 32, rows [32 N, 64 N) where
// t = 16 mod 32 (absolute time: a window is a line when the rows are line-aligned) -- and every 16 steps one half's next
// window (4 N loads, 16 N registers) is issued, a tile later written over that half's spent window.  The loads go out a
// tile before they are written to LDS, so their latency stays behind a tile of steps as the direct loads' did.  One
// window buffer per wave: 18 KB with two sequences per lane, eight waves' worth fit a CU's 160 KB.
// Only with 16-byte aligned rows (VEC4) and without the per-sample resistance channel (its tiles are half as long).
constexpr int kWinSteps = 32, kWinPhase = 16, kWinPitch = kWinSteps + 4;
template <int DYN_R, bool TM, bool VEC4> constexpr bool fused_x_window() { return !TM && VEC4 && DYN_R != 1; }
template <int FLOATS>
__device__ __forceinline__ float* wave_lds()
{
    __shared__ __attribute__((aligned(16))) float buf[FLOATS];
    return buf;
}
template <typename V, bool ON> struct XWindow {
    __device__ __forceinline__ XWindow(const float*, const LaneOwn<V>&, int64_t, int64_t) {}
};
template <typename V>
struct XWindow<V, true> {
    static constexpr int N = VT<V>::N, NR = kWinPhase / N;   // a tile: NR steps, N tiles to a phase of 16 steps
    static constexpr int RW = 64 * N, HR = RW / 2;           // the wave's rows; a half
#ifndef WDF_WIN_GRANULE
#define WDF_WIN_GRANULE 4                                     // floats per lane and load: 4 (eight lanes to a line) or 2 (sixteen)
#endif
    static constexpr int G = WDF_WIN_GRANULE, LPR = kWinSteps / G, RPI = 64 / LPR;   // lanes per row, rows per instruction
    static constexpr int NI = HR / RPI;                      // loads per lane and half window
    static constexpr int kFloats = RW * kWinPitch;
    typedef unsigned int raw_t __attribute__((ext_vector_type(G)));
    raw_t raw[NI];
    v4u tail[N * 2];            // (a chunk's start) the other half's 16 steps
    float* lds;
    float* wr;                  // where the lane's granule of a half window's load 0 goes (half 0)
    const float* rd;            // the lane's first sequence's row in LDS
    const float* x0;            // the wave's first row
    int64_t T, rows;            // rows of this wave (RW but for the batch's last wave)
    uint32_t voff, rstep;       // the lane's byte offset inside a load's rows; that many rows, bytes
    uint32_t hfoff;             // 16 where the lane's rows are in the upper half
    __device__ __forceinline__ XWindow(const float* __restrict__ x, const LaneOwn<V>& q, int64_t B, int64_t T_) : T(T_)
    {
        const int64_t row0 = (int64_t)blockIdx.x * RW;
        rows = B - row0 < RW ? B - row0 : RW;
        x0 = x + row0 * T;
        lds = wave_lds<kFloats>();
        const uint32_t lr = threadIdx.x / LPR, g = threadIdx.x % LPR;
        wr = lds + lr * kWinPitch + g * G;
        rd = lds + (q.b - row0) * kWinPitch;
        hfoff = (q.b - row0) >= HR ? (uint32_t)kWinPhase : 0u;
        voff = (lr * (uint32_t)T + g * G) * 4u;
        rstep = (uint32_t)RPI * (uint32_t)T * 4u;
#pragma unroll
        for (int m = 0; m < NI; ++m) raw[m] = raw_t{};
#pragma unroll
        for (int m = 0; m < N * 2; ++m) tail[m] = v4u{0, 0, 0, 0};
    }
    static __device__ __forceinline__ void buf_load_raw(v4u& v, __amdgpu_buffer_rsrc_t rs, uint32_t vo, uint32_t so)
    {
        v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
    }
    static __device__ __forceinline__ void buf_load_raw(v2u& v, __amdgpu_buffer_rsrc_t rs, uint32_t vo, uint32_t so)
    {
        v = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0);
    }
    // the window [ts, ts + 32) of one half's rows: global -> registers
    __device__ __forceinline__ void issue(int half, int64_t ts)
    {
        if (rows == RW && ts + kWinSteps <= T) {
            const __amdgpu_buffer_rsrc_t rs = row_rsrc(x0 + (int64_t)half * HR * T + ts);
#pragma unroll
            for (int m = 0; m < NI; ++m) buf_load_raw(raw[m], rs, voff, m * rstep);
        } else {                // the batch's last wave / a window across the sequence's end: rows and granules clamped (read, not used)
            const int64_t g = threadIdx.x % LPR, lr = threadIdx.x / LPR;
            const int64_t tg = ts + G * g + G <= T ? ts + G * g : T - G;
#pragma unroll
            for (int m = 0; m < NI; ++m) {
                const int64_t row = half * HR + m * RPI + lr < rows ? half * HR + m * RPI + lr : rows - 1;
                raw[m] = *reinterpret_cast<const raw_t*>(x0 + row * T + tg);
            }
        }
    }
    // registers -> LDS, over that half's previous window (its last tile has been read)
    __device__ __forceinline__ void commit(int half)
    {
        float* w = wr + half * (HR * kWinPitch);
#pragma unroll
        for (int m = 0; m < NI; ++m) *reinterpret_cast<raw_t*>(w + m * RPI * kWinPitch) = raw[m];
    }
    // the tile at t of the lane's sequences (the half whose window began this phase reads its head, the other half 16 steps in)
    __device__ __forceinline__ void read(int64_t t, V (&v)[NR])
    {
        const uint32_t sel = (uint32_t)(t / kWinPhase) & 1u, u = (uint32_t)(t / NR) % N;
        const float* p = rd + (((sel * kWinPhase) ^ hfoff) + u * NR);
#pragma unroll
        for (int j = 0; j < N; ++j) {
#pragma unroll
            for (int e = 0; e < NR / 4; ++e) {
                const float4 f = *reinterpret_cast<const float4*>(p + j * kWinPitch + e * 4);
                vset(v[4 * e], j, f.x); vset(v[4 * e + 1], j, f.y); vset(v[4 * e + 2], j, f.z); vset(v[4 * e + 3], j, f.w);
            }
        }
    }
    // A chunk's start at tw (a multiple of 16): the half whose window starts there takes it whole; the other half is 16 steps
    // into a window it never held, and takes its second 16 steps (four lanes to a row's 64 bytes, straight to LDS).
    __device__ __forceinline__ void first_issue(int64_t tw)
    {
        const int a = (int)(tw / kWinPhase) & 1, o = a ^ 1;
        issue(a, tw);
        const int64_t g = threadIdx.x & 3, lr = threadIdx.x >> 2;
        const int64_t tg = tw + 4 * g + 4 <= T ? tw + 4 * g : T - 4;
#pragma unroll
        for (int m = 0; m < N * 2; ++m) {
            const int64_t row = o * HR + m * 16 + lr < rows ? o * HR + m * 16 + lr : rows - 1;
            tail[m] = *reinterpret_cast<const v4u*>(x0 + row * T + tg);
        }
    }
    __device__ __forceinline__ void first(int64_t tw, int64_t t_end, V (&v)[NR])
    {
        const int a = (int)(tw / kWinPhase) & 1, o = a ^ 1;
        const int g = threadIdx.x & 3, lr = threadIdx.x >> 2;
        commit(a);
#pragma unroll
        for (int m = 0; m < N * 2; ++m)
            *reinterpret_cast<v4u*>(lds + (o * HR + m * 16 + lr) * kWinPitch + kWinPhase + g * 4) = tail[m];
        read(tw, v);
        if (N == 1 && tw + NR < t_end) issue(o, tw + NR);
    }
    // the tile at tn (called a tile ahead, as the direct loads are)
    __device__ __forceinline__ void next(int64_t tn, int64_t t_end, V (&v)[NR])
    {
        const int sel = (int)(tn / kWinPhase) & 1, u = (int)(tn / NR) % N;
        if (u == 0) commit(sel);
        read(tn, v);
        if (u == N - 1 && tn + NR < t_end) issue(sel ^ 1, tn + NR);
    }
};

// the lane's N adjacent elements of a [B] row (plain / published)
template <typename V>
__device__ __forceinline__ V load_own(const float* row, const LaneOwn<V>& q)
{
    V r = vsplat<V>(0.0f);
#pragma unroll
    for (int j = 0; j < VT<V>::N; ++j) vset(r, j, row[q.b + j]);
    return r;
}
template <typename V>
__device__ __forceinline__ void store_own(float* row, const LaneOwn<V>& q, V v)
{
#pragma unroll
    for (int j = 0; j < VT<V>::N; ++j) row[q.b + j] = vget(v, j);
}
template <typename V>
__device__ __forceinline__ void publish_own(float* row, const LaneOwn<V>& q, V v)
{
#pragma unroll
    for (int j = 0; j < VT<V>::N; ++j) __hip_atomic_store(row + q.b + j, vget(v, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one step's input of the lane's sequences (tail / masked tiles)
template <typename V, bool TM>
__device__ __forceinline__ V load_step(const float* __restrict__ x, const LaneOwn<V>& q, int64_t B, int64_t T, int64_t t)
{
    V r = vsplat<V>(0.0f);
#pragma unroll
    for (int j = 0; j < VT<V>::N; ++j) vset(r, j, load_one<TM>(x, q.b + j, B, T, t));
    return r;
}

// Tangent state of the lane's sequences inside a chunk.  G* accumulate sum_n s[n] (hg_n + hg_{n-1}) (summation
// by parts of sum_n hg_n (s[n+1] + s[n]), hg = g/2): one FMA per statistic and step.
template <typename V>
struct FusedTan {
    V A, cL, cV, cP;            // s_i = A sigma_i + c_i
    V GA, GL, GV, GP;           // running sums since the last flush
    V hg_prev;
    V sse;                      // hgs x sum of squared errors
    // MSE + ESR (LOSS = 2) only: the same sums weighted by y instead of (y - target), and hgs x sum y^2
    V HA, HL, HV, HP, hy_prev, syy;
    __device__ __forceinline__ void init()
    {
        A = vsplat<V>(1.0f);
        cL = cV = cP = GA = GL = GV = GP = hg_prev = sse = vsplat<V>(0.0f);
        HA = HL = HV = HP = hy_prev = syy = vsplat<V>(0.0f);
    }
    template <int LOSS>
    __device__ __forceinline__ void pin()
    {
        vpin(A); vpin(cL); vpin(cV); vpin(cP); vpin(GA); vpin(GL); vpin(GV); vpin(GP); vpin(sse);
        if constexpr (LOSS == 2) { vpin(HA); vpin(HL); vpin(HV); vpin(HP); vpin(syy); }
    }
};

// One step: forward (the arithmetic of fwd_step, same expressions) + partials (those of bwd_tp_step) +
// tangent update.  hgs = gscale / 2 (0 on masked steps).  Returns y.
// LOSS = 2 (MSE + ESR, clipper_pot.py:146-156,177): dLoss/dy = ga (y - target) + gb y with ga, gb functions of the GLOBAL
// sums S = sum (y - t)^2 and E = sum y^2, known only after the pass -- so the pass carries BOTH tangent-weighted sums,
// P_i = sum (y - t) dy/dtheta_i and Q_i = sum y dy/dtheta_i (hgs = 1/2 on live steps), and the last tile forms
// ga P + gb Q.  Seven more VALU per step.
template <int DYN_R, bool SYM, int FAST, typename V, int LOSS = 1>
__device__ __forceinline__ V fused_step(const ClipConsts& c, V xin, V rin, V tgt, float hgs, V& z, FusedTan<V>& s)
{
    V p, Rp, L;
    step_coeffs<DYN_R, V>(c, rin, p, Rp, L);
    const V b_diff = z - xin;
    const V b_temp = -p * b_diff;
    const V a = z + b_temp;
    const DiodeOutT<V> o = diode_pair<SYM, V, FAST>(a, L, c.d);
    const V zn = o.b + b_temp;
    const V y = 0.5f * (zn + z);
    // partials of the root (wdf_clipper.h, bwd_step / bwd_tp_step)
    const V w0p = o.w0 * vrcp(o.w0 + 1.0f);
    V Da, DL, DV;
    if constexpr (SYM && FAST == kRootLean && !DYN_R) {
        // LEAN (round 6: three packed instructions less): the constant factors of D_L (-2 nVt N) and D_V (-1 / nVt) are left out
        // of the per-step partials -- the tangent recurrences and their sums are linear in them, fused_publish_record puts the
        // factors back (FusedScale) -- and kappa = Da (1 - p) - p, cP = -(1 + Da) b_diff are formed without 1 + Da.
        const V w1p = vfma(-o.w1, o.w1, o.w1);                      // omega_1 <= 5.6e-4 (root_tier): w (1 - w)
        const V sp = w0p + w1p;
        const V tl = vsel_nonzero(a, -2.0f, 0.0f);                  // -2 lam^2
        const float tvm = c.d.two_v * c.d.m_dn;
        Da = vfma(tl, sp, 1.0f);
        DL = vcopysign(w0p - w1p, a);                               // x (-2 nVt N)
        DV = vfma(tl * a, sp, tvm * vcopysign(o.dw, a));            // x (-1 / nVt):  D_V = -(tl a sp + (a - b)) / nVt
        const V cPl = vfma(-Da, b_diff, -b_diff);
        const V kappa_l = vfma(Da, 1.0f - c.p, -c.p);
        const V d = y - tgt;
        const V hg = hgs * d;
        s.sse = vfma(hg, d, s.sse);
        const V hh = hg + s.hg_prev;
        s.hg_prev = hg;
        s.GA = vfma(hh, s.A, s.GA);
        s.GL = vfma(hh, s.cL, s.GL);
        s.GV = vfma(hh, s.cV, s.GV);
        s.GP = vfma(hh, s.cP, s.GP);
        if constexpr (LOSS == 2) {
            const V hy = hgs * y;
            s.syy = vfma(hy, y, s.syy);
            const V h2 = hy + s.hy_prev;
            s.hy_prev = hy;
            s.HA = vfma(h2, s.A, s.HA);
            s.HL = vfma(h2, s.cL, s.HL);
            s.HV = vfma(h2, s.cV, s.HV);
            s.HP = vfma(h2, s.cP, s.HP);
        }
        s.A = s.A * kappa_l;
        s.cL = vfma(kappa_l, s.cL, DL);
        s.cV = vfma(kappa_l, s.cV, DV);
        s.cP = vfma(kappa_l, s.cP, cPl);
        z = zn;
        return y;
    } else
    if constexpr (SYM && FAST) {
        // omega_1 <= omega(-4) = 0.018 here (series-only region, checked once per kernel): omega/(1 + omega) by
        // its alternating series to the cubic term (next term 1e-7 relative) instead of a reciprocal.
        // lam X = copysign(X, a) for the two differences, both >= 0 because omega and omega' are increasing and
        // both exactly 0 at a = 0 where w0 == w1 bit for bit (diode_pair, FAST); lam^2 = (a != 0).
        V w1p;
        if constexpr (FAST == kRootLean) w1p = vfma(-o.w1, o.w1, o.w1);      // omega_1 <= 5.6e-4 (root_tier): w (1 - w)
        else w1p = o.w1 * vfma(-o.w1, vfma(-o.w1, 1.0f - o.w1, 1.0f), 1.0f);
        const V sp = w0p + w1p;
        const V tl = vsel_nonzero(a, -2.0f, 0.0f);                  // -2 lam^2
        const float tvm = c.d.two_v * c.d.m_dn;
        Da = vfma(tl, sp, 1.0f);
        DL = (-tvm) * vcopysign(w0p - w1p, a);
        DV = vfma(tl * a, sp * (-1.0f / c.V), (-2.0f * c.d.m_dn) * vcopysign(o.dw, a));
    } else {
        const V w1p = o.w1 * vrcp(o.w1 + 1.0f);
        const V l2 = o.lam * o.lam;
        const V sp = w0p + w1p;
        const V tl = -2.0f * l2;
        Da = vfma(tl, sp, 1.0f);
        if constexpr (SYM) {
            const float tvm = c.d.two_v * c.d.m_dn;
            DL = (-tvm) * (o.lam * (w0p - w1p));
            DV = vfma(tl * a, sp * (-1.0f / c.V), (-2.0f * c.d.m_dn) * (o.lam * (o.w0 - o.w1)));
        } else {
            DL = (-c.d.two_v) * (o.lam * (o.m0 * w0p - o.m1 * w1p));
            DV = vfma(tl * a, sp * (-1.0f / c.V), -2.0f * (o.lam * (o.m0 * o.w0 - o.m1 * o.w1)));
        }
    }
    const V opd = Da + 1.0f;
    V cP = -opd * b_diff;
    if constexpr (DYN_R) cP = Rp * vfma(cP, p, DL);
    const V kappa = vfma(-p, opd, Da);
    // loss and tangent
    const V d = y - tgt;
    const V hg = hgs * d;
    s.sse = vfma(hg, d, s.sse);                          // hgs x the squared error (0 on masked steps); no branch here:
                                                         // a block boundary per step lets LLVM sink every step's tangent
                                                         // work to the end of the tile (6 live values per step)
    const V hh = hg + s.hg_prev;
    s.hg_prev = hg;
    s.GA = vfma(hh, s.A, s.GA);
    s.GL = vfma(hh, s.cL, s.GL);
    s.GV = vfma(hh, s.cV, s.GV);
    s.GP = vfma(hh, s.cP, s.GP);
    if constexpr (LOSS == 2) {
        const V hy = hgs * y;
        s.syy = vfma(hy, y, s.syy);
        const V h2 = hy + s.hy_prev;
        s.hy_prev = hy;
        s.HA = vfma(h2, s.A, s.HA);
        s.HL = vfma(h2, s.cL, s.HL);
        s.HV = vfma(h2, s.cV, s.HV);
        s.HP = vfma(h2, s.cP, s.HP);
    }
    s.A = s.A * kappa;
    s.cL = vfma(kappa, s.cL, DL);
    s.cV = vfma(kappa, s.cV, DV);
    s.cP = vfma(kappa, s.cP, cP);
    z = zn;
    return y;
}

// fp64 totals of a chunk, per sequence; the fp32 running sums are flushed into them every tile of steps
template <typename V>
struct FusedSums {
    static constexpr int N = VT<V>::N;
    double GA[N], GL[N], GV[N], GP[N], sse[N];
    double HA[N], HL[N], HV[N], HP[N], syy[N];       // LOSS = 2 only
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int j = 0; j < N; ++j) GA[j] = GL[j] = GV[j] = GP[j] = sse[j] = HA[j] = HL[j] = HV[j] = HP[j] = syy[j] = 0.0;
    }
    template <int LOSS>
    __device__ __forceinline__ void flush(FusedTan<V>& s)
    {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            GA[j] += (double)vget(s.GA, j); GL[j] += (double)vget(s.GL, j); GV[j] += (double)vget(s.GV, j);
            GP[j] += (double)vget(s.GP, j); sse[j] += (double)vget(s.sse, j);
            if constexpr (LOSS == 2) {
                HA[j] += (double)vget(s.HA, j); HL[j] += (double)vget(s.HL, j); HV[j] += (double)vget(s.HV, j);
                HP[j] += (double)vget(s.HP, j); syy[j] += (double)vget(s.syy, j);
            }
        }
        s.GA = s.GL = s.GV = s.GP = s.sse = vsplat<V>(0.0f);
        if constexpr (LOSS == 2) s.HA = s.HL = s.HV = s.HP = s.syy = vsplat<V>(0.0f);
    }
};

// The chunk's hand-over (write-through: another wave of this launch reads it): the per-sequence record and the wave's own
// sums.  wpart: double [tiles][K][NSEQ][NP]; `slot0`: the first sequence slot this call covers (the repair re-runs one slot
// of 64 sequences at a time with V = float).  Dead lanes (past the end of the batch) add nothing to the sums and keep a
// record slot of their own.
// What the LEAN step leaves out of its per-step partials (fused_step): cL, GL, HL carry D_L / (-2 nVt N), cV, GV, HV carry
// D_V / (-1 / nVt); every other tier runs with {1, 1}.
struct FusedScale { float L, V; };

template <typename V, int LOSS, int NSEQ>
__device__ __forceinline__ void fused_publish_record(float* rec, double* wpart, int64_t k, int64_t K, bool live,
                                                     int slot0, const FusedTan<V>& s, const FusedSums<V>& d, float hgs,
                                                     FusedScale fsc = FusedScale{1.0f, 1.0f})
{
    constexpr int NREC = FusedRec<LOSS>::N, NP = FusedRec<LOSS>::NP, NQ = FusedQuads<NSEQ, LOSS>::NQ;
    const double inv = hgs != 0.0f ? 1.0 / (double)hgs : 0.0;
    const uint32_t lanes = gridDim.x * 64u, lane = blockIdx.x * 64u + threadIdx.x;
    float* chunk = rec_chunk(rec, k);
    float f[NQ * 4];
#pragma unroll
    for (int i = 0; i < NQ * 4; ++i) f[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < VT<V>::N; ++j) {
        // the boundary term of the summation by parts: s[t1] hg_{t1-1}
        const double h = (double)vget(s.hg_prev, j), h2 = (double)vget(s.hy_prev, j);
        const float A = vget(s.A, j), cL = fsc.L * vget(s.cL, j), cV = fsc.V * vget(s.cV, j), cP = vget(s.cP, j);
        const float v[kFsOutEsr] = {A, cL, cV, cP, (float)(d.GA[j] + h * A), (float)(d.HA[j] + h2 * A)};
        if constexpr (VT<V>::N == NSEQ) {
#pragma unroll
            for (int i = 0; i < NREC; ++i) f[j * NREC + i] = v[i];
        } else {                                             // one slot of the lane's granules (the repair's re-run)
#pragma unroll
            for (int i = 0; i < NREC; ++i) {
                const int e = (slot0 + j) * NREC + i;
                __hip_atomic_store(chunk + ((size_t)(e >> 2) * lanes + lane) * 4 + (e & 3), v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        double p[kFsPartEsr] = {(double)fsc.L * d.GL[j] + h * cL, (double)fsc.V * d.GV[j] + h * cV, d.GP[j] + h * cP, d.sse[j] * inv,
                                (double)fsc.L * d.HL[j] + h2 * cL, (double)fsc.V * d.HV[j] + h2 * cV, d.HP[j] + h2 * cP, d.syy[j] * inv};
        double* w = wpart + (((int64_t)blockIdx.x * K + k) * NSEQ + slot0 + j) * NP;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const double t = wave_sum_dpp(live ? p[i] : 0.0);
            if (threadIdx.x == 0) __hip_atomic_store(w + i, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (VT<V>::N == NSEQ) {
        const __amdgpu_buffer_rsrc_t rs = row_rsrc(chunk);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float g[4] = {f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
            publish_quad(rs, lane * 16u, (uint32_t)q * lanes * 16u, g);
        }
    }
}

// Chunk geometry as clipper_fwd_tp_body (L, W multiples of kTile); target [T][B]; skip: steps below it carry no loss.
// LOSS = 0: the plain time-parallel forward (no target, no tangent, no record; STASH: the state before every step goes to
// zstash [T][B] for the reverse sweep) -- clipper_fwd_tp_kernel below runs this same body.
template <int DYN_R, bool SYM, bool TM, bool VEC4, int FAST, typename V, int LOSS, bool STASH = false>
__device__ __forceinline__ void clipper_fused_body(
    const ClipConsts& c_in, const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ target,
    float* __restrict__ y, float* __restrict__ zstash, const float* __restrict__ z0, float* __restrict__ zT, float* __restrict__ zwarm,
    float* __restrict__ zend, float* rec, const float* __restrict__ theta, const TpCtl* __restrict__ ctl,
    float* __restrict__ snap, int J, int64_t B, int64_t T, int64_t L, int64_t W, float hgs, int64_t skip, int64_t skew = 0,
    double* wpart = nullptr)
{
    constexpr int NR = FusedTile<V, DYN_R>::NR;
#ifdef WDF_DBG_TIMES
    unsigned long long dbg_p[5];
    dbg_p[0] = __builtin_amdgcn_s_memtime();
#endif
    const LaneOwn<V> q(B);
    // (DYN_R = 2: the lane's sequences' pot values, read once -- their first sample -- and calc_impedance's result with them)
    const ClipConstsSeq<V> c = make_seq_consts<DYN_R, V>(c_in, DYN_R == 2 ? load_step<V, TM>(r, q, B, T, 0) : vsplat<V>(1.0f));
    const int64_t k = blockIdx.y, K = gridDim.y;
    int64_t t0, t1;
    chunk_span(k, K, L, skew, T, t0, t1);
    int64_t tw = 0;
    V z = vsplat<V>(0.0f);
    // The controller first, all of it in one flight of loads (geometry tag, ring head, warm-up units, the three parameter
    // vectors of the extrapolation); then EVERYTHING the wave needs to take its first step in a second flight: the first
    // tile of x / target and the three snapshots the start state is extrapolated from.  (Read where they were used, these
    // were four dependent round trips at the head of every chunk.)
    TpCtl c0 = {};
    if (ctl != nullptr) c0 = *ctl;
    __builtin_amdgcn_sched_barrier(0);
    const bool stateful = ctl != nullptr && c0.geom == tp_geom_tag(K, J, skew != 0);
    const int valid = stateful ? c0.valid : 0;
    const int head = stateful ? c0.head : 0;
    const bool warm_start = k > 0 && valid > 0 && !(stateful && c0.cold_hold > 0);   // (cold_hold: tp_publish_status_and_steer)
    const int jw = warm_start ? c0.j_next : 0;
    if (warm_start) tw = t0 - (int64_t)kWarmStep * jw;
    else tw = (t0 > W) ? t0 - W : 0;
    float* __restrict__ snapw = (snap != nullptr && k + 1 < K) ? snap + ((int64_t)((head + 1) % kTpRing) * J * K + k) * B : nullptr;

    const uint32_t rowb = (uint32_t)B * 4u;
    const uint32_t boff = q.boff;
    V xc[NR], xn[NR], rc[NR], rn[NR], gc[NR], gn[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) { xc[i] = xn[i] = gc[i] = gn[i] = vsplat<V>(0.0f); rc[i] = rn[i] = vsplat<V>(1.0f); }
    const int64_t nfull_end = t1 - (t1 - tw) % NR;
    constexpr bool WIN = fused_x_window<DYN_R, TM, VEC4>();   // batch-major x: whole lines through LDS (XWindow)
    XWindow<V, WIN> xw(x, q, B, T);
    if constexpr (WIN) static_assert(NR == XWindow<V, true>::NR, "a window is N tiles");
    if (tw < nfull_end) {
        if constexpr (WIN) xw.first_issue(tw);
        else load_x_tile<V, TM, VEC4, NR>(x, q, B, T, tw, rowb, xn);
        if constexpr (DYN_R == 1) load_x_tile<V, TM, VEC4, NR>(r, q, B, T, tw, rowb, rn);
        if constexpr (LOSS != 0) { if (tw >= t0) load_rows<V, NR>(target + tw * B, boff, rowb, gn); }
    }
    if (warm_start) {
        // (valid = 1: one snapshot, taken as it is; 2: secant; 3: + parabola -- the missing ones alias the newest, unused)
        const int h2 = valid > 1 ? (head + kTpRing - 1) % kTpRing : head;
        const int h3 = valid > 2 ? (head + kTpRing - 2) % kTpRing : h2;
        z = load_own<V>(snap + (((int64_t)head * J + jw) * K + (k - 1)) * B, q);
        const V z2 = load_own<V>(snap + (((int64_t)h2 * J + jw) * K + (k - 1)) * B, q);
        const V z3 = load_own<V>(snap + (((int64_t)h3 * J + jw) * K + (k - 1)) * B, q);
        __builtin_amdgcn_sched_barrier(0);                  // (the tile's loads stay above: one flight)
        if (valid > 1) {                                    // extrapolated along the parameter path (tp_extrapolation)
            const TpExtrap e = tp_extrapolation(theta, c0, valid);
            // (written around z1 so that an unchanged theta returns the snapshot bit for bit)
            const V zs = vfma(vsplat<V>(e.lam), z - z2, z);                       // secant
            if (e.quad) {
                const V zq = vfma(vsplat<V>(e.w3), z3 - z, vfma(vsplat<V>(e.w2), z2 - z, z));
                const V c = zq - zs;                                              // what the parabola adds: kept where it is signal
                z = zs + vsel(vgt_c(vabs(c), 4.0e-7f), c, vsplat<V>(0.0f));
            } else {
                z = zs;
            }
        }
    } else if (tw == 0 && z0) {
        z = load_own<V>(z0, q);
    }
    int64_t t = tw;
    if constexpr (WIN) { if (tw < nfull_end) xw.first(tw, nfull_end, xn); }
#ifdef WDF_DBG_TIMES
    dbg_p[1] = __builtin_amdgcn_s_memtime();
#endif
    for (; t < t0 && t < nfull_end; t += NR) {              // ---- warm-up tiles: forward only, nothing stored
#pragma unroll
        for (int i = 0; i < NR; ++i) { xc[i] = xn[i]; if constexpr (DYN_R == 1) rc[i] = rn[i]; }
        if (t + NR < nfull_end) {
            if constexpr (WIN) xw.next(t + NR, nfull_end, xn);
            else load_x_tile<V, TM, VEC4, NR>(x, q, B, T, t + NR, rowb, xn);
            if constexpr (DYN_R == 1) load_x_tile<V, TM, VEC4, NR>(r, q, B, T, t + NR, rowb, rn);
            if constexpr (LOSS != 0) { if (t + NR >= t0) load_rows<V, NR>(target + (t + NR) * B, boff, rowb, gn); }   // the first owned tile's target
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) (void)fwd_step<DYN_R, SYM, V, FAST>(c, xc[i], rc[i], z);
    }
#ifdef WDF_DBG_TIMES
    dbg_p[2] = __builtin_amdgcn_s_memtime();
#endif
    publish_own<V>(zwarm + k * B, q, z);
    wait_vmcnt<0>();
#ifdef WDF_DBG_TIMES
    dbg_p[3] = __builtin_amdgcn_s_memtime();
#endif                                        // (the first owned tile's loads: see the wait on the back edge below)
    FusedTan<V> s;
    s.init();
    FusedSums<V> d;
    d.init();
#ifdef WDF_DBG_TIMES
    unsigned long long dbg_wait = 0;
#endif
    for (; t < nfull_end; t += NR) {                        // ---- owned tiles
#pragma unroll
        for (int i = 0; i < NR; ++i) { xc[i] = xn[i]; gc[i] = gn[i]; if constexpr (DYN_R == 1) rc[i] = rn[i]; }
        const bool more = t + NR < nfull_end;
        if (snapw != nullptr && t1 - t <= (int64_t)kWarmStep * (J - 1) && (t1 - t) % kWarmStep == 0)
            store_own<V>(snapw + ((t1 - t) / kWarmStep) * K * B, q, z);   // snapshot kWarmStep j steps before the chunk's end
        const __amdgpu_buffer_rsrc_t ry = row_rsrc(y + t * B);
        const __amdgpu_buffer_rsrc_t rz = row_rsrc(STASH ? zstash + t * B : y + t * B);
        // steps of this tile below `skip` carry no loss (skip_samples = 50, clipper_pot.py:232): their number, a scalar
        const int64_t below = skip - t;
        const int n_masked = __builtin_amdgcn_readfirstlane((int)(below < 0 ? 0 : (below > NR ? NR : below)));
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (i == kFusedPrefetchAt * NR / 2) {
                // The next tile's loads go out at the TOP of this tile, a whole tile before their first use (vmcnt is one
                // in-order counter for loads and stores: issued later they would sit behind this tile's first stores).
                // With 16-row tiles a tile's loads + stores (48) stay inside the counter's 6 bits.
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    if constexpr (WIN) xw.next(t + NR, nfull_end, xn);
                    else load_x_tile<V, TM, VEC4, NR>(x, q, B, T, t + NR, rowb, xn);
                    if constexpr (DYN_R == 1) load_x_tile<V, TM, VEC4, NR>(r, q, B, T, t + NR, rowb, rn);
                    if constexpr (LOSS != 0) load_rows<V, NR>(target + (t + NR) * B, boff, rowb, gn);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (LOSS == 0) {
                if constexpr (STASH) buf_store_nt(z, rz, boff, i * rowb);
                buf_store_nt(fwd_step<DYN_R, SYM, V, FAST>(c, xc[i], rc[i], z), ry, boff, i * rowb);
                continue;
            }
            buf_store_nt(fused_step<DYN_R, SYM, FAST, V, LOSS>(c, xc[i], rc[i], gc[i], step_loss_scale(i, n_masked, hgs), z, s), ry, boff,
                         i * rowb);
            // The tangent updates do not feed the next step's state, so left alone instruction selection
            // emits the z chain of the whole tile first and keeps every step's partials alive (200 VGPRs,
            // spills).  Pinning the tangent state (a chained, empty asm) before the scheduling barrier keeps
            // each group of steps' work inside the group.
            if (i % kFusedSchedGroup == kFusedSchedGroup - 1) {
                s.template pin<LOSS>();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // Everything but this tile's NR stores has returned -- in particular the next tile's loads, issued NR steps ago.
        // Said here, on the loop's back edge: left to the compiler, the wait sits at the loop header, where it must also
        // hold for the path from the warm-up loop (no stores behind the loads) and so becomes vmcnt(1): every tile would
        // start by draining the previous tile's stores (measured: 12 % of the waves' cycles parked in s_waitcnt).
#ifdef WDF_DBG_TIMES
        { const unsigned long long w0 = __builtin_amdgcn_s_memtime(); wait_vmcnt<NR>(); dbg_wait += __builtin_amdgcn_s_memtime() - w0; }
#else
        wait_vmcnt<NR * (STASH ? 2 : 1)>();
#endif
        // fp32 running sums go to the fp64 totals every kFlushSteps steps (a scalar branch per tile): flushed every tile of
        // 8 steps the 20 conversions + fp64 adds were 2 % of the kernel's VALU time
        if constexpr (LOSS != 0) { if ((t + NR - t0) % kFlushSteps == 0) d.template flush<LOSS>(s); }
    }
    for (int64_t tt = nfull_end; tt < t1; ++tt) {           // tail of the last chunk (T % NR)
        const V xin = load_step<V, TM>(x, q, B, T, tt);
        const V rin = DYN_R == 1 ? load_step<V, TM>(r, q, B, T, tt) : vsplat<V>(1.0f);
        if constexpr (LOSS == 0) {
            if constexpr (STASH) store_own<V>(zstash + tt * B, q, z);
            store_own<V>(y + tt * B, q, fwd_step<DYN_R, SYM, V, FAST>(c, xin, rin, z));
            continue;
        }
        const V tg = load_own<V>(target + tt * B, q);
        store_own<V>(y + tt * B, q, fused_step<DYN_R, SYM, FAST, V, LOSS>(c, xin, rin, tg, tt >= skip ? hgs : 0.0f, z, s));
    }
    d.template flush<LOSS>(s);
    publish_own<V>(zend + k * B, q, z);
    if (snapw != nullptr) store_own<V>(snapw, q, z);
    if (zT && t1 == T) store_own<V>(zT, q, z);
    if constexpr (LOSS != 0) {
        FusedScale fsc{1.0f, 1.0f};
        if constexpr (SYM && FAST == kRootLean && !DYN_R) fsc = FusedScale{-(c.d.two_v * c.d.m_dn), -1.0f / c.V};
        fused_publish_record<V, LOSS, VT<V>::N>(rec, wpart, k, K, q.live, 0, s, d, hgs, fsc);
    }
#ifdef WDF_DBG_TIMES
    dbg_p[4] = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && g_dbg_times) {
        unsigned long long* o = g_dbg_times + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        o[2] = dbg_wait;
        o[4] = dbg_p[1] - dbg_p[0]; o[5] = dbg_p[2] - dbg_p[1]; o[6] = dbg_p[3] - dbg_p[2]; o[7] = dbg_p[4] - dbg_p[3];
    }
#endif
}

// What the step hands back.  MSE: gtheta[4] (+=), sse.  MSE + ESR: sums10 = {S, E, gP[4], gQ[4]} of THIS rank (the chain
// rule applied to both tangent-weighted sums: what ranks all-reduce); and, when gtheta is not null, the step finished as
// a single rank: loss coefficients from S and E (esr_coef_kernel's formulas), gtheta = ga gP + gb gQ,
// loss3 = {mse, esr, mse + esr}, Adam.
struct FusedOut {
    float* gtheta; int accumulate; float* sse_out; AdamTail adam;
    double n_global, eps; float* sums10; float* loss3;
    TpFinishCtx fc;             // filled in by the kernels: the verification's deferred half
};

// MSE + ESR: the tile's eight sums -> its slot of ws (8 doubles per tile); the LAST tile reduces over the tiles in a fixed
// order (lane i takes tiles i, i + 64, ...; then the wave's shuffle tree), applies the chain rule to P and Q and finishes.
__device__ __forceinline__ void esr_tile_partial_and_finish(const double (&v)[8], double* ws, unsigned* gticket, const float* theta,
                                                            float fs, int dyn_r, const FusedOut& out)
{
    const unsigned ntiles = gridDim.x;
    double w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = wave_sum_dpp(v[i]);
    unsigned done = 0;
    if (threadIdx.x == 0) {
        double* o = ws + (int64_t)blockIdx.x * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) __hip_atomic_store(o + i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the partial has landed before the tile count moves
        done = atomicAdd(&gticket[0], 1u);
    }
    done = __builtin_amdgcn_readfirstlane(done);
    if (done != ntiles - 1) return;
    if (threadIdx.x == 0) gticket[0] = 0u;
    // (everything the finish needs requested at once, as tile_partial_and_finish does)
    double t[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (unsigned i = threadIdx.x; i < ntiles; i += 64)
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] += __hip_atomic_load(ws + (int64_t)i * 8 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const AdamFetched af = adam_tail_fetch(out.gtheta != nullptr ? out.adam : AdamTail{nullptr, nullptr, nullptr, nullptr, nullptr, 0.0f, 0.0f, 0.0f, nullptr, nullptr});
    const TpFinishFetched ff = tp_finish_fetch(out.fc);
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = wave_sum_dpp(t[j]);
    // t = {P_L, P_V, P_P, S, Q_L, Q_V, Q_P, E}; every lane forms the chain rule, lanes 0..3 keep their component
    double gP[4], gQ[4];
    grad_chain_rule_d(t[0], t[1], t[2], theta, fs, dyn_r, gP);
    grad_chain_rule_d(t[4], t[5], t[6], theta, fs, dyn_r, gQ);
    const int c = threadIdx.x < 4 ? threadIdx.x : 3;
    const double gPc = c == 0 ? gP[0] : (c == 1 ? gP[1] : (c == 2 ? gP[2] : gP[3]));
    const double gQc = c == 0 ? gQ[0] : (c == 1 ? gQ[1] : (c == 2 ? gQ[2] : gQ[3]));
    if (threadIdx.x == 0) { out.sums10[0] = (float)t[3]; out.sums10[1] = (float)t[7]; }
    if (threadIdx.x < 4) { out.sums10[2 + threadIdx.x] = (float)gPc; out.sums10[6 + threadIdx.x] = (float)gQc; }
    float gi = 0.0f;
    if (out.gtheta != nullptr) {
        const double n = out.n_global, S = t[3], E = t[7] + out.eps;
        const double mse = S / n, esr = sqrt(S / E / n);
        const double ga = 2.0 / n + (esr > 0.0 ? 1.0 / (esr * E * n) : 0.0), gb = -esr / E;
        gi = (out.accumulate ? out.gtheta[c] : 0.0f) + (float)(ga * gPc + gb * gQc);
        if (threadIdx.x < 4) out.gtheta[threadIdx.x] = gi;
        if (threadIdx.x == 0 && out.loss3) { out.loss3[0] = (float)mse; out.loss3[1] = (float)esr; out.loss3[2] = (float)(mse + esr); }
    }
    tp_finish_deferred(out.fc, ff, theta);                   // (reads theta: before the update below)
    if (out.gtheta != nullptr && out.adam.theta != nullptr) adam_tail_apply(out.adam, af, gi);
}

// The tile's K records in time order -> the tile's sums -> (last tile) the step's result.  NSEQ: sequences per lane.
template <int NSEQ, int LOSS>
__device__ __forceinline__ void fused_combine_tile(float* rec, double* wpart, int64_t K, int64_t B, double* ws,
                                                   unsigned* gticket, const float* theta, float fs, int dyn_r, const FusedOut& out,
                                                   double (*sh)[4])
{
    constexpr int NREC = FusedRec<LOSS>::N, NP = FusedRec<LOSS>::NP, NQ = FusedQuads<NSEQ, LOSS>::NQ;
    const int64_t raw = ((int64_t)blockIdx.x * 64 + threadIdx.x) * NSEQ;
    const bool live = raw < B;
    const uint32_t lanes = gridDim.x * 64u, lane = blockIdx.x * 64u + threadIdx.x;
    double dL = 0.0, dV = 0.0, dP = 0.0, dS = 0.0, qL = 0.0, qV = 0.0, qP = 0.0, qE = 0.0;
    double sL[NSEQ], sV[NSEQ], sP[NSEQ];                  // tangent entering the chunk (z0 does not depend on theta)
#pragma unroll
    for (int h = 0; h < NSEQ; ++h) sL[h] = sV[h] = sP[h] = 0.0;
    // The chunk waves' own sums (no dependence on sigma): lane i takes chunk i -- fetched FIRST, in flight under the walk.
    constexpr int NW = NSEQ * NP / 2;                     // 16-byte loads per chunk
    const __amdgpu_buffer_rsrc_t rw = row_rsrc(reinterpret_cast<float*>(wpart + (int64_t)blockIdx.x * K * NSEQ * NP));
    v4u wq[NW];
    const uint32_t kk0 = threadIdx.x < K ? threadIdx.x : (uint32_t)K - 1;      // (lanes past K re-read the last chunk, unused)
#pragma unroll
    for (int i = 0; i < NW; ++i) wq[i] = load_published_quad(rw, kk0 * (uint32_t)(NSEQ * NP * 8) + 16u * i, 0);
    auto add_part = [&](const v4u (&w)[NW]) {
#pragma unroll
        for (int h = 0; h < NSEQ; ++h) {
            double p[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int e = h * NP + i;                 // double e of the chunk's block = dwords 2e, 2e + 1
                p[i] = __hiloint2double((int)w[e / 2][(e % 2) * 2 + 1], (int)w[e / 2][(e % 2) * 2]);
            }
            dL += p[0]; dV += p[1]; dP += p[2]; dS += p[3];
            if constexpr (LOSS == 2) { qL += p[4]; qV += p[5]; qP += p[6]; qE += p[7]; }
        }
    };
    // records of kAhead chunks in flight together (3 x 16 bytes per lane and chunk: 48 loads + the sums' <= 8): the walk is
    // K / kAhead dependent round trips on the step's critical path (the last tile's tail)
    constexpr int kAhead = 16;
    auto step = [&](const v4u (&g)[NQ]) {
        float v[NQ * 4];
#pragma unroll
        for (int i = 0; i < NQ * 4; ++i) v[i] = __uint_as_float(g[i / 4][i % 4]);
#pragma unroll
        for (int h = 0; h < NSEQ; ++h) {
            const float* r = v + h * NREC;
            const double A = r[0], GA = r[4];
            dL += sL[h] * GA;
            dV += sV[h] * GA;
            dP += sP[h] * GA;
            if constexpr (LOSS == 2) {
                const double HA = r[5];
                qL += sL[h] * HA;
                qV += sV[h] * HA;
                qP += sP[h] * HA;
            }
            sL[h] = A * sL[h] + (double)r[1];
            sV[h] = A * sV[h] + (double)r[2];
            sP[h] = A * sP[h] + (double)r[3];
        }
    };
    int64_t k = 0;
    for (; k + kAhead <= K; k += kAhead) {
        v4u g[kAhead][NQ];
#pragma unroll
        for (int j = 0; j < kAhead; ++j) {
            const __amdgpu_buffer_rsrc_t rs = row_rsrc(rec_chunk(rec, k + j));
#pragma unroll
            for (int q = 0; q < NQ; ++q) g[j][q] = load_published_quad(rs, lane * 16u, (uint32_t)q * lanes * 16u);
        }
#pragma unroll
        for (int j = 0; j < kAhead; ++j) step(g[j]);
    }
    for (; k < K; ++k) {
        v4u g[NQ];
        const __amdgpu_buffer_rsrc_t rs = row_rsrc(rec_chunk(rec, k));
#pragma unroll
        for (int q = 0; q < NQ; ++q) g[q] = load_published_quad(rs, lane * 16u, (uint32_t)q * lanes * 16u);
        step(g);
    }
    WDF_DBG_STAMP(3);
    if (!live) { dL = dV = dP = dS = qL = qV = qP = qE = 0.0; }
    if (threadIdx.x < K) add_part(wq);
    for (int64_t kk = threadIdx.x + 64; kk < K; kk += 64) {   // (more than 64 chunks: the rest, one round trip each)
        v4u w2[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) w2[i] = load_published_quad(rw, (uint32_t)kk * (uint32_t)(NSEQ * NP * 8) + 16u * i, 0);
        add_part(w2);
    }
    if constexpr (LOSS == 2) {
        const double v8[8] = {dL, dV, dP, dS, qL, qV, qP, qE};
        esr_tile_partial_and_finish(v8, ws, gticket, theta, fs, dyn_r, out);
    } else {
        tile_partial_and_finish(dL, dV, dP, dS, ws, gticket, theta, fs, dyn_r, out.gtheta, out.accumulate, out.sse_out, out.adam, sh, out.fc);
    }
}

// tickets: the forward's verification area [TpAcc][per-tile tickets][per-tile repair flags];
// gticket: [tiles combined, 0, 0, 0] -- both zero before the first launch and left zero by every step.
// A tile is the 64 * VT<V>::N sequences of one wave.
template <int DYN_R, bool SYM, bool TM, bool VEC4, typename V, int LOSS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WDF_FUSED_WAVES, WDF_FUSED_WAVES))) void clipper_fused_tp_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* theta, float fs, int n_up, int n_down,
    const float* __restrict__ target, float hgs, int64_t skip, float* __restrict__ y, const float* __restrict__ z0,
    float* __restrict__ zT, float* zwarm, float* zend, float* rec, TpStatus* __restrict__ status, TpCtl* ctl, float* snap,
    int J, unsigned* tickets, unsigned* gticket, float tol, int64_t B, int64_t T, int64_t L, int64_t W, int general,
    double* ws, FusedOut out, int64_t skew, double* wpart, int finish_later)
{
    double (*sh)[4];                                        // the in-kernel tail's scratch; where x has a window in LDS, inside it
    if constexpr (fused_x_window<DYN_R, TM, VEC4>()) {
        sh = reinterpret_cast<double (*)[4]>(wave_lds<XWindow<V, true>::kFloats>());
    } else {
        __shared__ double sh_own[64][4];
        sh = sh_own;
    }
#ifdef WDF_DBG_TIMES
    const unsigned long long dbg_t0 = wall_clock64();
    const unsigned long long dbg_m0 = __builtin_amdgcn_s_memtime();
#endif
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { status->fallback_ran = 0; status->pad = 0u; }   // (the repair launch counts)
    const ClipConsts c = load_consts(theta, fs, n_up, n_down);
    const int tier = root_tier<DYN_R, SYM>(c, general);
    bool ran = false;
    if constexpr (SYM && DYN_R != 1) {
        if (tier == kRootLean) {
            clipper_fused_body<DYN_R, SYM, TM, VEC4, kRootLean, V, LOSS>(c, x, r, target, y, nullptr, z0, zT, zwarm, zend, rec, theta, ctl, snap,
                                                                         J, B, T, L, W, hgs, skip, skew, wpart);
            ran = true;
        }
    }
    if (!ran) {
        if (tier != kRootGeneral)
            clipper_fused_body<DYN_R, SYM, TM, VEC4, kRootFast, V, LOSS>(c, x, r, target, y, nullptr, z0, zT, zwarm, zend, rec, theta, ctl, snap,
                                                                         J, B, T, L, W, hgs, skip, skew, wpart);
        else
            clipper_fused_body<DYN_R, SYM, TM, VEC4, kRootGeneral, V, LOSS>(c, x, r, target, y, nullptr, z0, zT, zwarm, zend, rec, theta, ctl,
                                                                            snap, J, B, T, L, W, hgs, skip, skew, wpart);
    }
#ifdef WDF_DBG_TIMES
    if (threadIdx.x == 0 && g_dbg_times) {
        unsigned long long* dbg_o = g_dbg_times + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        dbg_o[0] = dbg_t0; dbg_o[1] = wall_clock64(); dbg_o[3] = __builtin_amdgcn_s_memtime() - dbg_m0;
#ifdef WDF_DBG_HWID                                          // where the wave ran instead of its back-edge wait: HW_ID | XCC_ID << 32
        dbg_o[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                   ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
#endif
    }
#endif
    WDF_DBG_STAMP(0);
    // finish_later (round 6, the default): the chunk waves are done here -- no drain, no tile ticket -- and the launch behind this
    // one (clipper_fused_finish_kernel) verifies, repairs, combines and finishes the step with several waves per tile
    if (finish_later) return;
    if (!tp_tile_last(tickets)) return;
    WDF_DBG_STAMP(1);
    // the tile adds its verification result to the accumulators; the status word and the warm-start steering are left to
    // the wave that finishes the STEP (out.fc), here or in the repair launch
    const bool failed = tp_verify_tile<DYN_R, VT<V>::N, true>(theta, zwarm, zend, status, ctl, J, tickets, tol, B, L, W);
    WDF_DBG_STAMP(2);
    if (failed) return;                                     // left to clipper_fused_repair_kernel
    out.fc = TpFinishCtx{status, ctl, J, tickets, tol, (int64_t)gridDim.y, L, W, skew != 0};
    fused_combine_tile<VT<V>::N, LOSS>(rec, wpart, gridDim.y, B, ws, gticket, theta, fs, DYN_R ? 1 : 0, out, sh);
}

// The time-parallel FORWARD (wdf_clipper_fwd_tp / _warm: y and the state stash for a reverse sweep that arbitrary losses
// drive) on the same body: buffer-descriptor rows, 16-row tiles prefetched a tile ahead, the wait on the back edge.
// Verification and warm-start steering in its last waves (tp_finish), repairs in clipper_tp_repair_kernel (wdf_clipper.h).
template <int DYN_R, bool SYM, bool TM, bool VEC4, bool STASH, typename V>
__global__ __launch_bounds__(64) void clipper_fwd_tp_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ theta,
    float fs, int n_up, int n_down, float* __restrict__ y, float* __restrict__ zstash,
    const float* __restrict__ z0, float* __restrict__ zT, float* zwarm, float* zend,
    TpStatus* __restrict__ status, TpCtl* ctl, float* snap, int J, unsigned* tickets, float tol, int64_t B,
    int64_t Bh, int64_t T, int64_t L, int64_t W, int general, int verify_later)
{
    static_assert(VT<V>::N == 1, "one sequence per lane (the repair kernel and tp_finish index tiles of 64)");
#ifdef WDF_DBG_TIMES
    const unsigned long long dbg_t0 = wall_clock64();
    const unsigned long long dbg_m0 = __builtin_amdgcn_s_memtime();
#endif
    // verify_later (a stateless call: no warm-start block to steer): the boundaries are checked by the launch behind this one
    // (clipper_tp_repair_kernel, verify_all) -- the in-kernel verification is a chain of ~6 dependent device-scope round trips
    // (drain, tile ticket, boundary loads, tile count, last tile, status) that cost the 1024 x 4096 forward 12 us of its 53
    // at 32 chunks and 26 of 62 at 64 (tools/dbg_fwd_times.py); across a kernel boundary they are plain loads.
    if (verify_later && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *status = TpStatus{0, 0.0f, 0, 0u};
    const ClipConsts c = load_consts(theta, fs, n_up, n_down);
    const int tier = root_tier<DYN_R, SYM>(c, general);                 // wave-uniform: every practical diode is LEAN / FAST
    bool ran = false;
    if constexpr (SYM && DYN_R != 1) {
        if (tier == kRootLean) {
            clipper_fused_body<DYN_R, SYM, TM, VEC4, kRootLean, V, 0, STASH>(c, x, r, nullptr, y, zstash, z0, zT, zwarm, zend, nullptr, theta,
                                                                             ctl, snap, J, B, T, L, W, 0.0f, 0);
            ran = true;
        }
    }
    if (!ran) {
        if (tier != kRootGeneral)
            clipper_fused_body<DYN_R, SYM, TM, VEC4, kRootFast, V, 0, STASH>(c, x, r, nullptr, y, zstash, z0, zT, zwarm, zend, nullptr, theta,
                                                                             ctl, snap, J, B, T, L, W, 0.0f, 0);
        else
            clipper_fused_body<DYN_R, SYM, TM, VEC4, kRootGeneral, V, 0, STASH>(c, x, r, nullptr, y, zstash, z0, zT, zwarm, zend, nullptr, theta,
                                                                                ctl, snap, J, B, T, L, W, 0.0f, 0);
    }
#ifdef WDF_DBG_TIMES                                         // (tools/dbg_fwd_times.py: the wave's start, the end of its body, where it ran)
    if (threadIdx.x == 0 && g_dbg_times) {
        unsigned long long* dbg_o = g_dbg_times + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        dbg_o[0] = dbg_t0; dbg_o[1] = wall_clock64(); dbg_o[3] = __builtin_amdgcn_s_memtime() - dbg_m0;
        dbg_o[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                   ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
    }
#endif
    if (!verify_later) tp_finish<DYN_R>(theta, zwarm, zend, status, ctl, J, tickets, tol, B, L, W);
#ifdef WDF_DBG_TIMES
    if (threadIdx.x == 0 && g_dbg_times) g_dbg_times[8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 4] = wall_clock64();   // after the verification
#endif
}

// Re-run of chunk [t0, t1) for 64 sequences (one per lane, index b) from the exact state z: outputs, snapshots, record.
template <int DYN_R, bool SYM, bool TM, int FAST, int LOSS, int NSEQ>
__device__ __forceinline__ void fused_rerun_chunk(const ClipConsts& c_in, const float* __restrict__ x, const float* __restrict__ r,
                                                  const float* __restrict__ target, float* __restrict__ y, float* rec, double* wpart,
                                                  float* __restrict__ snapw, int J, int64_t K, int64_t k, int64_t b, int64_t B,
                                                  bool live, int slot, int64_t T, int64_t t0, int64_t t1, float hgs, int64_t skip,
                                                  float& z)
{
    const ClipConstsSeq<float> c = make_seq_consts<DYN_R, float>(c_in, DYN_R == 2 ? load_one<TM>(r, b, B, T, 0) : 1.0f);
    FusedTan<float> s;
    s.init();
    FusedSums<float> d;
    d.init();
    for (int64_t t = t0; t < t1; t += kBlk) {
        if ((t - t0) % kWarmStep == 0) {
            d.template flush<LOSS>(s);
            if (snapw != nullptr && t1 - t <= (int64_t)kWarmStep * (J - 1) && (t1 - t) % kWarmStep == 0)
                snapw[((t1 - t) / kWarmStep) * K * B + b] = z;
        }
        float xv[kBlk], rv[kBlk], gv[kBlk];
#pragma unroll
        for (int i = 0; i < kBlk; ++i) {
            const int64_t tt = (t + i < t1) ? t + i : t1 - 1;
            xv[i] = load_one<TM>(x, b, B, T, tt);
            rv[i] = DYN_R == 1 ? load_one<TM>(r, b, B, T, tt) : 1.0f;
            gv[i] = target[tt * B + b];
        }
#pragma unroll
        for (int i = 0; i < kBlk; ++i) {
            if (t + i < t1)                                      // wave-uniform
                y[(t + i) * B + b] = fused_step<DYN_R, SYM, FAST, float, LOSS>(c, xv[i], rv[i], gv[i], (t + i >= skip) ? hgs : 0.0f, z, s);
        }
    }
    d.template flush<LOSS>(s);
    if (snapw != nullptr) snapw[b] = z;
    FusedScale fsc{1.0f, 1.0f};
    if constexpr (SYM && FAST == kRootLean && !DYN_R) fsc = FusedScale{-(c.d.two_v * c.d.m_dn), -1.0f / c.V};
    fused_publish_record<float, LOSS, NSEQ>(rec, wpart, k, K, live, slot, s, d, hgs, fsc);
}

// Launched behind every fused step; a block leaves at once unless the step flagged its tile (the common
// case).  For a flagged tile (64 NSEQ sequences; each of its NSEQ interleaved sets of 64 in turn): walk the chunk
// boundaries in time order, re-run every chunk one of whose sequences arrived more than tol off (from the exact
// state, one sequence per lane), then combine the tile.
template <int DYN_R, bool SYM, bool TM, int NSEQ, int LOSS>
__global__ __launch_bounds__(64) void clipper_fused_repair_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* theta, float fs, int n_up, int n_down,
    const float* __restrict__ target, float hgs, int64_t skip, float* __restrict__ y, float* __restrict__ zT,
    const float* zwarm, float* zend, float* rec, int64_t B, int64_t T, int64_t K, int64_t L, float tol,
    TpStatus* __restrict__ status, TpCtl* ctl, float* __restrict__ snap, int J, unsigned* tickets,
    unsigned* gticket, int general, double* ws, FusedOut out, int64_t skew, double* wpart, int64_t W)
{
    __shared__ double sh[64][4];
    unsigned* tile_bad = tickets + 4 + gridDim.x;
    if (tile_bad[blockIdx.x] == 0u) return;
    const int64_t raw = ((int64_t)blockIdx.x * 64 + threadIdx.x) * NSEQ;
    const bool live = raw < B;
    const int64_t b0 = live ? raw : B - NSEQ;
    const ClipConsts c = load_consts(theta, fs, n_up, n_down);
    const int tier = root_tier<DYN_R, SYM>(c, general);
    // The ring slot the step wrote its snapshots to.  The control block is advanced by the wave that FINISHES the step, and with
    // a flagged tile that wave is the last of THIS launch's blocks to combine -- after every block has read this:
    int slot = 0;
    if (ctl != nullptr && snap != nullptr) slot = ((ctl->geom == tp_geom_tag(K, J, skew != 0) ? ctl->head : 0) + 1) % kTpRing;
    int nrep = 0;
#pragma unroll 1
    for (int h = 0; h < NSEQ; ++h) {
        const int64_t b = b0 + h;
        bool fixed_prev = false;
        float ze_fix = 0.0f;
        for (int64_t k = 1; k < K; ++k) {
            const float e = fixed_prev ? ze_fix : load_published(zend + (k - 1) * B + b);
            const float m = fabsf(load_published(zwarm + k * B + b) - e);
            fixed_prev = false;
            if (__builtin_amdgcn_ballot_w64(!(m <= tol)) == 0) continue;
            int64_t t0, t1;
            chunk_span(k, K, L, skew, T, t0, t1);
            float* __restrict__ snapw = (snap != nullptr && k + 1 < K) ? snap + ((int64_t)slot * J * K + k) * B : nullptr;
            float z = e;
            bool ran = false;
            if constexpr (SYM && DYN_R != 1) {
                if (tier == kRootLean) {
                    fused_rerun_chunk<DYN_R, SYM, TM, kRootLean, LOSS, NSEQ>(c, x, r, target, y, rec, wpart, snapw, J, K, k, b, B, live, h, T, t0, t1, hgs, skip, z);
                    ran = true;
                }
            }
            if (!ran) {
                if (tier != kRootGeneral) fused_rerun_chunk<DYN_R, SYM, TM, kRootFast, LOSS, NSEQ>(c, x, r, target, y, rec, wpart, snapw, J, K, k, b, B, live, h, T, t0, t1, hgs, skip, z);
                else fused_rerun_chunk<DYN_R, SYM, TM, kRootGeneral, LOSS, NSEQ>(c, x, r, target, y, rec, wpart, snapw, J, K, k, b, B, live, h, T, t0, t1, hgs, skip, z);
            }
            zend[k * B + b] = z;
            if (zT && t1 == T) zT[b] = z;
            ze_fix = z;
            fixed_prev = true;
            ++nrep;
        }
    }
    if (threadIdx.x == 0) {
        tile_bad[blockIdx.x] = 0u;
        if (nrep) atomicAdd(&status->fallback_ran, nrep);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the re-written records have landed (write-through)
    out.fc = TpFinishCtx{status, ctl, J, tickets, tol, K, L, W, skew != 0};
    fused_combine_tile<NSEQ, LOSS>(rec, wpart, K, B, ws, gticket, theta, fs, DYN_R ? 1 : 0, out, sh);
}

// ---- the step's finish as a launch of its own (round 6) --------------------------------------------------------------------
// What the tile's last chunk wave did alone at the end of clipper_fused_tp_kernel -- drain its stores, take the tile's ticket,
// load K - 1 boundaries, walk K records in time order (K / 16 dependent round trips), publish the tile's sums, take the step's
// ticket, and as the last tile reduce, apply the chain rule, steer the warm start and run Adam: ~12 us of dependent device-scope
// round trips at 32 chunks, 30+ at 128 -- and what the idle repair launch behind every step cost on top (4.7 us), as ONE launch
// of 64 NW-thread workgroups, one per tile, behind the kernel boundary:
//   * wave w of a tile owns the chunks [k0, k1) (at most kFinSeg = 8 at a time in registers): their records, their boundaries
//     and the chunk waves' own sums are requested TOGETHER -- one round trip;
//   * every wave verifies its own boundaries; the tile's verdict goes through LDS.  A tile with a missed boundary is re-run by
//     its wave 0 exactly as clipper_fused_repair_kernel does it (sequentially, from the exact state: outputs, snapshots,
//     records), the other waves wait at the barrier and then load the re-written records;
//   * the walk is split: the tangent entering a chunk is affine in the tangent entering the tile, sigma' = A sigma + c, so a
//     wave composes its chunks into ONE map {P, q[3]} (4 floats per sequence through LDS), every wave then builds the tangent
//     entering ITS first chunk from the maps of the waves before it (<= 7 FMAs per component) and walks its own records with
//     the sums -- the same sums in the same order inside a wave; across waves the partials are added in wave order;
//   * the tile's sums go out through tile_partial_and_finish / esr_tile_partial_and_finish as before: the last tile to arrive
//     reduces, applies the chain rule, publishes the status, steers the warm start and runs Adam.
// Measured on the trial of round 5 (profiles/README.md): the one-pass kernel without its tail runs 0.089-0.090 ms instead of
// 0.107-0.109; the serial tail as a second launch cost 19 us and gave all of it back.
#ifdef WDF_DBG_TIMES      // tools/dbg_fin_times.py: wall-clock stamps of every tile's wave 0 along the finish launch: [8 tiles K + 8 tile + i]
#define WDF_FIN_STAMP(i)                                                                                      \
    do { if (threadIdx.x == 0 && g_dbg_times) g_dbg_times[8 * ((size_t)gridDim.x * (size_t)K) + 8 * (size_t)blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define WDF_FIN_STAMP(i) do {} while (0)
#endif
constexpr int kFinSeg = 8;          // chunk records a finishing wave holds in registers at a time
constexpr int kFinMaxWaves = 8;     // waves per tile (512 threads: the repair path needs up to 161 VGPRs)

template <int DYN_R, bool SYM, bool TM, int NSEQ, int LOSS>
__global__ __launch_bounds__(64 * kFinMaxWaves) void clipper_fused_finish_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* theta, float fs, int n_up, int n_down,
    const float* __restrict__ target, float hgs, int64_t skip, float* __restrict__ y, float* __restrict__ zT,
    const float* zwarm, float* zend, float* rec, int64_t B, int64_t T, int64_t K, int64_t L, float tol,
    TpStatus* __restrict__ status, TpCtl* ctl, float* __restrict__ snap, int J, unsigned* tickets,
    unsigned* gticket, int general, double* ws, FusedOut out, int64_t skew, double* wpart, int64_t W)
{
    constexpr int NREC = FusedRec<LOSS>::N, NP = FusedRec<LOSS>::NP, NQ = FusedQuads<NSEQ, LOSS>::NQ;
    constexpr int NWQ = NSEQ * NP / 2;                     // 16-byte loads of a chunk's own sums
    __shared__ double sh[64][4];
    __shared__ float s_pq[kFinMaxWaves][NSEQ][64][4];      // per wave and sequence: the composed map {P, qL, qV, qP}
    __shared__ double s_sum[kFinMaxWaves][8];
    __shared__ float s_miss[kFinMaxWaves];
    __shared__ int s_bad[kFinMaxWaves];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, NW = blockDim.x >> 6;
    const int64_t seg = (K + NW - 1) / NW;
    const int64_t k0 = (int64_t)w * seg < K ? (int64_t)w * seg : K, k1 = k0 + seg < K ? k0 + seg : K;
    const int64_t raw = ((int64_t)blockIdx.x * 64 + lane) * NSEQ;
    const bool live = raw < B;
    const int64_t b_first = live ? raw : B - NSEQ;
    const uint32_t lanes = gridDim.x * 64u, rl = blockIdx.x * 64u + (uint32_t)lane;
    const bool one_batch = seg <= kFinSeg;
    WDF_FIN_STAMP(0);

    v4u g[kFinSeg][NQ];
    auto load_batch = [&](int64_t kb) {                     // records of chunks kb .. kb + 7 (clamped to the wave's last)
#pragma unroll
        for (int j = 0; j < kFinSeg; ++j) {
            const int64_t k = (kb + j < k1) ? kb + j : (k1 > k0 ? k1 - 1 : 0);
            const __amdgpu_buffer_rsrc_t rs = row_rsrc(rec_chunk(rec, k));
#pragma unroll
            for (int q = 0; q < NQ; ++q) g[j][q] = load_published_quad(rs, rl * 16u, (uint32_t)q * lanes * 16u);
        }
    };
    const __amdgpu_buffer_rsrc_t rw = row_rsrc(reinterpret_cast<float*>(wpart + (int64_t)blockIdx.x * K * NSEQ * NP));
    v4u wq[NWQ];
    auto load_part = [&](int64_t kk) {
        const uint32_t kc = (uint32_t)(kk < k1 ? kk : (k1 > k0 ? k1 - 1 : 0));
#pragma unroll
        for (int i = 0; i < NWQ; ++i) wq[i] = load_published_quad(rw, kc * (uint32_t)(NSEQ * NP * 8) + 16u * i, 0);
    };
    // ---- one flight: the first batch of records, the chunk waves' own sums, the boundaries
    load_batch(k0);
    load_part(k0 + lane);
    float miss = 0.0f;
    int nbad = 0;
    for (int64_t kb = k0; kb < k1; kb += kFinSeg) {
        float zw[kFinSeg][NSEQ], ze[kFinSeg][NSEQ];
#pragma unroll
        for (int j = 0; j < kFinSeg; ++j) {
            int64_t k = (kb + j < k1) ? kb + j : k1 - 1;
            k = k < 1 ? 1 : k;                                  // (chunk 0 has no boundary before it; K >= 2 here)
            load_published_n<NSEQ>(zwarm + k * B + b_first, zw[j]);
            load_published_n<NSEQ>(zend + (k - 1) * B + b_first, ze[j]);
        }
#pragma unroll
        for (int j = 0; j < kFinSeg; ++j)
#pragma unroll
            for (int h = 0; h < NSEQ; ++h) {
                const float m = fabsf(zw[j][h] - ze[j][h]);
                if (kb + j < k1 && kb + j >= 1) {
                    miss = fmaxf(miss, m);
                    nbad += !(m <= tol) ? 1 : 0;                // NaN counts as bad
                }
            }
    }
    {
        const float wmax = wave_max_dpp(miss);
        const int wbad = wave_sum_dpp(nbad);
        if (lane == 0) { s_miss[w] = wmax; s_bad[w] = wbad; }
    }
    WDF_FIN_STAMP(1);
    __syncthreads();
    int tile_bad_pairs = 0;
    float tile_miss = 0.0f;
    for (int i = 0; i < NW; ++i) { tile_bad_pairs += s_bad[i]; tile_miss = fmaxf(tile_miss, s_miss[i]); }
    if (threadIdx.x == 0) {                                  // the tile's share of the verification totals (tp_verify_tile, DEFER)
        TpAcc* acc = reinterpret_cast<TpAcc*>(tickets);
        if (tile_miss > 0.0f) (void)atomicMax(&acc->max_miss_bits, __float_as_int(tile_miss));
        if (tile_bad_pairs) (void)atomicAdd(&acc->n_bad, (unsigned)tile_bad_pairs);
    }
    if (tile_bad_pairs != 0) {
        // ---- the rare path: wave 0 re-runs the chunks that were entered off (clipper_fused_repair_kernel's walk)
        if (w == 0) {
            const ClipConsts c = load_consts(theta, fs, n_up, n_down);
            const int tier = root_tier<DYN_R, SYM>(c, general);
            // the ring slot the step wrote its snapshots to (the control block is advanced by the wave that finishes the step:
            // the last of this launch's tiles through the ticket below -- after every tile has read this)
            int slot = 0;
            if (ctl != nullptr && snap != nullptr) slot = ((ctl->geom == tp_geom_tag(K, J, skew != 0) ? ctl->head : 0) + 1) % kTpRing;
            int nrep = 0;
#pragma unroll 1
            for (int h = 0; h < NSEQ; ++h) {
                const int64_t b = b_first + h;
                bool fixed_prev = false;
                float ze_fix = 0.0f;
                for (int64_t k = 1; k < K; ++k) {
                    const float e = fixed_prev ? ze_fix : load_published(zend + (k - 1) * B + b);
                    const float m = fabsf(load_published(zwarm + k * B + b) - e);
                    fixed_prev = false;
                    if (__builtin_amdgcn_ballot_w64(!(m <= tol)) == 0) continue;
                    int64_t t0, t1;
                    chunk_span(k, K, L, skew, T, t0, t1);
                    float* __restrict__ snapw = (snap != nullptr && k + 1 < K) ? snap + ((int64_t)slot * J * K + k) * B : nullptr;
                    float z = e;
                    bool ran = false;
                    if constexpr (SYM && DYN_R != 1) {
                        if (tier == kRootLean) {
                            fused_rerun_chunk<DYN_R, SYM, TM, kRootLean, LOSS, NSEQ>(c, x, r, target, y, rec, wpart, snapw, J, K, k, b, B, live, h, T, t0, t1, hgs, skip, z);
                            ran = true;
                        }
                    }
                    if (!ran) {
                        if (tier != kRootGeneral) fused_rerun_chunk<DYN_R, SYM, TM, kRootFast, LOSS, NSEQ>(c, x, r, target, y, rec, wpart, snapw, J, K, k, b, B, live, h, T, t0, t1, hgs, skip, z);
                        else fused_rerun_chunk<DYN_R, SYM, TM, kRootGeneral, LOSS, NSEQ>(c, x, r, target, y, rec, wpart, snapw, J, K, k, b, B, live, h, T, t0, t1, hgs, skip, z);
                    }
                    __hip_atomic_store(zend + k * B + b, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (zT && t1 == T) zT[b] = z;
                    ze_fix = z;
                    fixed_prev = true;
                    ++nrep;
                }
            }
            if (lane == 0 && nrep) atomicAdd(&status->fallback_ran, nrep);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the re-written records have landed (write-through)
        }
        __syncthreads();
        load_batch(k0);                                        // (agent-scope loads: past this CU's L1)
        load_part(k0 + lane);
    }
    // ---- phase A: this wave's chunks as one affine map of the tangent, per sequence
    {
        double P[NSEQ], qL[NSEQ], qV[NSEQ], qP[NSEQ];
#pragma unroll
        for (int h = 0; h < NSEQ; ++h) { P[h] = 1.0; qL[h] = qV[h] = qP[h] = 0.0; }
        for (int64_t kb = k0; kb < k1; kb += kFinSeg) {
            if (kb > k0) load_batch(kb);
#pragma unroll
            for (int j = 0; j < kFinSeg; ++j) {
                if (kb + j >= k1) break;                        // wave-uniform
                float v[NQ * 4];
#pragma unroll
                for (int i = 0; i < NQ * 4; ++i) v[i] = __uint_as_float(g[j][i / 4][i % 4]);
#pragma unroll
                for (int h = 0; h < NSEQ; ++h) {
                    const float* rr = v + h * NREC;
                    const double A = rr[0];
                    P[h] *= A;
                    qL[h] = A * qL[h] + (double)rr[1];
                    qV[h] = A * qV[h] + (double)rr[2];
                    qP[h] = A * qP[h] + (double)rr[3];
                }
            }
        }
#pragma unroll
        for (int h = 0; h < NSEQ; ++h) {
            s_pq[w][h][lane][0] = (float)P[h]; s_pq[w][h][lane][1] = (float)qL[h];
            s_pq[w][h][lane][2] = (float)qV[h]; s_pq[w][h][lane][3] = (float)qP[h];
        }
    }
    WDF_FIN_STAMP(2);
    __syncthreads();
    // ---- phase B: the tangent entering this wave's first chunk, then the walk with the sums
    double sL[NSEQ], sV[NSEQ], sP[NSEQ];                   // (the tangent entering the tile is 0: z0 does not depend on theta)
#pragma unroll
    for (int h = 0; h < NSEQ; ++h) {
        sL[h] = sV[h] = sP[h] = 0.0;
        for (int ww = 0; ww < w; ++ww) {
            const double Pw = s_pq[ww][h][lane][0];
            sL[h] = Pw * sL[h] + (double)s_pq[ww][h][lane][1];
            sV[h] = Pw * sV[h] + (double)s_pq[ww][h][lane][2];
            sP[h] = Pw * sP[h] + (double)s_pq[ww][h][lane][3];
        }
    }
    double dL = 0.0, dV = 0.0, dP = 0.0, dS = 0.0, qL2 = 0.0, qV2 = 0.0, qP2 = 0.0, qE = 0.0;
    for (int64_t kb = k0; kb < k1; kb += kFinSeg) {
        if (!one_batch) load_batch(kb);
#pragma unroll
        for (int j = 0; j < kFinSeg; ++j) {
            if (kb + j >= k1) break;                            // wave-uniform
            float v[NQ * 4];
#pragma unroll
            for (int i = 0; i < NQ * 4; ++i) v[i] = __uint_as_float(g[j][i / 4][i % 4]);
#pragma unroll
            for (int h = 0; h < NSEQ; ++h) {
                const float* rr = v + h * NREC;
                const double A = rr[0], GA = rr[4];
                dL += sL[h] * GA;
                dV += sV[h] * GA;
                dP += sP[h] * GA;
                if constexpr (LOSS == 2) {
                    const double HA = rr[5];
                    qL2 += sL[h] * HA;
                    qV2 += sV[h] * HA;
                    qP2 += sP[h] * HA;
                }
                sL[h] = A * sL[h] + (double)rr[1];
                sV[h] = A * sV[h] + (double)rr[2];
                sP[h] = A * sP[h] + (double)rr[3];
            }
        }
    }
    if (!live) { dL = dV = dP = dS = qL2 = qV2 = qP2 = qE = 0.0; }
    auto add_part = [&]() {                                 // the chunk wave's own sums (already masked to its live lanes)
#pragma unroll
        for (int h = 0; h < NSEQ; ++h) {
            double p[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int e = h * NP + i;                     // double e of the chunk's block = dwords 2e, 2e + 1
                p[i] = __hiloint2double((int)wq[e / 2][(e % 2) * 2 + 1], (int)wq[e / 2][(e % 2) * 2]);
            }
            dL += p[0]; dV += p[1]; dP += p[2]; dS += p[3];
            if constexpr (LOSS == 2) { qL2 += p[4]; qV2 += p[5]; qP2 += p[6]; qE += p[7]; }
        }
    };
    if (k0 + lane < k1) add_part();                         // lane i: chunk k0 + i
    for (int64_t kk = k0 + lane + 64; kk < k1; kk += 64) { load_part(kk); add_part(); }   // (more than 64 chunks per wave)
    {
        const double v8[8] = {dL, dV, dP, dS, qL2, qV2, qP2, qE};
#pragma unroll
        for (int i = 0; i < (LOSS == 2 ? 8 : 4); ++i) {
            const double t = wave_sum_dpp(v8[i]);
            if (lane == 0) s_sum[w][i] = t;
        }
    }
    WDF_FIN_STAMP(3);
    __syncthreads();
    if (w != 0) return;
    double tot[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (lane == 0) {
        for (int ww = 0; ww < NW; ++ww)
#pragma unroll
            for (int i = 0; i < (LOSS == 2 ? 8 : 4); ++i) tot[i] += s_sum[ww][i];
    }
    out.fc = TpFinishCtx{status, ctl, J, tickets, tol, K, L, W, skew != 0};
    WDF_FIN_STAMP(4);
    if constexpr (LOSS == 2) esr_tile_partial_and_finish(tot, ws, gticket, theta, fs, DYN_R ? 1 : 0, out);
    else tile_partial_and_finish(tot[0], tot[1], tot[2], tot[3], ws, gticket, theta, fs, DYN_R ? 1 : 0, out.gtheta, out.accumulate, out.sse_out,
                                 out.adam, sh, out.fc);
}

}  // namespace wdf
