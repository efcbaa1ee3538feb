// wdf_vec.h -- one-or-two-wide fp32 value types for the WDF kernels.
//
// The kernels are written once over a value type V: V = float runs one sequence per lane; V = v2f runs TWO independent
// sequences per lane and lets the compiler pack every add / mul / fma of the step into v_pk_* instructions
// (transcendentals, compares and selects stay per component).  On gfx950 a packed fp32 instruction costs TWO plain ones
// (tools/ubench/valu_rate.hip: 1.05 vs 2.0 ns per instruction and SIMD), so packing buys no arithmetic; what it buys is
// half the memory / scalar / addressing instructions per sequence and two independent chains per wave -- worth ~5 % in
// the VALU-bound one-pass step (wdf_clipper_fused.h), nothing in the memory-co-bound kernel pair (which stays at float).
#pragma once

#include <hip/hip_runtime.h>

namespace wdf {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

template <typename V> struct VT;
template <> struct VT<float> {
    using mask = bool;
    static constexpr int N = 1;
};
template <> struct VT<v2f> {
    using mask = v2i;
    static constexpr int N = 2;
};

template <typename V> __device__ __forceinline__ V vsplat(float x);
template <> __device__ __forceinline__ float vsplat<float>(float x) { return x; }
template <> __device__ __forceinline__ v2f vsplat<v2f>(float x) { return v2f{x, x}; }

__device__ __forceinline__ float vget(float v, int) { return v; }
__device__ __forceinline__ float vget(v2f v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ void vset(float& v, int, float x) { v = x; }
__device__ __forceinline__ void vset(v2f& v, int i, float x) { if (i == 0) v.x = x; else v.y = x; }

__device__ __forceinline__ float vfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f vfma(v2f a, float b, v2f c) { return __builtin_elementwise_fma(a, v2f{b, b}, c); }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, float c) { return __builtin_elementwise_fma(a, b, v2f{c, c}); }
__device__ __forceinline__ v2f vfma(v2f a, float b, float c) { return __builtin_elementwise_fma(a, v2f{b, b}, v2f{c, c}); }
__device__ __forceinline__ v2f vfma(float a, v2f b, v2f c) { return __builtin_elementwise_fma(v2f{a, a}, b, c); }
__device__ __forceinline__ v2f vfma(float a, v2f b, float c) { return __builtin_elementwise_fma(v2f{a, a}, b, v2f{c, c}); }

__device__ __forceinline__ float vabs(float a) { return fabsf(a); }
__device__ __forceinline__ v2f vabs(v2f a) { return v2f{fabsf(a.x), fabsf(a.y)}; }
// min / max against a constant, written as compare+select (no NaN canonicalisation ops)
__device__ __forceinline__ float vmin_c(float a, float c) { return a < c ? a : c; }
__device__ __forceinline__ v2f vmin_c(v2f a, float c) { return v2f{a.x < c ? a.x : c, a.y < c ? a.y : c}; }
__device__ __forceinline__ float vmax_c(float a, float c) { return a > c ? a : c; }
__device__ __forceinline__ v2f vmax_c(v2f a, float c) { return v2f{a.x > c ? a.x : c, a.y > c ? a.y : c}; }

__device__ __forceinline__ float vexp2(float a) { return __builtin_amdgcn_exp2f(a); }
__device__ __forceinline__ v2f vexp2(v2f a) { return v2f{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
__device__ __forceinline__ float vlog2(float a) { return __builtin_amdgcn_logf(a); }
__device__ __forceinline__ v2f vlog2(v2f a) { return v2f{__builtin_amdgcn_logf(a.x), __builtin_amdgcn_logf(a.y)}; }
__device__ __forceinline__ float vrcp(float a) { return __builtin_amdgcn_rcpf(a); }
__device__ __forceinline__ v2f vrcp(v2f a) { return v2f{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }

// compares against a constant -> mask
__device__ __forceinline__ bool vle_c(float a, float c) { return a <= c; }
__device__ __forceinline__ v2i vle_c(v2f a, float c) { return a <= v2f{c, c}; }
__device__ __forceinline__ bool vgt_c(float a, float c) { return a > c; }
__device__ __forceinline__ v2i vgt_c(v2f a, float c) { return a > v2f{c, c}; }
__device__ __forceinline__ bool vge_c(float a, float c) { return a >= c; }
__device__ __forceinline__ v2i vge_c(v2f a, float c) { return a >= v2f{c, c}; }
__device__ __forceinline__ bool vlt_c(float a, float c) { return a < c; }
__device__ __forceinline__ v2i vlt_c(v2f a, float c) { return a < v2f{c, c}; }

__device__ __forceinline__ bool mand(bool a, bool b) { return a && b; }
__device__ __forceinline__ v2i mand(v2i a, v2i b) { return a & b; }
__device__ __forceinline__ bool mor(bool a, bool b) { return a || b; }
__device__ __forceinline__ v2i mor(v2i a, v2i b) { return a | b; }
__device__ __forceinline__ bool many(bool a) { return a; }
__device__ __forceinline__ bool many(v2i a) { return (a.x | a.y) != 0; }

__device__ __forceinline__ float vsel(bool m, float a, float b) { return m ? a : b; }
__device__ __forceinline__ v2f vsel(v2i m, v2f a, v2f b) { return v2f{m.x ? a.x : b.x, m.y ? a.y : b.y}; }
__device__ __forceinline__ v2f vsel(v2i m, v2f a, float b) { return v2f{m.x ? a.x : b, m.y ? a.y : b}; }
__device__ __forceinline__ v2f vsel(v2i m, float a, float b) { return v2f{m.x ? a : b, m.y ? a : b}; }

// x where a != 0, else 0
__device__ __forceinline__ float vsel_nz(float a, float x) { return a != 0.0f ? x : 0.0f; }
__device__ __forceinline__ v2f vsel_nz(v2f a, v2f x) { return v2f{a.x != 0.0f ? x.x : 0.0f, a.y != 0.0f ? x.y : 0.0f}; }

// sign(a) in {-1, 0, +1}  (np.sign)
__device__ __forceinline__ float vsign(float a) { return (a > 0.0f) ? 1.0f : ((a < 0.0f) ? -1.0f : 0.0f); }
__device__ __forceinline__ v2f vsign(v2f a) { return v2f{vsign(a.x), vsign(a.y)}; }
// |m| with the sign of s: one v_bfi_b32
__device__ __forceinline__ float vcopysign(float m, float s) { return __builtin_copysignf(m, s); }
__device__ __forceinline__ v2f vcopysign(v2f m, v2f s) { return v2f{__builtin_copysignf(m.x, s.x), __builtin_copysignf(m.y, s.y)}; }

// keep a value as computed (opaque to the optimiser at this point)
__device__ __forceinline__ void vpin(float& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void vpin(v2f& a) { asm volatile("" : "+v"(a)); }

}  // namespace wdf
