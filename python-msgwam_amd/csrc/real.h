// Scalar type of the per-ray state: float64 (the reference's, bit-comparable with numpy) or float32
// (BASELINE config 5: throughput mode, half the bytes per ray).  Everything per ray is templated on
// `T`; the flux rows, their reduction over workgroups / ranks and the mean-flow column are float64
// in both modes.  A lane holds 2 rays in both modes (16-B / 8-B accesses per array), so a 256-thread
// workgroup advances a tile of 512 rays.  (Round 2 first ran float32 with 4 rays per lane -- 16-B accesses, 1024-ray
// tiles: 1.25e6 rays per GPU are then 2.47 tiles per workgroup, i.e. 407 workgroups of three tiles on 512 slots and
// wavefronts that span 256 neighbouring rays; with 2 rays per lane the split is 489 x 5 tiles, a wavefront's rays
// disperse over half as many levels, and FOUR fully resident tiles fit the registers of every float32 variant:
// config 5 54.7 -> 49.5 us per step.  The price: mostly-streamed runs lose ~6 % to the 8-B accesses, 5e6 rays
// 2.97 -> 2.78e10 ray-steps/s.  -DMSGW_F32_RPT=4 brings the old layout back.)
#pragma once
#include <hip/hip_runtime.h>
#include <limits>
#include <type_traits>

namespace msgw {

constexpr int BLOCK = 256;           // 4 wavefronts
constexpr int WAVES = BLOCK / 64;

template <typename T> struct Real;
template <> struct Real<double> {
    static constexpr int RPT = 2;                 // rays per lane -> 16-B global accesses
    static constexpr int TILE = BLOCK * RPT;      // rays per workgroup iteration
    typedef double2 pair_t;
    typedef double4 quad_t;
    static __device__ __forceinline__ pair_t pair(double x, double y) { return make_double2(x, y); }
    static __device__ __forceinline__ quad_t quad(double x, double y, double z, double w) { return make_double4(x, y, z, w); }
};
#ifndef MSGW_F32_RPT
#define MSGW_F32_RPT 2                            // see the header comment
#endif
template <> struct Real<float> {
    static constexpr int RPT = MSGW_F32_RPT;
    static constexpr int TILE = BLOCK * RPT;
    typedef float2 pair_t;
    typedef float4 quad_t;
    static __device__ __forceinline__ pair_t pair(float x, float y) { return make_float2(x, y); }
    static __device__ __forceinline__ quad_t quad(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
};

// Every per-ray array is allocated and initialised up to a whole number of tiles (the host pads
// with inert rays), so all accesses are UNCONDITIONAL 16-B vector accesses: a conditional
// load makes hipcc branch around it and wait vmcnt(0) per load, which serialises the nine
// streams of a tile into nine memory round trips (measured: 27 us vs 16 us per launch).
// Addressing: uniform base pointer (SGPR pair) + ONE 32-bit byte offset shared by all SoA
// arrays of a tile (global_load ... v_off, s[base] form) instead of a 64-bit VGPR address
// per array: saves ~26 VGPRs and the address arithmetic.  Limits a context to 2^29 rays.
__device__ __forceinline__ void loadv(const double *p, unsigned int off, double (&out)[2])
{
    const double2 t = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(p) + off);   // 16 B per lane
    out[0] = t.x; out[1] = t.y;
}
__device__ __forceinline__ void storev(double *p, unsigned int off, const double (&v)[2])
{
    *reinterpret_cast<double2 *>(reinterpret_cast<char *>(p) + off) = make_double2(v[0], v[1]);
}
__device__ __forceinline__ void loadv(const float *p, unsigned int off, float (&out)[4])
{
    const float4 t = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(p) + off);
    out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
}
__device__ __forceinline__ void storev(float *p, unsigned int off, const float (&v)[4])
{
    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(p) + off) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void loadv(const float *p, unsigned int off, float (&out)[2])
{
    const float2 t = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(p) + off);   // 8 B per lane
    out[0] = t.x; out[1] = t.y;
}
__device__ __forceinline__ void storev(float *p, unsigned int off, const float (&v)[2])
{
    *reinterpret_cast<float2 *>(reinterpret_cast<char *>(p) + off) = make_float2(v[0], v[1]);
}

// one ray per lane (the fixed-background kernel at small ray counts, see k_ray_step_fixed): 8-B / 4-B accesses
template <typename T> __device__ __forceinline__ void loadv(const T *p, unsigned int off, T (&out)[1])
{
    out[0] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(p) + off);
}
template <typename T> __device__ __forceinline__ void storev(T *p, unsigned int off, const T (&v)[1])
{
    *reinterpret_cast<T *>(reinterpret_cast<char *>(p) + off) = v[0];
}

// max / min of two values in ONE instruction (v_max / v_min; a `(a > b) ? a : b` costs a compare, its hazard slot and two
// v_cndmask per float64).  Used where the operands cannot be NaN (the overlap of a contributing ray volume with a grid
// cell, lib/libprop.py:157-158: rays with a NaN bound are out of domain and never reach the level loop); for such
// operands it equals the select form up to the sign of a zero, which the caller's |zmax - zmin| removes.
__device__ __forceinline__ double max1(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double min1(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max1(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float min1(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <typename T> __device__ __forceinline__ T real_inf() { return std::numeric_limits<T>::infinity(); }

// Division and square root of the per-ray arithmetic.  float64: correctly rounded (bit-comparable with numpy) on the
// operands of this physics -- the LEAN forms below are hipcc's own IEEE expansions WITHOUT the parts that only serve
// extreme exponents: the two v_div_scale / the scaling half of v_div_fmas of a division (12 -> 9 VALU instructions),
// the pre-scaling of arguments below 2^-767, its select and the two v_ldexp of a square root (19 -> 13).  Same
// operations in the same order otherwise, so wherever hipcc's sequence would not have scaled the bits are the same:
//   x / y        finite operands with 2^-968 <= |x|, 2^-1022 <= |y| <= 2^1021, 2^-1022 <= |x / y| < 2^1024 (beyond that the
//                Newton residual, the reciprocal or the quotient leave the normal range); zeros, infinities and NaNs in either operand are put
//                right by v_div_fixup as in hipcc's expansion;
//   sqrt(x)      2^-767 <= x < inf; +-0, +inf, negative arguments and NaN by one class test.
// The ray equations divide and take roots of frequencies, wavenumbers, densities and their squares (1e-30 .. 1e30);
// tests/test_gpu_parity.py::test_device_sqrt_and_division_are_correctly_rounded holds both to numpy bit for bit over those
// domains and the special values.  Why: the ray kernels are bound by VALU issue -- timing-only builds
// with a 1-instruction square root / a 2-instruction division put the three roots of a fixed-background RK3 step at 13 %
// and its nine divisions at 25 % of config 2's time (4.5 % / 7.5 % of config 3's).  -DMSGW_LEAN_ARITH=0: x / y and sqrt(x)
// as hipcc expands them.
// float32 is the throughput mode without a bit-level pin: v_rcp_f32 / v_sqrt_f32 (1 ulp each; x * rcp(y) <= 2 ulp)
// instead of the ~12-instruction IEEE sequences -- the hot loop has ~9 divisions and 2 square roots per ray-stage
// (measured at config 5: 84 -> 63 us per step).  Operands are far from the denormal range (wavenumbers 1e-6..1e-2,
// frequencies ~1e-3, densities 1e-15..1e20).
#ifndef MSGW_LEAN_ARITH
#define MSGW_LEAN_ARITH 1
#endif
__device__ __forceinline__ double div_(double x, double y)
{
#if MSGW_LEAN_ARITH
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = x * r;
    e = __builtin_fma(-y, q, x);                             // exact remainder
    return __builtin_amdgcn_div_fixup(__builtin_fma(e, r, q), y, x);
#else
    return x / y;
#endif
}
__device__ __forceinline__ float div_(float x, float y) { return x * __builtin_amdgcn_rcpf(y); }
__device__ __forceinline__ double sqrt_(double x)
{
#if MSGW_LEAN_ARITH
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return __builtin_amdgcn_class(x, 0x260) ? x : g;          // +-0 and +inf are their own roots (0x260: -0 | +0 | +inf)
#else
    return sqrt(x);
#endif
}
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
// v_div_fixup: the quotient q of x / d with the special cases of IEEE division put right (see div_const)
__device__ __forceinline__ double div_fixup_(double q, double d, double x) { return __builtin_amdgcn_div_fixup(q, d, x); }
__device__ __forceinline__ float div_fixup_(float q, float d, float x) { return __builtin_amdgcn_div_fixupf(q, d, x); }

}   // namespace msgw
