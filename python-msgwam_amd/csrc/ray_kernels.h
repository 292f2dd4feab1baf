// Device code of the ray-propagation path, written for CDNA4 / gfx950 only
// (wave64, DPP cross-lane reduction, LDS-staged column, 16-B/lane coalesced
// SoA loads).  Compiled with -ffp-contract=off: every expression below is in
// the reference's (numpy's) evaluation order so that, with T = double, per-ray
// results are bit-comparable with lib/libprop.py.  T = float is the throughput
// mode of BASELINE config 5 (same formulas in float32; flux rows, their reduction
// and the mean-flow column stay float64).  Citations are reference file:line.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdint>
#include "column_math.h"
#include "real.h"

#ifndef MSGW_PREFKL
#define MSGW_PREFKL 1       // resident tiles read kk, ll one tile ahead (tile_body.inc)
#endif

namespace msgw {

constexpr int SPAN_MAX = 8;          // widest per-wave level span on the shuffle path
constexpr int COL_BLOCK = 1024;      // the column kernel is one workgroup
constexpr int FUSE_ROWS = 16;        // most dense flux rows the fused prologue adds up (= reduce-1 groups)

// All per-ray arrays live in ONE slab, array k at slab + k * pitch: a kernel keeps one base pointer and the pitch in
// SGPRs instead of one 64-bit pointer per array (19 pairs: with them the persistent kernel spilled SGPRs in its
// hottest loop).  The k * pitch term is made opaque at every use so that the compiler re-forms it with a few SALU
// instructions instead of hoisting 19 loop-invariant pointers and spilling them again.
enum RayArray { A_DENS, A_RR, A_MM, A_DRR, A_KK, A_LL, A_DMM, A_VOL, A_FRAY, A_PVF, A_Q_RR, A_Q_MM, A_Q_DENS, A_RR0, A_MM0,
                A_SRC_DENS, A_SRC_RR, A_SRC_MM, A_CG, A_COUNT };
template <typename T>
struct RayPtrsT {
    char *slab;
    unsigned int pitch256;                          // bytes between two arrays / 256
    template <int K> __device__ __forceinline__ T *at() const
    {
        unsigned int p = pitch256;
        asm volatile("" : "+s"(p));
        const unsigned int t = (unsigned int)K * p;
        return reinterpret_cast<T *>(slab + (((unsigned long long)t) << 8));
    }
    __device__ __forceinline__ T *dens() const { return at<A_DENS>(); }
    __device__ __forceinline__ T *rr() const { return at<A_RR>(); }
    __device__ __forceinline__ T *mm() const { return at<A_MM>(); }
    __device__ __forceinline__ const T *drr() const { return at<A_DRR>(); }
    __device__ __forceinline__ const T *kk() const { return at<A_KK>(); }
    __device__ __forceinline__ const T *ll() const { return at<A_LL>(); }
    __device__ __forceinline__ const T *dmm() const { return at<A_DMM>(); }
    __device__ __forceinline__ const T *vol() const { return at<A_VOL>(); }
    __device__ __forceinline__ const T *fray() const { return at<A_FRAY>(); }
    __device__ __forceinline__ const T *pvf() const { return at<A_PVF>(); }
    __device__ __forceinline__ T *q_rr() const { return at<A_Q_RR>(); }
    __device__ __forceinline__ T *q_mm() const { return at<A_Q_MM>(); }
    __device__ __forceinline__ T *q_dens() const { return at<A_Q_DENS>(); }
    __device__ __forceinline__ T *rr0() const { return at<A_RR0>(); }
    __device__ __forceinline__ T *mm0() const { return at<A_MM0>(); }
    __device__ __forceinline__ const T *src_dens() const { return at<A_SRC_DENS>(); }
    __device__ __forceinline__ const T *src_rr() const { return at<A_SRC_RR>(); }
    __device__ __forceinline__ const T *src_mm() const { return at<A_SRC_MM>(); }
    __device__ __forceinline__ T *cg() const { return at<A_CG>(); }
};
typedef RayPtrsT<double> RayPtrs;

struct ColPtrs {                                   // the column is float64 in both modes
    const double *xg;                              // grid[1:-1], ni = ng-2
    const double *dudz, *dvdz;                     // first differences on xg (:352-353)
    const double *slu, *slv;                       // np.interp slopes, ni-1
    const double *grids, *rhobar, *slrho;          // nc = ng-1 (slopes nc-1)
};

template <typename T>
struct StageArgsT {
    long long n;
    int ng;
    int tiles_per_block;  // ceil(rays_per_block / TILE)
    int fixed_steps;      // k_ray_step_fixed: RK3 steps per launch (rays are independent: state stays in registers)
    int relaunch;         // EXTENSION (MSGW_RELAUNCH): recycle rays that left the column or broke, after stage 2
    T z_bot, z_top, relaunch_frac;
    long long rays_per_block;   // contiguous rays owned by a workgroup (multiple of 16: 128-B aligned
                          // starts); chosen so that the workgroups divide evenly over the CUs
    T dt;
    double dtc;           // dt of the mean-flow update (float64 in both modes)
    T bvf2;               // bvf**2
    T f_uni;              // per-ray f when it is the same for every ray
    T f0sq;               // (2*Omega*sin(phi0))**2, config latitude (:589, :597)
    int same_f;           // f_uni*f_uni == f0sq  -> omh == om
    T sat_c;              // kappa**2 * .5
    T sat_rr_div;         // 1.0 (driver quirk raytracer.py:184) or dt
    T xg0, inv_dzg;       // index guess on xg
    T gs0, inv_dzs;       // index guess on grids
    T xg_last, gs_last;   // last abscissa of each table (first = xg0 / gs0)
    T dzs;                // grids[1]-grids[0]  (:123 with G = grids)
    int mk_ok;            // RN(1/dzs) usable for the exact constant division (see div_const)
    RayPtrsT<T> r;
    ColPtrs c;
    double *partial;      // [blocks][2][ng-2] per-workgroup flux rows
    int *ranges;          // [blocks][2] touched level range of each row
    // In-kernel first-level flux reduction (GROUPRED kernels): workgroups [g*grp_size, ...) publish
    // dense rows to grp_part (write-through), take a ticket on grp_cnt[g]; the last arriver adds
    // the group's rows in row order into grp_rows[g].
    int grp_size;             // workgroups per group
    int row_stride;           // doubles per published row (multiple of 16: rows never share a 128-B line)
    double *grp_part;         // [blocks][row_stride]
    double *grp_rows;         // [ngroups][2*(ng-2)] dense group sums
    unsigned int *grp_cnt;    // [ngroups] arrival tickets, [63] completed groups (zero between launches)
    int ngroups;              // number of groups
    double *flux_out;         // [2*(ng-2)] final flux row of this launch (sum of the group sums)
    // Pending mean-flow update of the PREVIOUS RK stage, applied in this kernel's prologue by
    // every workgroup in LDS (workgroup 0 also publishes the new column to cout):
    int col_pending;      // 0: shear tables are read from c.dudz..; 1: apply the update below first
    int col_stage;        // RK stage (0, 1, 2) of that update
    const double *col_rows;   // [2*(ng-2)] the final flux row of that stage
    const double *pg;     // pressure gradient [2][ng-1]
    double f0, dzg;       // config Coriolis parameter (:535), grid[1]-grid[0] (:349, :662)
    ColIn cin;            // column before the update (never written by this launch)
    ColOut cout;          // column after the update (written by workgroup 0 only)
#ifdef MSGW_STAMP
    unsigned long long *stamps;   // diagnostic build only: [blocks][8] wall-clock stamps
#endif
};
typedef StageArgsT<double> StageArgs;
#ifdef MSGW_STAMP
#define MSGW_STAMP_AT(k) do { if (threadIdx.x == 0 && a.stamps) a.stamps[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define MSGW_STAMP_AT(k) do { } while (0)
#endif

// ------------------------------------------------------------------ cross-lane
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
    // every lane of these patterns has a valid source, so the "old" value is never used: with bound_ctrl and an
    // undefined old the compiler emits the bare v_mov_b32_dpp (no copy in front) and can fold it into a 32-bit VALU op
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(dpp_i32<CTRL>(hi), dpp_i32<CTRL>(lo));
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(dpp_i32<CTRL>(__float_as_int(v)));
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_f32(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// Sum over the 64 lanes in a FIXED order (xor 1, xor 2, half-mirror, mirror
// inside each 16-lane row, then the four rows): bit-reproducible, no LDS.
// Must be called with all 64 lanes active.
__device__ __forceinline__ double wave_sum(double v)
{
    v = v + dpp_f64<0xB1>(v);     // quad_perm [1,0,3,2]
    v = v + dpp_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    v = v + dpp_f64<0x141>(v);    // row_half_mirror
    v = v + dpp_f64<0x140>(v);    // row_mirror
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16);
    const double r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_sum(float v)
{
    v = v + dpp_f32<0xB1>(v);
    v = v + dpp_f32<0x4E>(v);
    v = v + dpp_f32<0x141>(v);
    v = v + dpp_f32<0x140>(v);
    const float r0 = readlane_f32(v, 0), r1 = readlane_f32(v, 16);
    const float r2 = readlane_f32(v, 32), r3 = readlane_f32(v, 48);
    return (r0 + r1) + (r2 + r3);
}
// Two sums at once (the two pseudo-momentum-flux components): v_permlane32_swap puts the
// half-sums of `a` in lanes 0-31 and those of `b` in lanes 32-63, four DPP steps reduce inside
// each 16-lane row, v_permlane16_swap adds the two rows of each half.  32 instructions for both
// sums instead of ~40 each; fixed order, bit-reproducible.  All 64 lanes must be active.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wave_sum2(double a, double b, double &ta, double &tb)
{
    const u32x2_t lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const u32x2_t hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    // {a.lanes0-31, b.lanes0-31} + {a.lanes32-63, b.lanes32-63}
    double v = __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
    v = v + dpp_f64<0xB1>(v);     // quad_perm [1,0,3,2]
    v = v + dpp_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    v = v + dpp_f64<0x141>(v);    // row_half_mirror
    v = v + dpp_f64<0x140>(v);    // row_mirror: every lane of a row holds the row sum
    const u32x2_t rlo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const u32x2_t rhi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    // {r0, r0, r2, r2} + {r1, r1, r3, r3}: rows 0,1 belong to `a`, rows 2,3 to `b`
    v = __hiloint2double((int)rhi.x, (int)rlo.x) + __hiloint2double((int)rhi.y, (int)rlo.y);
    ta = readlane_f64(v, 0);
    tb = readlane_f64(v, 32);
}
// The deposit's sums over a wavefront (NP = 2): only the first four steps of that butterfly -- every group of 8 lanes then
// holds its own sum of a (lanes 0-31) or b (lanes 32-63) -- and ONE lane per group adds it to the wave's private LDS row
// with ds_add_f64 (4 lane-operations per component and level).  The ray kernels are bound by VALU issue (DESIGN.md 4 K1f,
// 6): the remaining two butterfly steps and the bookkeeping of the lane-distributed level window (cmp, two selects, add)
// were 10 of 22 VALU instructions per level; the LDS atomics issue on the LDS port.  Measured at config 3 with 0 / 1 / 2 /
// 3 DPP steps before the atomic (32 / 16 / 8 / 4 lane-operations per component): 34.4 / 31.1 / 30.5-30.8 / 30.4-30.5 us
// per step against 31.6 for the full butterfly.  The row belongs to ONE wave and a wave's LDS operations are performed in
// program order, lanes in lane order: the sums are reproducible bit for bit from run to run (the repeated-run tests hold).
__device__ __forceinline__ void group_sum2_to_lds(double a, double b, double *row_c, int ncp, int lane)
{
    const u32x2_t lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const u32x2_t hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    double v = __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
    v = v + dpp_f64<0xB1>(v);     // quad_perm [1,0,3,2]
    v = v + dpp_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    v = v + dpp_f64<0x141>(v);    // row_half_mirror: every lane of a group of 8 holds the group's sum
    if ((lane & 7) == 0)
        __hip_atomic_fetch_add(row_c + (lane >> 5) * ncp, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void group_sum2_to_lds(float a, float b, double *row_c, int ncp, int lane)
{
    const u32x2_t s = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    float v = __uint_as_float(s.x) + __uint_as_float(s.y);
    v = v + dpp_f32<0xB1>(v);
    v = v + dpp_f32<0x4E>(v);
    v = v + dpp_f32<0x141>(v);
    if ((lane & 7) == 0)
        __hip_atomic_fetch_add(row_c + (lane >> 5) * ncp, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// float32 form of the same butterfly: one dword per value, and the DPP steps fold into v_add_f32_dpp
// (about 10 instructions for both sums).  Same fixed order.
__device__ __forceinline__ void wave_sum2(float a, float b, float &ta, float &tb)
{
    const u32x2_t s = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    float v = __uint_as_float(s.x) + __uint_as_float(s.y);
    v = v + dpp_f32<0xB1>(v);
    v = v + dpp_f32<0x4E>(v);
    v = v + dpp_f32<0x141>(v);
    v = v + dpp_f32<0x140>(v);
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
    ta = readlane_f32(v, 0);
    tb = readlane_f32(v, 32);
}
__device__ __forceinline__ int wave_min(int v)
{
    v = min(v, dpp_i32<0xB1>(v));
    v = min(v, dpp_i32<0x4E>(v));
    v = min(v, dpp_i32<0x141>(v));
    v = min(v, dpp_i32<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max(int v)
{
    v = max(v, dpp_i32<0xB1>(v));
    v = max(v, dpp_i32<0x4E>(v));
    v = max(v, dpp_i32<0x141>(v));
    v = max(v, dpp_i32<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// ------------------------------------------------------------------ np.interp on an LDS column
// numpy's arr_interp (the arithmetic behind lib/libprop.py:355-356, :595):
// end values outside [xp[0], xp[n-1]], fp[j] exactly when x == xp[j], otherwise
// slope[j]*(x - xp[j]) + fp[j] with slope[j] = (fp[j+1]-fp[j])/(xp[j+1]-xp[j])
// (precomputed by the column kernel with the same expression).
// locate: bracket index j with xp[j] <= x < xp[j+1] from a uniform-grid guess plus an exact
// fix-up against the real abscissae (so any monotone grid works; the fix-up loops run 0 times
// away from cell boundaries).  `flat` != 0 means "take fp[j] as is" (outside the table, on its
// last point, or exactly on a point).  NaN falls through to slope*(NaN - xj) + fp = NaN.
template <typename T> struct Bracket { int j; int flat; T xj; };
template <typename T>
__device__ __forceinline__ Bracket<T> interp_locate(T x, const T *xp, int n, T x_first, T x_last, T x0, T inv_dx)
{
    Bracket<T> b;
    int j = (int)((x - x0) * inv_dx);
    j = min(max(j, 0), n - 2);
    T xj = xp[j], xj1 = xp[j + 1];
    // On a uniform grid the guess is the bracket except within a rounding error of a grid point, so the fix-up loops
    // sit behind ONE test that practically no lane passes (as plain loops their first trips -- an LDS read, a wait
    // and exec-mask bookkeeping each -- ran for every ray: ~30 instructions per ray and RK stage).
    const bool low = (x < xj) & (j > 0), high = (x >= xj1) & (j < n - 2);
    if (__builtin_expect(low | high, 0)) {
        while (x < xj && j > 0) { --j; xj1 = xj; xj = xp[j]; }
        while (x >= xj1 && j < n - 2) { ++j; xj = xj1; xj1 = xp[j + 1]; }
    }
    const bool top = x >= x_last;                   // beyond or on the last point -> fp[n-1]
    b.j = top ? n - 1 : j;
    b.xj = xj;
    b.flat = (int)(top | (x < x_first) | (x == xj));   // (bitwise: no short-circuit branches)
    return b;
}
// The same look-up in two steps, so that a caller can issue the LDS reads of ALL its rays before waiting for any of
// them: `interp_guess_fetch` (no branch) reads the guessed bracket's abscissae and, speculatively, the table entry it
// points to; `interp_settle` makes the bracket exact (the rare fix-up re-reads the entry).  Same result as
// interp_locate followed by tab[b.j]; one LDS round trip per tile instead of two dependent ones per ray.
template <typename T, typename Q>
__device__ __forceinline__ void interp_guess_fetch(T x, const T *xp, const Q *tab, int n, T x_last, T x0, T inv_dx,
                                                   int &j, T &xj, T &xj1, Q &q)
{
    j = (int)((x - x0) * inv_dx);
    j = min(max(j, 0), n - 2);
    xj = xp[j]; xj1 = xp[j + 1];
    q = tab[(x >= x_last) ? n - 1 : j];
}
template <typename T, typename Q>
__device__ __forceinline__ Bracket<T> interp_settle(T x, const T *xp, const Q *tab, int n, T x_first, T x_last,
                                                    int j, T xj, T xj1, Q &q)
{
    Bracket<T> b;
    const bool top = x >= x_last;                   // beyond or on the last point -> fp[n-1]
    const bool low = (x < xj) & (j > 0), high = (x >= xj1) & (j < n - 2);
    if (__builtin_expect(low | high, 0)) {
        while (x < xj && j > 0) { --j; xj1 = xj; xj = xp[j]; }
        while (x >= xj1 && j < n - 2) { ++j; xj = xj1; xj1 = xp[j + 1]; }
        q = tab[top ? n - 1 : j];
    }
    b.j = top ? n - 1 : j;
    b.xj = xj;
    b.flat = (int)(top | (x < x_first) | (x == xj));
    return b;
}
template <typename T>
__device__ __forceinline__ T interp_eval(T x, const Bracket<T> &b, T fpj, T slj)
{
    const T lin = slj * (x - b.xj) + fpj;
    return b.flat ? fpj : lin;
}

// np.interp(x, grids, f) on arrays in GLOBAL memory (the standalone entry points; the stage kernels stage their tables
// in LDS): numpy's own slope expression (f[j+1]-f[j])/(x[j+1]-x[j]), end values beyond the table
__device__ __forceinline__ double interp_global(double x, const double *xp, const double *fp, int n, double x_first,
                                                double x_last, double inv_dx)
{
    const Bracket<double> b = interp_locate(x, xp, n, x_first, x_last, x_first, inv_dx);
    const int j0 = min(b.j, n - 2);                          // (b.j == n-1: flat, the slope is not used)
    const double sl = (fp[j0 + 1] - fp[j0]) / (xp[j0 + 1] - xp[j0]);
    return interp_eval(x, b, fp[b.j], sl);
}

// numpy `.astype(int)` on x86-64 (lib/libprop.py:124-125) kept in the floating-point
// domain: truncation toward zero; NaN/inf/out-of-range behave as INT64_MIN.
template <typename T>
__device__ __forceinline__ T np_trunc_index(T t)
{
    return (fabs(t) < T(9.2233720368547758e18)) ? trunc(t) : T(-9.3e18);
}

// Correctly rounded x / d for a wave-uniform constant d with c = RN(1/d) (Markstein):
// q = RN(x*c); r = x - d*q exactly (fma); RN(q + r*c) == RN(x/d) for every finite x unless d's
// significand is all ones (the host clears `ok` then).  3 VALU ops instead of the ~14 of an
// IEEE fp64 division; bit-identical to numpy's `x / d` (checked on 2.4e8 samples on the CPU and
// by the bit-exact parity tests).  Non-finite x goes through the plain product (inf stays inf).
// ANYNAN = true: the caller treats a NaN result like an infinite one (np_trunc_index maps both to INT64_MIN) or only
// uses the result for finite x (the overlap of a contributing ray volume), so the final finite-x select is dropped
// (a class compare, its hazard slot and two v_cndmask per call; twice per ray and once per ray and level).
template <bool ANYNAN = false, typename T>
__device__ __forceinline__ T div_const(T x, T d, T c, int ok)
{
    if constexpr (std::is_same<T, float>::value) return x * c;   // float32: no bit-level pin (see real.h)
    if (__builtin_expect(!ok, 0)) return x / d;
    const T q = x * c;
    const T r = fma(-d, q, x);
    const T q2 = fma(r, c, q);
    if constexpr (ANYNAN) return q2;
    // non-finite x: q2 is NaN (inf - inf) where x / d is +-inf.  v_div_fixup_f64 -- the last instruction of the IEEE
    // division sequence -- puts that right in ONE instruction (inf / finite -> +-inf, NaN -> NaN, else the quotient with
    // the sign of x / d) where a class test and two selects took three
    return div_fixup_(q2, d, x);
}
template <typename T> __device__ __forceinline__ T third_rn() { return T(1) / T(3); }

// ------------------------------------------------------------------ deposit (wave_projection)
// lib/libprop.py:123-163.  Each lane holds RPT rays with extent [lo, up],
// NP payload values and the phase-space volume.  Levels are accumulated per
// WAVE into that wave's private LDS row (plain read-modify-write by lane 0, so
// the order is fixed), after a DPP reduction over the 64 lanes per level.  If
// the wave's rays span more than SPAN_MAX levels (unsorted input) the lanes
// fall back to LDS float64 atomics on the same private row.
template <int NP, typename T>
__device__ __forceinline__ void deposit_indices(T lo, T up, bool valid, T dz, T cdz, int ok, int nzmax, int &nlo, int &nup)
{
    const T nl = np_trunc_index(div_const<true>(lo, dz, cdz, ok));         // :124 (inf / NaN -> INT64_MIN either way)
    const T nu = np_trunc_index(div_const<true>(up, dz, cdz, ok) + T(1));  // :125
    const T nz = (T)nzmax;                                   // :127
    const bool ood = ((nl >= nz) && (nu >= nz)) || ((nl <= T(0)) && (nu <= T(0)));   // :129-130
    nlo = (int)fmin(fmax(nl, T(0)), nz);                     // :133-134
    nup = (int)fmin(fmax(nu, T(0)), nz);
    if (ood || !valid) { nlo = 0; nup = 0; }                 // :135, :153-154
}

// Per-level sums of a wave.  NP = 2 (the pseudo-momentum fluxes of the RHS): group_sum2_to_lds -- four butterfly steps, then
// one lane of every group of 8 adds the group's sum to the wave's private LDS row (ds_add_f64).  (Rounds 1-3 ran the
// whole 64-lane butterfly and kept the sums of a window of 32 levels in one register per lane, folded into the row when
// the window moved: 22 VALU instructions per level against 12.)  Columns of any height take the same path.
// NP = 1 (diagnostic projections): plain butterfly, lane 0 adds to the LDS row.
// T = float: weights, payloads and the butterfly steps are float32; the rows are float64.
#ifdef MSGW_DBG_LEVELS
__device__ unsigned long long g_dbg_levels, g_dbg_tiles, g_dbg_wide;
#endif

template <int NP, typename T, int RPT = Real<T>::RPT>
__device__ __forceinline__ void deposit_tile(const T (&lo)[RPT], const T (&up)[RPT],
                                             const int (&nlo)[RPT], const int (&nup)[RPT],
                                             const T (&vol)[RPT], const T (&pay)[NP][RPT],
                                             const T *sG, T dz, T cdz, int ok,
                                             double *row, int ncp, int lane, int &wmin, int &wmax)
{
    int mylo = INT_MAX, myhi = INT_MIN;
#pragma unroll
    for (int r = 0; r < RPT; ++r)
        if (nup[r] > nlo[r]) { mylo = min(mylo, nlo[r]); myhi = max(myhi, nup[r]); }
#if defined(MSGW_ABLATE) && MSGW_ABLATE == 4
    const int wlo = __builtin_amdgcn_readfirstlane(mylo), whi = __builtin_amdgcn_readfirstlane(myhi);
#else
    const int wlo = wave_min(mylo), whi = wave_max(myhi);    // wave-uniform
#endif
    if (whi <= wlo) return;
    wmin = min(wmin, wlo);
    wmax = max(wmax, whi);
#ifdef MSGW_DBG_LEVELS
    if (lane == 0) { atomicAdd(&g_dbg_levels, (unsigned long long)(whi - wlo)); atomicAdd(&g_dbg_tiles, 1ull);
                     if (whi - wlo > SPAN_MAX) atomicAdd(&g_dbg_wide, 1ull); }
#endif
#if defined(MSGW_ABLATE) && MSGW_ABLATE == 1
    asm volatile("" :: "v"(lo[0]), "v"(up[1]), "v"(vol[0]), "v"(pay[0][0]), "v"(pay[NP - 1][1]));
    return;
#endif
    if (whi - wlo <= SPAN_MAX) {

        for (int c = wlo; c < whi; ++c) {                    // uniform trip count: all lanes stay
            const T g0 = sG[c], g1 = sG[c + 1];              // LDS broadcast reads
            T s[NP];
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const bool in = (c >= nlo[r]) & (c < nup[r]);
                const T zmin = max1(g0, lo[r]);                         // :157 (one instruction, see real.h)
                const T zmax = min1(g1, up[r]);                         // :158
                const T wv = div_const<true>(fabs(zmax - zmin), dz, cdz, ok) * vol[r];   // :160, :162
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const T term = in ? wv * pay[p][r] : T(0);
                    s[p] = (r == 0) ? term : s[p] + term;               // (no 0 + x: the compiler may not fold it)
                }
            }
            if (NP == 2) {
                group_sum2_to_lds(s[0], s[NP - 1], row + c, ncp, lane);
            } else {
                const double t = (double)wave_sum(s[0]);
                if (lane == 0) row[c] += t;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            for (int c = nlo[r]; c < nup[r]; ++c) {
                const T g0 = sG[c], g1 = sG[c + 1];
                const T zmin = max1(g0, lo[r]);
                const T zmax = min1(g1, up[r]);
                const T wv = div_const<true>(fabs(zmax - zmin), dz, cdz, ok) * vol[r];
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    __hip_atomic_fetch_add(&row[p * ncp + c], (double)(wv * pay[p][r]), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

// Sum the WAVES private rows in wave order and write this workgroup's row
// (only the touched level range) plus the range descriptor.
template <int NP>
__device__ __forceinline__ void flush_rows(const double *rows, int ncp, int *s_rng, int wave,
                                           int lane, int tid, int wmin, int wmax,
                                           double *partial, int *ranges)
{
    if (lane == 0) { s_rng[2 * wave] = wmin; s_rng[2 * wave + 1] = wmax; }
    __syncthreads();
    int bmin = INT_MAX, bmax = INT_MIN;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { bmin = min(bmin, s_rng[2 * w]); bmax = max(bmax, s_rng[2 * w + 1]); }
    if (bmax <= bmin) { bmin = 0; bmax = 0; }
    double *prow = partial + (size_t)blockIdx.x * NP * ncp;
    const int span = bmax - bmin;
    for (int i = tid; i < NP * span; i += BLOCK) {
        const int p = i / span, c = bmin + (i - p * span);
        double acc = rows[(0 * NP + p) * ncp + c];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) acc = acc + rows[(w * NP + p) * ncp + c];
        prow[p * ncp + c] = acc;
    }
    if (tid == 0) { ranges[2 * blockIdx.x] = bmin; ranges[2 * blockIdx.x + 1] = bmax; }
}

// Group variant of flush_rows (used by the RK-stage kernels): the workgroup's dense row is
// published with write-through (sc1) 8-byte stores, every storing wave drains its stores, one
// lane takes a ticket on the group's counter; the workgroup that draws the last ticket acquires
// (agent scope) and adds the group's rows in row order -- independent of arrival order, so the
// result is bitwise reproducible.  Protocol: cdna_hip_programming.md Guideline 16 / split-K
// recipe (sc1 payload + drained vmcnt + barrier + relaxed agent fetch_add; consumer acquire).
template <int NP, typename T>
__device__ __forceinline__ void flush_rows_group(const double *rows, int ncp, int *s_flag, double *s_scr,
                                                 int tid, const StageArgsT<T> a)
{
    typedef unsigned long long u64;
    const int ncols = NP * ncp;
    const int b = blockIdx.x, nb = gridDim.x;
    const int g = b / a.grp_size, r0 = g * a.grp_size, r1 = min(nb, r0 + a.grp_size);
    __syncthreads();                                          // all waves' rows complete in LDS
    double *mine = a.grp_part + (size_t)b * a.row_stride;
    for (int col = tid; col < ncols; col += BLOCK) {
        double acc = rows[col];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) acc = acc + rows[w * ncols + col];
        __hip_atomic_store(reinterpret_cast<u64 *>(mine + col), (u64)__double_as_longlong(acc),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // sc1 store
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains
    __syncthreads();
    if (tid == 0) {
        const unsigned int t = __hip_atomic_fetch_add(a.grp_cnt + g, 1u, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned int)(r1 - r0 - 1)) ? 1 : 0;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *s_flag = last;
    }
    __syncthreads();
    if (!*s_flag) return;
    // last arriver: two interleaved row sets (even / odd rows of the group), two columns per thread
    const int nc2 = (ncols + 1) / 2;                          // 16-B column pairs
    const int set = tid / nc2, c2 = tid - set * nc2;
    double2 acc2 = make_double2(0.0, 0.0);
    if (set < 2) {
        const double *src = a.grp_part + 2 * c2;
        for (int r = r0 + set; r < r1; r += 32) {             // 16 rows of this set per batch
            double2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                v[u] = *reinterpret_cast<const double2 *>(src + (size_t)min(r + 2 * u, r1 - 1) * a.row_stride);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool in = (r + 2 * u) < r1;
                acc2.x = acc2.x + (in ? v[u].x : 0.0);
                acc2.y = acc2.y + (in ? v[u].y : 0.0);
            }
        }
        if (set == 1) { s_scr[2 * c2] = acc2.x; s_scr[2 * c2 + 1] = acc2.y; }
    }
    __syncthreads();
    double *grow = a.grp_rows + (size_t)g * ncols;
    if (set == 0) {
        const double x = acc2.x + s_scr[2 * c2], y = acc2.y + s_scr[2 * c2 + 1];
        // group sums are re-read by another workgroup of this launch: write-through 8-byte stores
        if (2 * c2 < ncols)
            __hip_atomic_store(reinterpret_cast<u64 *>(grow + 2 * c2), (u64)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (2 * c2 + 1 < ncols)
            __hip_atomic_store(reinterpret_cast<u64 *>(grow + 2 * c2 + 1), (u64)__double_as_longlong(y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(a.grp_cnt + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm the ticket
        const unsigned int t2 = __hip_atomic_fetch_add(a.grp_cnt + 63, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t2 == (unsigned int)a.ngroups - 1u) ? 1 : 0;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *s_flag = last;
    }
    __syncthreads();
    if (!*s_flag) return;
    // third level: the reducer of the launch's last group adds the group sums (group order) into the
    // one row the next launch's prologue (or the all-reduce) reads
    if (tid < ncols) {
        double tot = 0.0;
        for (int r = 0; r < a.ngroups; r += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                v[u] = __longlong_as_double((long long)__hip_atomic_load(
                    reinterpret_cast<const u64 *>(a.grp_rows + (size_t)min(r + u, a.ngroups - 1) * ncols + tid),
                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
            for (int u = 0; u < 16; ++u) tot = tot + ((r + u < a.ngroups) ? v[u] : 0.0);
        }
        a.flux_out[tid] = tot;
    }
    if (tid == 0) __hip_atomic_store(a.grp_cnt + 63, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------ dispersion
// omega :383 and cg_rr :445-448 sharing sub-expressions (numpy recomputes the
// identical values; sharing them is bit-neutral).
template <typename T>
__device__ __forceinline__ void dispersion(T kk, T ll, T mm, T f2, T bvf2, T &kh2, T &m2, T &vk2, T &om, T &cgr)
{
    kh2 = kk * kk + ll * ll;
    m2 = mm * mm;
    vk2 = kh2 + m2;
    om = sqrt_(div_(bvf2 * kh2 + f2 * m2, vk2));
    cgr = div_(div_(-mm * (om * om - f2), om), vk2);
}

// saturation cap :601 (rho_f already interpolated)
template <typename T>
__device__ __forceinline__ T sat_cap(T sat_c, T rho_f, T omh, T bvf2, T mm_f, T f0sq)
{
    return div_(div_(sat_c * rho_f * omh * bvf2, mm_f * mm_f), omh * omh - f0sq);
}

// ------------------------------------------------------------------ K1: one RK stage over the rays
// STAGE 0,1,2: rhs_default (:618-651) + low-storage RK update (:693-698) of rr,
// mm (and dens when SAT) + pseudo-momentum-flux deposit (:654-658) of the
// stage's INPUT state.  STAGE 3: tendencies only, written to the q arrays
// (single-RHS probe).  DIRECT: the driver's post-step saturation
// (raytracer.py:182-188) fused into stage 2 (stage 0 keeps rr, mm copies).
// Registers of one tile (RPT rays per lane) -- loaded one tile ahead of use.
template <typename T>
struct TileRegs {
    static constexpr int RPT = Real<T>::RPT;
    T rr[RPT], mm[RPT], kk[RPT], ll[RPT], dens[RPT], drr[RPT], vol[RPT], ff[RPT], pvf[RPT];
    T qr[RPT], qm[RPT], qd[RPT], rr0[RPT], mm0[RPT];
    T cg[RPT];            // resident tiles only: cg_rr of the current state (see tile_body.inc)
    unsigned int off;     // byte offset of this lane's rays in every SoA array
    bool v[RPT];          // ray index < end of the workgroup's range
};

template <typename T, int STAGE, bool SAT, bool FVEC, bool DEPOSIT, bool DIRECT, bool CGMEM = false>
__device__ __forceinline__ void load_tile(TileRegs<T> &t, const StageArgsT<T> a, long long base, int tid,
                                          long long end)
{
    // Loads are unconditional (the arrays are padded by a tile); rays at or beyond `end` belong to
    // the next workgroup or to the padding: they are computed on but neither deposited nor stored.
    constexpr bool NEED_RHO = SAT || (DIRECT && STAGE == 2);
    constexpr int RPT = Real<T>::RPT;
    const long long i0 = base + RPT * tid;
    t.off = (unsigned int)(i0 * (long long)sizeof(T));
#pragma unroll
    for (int r = 0; r < RPT; ++r) t.v[r] = i0 + r < end;
    loadv(a.r.rr(), t.off, t.rr);
    loadv(a.r.mm(), t.off, t.mm);
    loadv(a.r.kk(), t.off, t.kk);
    loadv(a.r.ll(), t.off, t.ll);
    if (DEPOSIT || SAT || (DIRECT && STAGE == 2)) loadv(a.r.dens(), t.off, t.dens);
    if (DEPOSIT) {
        loadv(a.r.drr(), t.off, t.drr);
        loadv(a.r.vol(), t.off, t.vol);
    }
    if (FVEC) loadv(a.r.fray(), t.off, t.ff);
    if (NEED_RHO) loadv(a.r.pvf(), t.off, t.pvf);
    if (STAGE == 1 || STAGE == 2) {
        loadv(a.r.q_rr(), t.off, t.qr);
        loadv(a.r.q_mm(), t.off, t.qm);
        if (SAT) loadv(a.r.q_dens(), t.off, t.qd);
    }
    if (DIRECT && STAGE == 2) {
        loadv(a.r.rr0(), t.off, t.rr0);
        loadv(a.r.mm0(), t.off, t.mm0);
    }
    if (CGMEM) loadv(a.r.cg(), t.off, t.cg);
}

// Streamed tiles of the resident-tile flavours carry cg_rr of their state through memory instead of re-evaluating it
// (16 B per ray-stage for a sqrt and three divisions).  With four resident tiles a workgroup only has streamed
// tiles beyond 1e6 rays per GPU, where the kernel runs at the HBM ceiling: there the bytes are the dearer side.
#ifndef CGMEM_MAX_NRES
#define CGMEM_MAX_NRES 2
#endif

// LDS views shared by the stage kernels
template <typename T>
struct StageLds {
    const typename Real<T>::quad_t *sh;      // [ni] {dudz, slope, dvdz, slope}
    const typename Real<T>::pair_t *rho2;    // [nc] {rhobar, slope}
    const T *xg, *gs;       // abscissae: grid[1:-1] [ni], grids [nc]
    double *rows;           // [WAVES][2][ncp] per-wave flux rows
};

// Carve of the dynamic LDS of the stage kernels (and, with the column replica behind it, of the persistent
// kernel): [sh: ni x {dudz, slu, dvdz, slv}] [rho2: nc x {rhobar, slope}] [xg: ni] [gs: nc] in T, then (T = float
// only) a float64 copy of xg for the mean-flow arithmetic, then float64 [rows: WAVES x 2 x ncp] [rng].
// The prologue scratch of the fused column update aliases the rows: F [2][ng], u, v [nc], du, dv [ni]
// = 6*ng - 6 doubles inside 8*(ng - 2), which needs ng >= 5 (msgw_create enforces it).
template <typename T>
struct StageCarve {
    typename Real<T>::quad_t *sh;
    typename Real<T>::pair_t *rho2;
    T *xg, *gs;
    double *xgd;            // grid[1:-1] in float64 (== xg when T = double)
    double *rows;
    int *rng;
    double *F, *u, *v, *du, *dv;
    __device__ __forceinline__ StageCarve(void *lds, int ng)
    {
        const int ni = ng - 2, nc = ng - 1, ncp = ng - 2;
        T *t = reinterpret_cast<T *>(lds);
        sh = reinterpret_cast<typename Real<T>::quad_t *>(t);
        rho2 = reinterpret_cast<typename Real<T>::pair_t *>(t + 4 * ni);
        xg = t + 4 * ni + 2 * nc;
        gs = xg + ni;
        if constexpr (std::is_same<T, double>::value) {
            xgd = xg;
            rows = gs + nc;
        } else {
            const size_t bytes = (sizeof(T) * (size_t)(5 * ni + 3 * nc) + 15) & ~(size_t)15;
            xgd = reinterpret_cast<double *>(reinterpret_cast<char *>(lds) + bytes);
            rows = xgd + ni;
        }
        rng = reinterpret_cast<int *>(rows + WAVES * 2 * ncp);
        F = rows; u = F + 2 * ng; v = u + nc; du = v + nc; dv = du + ni;
    }
};

// NOTE: the kernel-argument structs are passed BY VALUE into these inlined helpers.  By
// reference (a pointer into the kernarg segment) hipcc no longer keeps the fields in SGPRs and
// the ray-stage kernel grows from 106 to 146 VGPRs (occupancy 4 -> 3).
// All tiles of this workgroup for one RK stage: physics + RK update + deposit (steps 2-3 of the
// kernel description in DESIGN.md).  `cur` holds the first tile's registers (already loaded);
// the shear/rho tables must be staged and the wave rows zeroed.  Ends with the wave's register
// accumulators folded into its LDS row.
// LAG = true ("lagged deposit", persistent kernel): instead of depositing the stage's INPUT state,
// deposit the state this stage has just PRODUCED, i.e. the next stage's wave_projection input.
// Same values (cg_rr is re-evaluated from the same kk, ll, new mm the next stage will load), but the
// flux of stage q+1 is then published one whole pass before it is needed.
template <typename T, int STAGE, bool SAT, bool FVEC, bool DEPOSIT, bool DIRECT, bool LAG = false, int NRES = 0,
          bool RELAUNCH = false>
__device__ __forceinline__ void process_tiles(const StageArgsT<T> a, const StageLds<T> L, TileRegs<T> &cur,
                                              long long start, long long end, int tid, int wave,
                                              int lane, int &wmin, int &wmax,
                                              TileRegs<T> (*res)[NRES > 0 ? NRES : 1] = nullptr,
                                              const unsigned int *poll_ctr = nullptr, unsigned int *polled = nullptr)
{
    constexpr int RPT = Real<T>::RPT;
    constexpr int TILE = Real<T>::TILE;
    typedef typename Real<T>::quad_t quad_t;
    typedef typename Real<T>::pair_t pair_t;
    const int ng = a.ng, ni = ng - 2, nc = ng - 1, ncp = ng - 2;
    const quad_t *s_sh = L.sh;
    const pair_t *s_rho2 = L.rho2;
    const T *s_xg = L.xg, *s_gs = L.gs;
    double *s_rows = L.rows;
    (void)s_rho2; (void)ng;
    // with resident tiles the kernel is FP64-VALU bound, not HBM bound: streamed tiles then carry cg_rr of
    // their state through memory (8 B store + 8 B load per ray-stage) instead of re-evaluating it
    constexpr bool CGMEM = NRES > 0 && NRES <= CGMEM_MAX_NRES && LAG && DEPOSIT && !SAT;
    // The exact constant division (div_const) has a run-time fall-back for grid spacings whose significand is all ones;
    // the resident-tile flavours of the persistent kernel are only launched when it is not needed (plan_persist), so
    // that the test and its branches leave their level loop (twice per ray, once per ray and level).
    const int mk_ok = NRES > 0 ? 1 : a.mk_ok;
    // resident tiles first: they are the workgroup's first NRES tiles, so the deposit order is ray order
    if constexpr (NRES > 0) {
        static_assert(NRES == 0 || STAGE != 3, "the single-RHS probe has no resident tiles");
#pragma unroll
        for (int i = 0; i < NRES; ++i) {
            // persistent kernel, PREFETCH (persist_kernel.h): lane 0 looks at the release counter one tile before the end
            if (i == NRES - 1 && poll_ctr && tid == 0)
                *polled = __hip_atomic_load(poll_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#define TB_T (*res)[i]
#define TB_NEXT (*res)[(i + 1) % NRES]
#define TB_RESIDENT true
#define TB_IDX i
#include "tile_body.inc"
#undef TB_T
#undef TB_NEXT
#undef TB_RESIDENT
#undef TB_IDX
        }
    }
    // streamed tiles: `start` is the first streamed ray, `cur` holds its tile (already loaded)
    for (int t = 0; t < a.tiles_per_block - NRES; ++t) {
        const long long base = start + (long long)t * TILE;
        if (base >= end) break;                              // workgroup-uniform
        const bool more = (t + 1 < a.tiles_per_block - NRES) && (base + TILE < end);
#define TB_T cur
#define TB_NEXT cur
#define TB_RESIDENT false
#define TB_IDX t
#include "tile_body.inc"
#undef TB_T
#undef TB_NEXT
#undef TB_RESIDENT
#undef TB_IDX
        MSGW_STAMP_AT(3 + 2 * (t & 1));
        if (more) load_tile<T, STAGE, SAT, FVEC, DEPOSIT, DIRECT, CGMEM>(cur, a, base + TILE, tid, end);
    }
}

// Deposit-only pass over this workgroup's rays (no stores): wave_projection(var=0) of the CURRENT
// state into the per-wave LDS rows.  Seeds the lagged-deposit pipeline of the persistent kernel.
template <typename T, bool FVEC, bool CGSTORE = false>
__device__ __forceinline__ void deposit_pass(const StageArgsT<T> a, const StageLds<T> L, long long start, long long end,
                                             int tid, int wave, int lane, long long cg_from = 0)
{
    constexpr int RPT = Real<T>::RPT;
    constexpr int TILE = Real<T>::TILE;
    const int nc = a.ng - 1, ncp = a.ng - 2;
    int wmin = INT_MAX, wmax = INT_MIN;
    for (int t = 0; t < a.tiles_per_block; ++t) {
        const long long base = start + (long long)t * TILE;
        if (base >= end) break;
        const long long e0 = base + RPT * tid;
        const unsigned int off = (unsigned int)(e0 * (long long)sizeof(T));
        bool valid[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) valid[r] = e0 + r < end;
        T rr[RPT], mm[RPT], kk[RPT], ll[RPT], dens[RPT], drr[RPT], vol[RPT], ff[RPT];
        loadv(a.r.rr(), off, rr); loadv(a.r.mm(), off, mm); loadv(a.r.kk(), off, kk); loadv(a.r.ll(), off, ll);
        loadv(a.r.dens(), off, dens); loadv(a.r.drr(), off, drr); loadv(a.r.vol(), off, vol);
        if (FVEC) loadv(a.r.fray(), off, ff);
        T lo[RPT], up[RPT], pay[2][RPT], cg2[RPT];
        int nlo[RPT], nup[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const T f = FVEC ? ff[r] : a.f_uni;
            T kh2, m2, vk2, om, cgr;
            dispersion(kk[r], ll[r], mm[r], f * f, a.bvf2, kh2, m2, vk2, om, cgr);
            cg2[r] = cgr;
            lo[r] = rr[r] - T(.5) * drr[r];
            up[r] = rr[r] + T(.5) * drr[r];
            deposit_indices<2>(lo[r], up[r], valid[r], a.dzs, a.inv_dzs, a.mk_ok, nc - 2, nlo[r], nup[r]);
            pay[0][r] = cgr * kk[r] * dens[r];
            pay[1][r] = cgr * ll[r] * dens[r];
        }
        if (CGSTORE && valid[0] && base >= cg_from) storev(a.r.cg(), off, cg2);   // streamed tiles of the persistent kernel
        deposit_tile<2, T>(lo, up, nlo, nup, vol, pay, L.gs, a.dzs, a.inv_dzs, a.mk_ok,
                           L.rows + wave * 2 * ncp, ncp, lane, wmin, wmax);
    }
}

// stage the static tables of the column (float64 in global memory) into the LDS views
template <typename T, int STRIDE = BLOCK>
__device__ __forceinline__ void stage_shear_table(const StageCarve<T> &C, const ColPtrs c, int ni, int tid)
{
    for (int i = tid; i < ni; i += STRIDE) {
        const bool in = i < ni - 1;                      // the last point has no slope (never used)
        C.sh[i] = Real<T>::quad((T)c.dudz[i], in ? (T)c.slu[i] : T(0), (T)c.dvdz[i], in ? (T)c.slv[i] : T(0));
    }
}
template <typename T, int STRIDE = BLOCK>
__device__ __forceinline__ void stage_xg(const StageCarve<T> &C, const ColPtrs c, int ni, int tid)
{
    for (int i = tid; i < ni; i += STRIDE) {
        const double x = c.xg[i];
        C.xg[i] = (T)x;
        if constexpr (!std::is_same<T, double>::value) C.xgd[i] = x;
    }
}
template <typename T, int STRIDE = BLOCK>
__device__ __forceinline__ void stage_rho(const StageCarve<T> &C, const ColPtrs c, int nc, int tid, bool rho)
{
    for (int i = tid; i < nc; i += STRIDE) {
        C.gs[i] = (T)c.grids[i];
        if (rho) C.rho2[i] = Real<T>::pair((T)c.rhobar[i], (i < nc - 1) ? (T)c.slrho[i] : T(0));
    }
}
// packed shear table from the float64 shear columns in LDS (fused column update)
template <typename T>
__device__ __forceinline__ void pack_shear_table(typename Real<T>::quad_t *sh, const double *du, const double *dv,
                                                 const double *xgd, int ni, int tid)
{
    for (int i = tid; i < ni; i += BLOCK) {
        const bool in = i < ni - 1;
        sh[i] = Real<T>::quad((T)du[i], in ? (T)column_slope(du, xgd, i) : T(0),
                              (T)dv[i], in ? (T)column_slope(dv, xgd, i) : T(0));
    }
}

// (A register double-buffered "prefetch next tile" variant was measured and dropped: its ~50
// extra VGPRs cost a wave of occupancy, capping it at 128 VGPRs spilled, and every spill reload
// carries s_waitcnt vmcnt(0), which drains the prefetch.  The other resident workgroups cover
// a workgroup's load latency instead.)
template <typename T, int STAGE, bool SAT, bool FVEC, bool DEPOSIT, bool DIRECT, bool GROUPRED = false,
          bool LAG = false, bool RELAUNCH = false>
__global__ void __launch_bounds__(BLOCK) k_ray_stage(const StageArgsT<T> a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ng = a.ng, ni = ng - 2, nc = ng - 1, ncp = ng - 2;
    constexpr bool NEED_RHO = SAT || (DIRECT && STAGE == 2);
    const StageCarve<T> C(lds, ng);
    double *s_rows = C.rows;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long start = (long long)blockIdx.x * a.rays_per_block;
    const long long end = min(a.n, start + a.rays_per_block);

    MSGW_STAMP_AT(0);
    // Load order matters (vmcnt retires in order): first the handful of loads of the pending
    // mean-flow update, THEN the first tile's ray loads, so that the column math below runs
    // while the ray data is in flight.  (Host guarantees 2*ncp <= BLOCK and nc <= BLOCK
    // whenever col_pending is set.)
    const bool fuse = DEPOSIT && a.col_pending;
    const int ncols = 2 * ncp;
    double c_tot = 0.0, c_u = 0, c_v = 0, c_qu = 0, c_qv = 0, c_rho = 1, c_pg0 = 0, c_pg1 = 0;
    if (fuse) {
        const int col = min(tid, ncols - 1), jc = min(tid, nc - 1);
        c_tot = a.col_rows[col];                             // the final (all-reduced) flux row
        c_u = a.cin.uu[jc]; c_v = a.cin.vv[jc];
        c_qu = a.cin.q_uu[jc]; c_qv = a.cin.q_vv[jc];
        c_rho = a.c.rhobar[jc]; c_pg0 = a.pg[jc]; c_pg1 = a.pg[nc + jc];
    }
    TileRegs<T> cur;
    load_tile<T, STAGE, SAT, FVEC, DEPOSIT, DIRECT>(cur, a, start, tid, end);

    stage_xg(C, a.c, ni, tid);
    if (fuse) {
        // (1) finish the flux reduction
        if (tid < ncols) {
            const int p = tid / ncp, c = tid - p * ncp;
            C.F[p * ng + 1 + c] = c_tot;                     // pm_flux[:, 1:-1]  (:654)
        }
        __syncthreads();
        column_flux_ends(tid, ng, C.F);
        __syncthreads();
        // (2) RK stage of uu, vv (:665-666, :693-698); workgroup 0 publishes the new column
        if (tid < nc) {
            double du, dv, un, vn, qu, qv;
            column_tendency(tid, ng, a.f0, a.dzg, 0, C.F, c_rho, c_pg0, c_pg1, c_u, c_v, du, dv);
            column_rk(a.col_stage, a.dtc, du, dv, c_u, c_v, c_qu, c_qv, un, vn, qu, qv);
            C.u[tid] = un; C.v[tid] = vn;
            if (blockIdx.x == 0) { a.cout.uu[tid] = un; a.cout.vv[tid] = vn; a.cout.q_uu[tid] = qu; a.cout.q_vv[tid] = qv; }
        }
        __syncthreads();
        // (3) shear + np.interp slopes straight into the packed LDS table
        column_shear(tid, BLOCK, ng, a.dzg, C.u, C.v, C.du, C.dv);
        __syncthreads();
        pack_shear_table<T>(C.sh, C.du, C.dv, C.xgd, ni, tid);
        __syncthreads();                                     // scratch is re-used as the wave rows below
    } else {
        stage_shear_table(C, a.c, ni, tid);
    }
    if (DEPOSIT || NEED_RHO) stage_rho(C, a.c, nc, tid, NEED_RHO);
    if (DEPOSIT)
        for (int i = tid; i < WAVES * 2 * ncp; i += BLOCK) s_rows[i] = 0.0;
    __syncthreads();
    MSGW_STAMP_AT(1);

    int wmin = INT_MAX, wmax = INT_MIN;
    const StageLds<T> L{C.sh, C.rho2, C.xg, C.gs, s_rows};
    process_tiles<T, STAGE, SAT, FVEC, DEPOSIT, DIRECT, LAG, 0, RELAUNCH>(a, L, cur, start, end, tid, wave, lane, wmin, wmax);
    if (DEPOSIT) {
        if (GROUPRED) flush_rows_group<2, T>(s_rows, ncp, C.rng, lds /* interp tables are dead by now */, tid, a);
        else flush_rows<2>(s_rows, ncp, C.rng, wave, lane, tid, wmin, wmax, a.partial, a.ranges);
    }
    MSGW_STAMP_AT(6);
}

// Deposit-only launch that seeds the lagged launch chain: F_0 = wave_projection(state_0) into the
// group rows (same LDS carve and group reduction as k_ray_stage).
template <typename T, bool FVEC>
__global__ void __launch_bounds__(BLOCK) k_deposit_only(const StageArgsT<T> a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ng = a.ng, nc = ng - 1, ncp = ng - 2;
    const StageCarve<T> C(lds, ng);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long start = (long long)blockIdx.x * a.rays_per_block;
    const long long end = min(a.n, start + a.rays_per_block);
    stage_rho(C, a.c, nc, tid, false);
    for (int i = tid; i < WAVES * 2 * ncp; i += BLOCK) C.rows[i] = 0.0;
    __syncthreads();
    const StageLds<T> L{nullptr, nullptr, C.xg, C.gs, C.rows};
    deposit_pass<T, FVEC>(a, L, start, end, tid, wave, lane);
    flush_rows_group<2, T>(C.rows, ncp, C.rng, lds, tid, a);
}

// ------------------------------------------------------------------ K1f: fixed background, whole RK3 step in registers
// The rhs hook that zeroes slots 9, 10 (BASELINE configs 1, 2): no inter-ray
// dependency, so the three stages run back to back per ray, and ALL the steps of
// a call run in one launch: rr, mm (dens) touch HBM once per call, not per step
// (same arithmetic in the same order: bit-identical to one launch per step).
// NARROW = false: 256-thread workgroups, 2 rays per lane (16-B / 8-B accesses), tiles of 512 rays.
// NARROW = true : ONE wavefront per workgroup, ONE ray per lane, tiles of 64 rays -- the geometry of small ray counts
// (BASELINE config 2: 1e5 rays).  The kernel is bound by the instruction issue of dependent float64 chains (three IEEE
// divisions and a square root per ray-stage, ~130 instructions each in sequence), not by memory: with 2 rays per lane
// 1e5 rays are 784 wavefronts -- one per SIMD on 784 of the chip's 1024 SIMDs, each issuing two rays' chains alone
// (profiles/r02_config2_summary.md: VALU 41 % busy).  With one ray per lane they are 1563 wavefronts on ALL SIMDs,
// half the chain per wavefront, and a SIMD that hosts two of them interleaves their issue.
template <typename T, bool SAT, bool FVEC, bool DIRECT, bool NARROW = false>
__global__ void __launch_bounds__(NARROW ? 64 : BLOCK) k_ray_step_fixed(const StageArgsT<T> a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RPT = NARROW ? 1 : Real<T>::RPT;
    constexpr int BLK = NARROW ? 64 : BLOCK;
    constexpr int TILE = BLK * RPT;
    typedef typename Real<T>::quad_t quad_t;
    typedef typename Real<T>::pair_t pair_t;
    const int ng = a.ng, ni = ng - 2, nc = ng - 1;
    constexpr bool NEED_RHO = SAT || DIRECT;
    const StageCarve<T> C(lds, ng);
    const quad_t *s_sh = C.sh;
    const pair_t *s_rho2 = C.rho2;
    const T *s_xg = C.xg, *s_gs = C.gs;
    const int tid = threadIdx.x;
    stage_xg<T, BLK>(C, a.c, ni, tid);
    stage_shear_table<T, BLK>(C, a.c, ni, tid);
    if (NEED_RHO) stage_rho<T, BLK>(C, a.c, nc, tid, true);
    __syncthreads();

    const long long start = (long long)blockIdx.x * a.rays_per_block;
    const long long end = min(a.n, start + a.rays_per_block);
    for (int t = 0; t < a.tiles_per_block; ++t) {
        const long long base = start + (long long)t * TILE;
        if (base >= end) break;
        const long long e0 = base + RPT * tid;
        const unsigned int i0 = (unsigned int)(e0 * (long long)sizeof(T));
        const bool own = e0 < end;
        T rr[RPT], mm[RPT], kk[RPT], ll[RPT], dens[RPT], ff[RPT], pvf[RPT];
        loadv(a.r.rr(), i0, rr);
        loadv(a.r.mm(), i0, mm);
        loadv(a.r.kk(), i0, kk);
        loadv(a.r.ll(), i0, ll);
        if (FVEC) loadv(a.r.fray(), i0, ff);
        if (NEED_RHO) { loadv(a.r.dens(), i0, dens); loadv(a.r.pvf(), i0, pvf); }
        T sd[RPT], sr[RPT], sm[RPT], drr[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) { sd[r] = T(0); sr[r] = T(0); sm[r] = T(0); drr[r] = T(0); }
        if (a.relaunch) {                                     // workgroup-uniform
            loadv(a.r.src_dens(), i0, sd); loadv(a.r.src_rr(), i0, sr); loadv(a.r.src_mm(), i0, sm);
            loadv(a.r.drr(), i0, drr);
            if (!NEED_RHO) loadv(a.r.dens(), i0, dens);
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const T f = FVEC ? ff[r] : a.f_uni;
            const T f2 = f * f;
            const T kh2 = kk[r] * kk[r] + ll[r] * ll[r];
          for (int step = 0; step < a.fixed_steps; ++step) {
            const T rr_old = rr[r], mm_old = mm[r];
            T q_r = T(0), q_m = T(0), q_d = T(0);
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const T m2 = mm[r] * mm[r];
                const T vk2 = kh2 + m2;
                const T om = sqrt_(div_(a.bvf2 * kh2 + f2 * m2, vk2));
                const T cgr = div_(div_(-mm[r] * (om * om - f2), om), vk2);
                const T st_rr = cgr;                                   // = .5 * (cgr + cgr) exactly (:640, see tile_body.inc)
                const Bracket<T> bk = interp_locate(rr[r], s_xg, ni, a.xg0, a.xg_last, a.xg0, a.inv_dzg);
                const quad_t sh = s_sh[bk.j];
                const T gu = interp_eval(rr[r], bk, sh.x, sh.y);
                const T gv = interp_eval(rr[r], bk, sh.z, sh.w);
                const T st_mm = -(kk[r] * gu + ll[r] * gv);                       // :519-520 (see tile_body.inc)
                T st_dens = T(0);
                if (SAT) {
                    const T rr_f = rr[r] + st_rr * a.dt;
                    const T mm_f = mm[r] + st_mm * a.dt;
                    const Bracket<T> br = interp_locate(rr_f, s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
                    const pair_t rh = s_rho2[br.j];                                    // {rhobar, slope}
                    const T rho_f = interp_eval(rr_f, br, rh.x, rh.y);        // :595
                    const T omh = a.same_f ? om : sqrt_(div_(a.bvf2 * kh2 + a.f0sq * m2, vk2));
                    const T maxd = sat_cap(a.sat_c, rho_f, omh, a.bvf2, mm_f, a.f0sq);
                    if (maxd < dens[r] * pvf[r]) st_dens = div_(maxd - dens[r], a.dt);
                }
                if (st == 0) {
                    q_r = a.dt * st_rr; q_m = a.dt * st_mm;
                    rr[r] = rr[r] + div_const(q_r, T(3), third_rn<T>(), 1);
                    mm[r] = mm[r] + div_const(q_m, T(3), third_rn<T>(), 1);
                    if (SAT) { q_d = a.dt * st_dens; dens[r] = dens[r] + div_const(q_d, T(3), third_rn<T>(), 1); }
                } else {
                    const T A = (st == 1) ? T(RK_A1) : T(RK_A2), B = (st == 1) ? T(RK_B1) : T(RK_B2);
                    q_r = a.dt * st_rr - A * q_r; q_m = a.dt * st_mm - A * q_m;
                    rr[r] = rr[r] + B * q_r; mm[r] = mm[r] + B * q_m;
                    if (SAT) { q_d = a.dt * st_dens - A * q_d; dens[r] = dens[r] + B * q_d; }
                }
            }
            if (DIRECT) {
                const T rr_st = div_(rr[r] - rr_old, a.sat_rr_div);
                const T mm_st = div_(mm[r] - mm_old, a.dt);
                const T rr_f = rr_old + rr_st * a.dt;
                const T mm_f = mm_old + mm_st * a.dt;
                const Bracket<T> br = interp_locate(rr_f, s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
                const pair_t rh = s_rho2[br.j];                                    // {rhobar, slope}
                const T rho_f = interp_eval(rr_f, br, rh.x, rh.y);        // :595
                const T m02 = mm_old * mm_old;
                const T omh = sqrt_(div_(a.bvf2 * kh2 + a.f0sq * m02, kh2 + m02));
                const T maxd = sat_cap(a.sat_c, rho_f, omh, a.bvf2, mm_f, a.f0sq);
                if (maxd < dens[r] * pvf[r]) dens[r] = maxd;
            }
            if (a.relaunch) {                                 // EXTENSION MSGW_RELAUNCH (include/msgwam_hip.h)
                const bool out = (rr[r] - T(.5) * drr[r] > a.z_top) || (rr[r] + T(.5) * drr[r] < a.z_bot) ||
                                 (dens[r] < a.relaunch_frac * sd[r]);
                if (out) { dens[r] = sd[r]; rr[r] = sr[r]; mm[r] = sm[r]; }
            }
          }
        }
        if (own) {
            storev(a.r.rr(), i0, rr);
            storev(a.r.mm(), i0, mm);
            if (NEED_RHO || a.relaunch) storev(a.r.dens(), i0, dens);
        }
    }
}

// ------------------------------------------------------------------ diagnostics: wave_projection(var) on any grid
struct ProjExplicit {     // caller-supplied arrays of lprop.wave_projection (:92-94), always float64
    const double *dens, *lo, *up, *kk, *ll, *mlo, *mup, *dkk, *dll, *dmm, *fray;
};
template <typename T>
struct ProjArgsT {
    long long n;
    int nG;               // points of G; output has nG-1 levels
    int tiles_per_block;
    int var;              // payload: 0 pseudo-momentum fluxes (NP = 2), 1 wave-action flux, 2 wave action
    int boundary;         // 1: var 3 / 4 (:199-219) -- sums at the interfaces nb with nlow < nb < nup, unit weight
    T bvf2, f_uni, dz, cdz;
    int mk_ok;
    RayPtrsT<T> r;        // resident rays   (EXPL = false)
    ProjExplicit e;       // explicit arrays (EXPL = true; T = double)
    const double *G;
    double *partial;      // [blocks][NP][nG-1]
    int *ranges;
    // EXTENSION: N as a column on `grids` (nullptr: the scalar in bvf2), taken at the ray centre; the resident rays'
    // dmm then evolves, so their phase-space volume is |dkk*dll * dmm| of the CURRENT dmm (dkdl), not the upload's
    const T *dkdl;
    const double *bvfcol, *grids;
    // HPROP_GLOBAL = True: the resident rays' latitude evolves, so their Coriolis parameter is 2 Omega sin(phi) of the
    // CURRENT phi (:382), not the per-ray f of the upload (nullptr: HPROP off)
    const T *phi;
    T two_rot;
    int nc;
    double gs0, gs_last, inv_dzs;
};
typedef ProjArgsT<double> ProjArgs;

template <typename T, int NP, bool FVEC, bool EXPL>
__global__ void __launch_bounds__(BLOCK) k_project(const ProjArgsT<T> a)
{
    static_assert(!EXPL || std::is_same<T, double>::value, "caller arrays are float64");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RPT = Real<T>::RPT;
    constexpr int TILE = Real<T>::TILE;
    const int nG = a.nG, ncp = nG - 1;
    double *s_rows = lds;                                    // [WAVES][NP][ncp] float64
    int *s_rng = reinterpret_cast<int *>(s_rows + WAVES * NP * ncp);   // [2*WAVES]
    T *s_G = reinterpret_cast<T *>(s_rng + 2 * WAVES);       // [nG]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < nG; i += BLOCK) s_G[i] = (T)a.G[i];
    for (int i = tid; i < WAVES * NP * ncp; i += BLOCK) s_rows[i] = 0.0;
    __syncthreads();
    int wmin = INT_MAX, wmax = INT_MIN;
    const long long tile0 = (long long)blockIdx.x * a.tiles_per_block;
    for (int t = 0; t < a.tiles_per_block; ++t) {
        const long long base = (tile0 + t) * (long long)TILE;
        if (base >= a.n) break;
        const long long e0 = base + RPT * tid;
        const unsigned int i0 = (unsigned int)(e0 * (long long)sizeof(T));
        bool valid[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) valid[r] = e0 + r < a.n;
        T kk[RPT], ll[RPT], dens[RPT], vol[RPT], ff[RPT], lo[RPT], up[RPT], mmid[RPT], zctr[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) zctr[r] = T(0);
        if constexpr (EXPL) {
            T mlo[RPT], mup[RPT], dkk[RPT], dll[RPT], dmm[RPT];
            loadv(a.e.dens, i0, dens);
            loadv(a.e.lo, i0, lo);
            loadv(a.e.up, i0, up);
            loadv(a.e.kk, i0, kk);
            loadv(a.e.ll, i0, ll);
            loadv(a.e.mlo, i0, mlo);
            loadv(a.e.mup, i0, mup);
            loadv(a.e.dkk, i0, dkk);
            loadv(a.e.dll, i0, dll);
            loadv(a.e.dmm, i0, dmm);
            loadv(a.e.fray, i0, ff);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                vol[r] = fabs(dkk[r] * dll[r] * dmm[r]);              // :137
                mmid[r] = T(.5) * (mlo[r] + mup[r]);                  // :141
            }
        } else {
            T rr[RPT], mm[RPT], drr[RPT], dmm[RPT];
            loadv(a.r.rr(), i0, rr);
            loadv(a.r.mm(), i0, mm);
            loadv(a.r.kk(), i0, kk);
            loadv(a.r.ll(), i0, ll);
            loadv(a.r.dens(), i0, dens);
            loadv(a.r.drr(), i0, drr);
            loadv(a.r.dmm(), i0, dmm);
            loadv(a.r.vol(), i0, vol);
            if (FVEC) loadv(a.r.fray(), i0, ff);
            if (a.phi) {                                              // (kernel-uniform) HPROP: the latitude has evolved
                T ph[RPT];
                loadv(a.phi, i0, ph);
#pragma unroll
                for (int r = 0; r < RPT; ++r) ff[r] = a.two_rot * sin(ph[r]);
            }
            if (a.dkdl) {                                             // (kernel-uniform) N(z) column: dmm has evolved
                T dk[RPT];
                loadv(a.dkdl, i0, dk);
#pragma unroll
                for (int r = 0; r < RPT; ++r) vol[r] = fabs(dk[r] * dmm[r]);       // :137
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                lo[r] = rr[r] - T(.5) * drr[r];                       // :655 / raytracer.py:200-201
                up[r] = rr[r] + T(.5) * drr[r];
                zctr[r] = rr[r];
                mmid[r] = T(.5) * ((mm[r] - T(.5) * dmm[r]) + (mm[r] + T(.5) * dmm[r]));   // :141, :656
            }
        }
        T pay[NP][RPT];
        int nlo[RPT], nup[RPT];
        T zlo[RPT], zup[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) { zlo[r] = lo[r]; zup[r] = up[r]; }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            deposit_indices<NP>(lo[r], up[r], valid[r], a.dz, a.cdz, a.mk_ok, nG - 2, nlo[r], nup[r]);
            if (a.boundary) {            // interfaces nlow+1 .. nup-1 (:205, :216); weight |G[nb+1]-G[nb]|/dz = 1
                nlo[r] += 1;
                lo[r] = T(-1e30); up[r] = T(1e30);
            }
            const T f = (FVEC || EXPL) ? ff[r] : (a.phi ? ff[r] : a.f_uni);
            T kh2, m2, vk2, om, cgr;
            T bvf2 = a.bvf2;
            if (a.bvfcol) {                                           // (kernel-uniform) N at the ray centre
                const double zc = EXPL ? .5 * ((double)zlo[r] + (double)zup[r]) : (double)zctr[r];
                const double N = interp_global(zc, a.grids, a.bvfcol, a.nc, a.gs0, a.gs_last, a.inv_dzs);
                bvf2 = (T)(N * N);
            }
            dispersion(kk[r], ll[r], mmid[r], f * f, bvf2, kh2, m2, vk2, om, cgr);
            if (NP == 2) { pay[0][r] = cgr * kk[r] * dens[r]; pay[NP - 1][r] = cgr * ll[r] * dens[r]; }   // :148-149
            else pay[0][r] = (a.var == 1) ? cgr * dens[r] : dens[r];                     // :167, :184
        }
        deposit_tile<NP, T>(lo, up, nlo, nup, vol, pay, s_G, a.dz, a.cdz, a.mk_ok,
                            s_rows + wave * NP * ncp, ncp, lane, wmin, wmax);
    }
    flush_rows<NP>(s_rows, ncp, s_rng, wave, lane, tid, wmin, wmax, a.partial, a.ranges);
}

// ------------------------------------------------------------------ lprop.saturation on caller arrays (:561-615)
struct SatArgs {
    long long n;
    int nc, direct;
    double dt, bvf2, f0sq, sat_c, gs0, gs_last, inv_dzs;
    const double *dens, *rr, *rr_st, *drr, *drr_st, *kk, *ll, *mm, *mm_st, *dkk, *dll, *area;
    const double *grids, *rhobar, *slrho;
    const double *bvfcol;   // EXTENSION: N on grids (nullptr: the scalar in bvf2); N at rr_center and at rr_final
    double *out;
};

// ------------------------------------------------------------------ upload / download helpers
// inert padding rays [from, to) so that whole-tile vector accesses stay in initialised memory
template <typename T>
__global__ void k_fill_range(T *p, long long from, long long to, T v)
{
    const long long i = from + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < to) p[i] = v;
}

// vol = |dkk*dll*dmm| (:137) and pvf = dkk*dll*(rr_mm_area/drr) (:594, :599 with
// drr_final = drr + 0*dt) once per upload, from the caller's float64 values.
template <typename T>
__global__ void k_prepare(long long n, const double *dkk, const double *dll, const double *area,
                          const double *drr, const double *dmm, T *vol, T *pvf)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dkdl = dkk[i] * dll[i];
    vol[i] = (T)fabs(dkdl * dmm[i]);
    pvf[i] = (T)(dkdl * (area[i] / drr[i]));
}

// float32 state: the C ABI speaks float64 (the reference's arrays); convert on the device
template <typename A, typename B>
__global__ void k_convert(long long n, const A *src, B *dst)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (B)src[i];
}

}   // namespace msgw
