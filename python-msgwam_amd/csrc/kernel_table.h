// Look-up of kernel entry points across translation units.  The device code is built as several
// objects (one per ray type x RK stage, one per ray type x persistent-kernel flavour, one for the
// float64-only kernels) so that it compiles in parallel; the host side (msgwam_hip.hip) gets the
// entry point of a variant as an opaque pointer and launches it with hipLaunchKernel /
// hipExtLaunchKernel.  nullptr = that combination is not built (the host reports MSGW_ERR_UNSUP).
#pragma once
#include <type_traits>

namespace msgw {

// `form` of a ray-stage kernel
constexpr int FORM_TALL = 0;     // columns with more than 128 levels: sparse rows (the same kernel as FORM_PLAIN)
constexpr int FORM_LAG = 1;      // lagged launch chain: deposits the state it produces, in-kernel group reduction
constexpr int FORM_GROUP = 2;    // fused chain: deposits its input state, in-kernel group reduction
constexpr int FORM_PLAIN = 3;    // sparse per-workgroup rows (probe, tall-column chain)

// k_ray_stage<T, STAGE, ...>; one explicit specialisation per (T, STAGE) lives in kern_stage.hip
template <typename T, int STAGE>
const void *stage_kernel_impl(bool sat, bool fvec, bool deposit, bool direct, int form, bool relaunch);

template <typename T>
inline const void *stage_kernel(int stage, bool sat, bool fvec, bool deposit, bool direct, int form, bool relaunch)
{
    switch (stage) {
    case 0: return stage_kernel_impl<T, 0>(sat, fvec, deposit, direct, form, relaunch);
    case 1: return stage_kernel_impl<T, 1>(sat, fvec, deposit, direct, form, relaunch);
    case 2: return stage_kernel_impl<T, 2>(sat, fvec, deposit, direct, form, relaunch);
    case 3: return stage_kernel_impl<T, 3>(sat, fvec, deposit, direct, form, relaunch);
    }
    return nullptr;
}

// defined next to the STAGE = 3 (probe) kernels of each ray type
template <typename T> const void *fixed_kernel(bool sat, bool fvec, bool direct, bool narrow = false);   // k_ray_step_fixed
template <typename T> const void *deposit_only_kernel(bool fvec);                    // k_deposit_only
template <typename T> const void *project_kernel(int np, bool fvec);                 // k_project on resident rays
template <typename T> const void *fill_kernel();                                     // k_fill_range
template <typename T> const void *prepare_kernel();                                  // k_prepare
const void *convert_kernel_d2f();                                                    // k_convert<double, float>
const void *convert_kernel_f2d();                                                    // k_convert<float, double>

// k_rk3_persist<T, ...>; one explicit specialisation per (T, NRES) lives in kern_persist.hip
template <typename T, int NRES>
const void *persist_kernel_impl(bool sat, bool fvec, bool direct, bool relaunch);
// the LEAN LDS layout (persist_kernel.h): float64, two or four resident tiles
template <int NRES>
const void *persist_kernel_lean_impl(bool sat, bool fvec, bool direct, bool relaunch);
template <typename T>
inline const void *persist_kernel(bool sat, bool fvec, bool direct, int nres, bool relaunch, bool lean = false)
{
    if (lean) {
        if (!std::is_same<T, double>::value) return nullptr;
        return nres >= 4 ? persist_kernel_lean_impl<4>(sat, fvec, direct, relaunch)
             : nres == 2 ? persist_kernel_lean_impl<2>(sat, fvec, direct, relaunch) : nullptr;
    }
    return nres >= 4 ? persist_kernel_impl<T, 4>(sat, fvec, direct, relaunch)
         : nres == 3 ? persist_kernel_impl<T, 3>(sat, fvec, direct, relaunch)
         : nres > 0 ? persist_kernel_impl<T, 2>(sat, fvec, direct, relaunch)
                    : persist_kernel_impl<T, 0>(sat, fvec, direct, relaunch);
}

// k_ray_stage_chain<T, STAGE, SAT, HPROP, NZ> (HPROP_GLOBAL = True and / or the N(z) column extension); one explicit
// specialisation per (T, HPROP, NZ) lives in kern_chain.hip
template <typename T, bool HPROP, bool NZ>
const void *chain_kernel_impl(int stage, bool sat);
template <typename T>
inline const void *chain_kernel(int stage, bool sat, bool hprop, bool nz)
{
    return hprop ? (nz ? chain_kernel_impl<T, true, true>(stage, sat) : chain_kernel_impl<T, true, false>(stage, sat))
                 : (nz ? chain_kernel_impl<T, false, true>(stage, sat) : nullptr);
}
template <typename T> const void *nz_prepare_kernel();                               // k_nz_prepare<T>

// float64-only kernels (kern_misc.hip)
const void *column_kernel(int stage, int mode);          // k_column<STAGE, MODE>
const void *project_arrays_kernel(int np);               // k_project<double, NP, true, true>
const void *saturation_kernel();
const void *flux_reduce1_kernel();
const void *rho_slopes_kernel();
const void *xch_selftest_kernel();
const void *probe_arith_kernel();                         // k_probe_arith

// compile-time bool dispatch used by the look-up functions
template <class F>
inline const void *bsel(bool b, F f)
{
    return b ? f(std::true_type{}) : f(std::false_type{});
}

}   // namespace msgw
