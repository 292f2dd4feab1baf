// Persistent RK3 kernels of ONE ray type (MSGW_REAL = double | float) and ONE flavour (MSGW_NRES = 0: all rays
// streamed, 4 workgroups per CU; 2 / 4: that many register-resident tiles per workgroup, 2 workgroups per CU).
#include <type_traits>
#include "kernel_table.h"
#include "persist_kernel.h"

#if !defined(MSGW_REAL) || !defined(MSGW_NRES)
#error "compile with -DMSGW_REAL=double|float -DMSGW_NRES=0|2|3|4"
#endif

namespace msgw {

typedef MSGW_REAL real_t;

#ifndef MSGW_LEAN
#define MSGW_LEAN 0
#endif
#if MSGW_LEAN
// the lean-LDS-layout variants (float64; two or four resident tiles; no direct saturation)
template <>
const void *persist_kernel_lean_impl<MSGW_NRES>(bool sat, bool fvec, bool direct, bool relaunch)
{
    static_assert(std::is_same<real_t, double>::value && (MSGW_NRES == 2 || MSGW_NRES == 4), "lean: float64, 2 or 4 tiles");
    if (direct) return nullptr;
    return bsel(sat, [&](auto SAT) { return bsel(fvec, [&](auto FVEC) {
        return bsel(relaunch, [&](auto RL) -> const void * {
            return reinterpret_cast<const void *>(
                &k_rk3_persist<real_t, decltype(SAT)::value, decltype(FVEC)::value, false, MSGW_NRES, decltype(RL)::value, true>);
        }); }); });
}
#else
template <>
const void *persist_kernel_impl<real_t, MSGW_NRES>(bool sat, bool fvec, bool direct, bool relaunch)
{
    return bsel(sat, [&](auto SAT) { return bsel(fvec, [&](auto FVEC) { return bsel(direct, [&](auto DIR) {
        return bsel(relaunch, [&](auto RL) -> const void * {
            if constexpr (decltype(SAT)::value && decltype(DIR)::value) return nullptr;
            // four resident tiles: every float32 variant (fully resident); float64 all but the direct-saturation
            // variants (whose resident state also holds rr0, mm0: 49-97 spilled VGPRs), see tile_body.inc
            else if constexpr (MSGW_NRES > 3 && std::is_same<real_t, double>::value && decltype(DIR)::value) return nullptr;
            // three: the float64 direct-saturation variants only (evolving slots incl. rr0, mm0: 32 VGPRs per tile,
            // 0-6 spilled; 53.0 -> 50.8 us per step at 1e6 rays.  For online saturation three measured like four.)
            else if constexpr (MSGW_NRES == 3 && !(std::is_same<real_t, double>::value && decltype(DIR)::value)) return nullptr;
            else return reinterpret_cast<const void *>(
                &k_rk3_persist<real_t, decltype(SAT)::value, decltype(FVEC)::value, decltype(DIR)::value, MSGW_NRES,
                               decltype(RL)::value>);
        }); }); }); });
}

#endif

}   // namespace msgw
