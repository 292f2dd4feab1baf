// Ray-stage kernels of ONE ray type (MSGW_REAL = double | float) and ONE RK stage (MSGW_STAGE = 0..3;
// 3 = the single-RHS probe, which also carries the small per-type kernels).  See kernel_table.h.
#include <type_traits>
#include "kernel_table.h"
#include "ray_kernels.h"

#if !defined(MSGW_REAL) || !defined(MSGW_STAGE)
#error "compile with -DMSGW_REAL=double|float -DMSGW_STAGE=0..3"
#endif

namespace msgw {

typedef MSGW_REAL real_t;
#define KPTR(...) reinterpret_cast<const void *>(&__VA_ARGS__)

template <typename T, int STAGE, bool SAT, bool FVEC, bool DEPOSIT, bool DIRECT, bool RL>
static const void *pick_form(int form)
{
    constexpr bool GR = DEPOSIT && STAGE != 3;
    switch (form) {
    case FORM_LAG:
        if constexpr (GR) return KPTR(k_ray_stage<T, STAGE, SAT, FVEC, DEPOSIT, DIRECT, true, true, RL>);
        else return nullptr;
    case FORM_GROUP:
        if constexpr (GR) return KPTR(k_ray_stage<T, STAGE, SAT, FVEC, DEPOSIT, DIRECT, true, false, RL>);
        else return nullptr;
    case FORM_TALL:              // tall columns take the same kernel (the level window is relative)
    case FORM_PLAIN: return KPTR(k_ray_stage<T, STAGE, SAT, FVEC, DEPOSIT, DIRECT, false, false, RL>);
    }
    return nullptr;
}

template <>
const void *stage_kernel_impl<real_t, MSGW_STAGE>(bool sat, bool fvec, bool deposit, bool direct, int form, bool relaunch)
{
    constexpr int S = MSGW_STAGE;
    return bsel(sat, [&](auto SAT) { return bsel(fvec, [&](auto FVEC) { return bsel(deposit, [&](auto DEP) {
        return bsel(direct, [&](auto DIR) { return bsel(relaunch, [&](auto RL) -> const void * {
            // what the host ever asks for: online and direct saturation exclude each other; the direct form only
            // touches stages 0 (keeps rr, mm) and 2; the relaunch extension lives in stage 2; only the probe runs
            // without a deposit, and it has neither the direct form nor the relaunch
            constexpr bool ok = !(decltype(SAT)::value && decltype(DIR)::value) &&
                                (!decltype(DIR)::value || S == 0 || S == 2) && (!decltype(RL)::value || S == 2) &&
                                (decltype(DEP)::value || S == 3);
            if constexpr (ok)
                return pick_form<real_t, S, decltype(SAT)::value, decltype(FVEC)::value, decltype(DEP)::value,
                                 decltype(DIR)::value, decltype(RL)::value>(form);
            else
                return nullptr;
        }); }); }); }); });
}

#if MSGW_STAGE == 3
template <>
const void *fixed_kernel<real_t>(bool sat, bool fvec, bool direct, bool narrow)
{
    return bsel(sat, [&](auto SAT) { return bsel(fvec, [&](auto FVEC) { return bsel(direct, [&](auto DIR) {
        return bsel(narrow, [&](auto NAR) -> const void * {
            if constexpr (decltype(SAT)::value && decltype(DIR)::value) return nullptr;
            else return KPTR(k_ray_step_fixed<real_t, decltype(SAT)::value, decltype(FVEC)::value, decltype(DIR)::value,
                                              decltype(NAR)::value>);
        }); }); }); });
}
template <>
const void *deposit_only_kernel<real_t>(bool fvec)
{
    return fvec ? KPTR(k_deposit_only<real_t, true>) : KPTR(k_deposit_only<real_t, false>);
}
template <>
const void *project_kernel<real_t>(int np, bool fvec)
{
    if (np == 2) return fvec ? KPTR(k_project<real_t, 2, true, false>) : KPTR(k_project<real_t, 2, false, false>);
    return fvec ? KPTR(k_project<real_t, 1, true, false>) : KPTR(k_project<real_t, 1, false, false>);
}
template <> const void *fill_kernel<real_t>() { return KPTR(k_fill_range<real_t>); }
template <> const void *prepare_kernel<real_t>() { return KPTR(k_prepare<real_t>); }
#endif

}   // namespace msgw
