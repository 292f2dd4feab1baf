// The general per-stage ray kernel (chain_kernels.h) of ONE ray type (MSGW_REAL = double | float) and ONE mode
// (MSGW_HPROP, MSGW_NZ = 0 | 1, not both 0): four RK stages (3 = the single-RHS probe) x online saturation off / on.
// See kernel_table.h.
#include "kernel_table.h"
#include "chain_kernels.h"

#if !defined(MSGW_REAL) || !defined(MSGW_HPROP) || !defined(MSGW_NZ)
#error "compile with -DMSGW_REAL=double|float -DMSGW_HPROP=0|1 -DMSGW_NZ=0|1"
#endif

namespace msgw {

typedef MSGW_REAL real_t;
#define KPTR(...) reinterpret_cast<const void *>(&__VA_ARGS__)

template <>
const void *chain_kernel_impl<real_t, MSGW_HPROP != 0, MSGW_NZ != 0>(int stage, bool sat)
{
    constexpr bool H = MSGW_HPROP != 0, N = MSGW_NZ != 0;
    switch (stage) {
    case 0: return sat ? KPTR(k_ray_stage_chain<real_t, 0, true, H, N>) : KPTR(k_ray_stage_chain<real_t, 0, false, H, N>);
    case 1: return sat ? KPTR(k_ray_stage_chain<real_t, 1, true, H, N>) : KPTR(k_ray_stage_chain<real_t, 1, false, H, N>);
    case 2: return sat ? KPTR(k_ray_stage_chain<real_t, 2, true, H, N>) : KPTR(k_ray_stage_chain<real_t, 2, false, H, N>);
    case 3: return sat ? KPTR(k_ray_stage_chain<real_t, 3, true, H, N>) : KPTR(k_ray_stage_chain<real_t, 3, false, H, N>);
    }
    return nullptr;
}

}   // namespace msgw
