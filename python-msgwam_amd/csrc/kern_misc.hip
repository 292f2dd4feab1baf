// The float64-only kernels: standalone column kernel, projections and
// saturation on caller arrays, first-level flux reduction, exchange self-test, float32 <-> float64 conversion.
#include <type_traits>
#include "kernel_table.h"
#include "column_kernels.h"
#include "misc_kernels.h"

namespace msgw {

#define KPTR(...) reinterpret_cast<const void *>(&__VA_ARGS__)

template <int STAGE>
static const void *column_mode(int mode)
{
    switch (mode) {
    case COL_REDUCE: return KPTR(k_column<STAGE, COL_REDUCE>);
    case COL_UPDATE: return KPTR(k_column<STAGE, COL_UPDATE>);
    case COL_REDUCE | COL_UPDATE: return KPTR(k_column<STAGE, COL_REDUCE | COL_UPDATE>);
    }
    return nullptr;
}
const void *column_kernel(int stage, int mode)
{
    switch (stage) {
    case 0: return column_mode<0>(mode);
    case 1: return column_mode<1>(mode);
    case 2: return column_mode<2>(mode);
    case 3: return column_mode<3>(mode);
    case 4: return column_mode<4>(mode);
    }
    return nullptr;
}
template <> const void *nz_prepare_kernel<double>() { return KPTR(k_nz_prepare<double>); }
template <> const void *nz_prepare_kernel<float>() { return KPTR(k_nz_prepare<float>); }
const void *project_arrays_kernel(int np)
{
    return np == 2 ? KPTR(k_project<double, 2, true, true>) : KPTR(k_project<double, 1, true, true>);
}
const void *saturation_kernel() { return KPTR(k_saturation); }
const void *flux_reduce1_kernel() { return KPTR(k_flux_reduce1); }
const void *rho_slopes_kernel() { return KPTR(k_rho_slopes); }
const void *xch_selftest_kernel() { return KPTR(k_xch_selftest); }
const void *probe_arith_kernel() { return KPTR(k_probe_arith); }
const void *convert_kernel_d2f() { return KPTR(k_convert<double, float>); }
const void *convert_kernel_f2d() { return KPTR(k_convert<float, double>); }

}   // namespace msgw
