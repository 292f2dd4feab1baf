// Mean-flow arithmetic shared by the standalone column kernel (k_column) and the
// prologue of the ray-stage kernel (which applies the previous stage's pending
// mean-flow update in LDS while its ray loads are in flight).  One definition,
// so both paths round identically.  Reference: lib/libprop.py:653-666, :523-558,
// :693-698 (RK stage of slots 9, 10), :349-353 (shear), np.interp slopes.
#pragma once
#include <hip/hip_runtime.h>

namespace msgw {

// Williamson RK3 as python evaluates it (lib/libprop.py:693-698)
constexpr double RK_A1 = 5.0 / 9.0;
constexpr double RK_A2 = 153.0 / 128.0;
constexpr double RK_B1 = 15.0 / 16.0;
constexpr double RK_B2 = 8.0 / 15.0;

struct ColIn { const double *uu, *vv, *q_uu, *q_vv; };
struct ColOut { double *uu, *vv, *q_uu, *q_vv; };

// pm_flux end points (:659-660).  s_F is [2][ng] with pm_flux[:, 1:-1] already at [p*ng + 1 + c].
// Call with a barrier before (s_F filled) and after.
__device__ __forceinline__ void column_flux_ends(int tid, int ng, double *s_F)
{
    if (tid < 2) { s_F[tid * ng] = s_F[tid * ng + 1]; s_F[tid * ng + ng - 1] = s_F[tid * ng + ng - 2]; }
}

// du_dt, dv_dt (:523-558) of level j from the interface flux.
__device__ __forceinline__ void column_tendency(int j, int ng, double f0, double dzg, int fixed_bg,
                                                const double *s_F, double rho, double pg0,
                                                double pg1, double u, double v,
                                                double &du, double &dv)
{
    const double gx = (s_F[j + 1] - s_F[j]) / dzg;                       // :663
    const double gy = (s_F[ng + j + 1] - s_F[ng + j]) / dzg;
    const double rinv = 1.0 / rho;                                       // rhobar**-1
    du = f0 * v - rinv * (pg0 + gx);                                     // :537
    dv = -f0 * u - rinv * (pg1 + gy);                                    // :556
    if (fixed_bg) { du = 0.0; dv = 0.0; }
}

// One RK stage (0, 1, 2) of the winds of level j; q is the low-storage register.
__device__ __forceinline__ void column_rk(int stage, double dt, double du, double dv, double u,
                                          double v, double qu_old, double qv_old, double &un,
                                          double &vn, double &qu, double &qv)
{
    if (stage == 0) {
        qu = dt * du; qv = dt * dv;
        un = u + qu / 3; vn = v + qv / 3;
    } else if (stage == 1) {
        qu = dt * du - RK_A1 * qu_old; qv = dt * dv - RK_A1 * qv_old;
        un = u + RK_B1 * qu; vn = v + RK_B1 * qv;
    } else {
        qu = dt * du - RK_A2 * qu_old; qv = dt * dv - RK_A2 * qv_old;
        un = u + RK_B2 * qu; vn = v + RK_B2 * qv;
    }
}

// Full stage for all levels: reads the old column from `in`, leaves the new winds in
// s_u, s_v (LDS) and, if `write`, stores the new column to `out`.  Barrier after.
__device__ __forceinline__ void column_stage_all(int stage, int tid, int nthr, int ng, double dt,
                                                 double f0, double dzg, int fixed_bg,
                                                 const double *s_F, const double *rhobar,
                                                 const double *pg, const ColIn &in,
                                                 const ColOut &out, bool write, double *s_u,
                                                 double *s_v)
{
    const int nc = ng - 1;
    for (int j = tid; j < nc; j += nthr) {
        const double u = in.uu[j], v = in.vv[j];
        double du, dv, un, vn, qu, qv;
        column_tendency(j, ng, f0, dzg, fixed_bg, s_F, rhobar[j], pg[j], pg[nc + j], u, v, du, dv);
        const double quo = (stage == 0) ? 0.0 : in.q_uu[j], qvo = (stage == 0) ? 0.0 : in.q_vv[j];
        column_rk(stage, dt, du, dv, u, v, quo, qvo, un, vn, qu, qv);
        s_u[j] = un; s_v[j] = vn;
        if (write) { out.uu[j] = un; out.vv[j] = vn; out.q_uu[j] = qu; out.q_vv[j] = qv; }
    }
}

// Shear on the interior interfaces (:352-353) from the winds in LDS.  Barrier after.
__device__ __forceinline__ void column_shear(int tid, int nthr, int ng, double dzg, const double *s_u,
                                             const double *s_v, double *s_du, double *s_dv)
{
    const int ni = ng - 2;
    for (int j = tid; j < ni; j += nthr) {
        s_du[j] = (s_u[j + 1] - s_u[j]) / dzg;
        s_dv[j] = (s_v[j + 1] - s_v[j]) / dzg;
    }
}

// np.interp slope between table points j and j+1
__device__ __forceinline__ double column_slope(const double *s_f, const double *xg, int j)
{
    return (s_f[j + 1] - s_f[j]) / (xg[j + 1] - xg[j]);
}

}   // namespace msgw
