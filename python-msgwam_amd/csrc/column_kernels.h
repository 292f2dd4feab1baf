// K2: flux finalise + mean-flow update, one workgroup (the column has
// O(100..1000) levels).  Deterministic: per-workgroup flux rows are summed in
// a fixed (segment, workgroup) order, independent of scheduling.
#pragma once
#include "column_math.h"
#include "ray_kernels.h"

namespace msgw {

struct ColArgs {
    int ng;               // interfaces
    int nblocks;          // rows in `partial`
    int nseg;             // row segments summed in parallel
    int npay;             // payload rows per workgroup row (2 for the RHS, 1 or 2 for diagnostics)
    int ncp;              // levels per payload row
    int fixed_background;
    double dt, f0, dzg;   // f0 = 2*Omega*sin(phi0) (:535); dzg = grid[1]-grid[0] (:349, :662)
    const double *partial;
    const int *ranges;
    double *flux;         // [npay][ncp] reduced (this rank's) flux profile
    const double *rhobar, *pg;
    ColIn in;             // column read  (may alias `out`: each level is read, then written, by one thread)
    ColOut out;           // column written
    const double *xg;
    double *dudz, *dvdz, *slu, *slv;
    double *out_du, *out_dv, *out_flux;   // probe outputs (device)
};

constexpr int COL_REDUCE = 1;   // sum the workgroup rows into `flux`
constexpr int COL_UPDATE = 2;   // mean-flow tendencies / RK update / shear columns from `flux`

// STAGE 0,1,2: RK stage of slots 9, 10 (lib/libprop.py:693-698 applied to uu, vv)
// STAGE 3    : probe -- du_dt, dv_dt and pm_flux written out, nothing updated
// STAGE 4    : derive dudz/dvdz/slopes from the current uu, vv only
template <int STAGE, int MODE>
__global__ void __launch_bounds__(COL_BLOCK) k_column(const ColArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ng = a.ng, nc = ng - 1, ni = ng - 2;
    const int ncols = a.npay * a.ncp;
    double *s_F = lds;                      // [2][ng]
    double *s_u = s_F + 2 * ng;             // [nc]
    double *s_v = s_u + nc;                 // [nc]
    double *s_du = s_v + nc;                // [ni]
    double *s_dv = s_du + ni;               // [ni]
    double *s_seg = s_dv + ni;              // [nseg][ncols]
    int *s_rng = reinterpret_cast<int *>(s_seg + (size_t)a.nseg * ncols);   // [2*nblocks]
    const int tid = threadIdx.x;

    if (MODE & COL_REDUCE) {
        // second-level rows (written by k_flux_reduce1) are dense: no range descriptors
        for (int i = tid; i < 2 * a.nblocks; i += COL_BLOCK)
            s_rng[i] = a.ranges ? a.ranges[i] : ((i & 1) ? INT_MAX : INT_MIN);
        __syncthreads();
        for (int idx = tid; idx < a.nseg * ncols; idx += COL_BLOCK) {
            const int seg = idx / ncols, col = idx - seg * ncols;
            const int p = col / a.ncp, c = col - p * a.ncp;
            const int b0 = (int)((long long)seg * a.nblocks / a.nseg);
            const int b1 = (int)((long long)(seg + 1) * a.nblocks / a.nseg);
            const double *src = a.partial + (size_t)p * a.ncp + c;
            const size_t stride = (size_t)a.npay * a.ncp;
            double acc = 0.0;
            // batches of 8 unconditional loads (row index clamped, value masked): one memory
            // latency per batch instead of one per row; fixed add order
            for (int b = b0; b < b1; b += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(size_t)min(b + u, b1 - 1) * stride];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int bb = min(b + u, b1 - 1);
                    const bool in = (b + u < b1) && (c >= s_rng[2 * bb]) && (c < s_rng[2 * bb + 1]);
                    acc = acc + (in ? v[u] : 0.0);
                }
            }
            s_seg[idx] = acc;
        }
        __syncthreads();
        for (int col = tid; col < ncols; col += COL_BLOCK) {
            double tot = s_seg[col];
            for (int s = 1; s < a.nseg; ++s) tot = tot + s_seg[s * ncols + col];
            a.flux[col] = tot;
            if ((MODE & COL_UPDATE) && STAGE != 4) {        // pm_flux[:, 1:-1] straight into LDS (:654)
                const int p = col / a.ncp, c = col - p * a.ncp;
                s_F[p * ng + 1 + c] = tot;
            }
        }
        if (!(MODE & COL_UPDATE)) return;
        __syncthreads();
    }

    if (STAGE != 4) {
        // pm_flux on interfaces, lib/libprop.py:653-660 (flux has ng-2 levels per component)
        if (!(MODE & COL_REDUCE)) {
            for (int i = tid; i < 2 * (ng - 2); i += COL_BLOCK) {
                const int p = i / (ng - 2), c = i - p * (ng - 2);
                s_F[p * ng + 1 + c] = a.flux[i];
            }
            __syncthreads();
        }
        column_flux_ends(tid, ng, s_F);
        __syncthreads();
        if (STAGE == 3) {                                   // probe: tendencies + flux out, no update
            if (a.out_flux)
                for (int i = tid; i < 2 * ng; i += COL_BLOCK) a.out_flux[i] = s_F[i];
            for (int j = tid; j < nc; j += COL_BLOCK) {
                double du, dv;
                column_tendency(j, ng, a.f0, a.dzg, a.fixed_background, s_F, a.rhobar[j], a.pg[j],
                                a.pg[nc + j], a.in.uu[j], a.in.vv[j], du, dv);
                if (a.out_du) a.out_du[j] = du;
                if (a.out_dv) a.out_dv[j] = dv;
            }
            return;
        }
        column_stage_all(STAGE, tid, COL_BLOCK, ng, a.dt, a.f0, a.dzg, a.fixed_background, s_F,
                         a.rhobar, a.pg, a.in, a.out, true, s_u, s_v);
    } else {
        for (int j = tid; j < nc; j += COL_BLOCK) { s_u[j] = a.in.uu[j]; s_v[j] = a.in.vv[j]; }
    }
    __syncthreads();
    // shear on the interior interfaces, lib/libprop.py:352-353, and np.interp's slopes
    column_shear(tid, COL_BLOCK, ng, a.dzg, s_u, s_v, s_du, s_dv);
    __syncthreads();
    for (int j = tid; j < ni; j += COL_BLOCK) {
        a.dudz[j] = s_du[j]; a.dvdz[j] = s_dv[j];
        if (j < ni - 1) { a.slu[j] = column_slope(s_du, a.xg, j); a.slv[j] = column_slope(s_dv, a.xg, j); }
    }
}

// First level of the flux reduction when there are many workgroup rows: G1
// workgroups each sum a contiguous slice of rows (fixed order) into one dense
// row of `out` [G1][ncols].  Keeps the single-workgroup k_column off the
// one-CU bandwidth limit (977 rows x 198 columns = 1.5 MB at 1e6 rays).
struct Red1Args {
    int nblocks, ncols, ncp, npay, nseg;
    const double *partial;
    const int *ranges;
    double *out;
};
}   // namespace msgw
