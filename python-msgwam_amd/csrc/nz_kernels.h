// EXTENSION (north_star "U(z)/N2(z) column"): buoyancy frequency N as a COLUMN on `grids`, staged
// in LDS next to the shear table and interpolated per ray.  The reference has a scalar bvf only, so the semantics are
// build-defined and pinned to the reference in the limit
// N(z) = const, where every expression reduces to lib/libprop.py's and the results match the reference's goldens.
// With N(z) the vertical group velocity differs at rr +- drr/2 (lib/libprop.py:635-636), so ddrr_st != 0 (:641) and
// drr, dmm evolve (:645): five evolving per-ray slots (dens, rr, drr, mm, dmm).  Like HPROP_GLOBAL = True this is a
// plain kernel of its own -- one launch per RK stage, deposit of the stage's INPUT state, per-workgroup flux rows ->
// k_column -- so that the tuned kernels carry none of it.  float64, HPROP off.
#pragma once
#include "ray_kernels.h"

namespace msgw {

struct NzArgs {
    StageArgs s;                                   // rays, constants, static column tables, flux rows
    double *drr, *dmm;                             // evolving slots 4, 8
    double *q_drr, *q_dmm;                         // their low-storage RK registers (STAGE 3: the tendencies)
    const double *dkdl, *area;                     // dkk*dll and rr_mm_area per ray (:594, :599, :137)
    const double *bvf;                             // [ng-1] N on grids
    int group_reduce;                              // 1: in-kernel reduction of the flux rows (flush_rows_group)
};

template <int STAGE, bool SAT, int RPT = 2>
__global__ void __launch_bounds__(BLOCK, RPT == 1 ? 4 : 2) k_ray_stage_nz(const NzArgs h)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const StageArgs a = h.s;
    constexpr int TILE = (BLOCK * RPT);
    const int ng = a.ng, ni = ng - 2, nc = ng - 1, ncp = ng - 2;
    double4 *s_sh = reinterpret_cast<double4 *>(lds);                     // [ni] {dudz, slope, dvdz, slope}
    double2 *s_rho2 = reinterpret_cast<double2 *>(lds + 4 * ni);          // [nc] {rhobar, slope}
    double *s_xg = lds + 4 * ni + 2 * nc;                                  // [ni] grid[1:-1]
    double *s_gs = s_xg + ni;                                              // [nc] grids
    double2 *s_n2 = reinterpret_cast<double2 *>(s_gs + nc + ((ni + nc) & 1));   // [nc] {N, slope} (16-B aligned)
    double *s_rows = reinterpret_cast<double *>(s_n2 + nc);                // [WAVES][2][ncp]
    int *s_rng = reinterpret_cast<int *>(s_rows + WAVES * 2 * ncp);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    for (int i = tid; i < ni; i += BLOCK) {
        s_xg[i] = a.c.xg[i];
        const bool in = i < ni - 1;
        s_sh[i] = make_double4(a.c.dudz[i], in ? a.c.slu[i] : 0.0, a.c.dvdz[i], in ? a.c.slv[i] : 0.0);
    }
    for (int i = tid; i < nc; i += BLOCK) {
        s_gs[i] = a.c.grids[i];
        s_rho2[i] = make_double2(a.c.rhobar[i], (i < nc - 1) ? a.c.slrho[i] : 0.0);
        const bool in = i < nc - 1;                            // np.interp slope (f[j+1]-f[j])/(x[j+1]-x[j])
        s_n2[i] = make_double2(h.bvf[i], in ? (h.bvf[i + 1] - h.bvf[i]) / (a.c.grids[i + 1] - a.c.grids[i]) : 0.0);
    }
    for (int i = tid; i < WAVES * 2 * ncp; i += BLOCK) s_rows[i] = 0.0;
    __syncthreads();

    const long long start = (long long)blockIdx.x * a.rays_per_block;
    const long long end = min(a.n, start + a.rays_per_block);
    int wmin = INT_MAX, wmax = INT_MIN;
    DepWindow acc;
    acc.clear();
    for (int t = 0; t < a.tiles_per_block; ++t) {
        const long long base = start + (long long)t * TILE;
        if (base >= end) break;
        const long long e0 = base + RPT * tid;
        const unsigned int i0 = (unsigned int)(e0 * 8);
        bool valid[RPT];
        double dens[RPT], rr[RPT], drr[RPT], kk[RPT], ll[RPT], mm[RPT], dmm[RPT], ff[RPT], dkdl[RPT], area[RPT];
        double qd[RPT], qr[RPT], qdr[RPT], qm[RPT], qdm[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            valid[r] = e0 + r < end; area[r] = 1.0;
            qd[r] = 0; qr[r] = 0; qdr[r] = 0; qm[r] = 0; qdm[r] = 0;
        }
        loadv(a.r.dens(), i0, dens); loadv(a.r.rr(), i0, rr); loadv(h.drr, i0, drr); loadv(a.r.kk(), i0, kk);
        loadv(a.r.ll(), i0, ll); loadv(a.r.mm(), i0, mm); loadv(h.dmm, i0, dmm); loadv(a.r.fray(), i0, ff);
        loadv(h.dkdl, i0, dkdl);
        if (SAT) loadv(h.area, i0, area);
        if (STAGE == 1 || STAGE == 2) {
            loadv(a.r.q_rr(), i0, qr); loadv(h.q_drr, i0, qdr); loadv(a.r.q_mm(), i0, qm); loadv(h.q_dmm, i0, qdm);
            if (SAT) loadv(a.r.q_dens(), i0, qd);
        }
        double lo[RPT], up[RPT], pay[2][RPT], vol[RPT];
        double n_dens[RPT], n_rr[RPT], n_drr[RPT], n_mm[RPT], n_dmm[RPT];
        int nlo[RPT], nup[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const double f2 = ff[r] * ff[r];
            lo[r] = rr[r] - .5 * drr[r];                                            // :636, :655
            up[r] = rr[r] + .5 * drr[r];                                            // :635
            // N at the ray centre and at both ends of its extent (np.interp on grids, end values outside)
            const Bracket<double> bc = interp_locate(rr[r], s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
            const Bracket<double> bu = interp_locate(up[r], s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
            const Bracket<double> bd = interp_locate(lo[r], s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
            const double2 tc = s_n2[bc.j], tu = s_n2[bu.j], td = s_n2[bd.j];
            const double N_c = interp_eval(rr[r], bc, tc.x, tc.y);
            const double N_u = interp_eval(up[r], bu, tu.x, tu.y);
            const double N_d = interp_eval(lo[r], bd, td.x, td.y);
            // derivative of that interpolant at the centre: the segment's slope, 0 where np.interp clamps
            const double dNdz = (rr[r] >= a.gs_last || rr[r] < a.gs0) ? 0.0 : tc.y;
            double kh2, m2, vk2, om, cgr, t0, t1, t2, t3, cg_u, cg_d;
            dispersion(kk[r], ll[r], mm[r], f2, N_u * N_u, t0, t1, t2, t3, cg_u);   // :635
            dispersion(kk[r], ll[r], mm[r], f2, N_d * N_d, t0, t1, t2, t3, cg_d);   // :636
            dispersion(kk[r], ll[r], mm[r], f2, N_c * N_c, kh2, m2, vk2, om, cgr);
            const double st_rr = .5 * (cg_d + cg_u);                                // :640
            const double st_drr = cg_u - cg_d;                                      // :641
            const Bracket<double> bk = interp_locate(rr[r], s_xg, ni, a.xg0, a.xg_last, a.xg0, a.inv_dzg);
            const double4 sh = s_sh[bk.j];
            const double gu = interp_eval(rr[r], bk, sh.x, sh.y);                   // :355
            const double gv = interp_eval(rr[r], bk, sh.z, sh.w);                   // :356
            double gradient = kk[r] * gu + ll[r] * gv;                              // :517
            gradient = gradient + N_c * kh2 / om / vk2 * dNdz;                      // refraction by dN/dz (extension)
            const double st_mm = (kk[r] * 0.0 + ll[r] * 0.0) - gradient;            // :519-520 (HPROP off)
            const double st_dmm = dmm[r] / drr[r] * st_drr;                         // :645
            double st_dens = 0.0;
            if (SAT) {                                                              // :647-651 -> :561-615
                const double rr_f = rr[r] + st_rr * a.dt;                           // :591
                const double drr_f = drr[r] + st_drr * a.dt;                        // :592
                const double mm_f = mm[r] + st_mm * a.dt;                           // :593
                const double dmm_f = area[r] / drr_f;                               // :594
                const Bracket<double> br = interp_locate(rr_f, s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
                const double2 rh = s_rho2[br.j], tn = s_n2[br.j];
                const double rho_f = interp_eval(rr_f, br, rh.x, rh.y);             // :595
                const double N_f = interp_eval(rr_f, br, tn.x, tn.y);               // NN at rr_final
                const double omh = sqrt((N_c * N_c * kh2 + a.f0sq * m2) / vk2);     // omega(kk, ll, mm, phi0) (:597)
                const double pv = dkdl[r] * dmm_f;                                  // :599
                const double maxd = sat_cap(a.sat_c, rho_f, omh, N_f * N_f, mm_f, a.f0sq);   // :601
                if (maxd < dens[r] * pv) st_dens = (maxd - dens[r]) / a.dt;         // :604, :613
            }
            deposit_indices<2>(lo[r], up[r], valid[r], a.dzs, a.inv_dzs, a.mk_ok, nc - 2, nlo[r], nup[r]);
            vol[r] = fabs(dkdl[r] * dmm[r]);                                        // :137
            const double mmid = .5 * ((mm[r] - .5 * dmm[r]) + (mm[r] + .5 * dmm[r]));   // :141, :656
            double cgm;
            dispersion(kk[r], ll[r], mmid, f2, N_c * N_c, t0, t1, t2, t3, cgm);    // :139-144
            pay[0][r] = cgm * kk[r] * dens[r];                                      // :148-149
            pay[1][r] = cgm * ll[r] * dens[r];
            if (STAGE == 3) {
                n_dens[r] = st_dens; n_rr[r] = st_rr; n_drr[r] = st_drr; n_mm[r] = st_mm; n_dmm[r] = st_dmm;
            } else {
                const double st[5] = {st_dens, st_rr, st_drr, st_mm, st_dmm};
                const double y[5] = {dens[r], rr[r], drr[r], mm[r], dmm[r]};
                double q[5] = {qd[r], qr[r], qdr[r], qm[r], qdm[r]};
                double yn[5];
#pragma unroll
                for (int v = 0; v < 5; ++v) {                                       // :693-698
                    if (STAGE == 0) { q[v] = a.dt * st[v]; yn[v] = y[v] + div_const(q[v], 3.0, third_rn<double>(), 1); }
                    else if (STAGE == 1) { q[v] = a.dt * st[v] - RK_A1 * q[v]; yn[v] = y[v] + RK_B1 * q[v]; }
                    else { q[v] = a.dt * st[v] - RK_A2 * q[v]; yn[v] = y[v] + RK_B2 * q[v]; }
                }
                n_dens[r] = SAT ? yn[0] : dens[r];
                n_rr[r] = yn[1]; n_drr[r] = yn[2]; n_mm[r] = yn[3]; n_dmm[r] = yn[4];
                qd[r] = q[0]; qr[r] = q[1]; qdr[r] = q[2]; qm[r] = q[3]; qdm[r] = q[4];
            }
        }
        if (valid[0]) {                                        // only the owner stores (pairs never straddle)
            if (STAGE == 3) {
                storev(a.r.q_dens(), i0, n_dens); storev(a.r.q_rr(), i0, n_rr); storev(h.q_drr, i0, n_drr);
                storev(a.r.q_mm(), i0, n_mm); storev(h.q_dmm, i0, n_dmm);
            } else {
                if (SAT) storev(a.r.dens(), i0, n_dens);
                storev(a.r.rr(), i0, n_rr); storev(h.drr, i0, n_drr); storev(a.r.mm(), i0, n_mm); storev(h.dmm, i0, n_dmm);
                if (STAGE != 2) {
                    if (SAT) storev(a.r.q_dens(), i0, qd);
                    storev(a.r.q_rr(), i0, qr); storev(h.q_drr, i0, qdr); storev(a.r.q_mm(), i0, qm); storev(h.q_dmm, i0, qdm);
                }
            }
        }
        deposit_tile<2, double, RPT>(lo, up, nlo, nup, vol, pay, s_gs, a.dzs, a.inv_dzs, a.mk_ok, s_rows + wave * 2 * ncp,
                                   ncp, lane, wmin, wmax, acc);
    }
    flush_window(acc, s_rows + wave * 2 * ncp, ncp, lane);
    // the workgroups' rows are reduced inside the launch (ticketed, fixed order: the per-stage kernel's protocol) to the
    // one row the column kernel reads -- or, for the single-RHS probe, left as sparse rows for k_flux_reduce1
    if (STAGE != 3 && h.group_reduce) flush_rows_group<2, double>(s_rows, ncp, s_rng, lds, tid, a);
    else flush_rows<2>(s_rows, ncp, s_rng, wave, lane, tid, wmin, wmax, a.partial, a.ranges);
}

}   // namespace msgw
