// HPROP_GLOBAL = True (lib/libprop.py:5): horizontal propagation on the sphere.  lam, phi, kk, ll evolve as
// well as rr, mm (dens); drr, dmm stay frozen (ddrr_st = cgr_up - cgr_down = 0 exactly, :641, :645).
// raytracer.py switches this off (:38) and the tuned kernels serve that case; this is the reference's own
// default mode, kept in a kernel of its own so that it costs the tuned ones nothing: one launch per RK stage
// (deposit of the stage's INPUT state, per-workgroup flux rows -> k_column), 2 rays per lane, plain loads.
// Operation order follows the reference line by line; sin/cos/tan come from the device math library, so
// per-ray results agree with numpy to a few ulp, not bit for bit (rtol 1e-10 is asserted).
#pragma once
#include "ray_kernels.h"

namespace msgw {

struct HpropArgs {
    StageArgs s;                                   // rays (kk, ll are written here), constants, tables, flux rows
    double *lam, *phi, *kk, *ll;                   // evolving slots 1, 2, 5, 6
    double *q_lam, *q_phi, *q_kk, *q_ll;           // their low-storage RK registers (STAGE 3: the tendencies)
    const double *uu, *vv;                         // the column on grids, for uu_ray / vv_ray (:357-358)
    double rad_earth, two_rot, df2c;               // RAD_EARTH, 2*ROT_EARTH, 8*ROT_EARTH**2 (:489)
    int group_reduce;                              // 1: in-kernel reduction of the flux rows (flush_rows_group)
};

// (three workgroups per CU -- 168 VGPRs, 20-38 of them spilled in stages 1, 2 -- measured slower: 173.7 vs 161.5 us per step)
#ifndef HPROP_WG_PER_CU
#define HPROP_WG_PER_CU 2
#endif
template <int STAGE, bool SAT, int RPT = 2>
__global__ void __launch_bounds__(BLOCK, RPT == 1 ? 4 : HPROP_WG_PER_CU) k_ray_stage_hprop(const HpropArgs h)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const StageArgs a = h.s;
    const int ng = a.ng, ni = ng - 2, nc = ng - 1, ncp = ng - 2;
    double4 *s_sh = reinterpret_cast<double4 *>(lds);                     // [ni] {dudz, slope, dvdz, slope}
    double2 *s_rho2 = reinterpret_cast<double2 *>(lds + 4 * ni);          // [nc] {rhobar, slope}
    double *s_xg = lds + 4 * ni + 2 * nc;                                  // [ni] grid[1:-1]
    double *s_gs = s_xg + ni;                                              // [nc] grids
    double4 *s_uv = reinterpret_cast<double4 *>(s_gs + nc + ((ni + nc) & 1));   // [nc] {uu, slope, vv, slope} (16-B aligned)
    double *s_rows = reinterpret_cast<double *>(s_uv + nc);                // [WAVES][2][ncp]
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
        const double dx = in ? a.c.grids[i + 1] - a.c.grids[i] : 1.0;
        s_uv[i] = make_double4(h.uu[i], in ? (h.uu[i + 1] - h.uu[i]) / dx : 0.0,
                               h.vv[i], in ? (h.vv[i + 1] - h.vv[i]) / dx : 0.0);
    }
    for (int i = tid; i < WAVES * 2 * ncp; i += BLOCK) s_rows[i] = 0.0;
    __syncthreads();

    const long long start = (long long)blockIdx.x * a.rays_per_block;
    const long long end = min(a.n, start + a.rays_per_block);
    int wmin = INT_MAX, wmax = INT_MIN;
    DepWindow acc;
    acc.clear();
    for (int t = 0; t < a.tiles_per_block; ++t) {
        const long long base = start + (long long)t * (BLOCK * RPT);
        if (base >= end) break;
        const long long e0 = base + RPT * tid;
        const unsigned int i0 = (unsigned int)(e0 * 8);
        bool valid[RPT];
        double dens[RPT], lam[RPT], phi[RPT], rr[RPT], drr[RPT], kk[RPT], ll[RPT], mm[RPT], vol[RPT], pvf[RPT];
        double qd[RPT], qla[RPT], qph[RPT], qr[RPT], qk[RPT], ql[RPT], qm[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            valid[r] = e0 + r < end; pvf[r] = 1.0;
            qd[r] = 0; qla[r] = 0; qph[r] = 0; qr[r] = 0; qk[r] = 0; ql[r] = 0; qm[r] = 0;
        }
        loadv(a.r.dens(), i0, dens); loadv(h.lam, i0, lam); loadv(h.phi, i0, phi); loadv(a.r.rr(), i0, rr);
        loadv(a.r.drr(), i0, drr); loadv(h.kk, i0, kk); loadv(h.ll, i0, ll); loadv(a.r.mm(), i0, mm);
        loadv(a.r.vol(), i0, vol);
        if (SAT) loadv(a.r.pvf(), i0, pvf);
        // (the RK registers are loaded after the physics; the kernel is latency-bound)
        double lo[RPT], up[RPT], pay[2][RPT];
        double n_dens[RPT], n_lam[RPT], n_phi[RPT], n_rr[RPT], n_kk[RPT], n_ll[RPT], n_mm[RPT];
        double tend[7][RPT];
        int nlo[RPT], nup[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            double sinp, cosp;
            sincos(phi[r], &sinp, &cosp);                      // one argument reduction for both
            const double tanp = sinp / cosp;                   // (np.tan to ~1 ulp: this path is held to rtol 1e-10)
            const double f = h.two_rot * sinp;                                      // :382
            const double f2 = f * f;
            double kh2, m2, vk2, om, cgr;
            dispersion(kk[r], ll[r], mm[r], f2, a.bvf2, kh2, m2, vk2, om, cgr);   // :369-383, :434-448
            const Bracket<double> bk = interp_locate(rr[r], s_xg, ni, a.xg0, a.xg_last, a.xg0, a.inv_dzg);
            const double4 sh = s_sh[bk.j];
            const double gu = interp_eval(rr[r], bk, sh.x, sh.y);                   // du/dz at the ray (:355)
            const double gv = interp_eval(rr[r], bk, sh.z, sh.w);                   // dv/dz at the ray (:356)
            const Bracket<double> bu = interp_locate(rr[r], s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
            const double4 uv = s_uv[bu.j];
            const double uu_ray = interp_eval(rr[r], bu, uv.x, uv.y);               // :357
            const double vv_ray = interp_eval(rr[r], bu, uv.z, uv.w);               // :358
            const double disp = a.bvf2 - om * om;
            const double cg_lam = kk[r] / om / vk2 * disp + uu_ray;                 // :404
            const double cg_ph = ll[r] / om / vk2 * disp + vv_ray;                  // :428
            const double R = h.rad_earth + rr[r];
            const double st_lam = cg_lam / R / cosp;                                // :638
            const double st_phi = cg_ph / R;                                        // :639
            const double st_rr = .5 * (cgr + cgr);                                  // :640
            const double zero_grad = kk[r] * 0.0 + ll[r] * 0.0;                     // no horizontal wind gradients (:360-364)
            const double st_kk = kk[r] / R * (tanp * cg_ph - cgr) - zero_grad / R / cosp;      // :463-468
            const double df2 = h.df2c * sinp * cosp * 1;                            // :489
            const double st_ll = -(ll[r] * cgr + kk[r] * tanp * cg_lam + m2 / 2 / om / vk2 * df2) / R
                                 - zero_grad / R;                                   // :487-496
            const double st_mm = (kk[r] * cg_lam + ll[r] * cg_ph) / R - (kk[r] * gu + ll[r] * gv);   // :517-520
            double st_dens = 0.0;
            if (SAT) {                                                              // :647-651 -> :561-615
                const double rr_f = rr[r] + st_rr * a.dt;
                const double mm_f = mm[r] + st_mm * a.dt;
                const Bracket<double> br = interp_locate(rr_f, s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
                const double2 rh = s_rho2[br.j];
                const double rho_f = interp_eval(rr_f, br, rh.x, rh.y);
                const double omh = sqrt((a.bvf2 * kh2 + a.f0sq * m2) / vk2);        // omega(kk, ll, mm, phi0) (:597)
                const double maxd = sat_cap(a.sat_c, rho_f, omh, a.bvf2, mm_f, a.f0sq);
                if (maxd < dens[r] * pvf[r]) st_dens = (maxd - dens[r]) / a.dt;
            }
            lo[r] = rr[r] - .5 * drr[r];                                            // :655
            up[r] = rr[r] + .5 * drr[r];
            deposit_indices<2>(lo[r], up[r], valid[r], a.dzs, a.inv_dzs, a.mk_ok, nc - 2, nlo[r], nup[r]);
            pay[0][r] = cgr * kk[r] * dens[r];                                      // :148-149
            pay[1][r] = cgr * ll[r] * dens[r];
            tend[0][r] = st_dens; tend[1][r] = st_lam; tend[2][r] = st_phi; tend[3][r] = st_rr;
            tend[4][r] = st_kk; tend[5][r] = st_ll; tend[6][r] = st_mm;
        }
        asm volatile("" ::: "memory");                         // keep the loads below from being hoisted over the physics
        if (STAGE == 1 || STAGE == 2) {
            loadv(h.q_lam, i0, qla); loadv(h.q_phi, i0, qph); loadv(a.r.q_rr(), i0, qr); loadv(h.q_kk, i0, qk);
            loadv(h.q_ll, i0, ql); loadv(a.r.q_mm(), i0, qm);
            if (SAT) loadv(a.r.q_dens(), i0, qd);
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (STAGE == 3) {
                n_dens[r] = tend[0][r]; n_lam[r] = tend[1][r]; n_phi[r] = tend[2][r]; n_rr[r] = tend[3][r];
                n_kk[r] = tend[4][r]; n_ll[r] = tend[5][r]; n_mm[r] = tend[6][r];
            } else {
                const double y[7] = {dens[r], lam[r], phi[r], rr[r], kk[r], ll[r], mm[r]};
                double q[7] = {qd[r], qla[r], qph[r], qr[r], qk[r], ql[r], qm[r]};
                double yn[7];
#pragma unroll
                for (int v = 0; v < 7; ++v) {                                       // :693-698
                    if (STAGE == 0) { q[v] = a.dt * tend[v][r]; yn[v] = y[v] + div_const(q[v], 3.0, third_rn<double>(), 1); }
                    else if (STAGE == 1) { q[v] = a.dt * tend[v][r] - RK_A1 * q[v]; yn[v] = y[v] + RK_B1 * q[v]; }
                    else { q[v] = a.dt * tend[v][r] - RK_A2 * q[v]; yn[v] = y[v] + RK_B2 * q[v]; }
                }
                n_dens[r] = SAT ? yn[0] : dens[r];
                n_lam[r] = yn[1]; n_phi[r] = yn[2]; n_rr[r] = yn[3]; n_kk[r] = yn[4]; n_ll[r] = yn[5]; n_mm[r] = yn[6];
                qd[r] = q[0]; qla[r] = q[1]; qph[r] = q[2]; qr[r] = q[3]; qk[r] = q[4]; ql[r] = q[5]; qm[r] = q[6];
            }
        }
        if (valid[0]) {                                        // only the owner stores (pairs never straddle)
            if (STAGE == 3) {
                storev(a.r.q_dens(), i0, n_dens); storev(h.q_lam, i0, n_lam); storev(h.q_phi, i0, n_phi);
                storev(a.r.q_rr(), i0, n_rr); storev(h.q_kk, i0, n_kk); storev(h.q_ll, i0, n_ll);
                storev(a.r.q_mm(), i0, n_mm);
            } else {
                if (SAT) storev(a.r.dens(), i0, n_dens);
                storev(h.lam, i0, n_lam); storev(h.phi, i0, n_phi); storev(a.r.rr(), i0, n_rr);
                storev(h.kk, i0, n_kk); storev(h.ll, i0, n_ll); storev(a.r.mm(), i0, n_mm);
                if (STAGE != 2) {
                    if (SAT) storev(a.r.q_dens(), i0, qd);
                    storev(h.q_lam, i0, qla); storev(h.q_phi, i0, qph); storev(a.r.q_rr(), i0, qr);
                    storev(h.q_kk, i0, qk); storev(h.q_ll, i0, ql); storev(a.r.q_mm(), i0, qm);
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
