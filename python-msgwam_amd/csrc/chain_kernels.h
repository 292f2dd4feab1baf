// The GENERAL per-stage ray kernel: every evolving slot of the reference's state vector.
//   HPROP  HPROP_GLOBAL = True (lib/libprop.py:5): horizontal propagation on the sphere.  lam, phi, kk, ll evolve as
//          well as rr, mm (dens); raytracer.py switches this off (:38) and the tuned kernels serve that case.
//   NZ     EXTENSION (north_star "U(z)/N2(z) column"): buoyancy frequency N as a COLUMN on `grids`,
//          interpolated per ray.  The reference has a scalar bvf only, so the semantics are build-defined
//          and pinned to the reference in the limit N(z) = const.  With N(z) the vertical group
//          velocity differs at rr +- drr/2 (:635-636), so ddrr_st != 0 (:641) and drr, dmm evolve (:645).
// Both at once (all nine per-ray slots evolve) is the union of the two.
// One launch per RK stage -- deposit of the stage's INPUT state, per-workgroup flux rows -> k_column -- 2 rays per lane,
// plain loads: kept apart from the tuned kernels so that it costs them nothing.  On top of the reference's online
// saturation (SAT, :647-651) the kernel of stage 2 can apply the driver's direct saturation (raytracer.py:182-188;
// `direct`, the start-of-step rr, mm [, drr] are kept by stage 0) and the MSGW_RELAUNCH extension (`s.relaunch`): both
// wave-uniform run-time branches, these kernels are bound by the bytes they stream.
// T = double: operation order follows the reference line by line; sin/cos/tan come from the device math library, so
// per-ray results agree with numpy to a few ulp, not bit for bit (rtol 1e-10 is asserted).  T = float: the
// throughput mode's arithmetic (real.h), held to float64 results at float32 tolerance by the tests.
#pragma once
#include "ray_kernels.h"

namespace msgw {

template <typename T>
struct ChainArgsT {
    StageArgsT<T> s;                               // rays, constants, static column tables, flux rows
    // HPROP: evolving slots 1, 2, 5, 6 (kk, ll are the slab's arrays, written here) and their RK registers
    T *lam, *phi, *kk, *ll;
    T *q_lam, *q_phi, *q_kk, *q_ll;                // (STAGE 3: the tendencies)
    const double *uu, *vv;                         // the column on grids, for uu_ray / vv_ray (:357-358)
    T rad_earth, two_rot, df2c;                    // RAD_EARTH, 2*ROT_EARTH, 8*ROT_EARTH**2 (:489)
    // N(z): evolving slots 4, 8 and their RK registers
    T *drr, *dmm, *q_drr, *q_dmm;
    T *drr0;                                       // start-of-step drr (direct saturation with an N(z) column)
    const T *dkdl, *area;                          // dkk*dll and rr_mm_area per ray (:594, :599, :137)
    const double *bvf;                             // [ng-1] N on grids
    int group_reduce;                              // 1: in-kernel reduction of the flux rows (flush_rows_group)
    int direct;                                    // 1: direct saturation after stage 2 (s.sat_rr_div: the driver's quirk)
};

__device__ __forceinline__ void sincos_(double x, double &s, double &c) { sincos(x, &s, &c); }
__device__ __forceinline__ void sincos_(float x, float &s, float &c) { sincosf(x, &s, &c); }

// bytes of dynamic LDS the kernel carves (host side: the same expression)
template <typename T>
__host__ __device__ inline size_t chain_lds_bytes(int ng, bool hprop, bool nz)
{
    const size_t ni = ng - 2, nc = ng - 1, ncp = ng - 2;
    size_t b = ni * 4 * sizeof(T) + (hprop ? nc * 4 * sizeof(T) : 0) + nc * 2 * sizeof(T) + (nz ? nc * 2 * sizeof(T) : 0) +
               (ni + nc) * sizeof(T);
    b = (b + 15) & ~(size_t)15;
    return b + WAVES * 2 * ncp * sizeof(double) + 16 * sizeof(int);
}

template <typename T, int STAGE, bool SAT, bool HPROP, bool NZ>
__global__ void __launch_bounds__(BLOCK, 2) k_ray_stage_chain(const ChainArgsT<T> h)
{
    typedef typename Real<T>::quad_t quad_t;
    typedef typename Real<T>::pair_t pair_t;
    constexpr int RPT = 2;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const StageArgsT<T> a = h.s;
    const int ng = a.ng, ni = ng - 2, nc = ng - 1, ncp = ng - 2;
    char *cur = reinterpret_cast<char *>(lds);
    quad_t *s_sh = reinterpret_cast<quad_t *>(cur); cur += (size_t)ni * sizeof(quad_t);      // [ni] {dudz, slope, dvdz, slope}
    quad_t *s_uv = reinterpret_cast<quad_t *>(cur); if (HPROP) cur += (size_t)nc * sizeof(quad_t);   // [nc] {uu, slope, vv, slope}
    pair_t *s_rho2 = reinterpret_cast<pair_t *>(cur); cur += (size_t)nc * sizeof(pair_t);    // [nc] {rhobar, slope}
    pair_t *s_n2 = reinterpret_cast<pair_t *>(cur); if (NZ) cur += (size_t)nc * sizeof(pair_t);      // [nc] {N, slope}
    T *s_xg = reinterpret_cast<T *>(cur); cur += (size_t)ni * sizeof(T);                     // [ni] grid[1:-1]
    T *s_gs = reinterpret_cast<T *>(cur); cur += (size_t)nc * sizeof(T);                     // [nc] grids
    cur = reinterpret_cast<char *>(lds) + ((cur - reinterpret_cast<char *>(lds) + 15) & ~(ptrdiff_t)15);
    double *s_rows = reinterpret_cast<double *>(cur);                                       // [WAVES][2][ncp]
    int *s_rng = reinterpret_cast<int *>(s_rows + WAVES * 2 * ncp);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    for (int i = tid; i < ni; i += BLOCK) {
        s_xg[i] = (T)a.c.xg[i];
        const bool in = i < ni - 1;
        s_sh[i] = Real<T>::quad((T)a.c.dudz[i], in ? (T)a.c.slu[i] : T(0), (T)a.c.dvdz[i], in ? (T)a.c.slv[i] : T(0));
    }
    for (int i = tid; i < nc; i += BLOCK) {
        s_gs[i] = (T)a.c.grids[i];
        const bool in = i < nc - 1;
        s_rho2[i] = Real<T>::pair((T)a.c.rhobar[i], in ? (T)a.c.slrho[i] : T(0));
        const double dx = in ? a.c.grids[i + 1] - a.c.grids[i] : 1.0;      // np.interp slope (f[j+1]-f[j])/(x[j+1]-x[j])
        if (HPROP)
            s_uv[i] = Real<T>::quad((T)h.uu[i], in ? (T)((h.uu[i + 1] - h.uu[i]) / dx) : T(0),
                                    (T)h.vv[i], in ? (T)((h.vv[i + 1] - h.vv[i]) / dx) : T(0));
        if (NZ) s_n2[i] = Real<T>::pair((T)h.bvf[i], in ? (T)((h.bvf[i + 1] - h.bvf[i]) / dx) : T(0));
    }
    for (int i = tid; i < WAVES * 2 * ncp; i += BLOCK) s_rows[i] = 0.0;
    __syncthreads();

    const bool direct = STAGE != 3 && !SAT && h.direct != 0;           // raytracer.py:182: only if not online
    const bool relaunch = STAGE == 2 && a.relaunch != 0;
    const long long start = (long long)blockIdx.x * a.rays_per_block;
    const long long end = min(a.n, start + a.rays_per_block);
    int wmin = INT_MAX, wmax = INT_MIN;
    for (int t = 0; t < a.tiles_per_block; ++t) {
        const long long base = start + (long long)t * (BLOCK * RPT);
        if (base >= end) break;
        const long long e0 = base + RPT * tid;
        const unsigned int off = (unsigned int)(e0 * (long long)sizeof(T));   // byte offset shared by all arrays
        bool valid[RPT];
        // the nine per-ray slots (evolving ones depend on the mode), the static per-ray factors and the RK registers
        T dens[RPT], lam[RPT], phi[RPT], rr[RPT], drr[RPT], kk[RPT], ll[RPT], mm[RPT], dmm[RPT];
        T vol[RPT], pvf[RPT], ff[RPT], dkdl[RPT], area[RPT];
        T q[9][RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            valid[r] = e0 + r < end; pvf[r] = T(1); area[r] = T(1); dkdl[r] = T(1); ff[r] = T(0); lam[r] = T(0); phi[r] = T(0);
            dmm[r] = T(1);
#pragma unroll
            for (int v = 0; v < 9; ++v) q[v][r] = T(0);
        }
        loadv(a.r.dens(), off, dens); loadv(a.r.rr(), off, rr); loadv(a.r.kk(), off, kk); loadv(a.r.ll(), off, ll);
        loadv(a.r.mm(), off, mm);
        if (HPROP) { loadv(h.lam, off, lam); loadv(h.phi, off, phi); }
        else loadv(a.r.fray(), off, ff);
        if (NZ) {
            loadv(h.drr, off, drr); loadv(h.dmm, off, dmm); loadv(h.dkdl, off, dkdl);
            if (SAT || direct) loadv(h.area, off, area);
        } else {
            loadv(a.r.drr(), off, drr); loadv(a.r.vol(), off, vol);
            if (SAT || direct) loadv(a.r.pvf(), off, pvf);
        }
        T tend[9][RPT];                                        // dens, lam, phi, rr, drr, kk, ll, mm, dmm
        T lo[RPT], up[RPT], pay[2][RPT], kh2s[RPT], Ncs[RPT];
        int nlo[RPT], nup[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            T sinp = T(0), cosp = T(1), tanp = T(0), f;
            if (HPROP) {
                sincos_(phi[r], sinp, cosp);                   // one argument reduction for both
                tanp = div_(sinp, cosp);                       // (np.tan to ~1 ulp: this path is held to rtol 1e-10)
                f = h.two_rot * sinp;                                               // :382
            } else f = ff[r];
            const T f2 = f * f;
            lo[r] = rr[r] - T(.5) * drr[r];                                         // :636, :655
            up[r] = rr[r] + T(.5) * drr[r];                                         // :635
            T kh2, m2, vk2, om, cgr, cg_u, cg_d, N_c = T(0), dNdz = T(0), bvf2c = a.bvf2;
            if (NZ) {
                // N at the ray centre and at both ends of its extent (np.interp on grids, end values outside)
                const Bracket<T> bc = interp_locate(rr[r], s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
                const Bracket<T> bu = interp_locate(up[r], s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
                const Bracket<T> bd = interp_locate(lo[r], s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
                const pair_t tc = s_n2[bc.j], tu = s_n2[bu.j], td = s_n2[bd.j];
                N_c = interp_eval(rr[r], bc, tc.x, tc.y);
                const T N_u = interp_eval(up[r], bu, tu.x, tu.y);
                const T N_d = interp_eval(lo[r], bd, td.x, td.y);
                // derivative of that interpolant at the centre: the segment's slope, 0 where np.interp clamps
                dNdz = (rr[r] >= a.gs_last || rr[r] < a.gs0) ? T(0) : tc.y;
                T t0, t1, t2, t3;
                dispersion(kk[r], ll[r], mm[r], f2, N_u * N_u, t0, t1, t2, t3, cg_u);   // :635
                dispersion(kk[r], ll[r], mm[r], f2, N_d * N_d, t0, t1, t2, t3, cg_d);   // :636
                bvf2c = N_c * N_c;
            }
            dispersion(kk[r], ll[r], mm[r], f2, bvf2c, kh2, m2, vk2, om, cgr);       // :369-383, :434-448
            if (!NZ) { cg_u = cgr; cg_d = cgr; }
            kh2s[r] = kh2; Ncs[r] = N_c;
            const T st_rr = NZ ? T(.5) * (cg_d + cg_u) : cgr;                       // :640 (scalar bvf: exactly cgr)
            const T st_drr = NZ ? cg_u - cg_d : T(0);                               // :641
            const Bracket<T> bk = interp_locate(rr[r], s_xg, ni, a.xg0, a.xg_last, a.xg0, a.inv_dzg);
            const quad_t sh = s_sh[bk.j];
            const T gu = interp_eval(rr[r], bk, sh.x, sh.y);                        // du/dz at the ray (:355)
            const T gv = interp_eval(rr[r], bk, sh.z, sh.w);                        // dv/dz at the ray (:356)
            T gradient = kk[r] * gu + ll[r] * gv;                                   // :517
            if (NZ) gradient = gradient + div_(div_(N_c * kh2, om), vk2) * dNdz;    // refraction by dN/dz (extension)
            T st_lam = T(0), st_phi = T(0), st_kk = T(0), st_ll = T(0), st_mm;
            if (HPROP) {
                const Bracket<T> bu = interp_locate(rr[r], s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
                const quad_t uv = s_uv[bu.j];
                const T uu_ray = interp_eval(rr[r], bu, uv.x, uv.y);                // :357
                const T vv_ray = interp_eval(rr[r], bu, uv.z, uv.w);                // :358
                const T disp = bvf2c - om * om;
                const T cg_lam = div_(div_(kk[r], om), vk2) * disp + uu_ray;        // :404
                const T cg_ph = div_(div_(ll[r], om), vk2) * disp + vv_ray;         // :428
                const T R = h.rad_earth + rr[r];
                st_lam = div_(div_(cg_lam, R), cosp);                               // :638
                st_phi = div_(cg_ph, R);                                            // :639
                const T zero_grad = kk[r] * T(0) + ll[r] * T(0);                    // no horizontal wind gradients (:360-364)
                st_kk = div_(kk[r], R) * (tanp * cg_ph - cgr) - div_(div_(zero_grad, R), cosp);      // :463-468
                const T df2 = h.df2c * sinp * cosp * T(1);                          // :489
                st_ll = -div_(ll[r] * cgr + kk[r] * tanp * cg_lam + div_(div_(div_(m2, T(2)), om), vk2) * df2, R)
                        - div_(zero_grad, R);                                       // :487-496
                st_mm = div_(kk[r] * cg_lam + ll[r] * cg_ph, R) - gradient;         // :517-520
            } else {
                st_mm = (kk[r] * T(0) + ll[r] * T(0)) - gradient;                   // :519-520 (HPROP off)
            }
            const T st_dmm = NZ ? div_(dmm[r], drr[r]) * st_drr : T(0);             // :645
            T st_dens = T(0);
            if (SAT) {                                                              // :647-651 -> :561-615
                const T rr_f = rr[r] + st_rr * a.dt;                                // :591
                const T mm_f = mm[r] + st_mm * a.dt;                                // :593
                const Bracket<T> br = interp_locate(rr_f, s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
                const pair_t rh = s_rho2[br.j];
                const T rho_f = interp_eval(rr_f, br, rh.x, rh.y);                  // :595
                const T omh = sqrt_(div_(bvf2c * kh2 + a.f0sq * m2, vk2));          // omega(kk, ll, mm, phi0) (:597)
                T pv = pvf[r], nf2 = a.bvf2;
                if (NZ) {
                    const T drr_f = drr[r] + st_drr * a.dt;                         // :592
                    pv = dkdl[r] * div_(area[r], drr_f);                            // :594, :599
                    const pair_t tn = s_n2[br.j];
                    const T N_f = interp_eval(rr_f, br, tn.x, tn.y);                // NN at rr_final
                    nf2 = N_f * N_f;
                }
                const T maxd = sat_cap(a.sat_c, rho_f, omh, nf2, mm_f, a.f0sq);     // :601
                if (maxd < dens[r] * pv) st_dens = div_(maxd - dens[r], a.dt);      // :604, :613
            }
            deposit_indices<2>(lo[r], up[r], valid[r], a.dzs, a.inv_dzs, a.mk_ok, nc - 2, nlo[r], nup[r]);
            T cgm = cgr;                                                            // (HPROP alone: the stage's own cg_rr,
            if (NZ) {                                                               //  equal to within 1 ulp)
                vol[r] = fabs(dkdl[r] * dmm[r]);                                    // :137
                const T mmid = T(.5) * ((mm[r] - T(.5) * dmm[r]) + (mm[r] + T(.5) * dmm[r]));   // :141, :656
                T t0, t1, t2, t3;
                dispersion(kk[r], ll[r], mmid, f2, bvf2c, t0, t1, t2, t3, cgm);     // :139-144
            }
            pay[0][r] = cgm * kk[r] * dens[r];                                      // :148-149
            pay[1][r] = cgm * ll[r] * dens[r];
            tend[0][r] = st_dens; tend[1][r] = st_lam; tend[2][r] = st_phi; tend[3][r] = st_rr; tend[4][r] = st_drr;
            tend[5][r] = st_kk; tend[6][r] = st_ll; tend[7][r] = st_mm; tend[8][r] = st_dmm;
        }
        asm volatile("" ::: "memory");                         // keep the loads below from being hoisted over the physics
        if (STAGE == 1 || STAGE == 2) {
            loadv(a.r.q_rr(), off, q[3]); loadv(a.r.q_mm(), off, q[7]);
            if (SAT) loadv(a.r.q_dens(), off, q[0]);
            if (HPROP) { loadv(h.q_lam, off, q[1]); loadv(h.q_phi, off, q[2]); loadv(h.q_kk, off, q[5]); loadv(h.q_ll, off, q[6]); }
            if (NZ) { loadv(h.q_drr, off, q[4]); loadv(h.q_dmm, off, q[8]); }
        }
        T y[9][RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            y[0][r] = dens[r]; y[1][r] = lam[r]; y[2][r] = phi[r]; y[3][r] = rr[r]; y[4][r] = drr[r];
            y[5][r] = kk[r]; y[6][r] = ll[r]; y[7][r] = mm[r]; y[8][r] = dmm[r];
        }
        // slots that evolve in this mode (the others keep their values; their tendencies are exact zeros)
        constexpr bool EV[9] = {SAT, HPROP, HPROP, true, NZ, HPROP, HPROP, true, NZ};
        if (direct && STAGE == 0 && valid[0]) {                // keep the start-of-step rr, mm [, drr]
            storev(a.r.rr0(), off, rr); storev(a.r.mm0(), off, mm);
            if (NZ) storev(h.drr0, off, drr);
        }
#pragma unroll
        for (int v = 0; v < 9; ++v) {
            if (!EV[v]) continue;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                if (STAGE == 3) { y[v][r] = tend[v][r]; continue; }
                T qq;                                                               // :693-698
                if (STAGE == 0) { qq = a.dt * tend[v][r]; y[v][r] = y[v][r] + div_const(qq, T(3), third_rn<T>(), 1); }
                else if (STAGE == 1) { qq = a.dt * tend[v][r] - T(RK_A1) * q[v][r]; y[v][r] = y[v][r] + T(RK_B1) * qq; }
                else { qq = a.dt * tend[v][r] - T(RK_A2) * q[v][r]; y[v][r] = y[v][r] + T(RK_B2) * qq; }
                q[v][r] = qq;
            }
        }
        bool dens_out = SAT;
        if (STAGE == 2 && direct) {                            // raytracer.py:182-188 -> :561-610, on the NEW kk, ll
            T rr0[RPT], mm0[RPT], drr0[RPT];
            loadv(a.r.rr0(), off, rr0); loadv(a.r.mm0(), off, mm0);
            if (NZ) loadv(h.drr0, off, drr0);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const T rr_st = div_(y[3][r] - rr0[r], a.sat_rr_div);               // raytracer.py:184
                const T mm_st = div_(y[7][r] - mm0[r], a.dt);                       // raytracer.py:187
                const T rr_f = rr0[r] + rr_st * a.dt;
                const T mm_f = mm0[r] + mm_st * a.dt;
                const Bracket<T> br = interp_locate(rr_f, s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
                const pair_t rh = s_rho2[br.j];
                const T rho_f = interp_eval(rr_f, br, rh.x, rh.y);                  // :595
                const T kh2 = y[5][r] * y[5][r] + y[6][r] * y[6][r];
                const T m02 = mm0[r] * mm0[r];
                T nc2 = a.bvf2, nf2 = a.bvf2, pv = pvf[r];
                if (NZ) {
                    const Bracket<T> b0 = interp_locate(rr0[r], s_gs, nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
                    const pair_t t0 = s_n2[b0.j], tf = s_n2[br.j];
                    const T N_0 = interp_eval(rr0[r], b0, t0.x, t0.y);              // N at rr_center
                    const T N_f = interp_eval(rr_f, br, tf.x, tf.y);                // N at rr_final
                    nc2 = N_0 * N_0; nf2 = N_f * N_f;
                    const T drr_st = div_(y[4][r] - drr0[r], a.dt);                 // raytracer.py:185
                    pv = dkdl[r] * div_(area[r], drr0[r] + drr_st * a.dt);          // :592, :594, :599
                }
                const T omh = sqrt_(div_(nc2 * kh2 + a.f0sq * m02, kh2 + m02));     // :597 (old mm)
                const T maxd = sat_cap(a.sat_c, rho_f, omh, nf2, mm_f, a.f0sq);
                y[0][r] = (maxd < y[0][r] * pv) ? maxd : y[0][r];                   // :604-608
            }
            dens_out = true;
        }
        if (relaunch) {                                        // EXTENSION MSGW_RELAUNCH (include/msgwam_hip.h)
            T sd[RPT], sr[RPT], sm[RPT];
            loadv(a.r.src_dens(), off, sd); loadv(a.r.src_rr(), off, sr); loadv(a.r.src_mm(), off, sm);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const bool out = (y[3][r] - T(.5) * y[4][r] > a.z_top) || (y[3][r] + T(.5) * y[4][r] < a.z_bot) ||
                                 (y[0][r] < a.relaunch_frac * sd[r]);
                y[0][r] = out ? sd[r] : y[0][r];
                y[3][r] = out ? sr[r] : y[3][r];
                y[7][r] = out ? sm[r] : y[7][r];
            }
            dens_out = true;
        }
        if (valid[0]) {                                        // only the owner stores (pairs never straddle)
            T *const yp[9] = {a.r.dens(), h.lam, h.phi, a.r.rr(), h.drr, h.kk, h.ll, a.r.mm(), h.dmm};
            T *const qp[9] = {a.r.q_dens(), h.q_lam, h.q_phi, a.r.q_rr(), h.q_drr, h.q_kk, h.q_ll, a.r.q_mm(), h.q_dmm};
            if (STAGE == 3) {
                storev(qp[0], off, tend[0]);                   // (the probe always reports dens_st: zeros without SAT)
#pragma unroll
                for (int v = 1; v < 9; ++v) if (EV[v]) storev(qp[v], off, y[v]);
            } else {
                if (dens_out) storev(yp[0], off, y[0]);
#pragma unroll
                for (int v = 1; v < 9; ++v) if (EV[v]) storev(yp[v], off, y[v]);
                if (STAGE != 2) {
#pragma unroll
                    for (int v = 0; v < 9; ++v) if (EV[v]) storev(qp[v], off, q[v]);
                }
            }
        }
        deposit_tile<2, T, RPT>(lo, up, nlo, nup, vol, pay, s_gs, a.dzs, a.inv_dzs, a.mk_ok, s_rows + wave * 2 * ncp,
                                ncp, lane, wmin, wmax);
    }
    // the workgroups' rows are reduced inside the launch (ticketed, fixed order: the per-stage kernel's protocol) to the
    // one row the column kernel reads -- or, for the single-RHS probe, left as sparse rows for k_flux_reduce1
    if (STAGE != 3 && h.group_reduce) flush_rows_group<2, T>(s_rows, ncp, s_rng, lds, tid, a);
    else flush_rows<2>(s_rows, ncp, s_rng, wave, lane, tid, wmin, wmax, a.partial, a.ranges);
}

}   // namespace msgw
