/*
 * ORACLE -- test infrastructure, NOT the product.
 *
 * Plain-C, single-thread restatement of python-msgwam's ray-propagation hot
 * path (HPROP_GLOBAL = False branch; scalar bvf, plus the build-defined N(z)
 * column extension at the end of the file), written from the reference's
 * algorithm, operation by operation in numpy's evaluation order, so that with
 * -ffp-contract=off it is bit-comparable with the reference.  It exists so the
 * GPU parity tests can check 1e5..1e6-ray cases in seconds.  It is validated
 * against oracle/msgwam_oracle.py and the golden vectors produced from the real
 * reference (tests/test_oracle_c.py).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.
 *
 * Citations are file:line in the reference (lib/libprop.py unless noted).
 *
 * Build:  make -C oracle        (gcc -O2 -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RAD_EARTH 6378e3 /* :3 */

/* ---- numpy's np.interp (numpy/_core/src/multiarray/compiled_base.c), the
 * third-party arithmetic behind :355-356 and :595.  Piecewise linear, end
 * values outside [xp[0], xp[n-1]], exact fp[j] when x == xp[j]. ---- */
static double np_interp1(double x, const double *xp, const double *fp, int n)
{
    if (isnan(x)) return x;
    if (x > xp[n - 1]) return fp[n - 1];
    if (x < xp[0]) return fp[0];
    int lo = 0, hi = n - 1;             /* find j: xp[j] <= x < xp[j+1] */
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (x >= xp[mid]) lo = mid; else hi = mid;
    }
    int j = (x >= xp[n - 1]) ? n - 1 : lo;
    if (j == n - 1) return fp[j];
    if (xp[j] == x) return fp[j];
    double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    double r = slope * (x - xp[j]) + fp[j];
    if (isnan(r)) {
        r = slope * (x - xp[j + 1]) + fp[j + 1];
        if (isnan(r) && fp[j] == fp[j + 1]) r = fp[j];
    }
    return r;
}

/* numpy `.astype(int)` of a float64 on x86-64 (cvttsd2si): truncation toward
 * zero; NaN, inf and out-of-range give INT64_MIN (:124-125). */
static int64_t np_astype_int(double t)
{
    if (!(t > -9223372036854775808.0 && t < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)t;
}

/* :369-383 with ff = 2*Omega*sin(phi) supplied by the caller (computed in
 * numpy so that sin() is bit-identical). */
static double omega_(double kk, double ll, double mm, double ff, double bvf)
{
    return sqrt((bvf * bvf * (kk * kk + ll * ll) + ff * ff * (mm * mm))
                / (kk * kk + ll * ll + mm * mm));
}

/* :434-448 */
static double cg_rr_(double kk, double ll, double mm, double ff, double bvf)
{
    double vk_square = kk * kk + ll * ll + mm * mm;
    double om = omega_(kk, ll, mm, ff, bvf);
    return -mm * (om * om - ff * ff) / om / vk_square;
}

typedef struct {
    int ngrid;              /* interfaces */
    const double *grid;     /* [ngrid]     */
    const double *grids;    /* [ngrid-1]   */
    const double *rhobar;   /* [ngrid-1]   */
    const double *pg;       /* [2][ngrid-1] */
    double bvf, f0, kappa;  /* f0 = 2*Omega*sin(phi0) */
    int saturate_online;
    int fixed_background;
} orc_setup;

/* :92-197 for one ray, var 0/1/2; adds into out[npay][ncell]. */
static void project_ray(double dens, double lo, double up, double kk, double ll,
                        double mm_mid, double ff, double vol_abs, double bvf,
                        const double *G, int nG, int var, double *out)
{
    double dz = G[1] - G[0];                                   /* :123 */
    int64_t nlow = np_astype_int(lo / dz);                     /* :124 */
    int64_t nup = np_astype_int(up / dz + 1.);                 /* :125 */
    int64_t nzmax = nG - 2;                                    /* :127 */
    if ((nlow >= nzmax && nup >= nzmax) || (nlow <= 0 && nup <= 0)) return; /* :129-135 */
    if (nlow < 0) nlow = 0;
    if (nlow >= nzmax) nlow = nzmax;
    if (nup < 0) nup = 0;
    if (nup >= nzmax) nup = nzmax;
    double cgr = cg_rr_(kk, ll, mm_mid, ff, bvf);              /* :139-144 */
    double v0, v1 = 0.0;
    if (var == 0) { v0 = cgr * kk * dens; v1 = cgr * ll * dens; }   /* :148-149 */
    else if (var == 1) v0 = cgr * dens;                              /* :167 */
    else v0 = dens;                                                   /* :184 */
    int ncell = nG - 1;
    for (int64_t c = nlow; c < nup; ++c) {                     /* :156 */
        double zmin = (G[c] > lo || isnan(G[c])) ? G[c] : lo;  /* np.max of 2 (:157) */
        if (isnan(lo)) zmin = lo;
        double zmax = (G[c + 1] < up || isnan(G[c + 1])) ? G[c + 1] : up; /* np.min (:158) */
        if (isnan(up)) zmax = up;
        double w = fabs(zmax - zmin) / dz;                     /* :160 */
        out[c] += w * vol_abs * v0;                            /* :162 */
        if (var == 0) out[ncell + c] += w * vol_abs * v1;      /* :163 */
    }
}

/* wave_projection(var) over n rays (ray order = reference order).
 * out: [2][nG-1] for var 0, [nG-1] otherwise; zeroed here. */
void orc_project(int64_t n, const double *dens, const double *rr, const double *drr,
                 const double *kk, const double *ll, const double *mm, const double *dmm,
                 const double *fray, const double *dkk, const double *dll,
                 const double *G, int nG, double bvf, int var, double *out)
{
    int ncell = nG - 1;
    memset(out, 0, sizeof(double) * (size_t)ncell * (var == 0 ? 2 : 1));
    for (int64_t i = 0; i < n; ++i) {
        double lo = rr[i] - .5 * drr[i], up = rr[i] + .5 * drr[i];        /* :655 */
        double mlo = mm[i] - .5 * dmm[i], mup = mm[i] + .5 * dmm[i];      /* :656 */
        double vol = fabs(dkk[i] * dll[i] * dmm[i]);                      /* :137 */
        project_ray(dens[i], lo, up, kk[i], ll[i], .5 * (mlo + mup), fray[i], vol, bvf,
                    G, nG, var, out);
    }
}

/* One evaluation of rhs_default (:618-676), fast-path slots only:
 * st_dens, st_rr (= drr_st), st_mm (= dmm_st) per ray; du, dv per level;
 * pm_flux [2][ngrid] (optional, may be NULL). */
void orc_rhs(const orc_setup *s, double dt, int64_t n,
             const double *dens, const double *rr, const double *drr, const double *kk,
             const double *ll, const double *mm, const double *dmm, const double *fray,
             const double *dkk, const double *dll, const double *area,
             const double *uu, const double *vv,
             double *st_dens, double *st_rr, double *st_mm, double *du, double *dv,
             double *pm_flux_out)
{
    int ng = s->ngrid, nc = ng - 1, ni = ng - 2;     /* levels, interior interfaces */
    double dz = s->grid[1] - s->grid[0];             /* :349 */
    double *dudz = (double *)malloc(sizeof(double) * (size_t)ni * 2);
    double *dvdz = dudz + ni;
    for (int j = 0; j < ni; ++j) {                   /* :352-353 */
        dudz[j] = (uu[j + 1] - uu[j]) / dz;
        dvdz[j] = (vv[j + 1] - vv[j]) / dz;
    }
    double *P = (double *)calloc((size_t)2 * (nc - 1), sizeof(double));   /* (2, len(grids)-1) */
    double *F = (double *)calloc((size_t)2 * ng, sizeof(double));         /* :653 */
    const double *xp = s->grid + 1;                  /* grid[1:-1] */
    for (int64_t i = 0; i < n; ++i) {
        double ff = fray[i];
        double cgr = cg_rr_(kk[i], ll[i], mm[i], ff, s->bvf);             /* :635-636 */
        double drr_st = .5 * (cgr + cgr);                                  /* :640 */
        double ddrr_st = cgr - cgr;                                        /* :641 */
        double gu = np_interp1(rr[i], xp, dudz, ni);                       /* :355 */
        double gv = np_interp1(rr[i], xp, dvdz, ni);                       /* :356 */
        double gradient = kk[i] * gu + ll[i] * gv;                         /* :517 */
        double dmm_st = (kk[i] * 0.0 + ll[i] * 0.0) / (RAD_EARTH + rr[i]) - gradient; /* :519-520 */
        st_rr[i] = drr_st;
        st_mm[i] = dmm_st;
        /* :647-651, saturation(...) :561-615 in tendency form */
        double rr_final = rr[i] + drr_st * dt;                             /* :591 */
        double drr_final = drr[i] + ddrr_st * dt;                          /* :592 */
        double mm_final = mm[i] + dmm_st * dt;                             /* :593 */
        double dmm_final = area[i] / drr_final;                            /* :594 */
        double rho_f = np_interp1(rr_final, s->grids, s->rhobar, nc);      /* :595 */
        double omh = omega_(kk[i], ll[i], mm[i], s->f0, s->bvf);           /* :597 (phi0!) */
        double pv = dkk[i] * dll[i] * dmm_final;                           /* :599 */
        double maxd = s->kappa * s->kappa * .5 * rho_f * omh * (s->bvf * s->bvf)
                      / (mm_final * mm_final) / (omh * omh - s->f0 * s->f0); /* :601 */
        double dst = 0.0;
        if (maxd < dens[i] * pv) dst = (maxd - dens[i]) / dt;              /* :604, :613 */
        st_dens[i] = (double)s->saturate_online * dst;                     /* :647 */
        /* :654-658 deposit on G = grids */
        double lo = rr[i] - .5 * drr[i], up = rr[i] + .5 * drr[i];
        double mlo = mm[i] - .5 * dmm[i], mup = mm[i] + .5 * dmm[i];
        double vol = fabs(dkk[i] * dll[i] * dmm[i]);
        project_ray(dens[i], lo, up, kk[i], ll[i], .5 * (mlo + mup), ff, vol, s->bvf,
                    s->grids, nc, 0, P);
    }
    int np_ = nc - 1;                                /* = ngrid-2 */
    for (int c = 0; c < 2; ++c) {
        for (int j = 0; j < np_; ++j) F[c * ng + 1 + j] = P[c * np_ + j];  /* :654 */
        F[c * ng + 0] = F[c * ng + 1];                                      /* :659 */
        F[c * ng + ng - 1] = F[c * ng + ng - 2];                            /* :660 */
    }
    for (int j = 0; j < nc; ++j) {
        double gx = (F[j + 1] - F[j]) / dz;                                 /* :663 */
        double gy = (F[ng + j + 1] - F[ng + j]) / dz;
        double rinv = 1.0 / s->rhobar[j];                                   /* rhobar**-1 -> np.reciprocal */
        du[j] = s->f0 * vv[j] - rinv * (s->pg[j] + gx);                     /* :537 */
        dv[j] = -s->f0 * uu[j] - rinv * (s->pg[nc + j] + gy);               /* :556 */
        if (s->fixed_background) { du[j] = 0.0; dv[j] = 0.0; }
    }
    if (pm_flux_out) memcpy(pm_flux_out, F, sizeof(double) * (size_t)2 * ng);
    free(dudz); free(P); free(F);
}

/* direct saturation (:606-610) as called by the driver (raytracer.py:183-188).
 * rr_div = 1.0 reproduces the driver's `/ 1` quirk, rr_div = dt the intent. */
static void saturate_direct(const orc_setup *s, double dt, double rr_div, int64_t n,
                            double *dens, const double *rr_old, const double *rr_new,
                            const double *drr, const double *kk, const double *ll,
                            const double *mm_old, const double *mm_new,
                            const double *dkk, const double *dll, const double *area)
{
    int nc = s->ngrid - 1;
    for (int64_t i = 0; i < n; ++i) {
        double rr_st = (rr_new[i] - rr_old[i]) / rr_div;       /* raytracer.py:184 */
        double drr_st = (drr[i] - drr[i]) / dt;                /* :185 (drr never moves) */
        double mm_st = (mm_new[i] - mm_old[i]) / dt;           /* :187 */
        double rr_final = rr_old[i] + rr_st * dt;
        double drr_final = drr[i] + drr_st * dt;
        double mm_final = mm_old[i] + mm_st * dt;
        double dmm_final = area[i] / drr_final;
        double rho_f = np_interp1(rr_final, s->grids, s->rhobar, nc);
        double omh = omega_(kk[i], ll[i], mm_old[i], s->f0, s->bvf);
        double pv = dkk[i] * dll[i] * dmm_final;
        double maxd = s->kappa * s->kappa * .5 * rho_f * omh * (s->bvf * s->bvf)
                      / (mm_final * mm_final) / (omh * omh - s->f0 * s->f0);
        if (maxd < dens[i] * pv) dens[i] = maxd;
    }
}

/* nsteps of RK3 (:680-700), dens/rr/mm/uu/vv advanced in place.
 * direct_sat: 0 none, 1 driver's post-step saturation with the `/1` quirk,
 * 2 the same with `/dt`.  Applied only when !saturate_online, as the driver. */
int orc_step(const orc_setup *s, double dt, int nsteps, int direct_sat, int64_t n,
             double *dens, double *rr, const double *drr, const double *kk, const double *ll,
             double *mm, const double *dmm, const double *fray,
             const double *dkk, const double *dll, const double *area,
             double *uu, double *vv)
{
    int nc = s->ngrid - 1;
    size_t nb = sizeof(double) * (size_t)(n ? n : 1);
    double *sd = malloc(nb), *sr = malloc(nb), *sm = malloc(nb);
    double *qd = malloc(nb), *qr = malloc(nb), *qm = malloc(nb);
    double *ro = malloc(nb), *mo = malloc(nb);
    double *du = malloc(sizeof(double) * nc * 4), *dv = du + nc, *qu = du + 2 * nc, *qv = du + 3 * nc;
    if (!sd || !sr || !sm || !qd || !qr || !qm || !ro || !mo || !du) return -1;
    const double A[3] = {0.0, 5.0 / 9.0, 153.0 / 128.0};
    const double B23[3] = {0.0, 15.0 / 16.0, 8.0 / 15.0};
    for (int it = 0; it < nsteps; ++it) {
        memcpy(ro, rr, sizeof(double) * (size_t)n);
        memcpy(mo, mm, sizeof(double) * (size_t)n);
        for (int st = 0; st < 3; ++st) {
            orc_rhs(s, dt, n, dens, rr, drr, kk, ll, mm, dmm, fray, dkk, dll, area, uu, vv,
                    sd, sr, sm, du, dv, NULL);
            for (int64_t i = 0; i < n; ++i) {
                if (st == 0) {                          /* :693-694 */
                    qd[i] = dt * sd[i]; qr[i] = dt * sr[i]; qm[i] = dt * sm[i];
                    dens[i] = dens[i] + qd[i] / 3; rr[i] = rr[i] + qr[i] / 3; mm[i] = mm[i] + qm[i] / 3;
                } else {                                /* :695-698 */
                    qd[i] = dt * sd[i] - A[st] * qd[i];
                    qr[i] = dt * sr[i] - A[st] * qr[i];
                    qm[i] = dt * sm[i] - A[st] * qm[i];
                    dens[i] = dens[i] + B23[st] * qd[i];
                    rr[i] = rr[i] + B23[st] * qr[i];
                    mm[i] = mm[i] + B23[st] * qm[i];
                }
            }
            for (int j = 0; j < nc; ++j) {
                if (st == 0) {
                    qu[j] = dt * du[j]; qv[j] = dt * dv[j];
                    uu[j] = uu[j] + qu[j] / 3; vv[j] = vv[j] + qv[j] / 3;
                } else {
                    qu[j] = dt * du[j] - A[st] * qu[j];
                    qv[j] = dt * dv[j] - A[st] * qv[j];
                    uu[j] = uu[j] + B23[st] * qu[j];
                    vv[j] = vv[j] + B23[st] * qv[j];
                }
            }
        }
        if (direct_sat && !s->saturate_online)
            saturate_direct(s, dt, direct_sat == 1 ? 1.0 : dt, n, dens, ro, rr, drr, kk, ll,
                            mo, mm, dkk, dll, area);
    }
    free(sd); free(sr); free(sm); free(qd); free(qr); free(qm); free(ro); free(mo); free(du);
    return 0;
}

/* ======================================================================================================
 * EXTENSION, no reference behaviour: buoyancy frequency as a column N on `grids`.
 * C restatement of the definition in oracle/msgwam_oracle.py (bvf_at, bvf_gradient_at, rhs, saturation),
 * operation by operation, so that it agrees with it bit for bit (tests/test_oracle_c.py); with a constant
 * column every expression reduces to the reference's, which is what pins this code path.
 * Five evolving per-ray slots: dens, rr, drr, mm, dmm (lib/libprop.py:640-645 with cgr_up != cgr_down).
 * ====================================================================================================== */

/* slope of the np.interp segment grids[j] <= z < grids[j+1]; 0 where np.interp clamps; NaN stays NaN */
static double bvf_slope_at_(const double *g, const double *b, int n, double z)
{
    if (isnan(z)) return z;
    if (!(z >= g[0] && z < g[n - 1])) return 0.0;
    int lo = 0, hi = n - 1;                 /* searchsorted(g, z, 'right') - 1, clipped to [0, n-2] */
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (z >= g[mid]) lo = mid; else hi = mid;
    }
    if (lo > n - 2) lo = n - 2;
    return (b[lo + 1] - b[lo]) / (g[lo + 1] - g[lo]);
}

void orc_rhs_nz(const orc_setup *s, const double *bvfcol, double dt, int64_t n,
                const double *dens, const double *rr, const double *drr, const double *kk,
                const double *ll, const double *mm, const double *dmm, const double *fray,
                const double *dkk, const double *dll, const double *area,
                const double *uu, const double *vv,
                double *st_dens, double *st_rr, double *st_drr, double *st_mm, double *st_dmm,
                double *du, double *dv)
{
    int ng = s->ngrid, nc = ng - 1, ni = ng - 2;
    double dz = s->grid[1] - s->grid[0];
    double *dudz = (double *)malloc(sizeof(double) * (size_t)ni * 2);
    double *dvdz = dudz + ni;
    for (int j = 0; j < ni; ++j) {
        dudz[j] = (uu[j + 1] - uu[j]) / dz;
        dvdz[j] = (vv[j + 1] - vv[j]) / dz;
    }
    double *P = (double *)calloc((size_t)2 * (nc - 1), sizeof(double));
    double *F = (double *)calloc((size_t)2 * ng, sizeof(double));
    const double *xp = s->grid + 1;
    for (int64_t i = 0; i < n; ++i) {
        double ff = fray[i];
        double bvf = np_interp1(rr[i], s->grids, bvfcol, nc);                       /* N at the ray */
        double cgr_up = cg_rr_(kk[i], ll[i], mm[i], ff, np_interp1(rr[i] + .5 * drr[i], s->grids, bvfcol, nc));   /* :635 */
        double cgr_down = cg_rr_(kk[i], ll[i], mm[i], ff, np_interp1(rr[i] - .5 * drr[i], s->grids, bvfcol, nc)); /* :636 */
        double gu = np_interp1(rr[i], xp, dudz, ni);
        double gv = np_interp1(rr[i], xp, dvdz, ni);
        double gradient = kk[i] * gu + ll[i] * gv;                                  /* :517 */
        gradient = gradient + (bvf * (kk[i] * kk[i] + ll[i] * ll[i]) / omega_(kk[i], ll[i], mm[i], ff, bvf)
                               / (kk[i] * kk[i] + ll[i] * ll[i] + mm[i] * mm[i])
                               * bvf_slope_at_(s->grids, bvfcol, nc, rr[i]));       /* refraction by dN/dz */
        double dmm_st = (kk[i] * 0.0 + ll[i] * 0.0) / (RAD_EARTH + rr[i]) - gradient;
        double drr_st = .5 * (cgr_down + cgr_up);                                   /* :640 */
        double ddrr_st = cgr_up - cgr_down;                                         /* :641 */
        double ddmm_st = dmm[i] / drr[i] * ddrr_st;                                 /* :645 */
        st_rr[i] = drr_st; st_drr[i] = ddrr_st; st_mm[i] = dmm_st; st_dmm[i] = ddmm_st;
        double rr_final = rr[i] + drr_st * dt;
        double drr_final = drr[i] + ddrr_st * dt;
        double mm_final = mm[i] + dmm_st * dt;
        double dmm_final = area[i] / drr_final;
        double rho_f = np_interp1(rr_final, s->grids, s->rhobar, nc);
        double NN = np_interp1(rr_final, s->grids, bvfcol, nc);                     /* N at the projected height */
        double omh = omega_(kk[i], ll[i], mm[i], s->f0, bvf);
        double pv = dkk[i] * dll[i] * dmm_final;
        double maxd = s->kappa * s->kappa * .5 * rho_f * omh * (NN * NN)
                      / (mm_final * mm_final) / (omh * omh - s->f0 * s->f0);
        double dst = 0.0;
        if (maxd < dens[i] * pv) dst = (maxd - dens[i]) / dt;
        st_dens[i] = (double)s->saturate_online * dst;
        double lo = rr[i] - .5 * drr[i], up = rr[i] + .5 * drr[i];
        double mlo = mm[i] - .5 * dmm[i], mup = mm[i] + .5 * dmm[i];
        double vol = fabs(dkk[i] * dll[i] * dmm[i]);
        project_ray(dens[i], lo, up, kk[i], ll[i], .5 * (mlo + mup), ff, vol, bvf, s->grids, nc, 0, P);
    }
    int np_ = nc - 1;
    for (int c = 0; c < 2; ++c) {
        for (int j = 0; j < np_; ++j) F[c * ng + 1 + j] = P[c * np_ + j];
        F[c * ng + 0] = F[c * ng + 1];
        F[c * ng + ng - 1] = F[c * ng + ng - 2];
    }
    for (int j = 0; j < nc; ++j) {
        double gx = (F[j + 1] - F[j]) / dz;
        double gy = (F[ng + j + 1] - F[ng + j]) / dz;
        double rinv = 1.0 / s->rhobar[j];
        du[j] = s->f0 * vv[j] - rinv * (s->pg[j] + gx);
        dv[j] = -s->f0 * uu[j] - rinv * (s->pg[nc + j] + gy);
        if (s->fixed_background) { du[j] = 0.0; dv[j] = 0.0; }
    }
    free(dudz); free(P); free(F);
}

/* nsteps of RK3 with the N(z) column: dens, rr, drr, mm, dmm, uu, vv advanced in place */
int orc_step_nz(const orc_setup *s, const double *bvfcol, double dt, int nsteps, int64_t n,
                double *dens, double *rr, double *drr, const double *kk, const double *ll,
                double *mm, double *dmm, const double *fray,
                const double *dkk, const double *dll, const double *area,
                double *uu, double *vv)
{
    int nc = s->ngrid - 1;
    size_t nb = sizeof(double) * (size_t)(n ? n : 1);
    double *st = malloc(nb * 5), *q = malloc(nb * 5);
    double *du = malloc(sizeof(double) * nc * 4), *dv = du + nc, *qu = du + 2 * nc, *qv = du + 3 * nc;
    if (!st || !q || !du) return -1;
    const double A[3] = {0.0, 5.0 / 9.0, 153.0 / 128.0};
    const double B23[3] = {0.0, 15.0 / 16.0, 8.0 / 15.0};
    double *y[5] = {dens, rr, drr, mm, dmm};
    for (int it = 0; it < nsteps; ++it) {
        for (int sg = 0; sg < 3; ++sg) {
            orc_rhs_nz(s, bvfcol, dt, n, dens, rr, drr, kk, ll, mm, dmm, fray, dkk, dll, area, uu, vv,
                       st, st + n, st + 2 * n, st + 3 * n, st + 4 * n, du, dv);
            for (int v = 0; v < 5; ++v) {
                double *yy = y[v], *qq = q + (size_t)v * n; const double *ss = st + (size_t)v * n;
                for (int64_t i = 0; i < n; ++i) {
                    if (sg == 0) { qq[i] = dt * ss[i]; yy[i] = yy[i] + qq[i] / 3; }
                    else { qq[i] = dt * ss[i] - A[sg] * qq[i]; yy[i] = yy[i] + B23[sg] * qq[i]; }
                }
            }
            for (int j = 0; j < nc; ++j) {
                if (sg == 0) {
                    qu[j] = dt * du[j]; qv[j] = dt * dv[j];
                    uu[j] = uu[j] + qu[j] / 3; vv[j] = vv[j] + qv[j] / 3;
                } else {
                    qu[j] = dt * du[j] - A[sg] * qu[j];
                    qv[j] = dt * dv[j] - A[sg] * qv[j];
                    uu[j] = uu[j] + B23[sg] * qu[j];
                    vv[j] = vv[j] + B23[sg] * qv[j];
                }
            }
        }
    }
    free(st); free(q); free(du);
    return 0;
}
