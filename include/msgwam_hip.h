/*
 * msgwam_hip.h -- C ABI of the MI355X (gfx950) ray-propagation library.
 *
 * The reference (dsconnelly/python-msgwam) has NO FFI: its boundary for this
 * path is the Python module surface of lib/libprop.py as used by raytracer.py.
 * Each entry point below names the reference interface it stands behind
 * (file:line relative to the reference root).  The Python mirror of that
 * module surface lives in python-msgwam_amd/msgwam_amd/libprop.py and reaches
 * these functions through ctypes.
 *
 * Conventions: plain pointers + sizes; every host pointer is caller-owned and
 * only read/written during the call; device memory is owned by the opaque
 * context.  Return value 0 = ok, < 0 = error (text via msgw_last_error).
 * No C++ exceptions cross this boundary.  One context per GPU; a context is
 * not thread-safe.  NaN/inf propagate as in numpy, nothing traps.
 *
 * All floating-point data crossing this boundary is float64 (the reference is
 * float64 throughout).  The resident ray state is float64 by default -- that is
 * the reference-parity mode -- or float32 when the context is created with
 * MSGW_DTYPE_F32 (BASELINE config 5: half the bytes per ray, float32 per-ray
 * arithmetic; flux rows, their reduction and the mean-flow column stay float64).
 * Scope: scalar bvf as in the reference (a bvf COLUMN on grids is an extension, msgw_set_bvf_column);
 * both HPROP_GLOBAL branches (lib/libprop.py:5).  With
 * HPROP_GLOBAL = False, the driver's configuration (raytracer.py:38), only dens,
 * rr, mm and the uu, vv columns evolve, so only those are copied
 * back; HPROP_GLOBAL = True adds lam, phi, kk, ll (msgw_upload_hprop /
 * msgw_download_hprop), a bvf column drr, dmm (msgw_download_extents).  Every
 * combination of HPROP, bvf column, online / direct saturation, relaunch and
 * ray type (float64, float32) is served; HPROP_GLOBAL = False with a scalar bvf
 * by the tuned kernels (persistent form), everything else by the general
 * per-stage kernel.
 */
#ifndef MSGWAM_HIP_H
#define MSGWAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msgw_ctx msgw_ctx;

#define MSGW_ABI_VERSION 3

/* error codes */
#define MSGW_OK            0
#define MSGW_ERR_ARG      -1   /* bad argument / call order                 */
#define MSGW_ERR_HIP      -2   /* a HIP runtime call failed                  */
#define MSGW_ERR_NOGPU    -3   /* no usable gfx950 device                    */
#define MSGW_ERR_RCCL     -4   /* RCCL missing or a collective failed        */
#define MSGW_ERR_UNSUP    -5   /* outside the supported scope (e.g. HPROP)   */

/* flags of msgw_create_ex */
#define MSGW_DTYPE_F32          1u  /* resident ray state and per-ray arithmetic in float32 (default: float64) */

/* msgw_counters_t.transport: how several ranks sum their flux rows */
#define MSGW_TRANSPORT_NONE       0   /* one rank */
#define MSGW_TRANSPORT_RCCL       1   /* ncclAllReduce once per RK stage (lagged launch chain) */
#define MSGW_TRANSPORT_HOST_SHM   2   /* inside the persistent kernel, through a node-shared host segment */
#define MSGW_TRANSPORT_DEVICE_IPC 3   /* inside the persistent kernel, peer writes into HIP-IPC-mapped HBM (xGMI) */

/* flags of msgw_step / msgw_rhs */
#define MSGW_FIXED_BACKGROUND   1u  /* rhs hook that zeroes slots 9,10: no deposit, column frozen */
#define MSGW_DIRECT_SAT_QUIRK   2u  /* driver's post-step saturation, rr tendency "/ 1" (raytracer.py:184) */
#define MSGW_DIRECT_SAT         4u  /* the same with "/ dt"                                         */
#define MSGW_NO_GRAPH           8u  /* launch kernels eagerly instead of replaying a hipGraph       */
#define MSGW_TIME_KERNELS      16u  /* bracket every ray-stage kernel with HIP events (implies NO_GRAPH) */
#define MSGW_RELAUNCH          32u  /* EXTENSION, not in the reference (BASELINE config 5): after every RK3 step
                                       a ray whose volume has left the column (rr - drr/2 > grid[-1] or
                                       rr + drr/2 < grid[0]) or whose dens fell below frac x its source value
                                       (msgw_set_relaunch) is recycled to the (dens, rr, mm) it was uploaded with.
                                       Comparisons with NaN are false.  Checked once per RK3 step, after the
                                       direct saturation when that is on.  A compile-time variant of the persistent
                                       kernel and of the per-stage kernels.  */

/* counters filled by msgw_counters */
typedef struct {
    double  last_step_ms;        /* HIP-event time of the last msgw_step call (whole call)       */
    double  ray_kernel_ms_sum;   /* sum of ray-stage kernel durations (MSGW_TIME_KERNELS only)   */
    int64_t ray_kernel_launches; /* number of ray-stage launches behind that sum                 */
    int64_t ray_steps_total;     /* rays x RK3 steps advanced since create                       */
    int64_t nray;                /* rays resident                                               */
    int32_t ngrid;               /* interfaces                                                  */
    int32_t blocks;              /* workgroups of the ray-stage kernel                          */
    int32_t graph_steps;         /* RK3 steps per captured graph (0 = eager)                    */
    int32_t nranks;              /* communicator size (1 = no collective)                       */
    int32_t persist_steps;       /* RK3 steps covered by ONE launch in the last msgw_step call: persistent coupled kernel
                                    or fused fixed-background kernel (0 = one launch per RK stage)       */
    int32_t exchange;            /* 1: multi-rank steps use the in-kernel node-level flux exchange */
    int32_t persist_resident_tiles; /* tiles per workgroup the last persistent launch kept in registers */
    int32_t elem_bytes;          /* bytes per element of the resident ray state: 8 (float64) or 4 (float32) */
    int32_t transport;           /* MSGW_TRANSPORT_* of the communicator                         */
    int32_t tenants;             /* ranks of the communicator that share this rank's device (1 in production) */
    /* ABI 3 */
    int32_t launch_grid;         /* workgroups of the LAST ray-kernel launch (persistent kernel: ray workgroups + reducers
                                    + column workgroup; `blocks` is the per-stage kernels' geometry) */
    int32_t launch_ray_workgroups; /* of those, the workgroups that own rays */
    int32_t launch_reducers;     /* persistent kernel: reducer workgroups of the last launch (0: the last arriver reduces) */
    int32_t fixed_narrow;        /* fixed-background kernel: 1 = one ray per lane, one wavefront per workgroup */
    int32_t carried_flux;        /* 1: the last persistent launch took the flux of its initial state from the previous launch
                                    (same resident state, same kernel flavour) instead of a deposit-only pre-pass */
    double  algorithmic_bytes_total; /* SURVEY 8d bytes moved since create: words per ray-step of the path taken (35 coupled,
                                    45 with online saturation, 6 fixed background; the general chain 3 L + 7 E, i.e. 69
                                    HPROP, 55 N(z), 86 both) x elem_bytes x rays x steps -- the yardstick of
                                    the roofline, not a hardware counter */
    int32_t cooperative;         /* 1: the last persistent launch went through hipLaunchCooperativeKernel (the runtime
                                    vouches for co-residency of the grid); 0: plain launch (several ranks, MSGW_COOP=0, under a
                                    rocprofiler tool, or a device without cooperative launches) */
    int32_t coop_refused;        /* cooperative launches the runtime refused since create (that call took the launch chain,
                                    later ones the plain launch) */
} msgw_counters_t;

/* HPROP on: slots 1 and 2 (lam, phi) of the rays uploaded by the last msgw_upload_rays (same n). */
int msgw_upload_hprop(msgw_ctx *ctx, int64_t n, const double *lam, const double *phi);
/* HPROP on: lam, phi, kk, ll (any pointer may be NULL); tendencies != 0: their tendencies left by msgw_rhs. */
int msgw_download_hprop(msgw_ctx *ctx, int64_t n, int tendencies, double *lam, double *phi, double *kk, double *ll);

/* EXTENSION (the reference has a scalar bvf only, lib/libprop.py:380, :398, :422, :583): buoyancy
 * frequency as a column bvf[ngrid-1] on lprop.grids, np.interp'ed to the height each expression is about (definition: DESIGN.md 6d).  The vertical group velocity then differs at rr +- drr/2
 * (lib/libprop.py:635-636), so drr and dmm evolve as well (:641, :645).  Call BEFORE msgw_upload_rays; NULL returns
 * to the scalar of msgw_set_config.  Steps run in the general per-stage kernel (any ray type, HPROP on or off).
 * In the limit N(z) = const the results are the reference's. */
int msgw_set_bvf_column(msgw_ctx *ctx, const double *bvf);
/* Slots 4, 8 (drr, dmm) of the resident rays -- they only change with an N(z) column -- or, tendencies != 0, their
 * tendencies left by msgw_rhs.  Any pointer may be NULL. */
int msgw_download_extents(msgw_ctx *ctx, int64_t n, int tendencies, double *drr, double *dmm);

/* EXTENSION: the "broken ray" fraction of MSGW_RELAUNCH (default 1e-6; 0 disables that criterion). */
int msgw_set_relaunch(msgw_ctx *ctx, double frac);
/* EXTENSION: the (dens, rr, mm) a recycled slot returns to.  msgw_upload_rays sets them to the state it uploads; this
 * call replaces them afterwards -- resuming from a checkpoint (the uploaded state is then not the source any more), or
 * a source that changes in time.  n must be the resident ray count. */
int msgw_set_relaunch_source(msgw_ctx *ctx, int64_t n, const double *dens, const double *rr, const double *mm);

/* ABI version of the loaded library (== MSGW_ABI_VERSION). */
int msgw_abi_version(void);

/* Last error text; ctx may be NULL for errors of msgw_create itself. */
const char *msgw_last_error(const msgw_ctx *ctx);

/* Create a context on HIP device `device` for at most nray_cap rays on a
 * column with ngrid interfaces (len(lprop.grid), raytracer.py:74-77). */
int msgw_create(msgw_ctx **out, int device, int64_t nray_cap, int ngrid);
/* The same with MSGW_DTYPE_* flags (no reference counterpart: the reference is float64 numpy). */
int msgw_create_ex(msgw_ctx **out, int device, int64_t nray_cap, int ngrid, unsigned create_flags);
int msgw_destroy(msgw_ctx *ctx);

/* model_config scalars read by the hot path (lib/libprop.py:380, :534, :582-584,
 * :633) and the module flag HPROP_GLOBAL (lib/libprop.py:5, raytracer.py:38).
 * f0 = 2*ROT_EARTH*sin(phi0) is computed by the caller in numpy so that sin()
 * is bit-identical to the reference.  hprop != 0 (lib/libprop.py:5 HPROP_GLOBAL = True, libprop's own
 * default; raytracer.py:38 switches it off): lam, phi, kk, ll evolve as well; the steps then run in a
 * kernel of their own (one launch per RK stage) and need msgw_upload_hprop after msgw_upload_rays. */
int msgw_set_config(msgw_ctx *ctx, double bvf, double f0, double kappa,
                    int saturate_online, int hprop);

/* The column: lprop.grid [ngrid], lprop.grids [ngrid-1] (raytracer.py:76-77),
 * lprop.rhobar [ngrid-1] (set_hydrostatics, lib/libprop.py:47-62),
 * lprop.pressure_gradient [2][ngrid-1] (lib/libprop.py:65-82), and the wind
 * columns uu, vv [ngrid-1] = state slots 9, 10 (lib/libprop.py:629). */
int msgw_set_column(msgw_ctx *ctx, const double *grid, const double *grids,
                    const double *rhobar, const double *pgrad,
                    const double *uu, const double *vv);

/* Per-ray state slots 0,3,4,5,6,7,8 of the RK3 state vector (lib/libprop.py:629;
 * raytracer.py:160-172) plus the per-ray statics dkk, dll, rr_mm_area
 * (set_statics, lib/libprop.py:14-27; raytracer.py:105-109) and
 * fray[i] = 2*ROT_EARTH*sin(phi[i]) (slot 2, computed in numpy).  Slots 1, 2
 * (lam, phi) never change with HPROP off and stay on the host. */
int msgw_upload_rays(msgw_ctx *ctx, int64_t n,
                     const double *dens, const double *rr, const double *drr,
                     const double *kk, const double *ll, const double *mm,
                     const double *dmm, const double *fray,
                     const double *dkk, const double *dll, const double *rr_mm_area);

/* lprop.RK3(dt, state) applied nsteps times (lib/libprop.py:680-700 calling
 * rhs_default :618-676 three times per step), state resident on the device.
 * With MSGW_DIRECT_SAT* each step is followed by the driver's
 * lprop.saturation(..., direct=True) (raytracer.py:182-188). */
int msgw_step(msgw_ctx *ctx, double dt, int nsteps, unsigned flags);

/* One evaluation of rhs_default (lib/libprop.py:618-676) on the resident state:
 * tendencies of slots 0 (dens), 3 (rr), 7 (mm) [n each], 9, 10 (uu, vv)
 * [ngrid-1 each] and pm_flux [2][ngrid] (lib/libprop.py:653-660).  Any output
 * pointer may be NULL.  Tendencies of the other slots are identically zero. */
int msgw_rhs(msgw_ctx *ctx, double dt, unsigned flags,
             double *st_dens, double *st_rr, double *st_mm,
             double *st_uu, double *st_vv, double *pm_flux);

/* lprop.wave_projection(..., grid, var) (lib/libprop.py:92-219) of the resident
 * rays on an arbitrary uniform grid G [nG].  var 0, 1, 2 (cell centres): out is
 * [2][nG-1] for var 0 and [nG-1] otherwise (raytracer.py:213, :227); var 3, 4
 * (interfaces, :199-219): [nG] and [2][nG].  With an N(z) column the group velocity uses N at the ray centre rr.
 * With HPROP on the Coriolis parameter of a ray is that of its CURRENT latitude (lib/libprop.py:382). */
int msgw_project(msgw_ctx *ctx, int var, const double *G, int nG, double *out);

/* lprop.wave_projection(dens, lam, phi, rr_low, rr_up, kk, ll, mm_low, mm_up, dkk,
 * dll, dmm, grid, var) (lib/libprop.py:92-197) on caller-supplied host arrays
 * [n each]; fray[i] = 2*ROT_EARTH*sin(phi[i]); lam is unused by the reference.
 * Independent of the resident state (only the stream and scratch are used).
 * bvf: the scalar N of model_config (:139-144 -> :380); EXTENSION: NaN = the context's N(z) column
 * (msgw_set_bvf_column + msgw_set_column), np.interp'ed to the ray centre .5 * (rr_low + rr_up). */
int msgw_project_arrays(msgw_ctx *ctx, int64_t n, int var, double bvf,
                        const double *dens, const double *rr_low, const double *rr_up,
                        const double *kk, const double *ll, const double *mm_low,
                        const double *mm_up, const double *dkk, const double *dll,
                        const double *dmm, const double *fray,
                        const double *G, int nG, double *out);

/* lprop.saturation(dt, dens, rr_center, rr_center_st, drr, drr_st, kk, ll,
 * mm_center, mm_center_st, direct) (lib/libprop.py:561-615) on caller arrays,
 * with statics dkk, dll, rr_mm_area (:585-587) passed explicitly; uses the
 * resident config (bvf, f0, kappa) and column (grids, rhobar).  out [n]:
 * the saturated density (direct != 0, :606-610) or its tendency (:612-615).
 * With an N(z) column (msgw_set_bvf_column; extension) omega uses N at rr_center and the cap N at the projected height
 * rr_center + rr_center_st * dt, as in the stage kernels. */
int msgw_saturation(msgw_ctx *ctx, int64_t n, double dt, int direct,
                    const double *dens, const double *rr_center, const double *rr_center_st,
                    const double *drr, const double *drr_st, const double *kk, const double *ll,
                    const double *mm_center, const double *mm_center_st,
                    const double *dkk, const double *dll, const double *rr_mm_area,
                    double *out);

/* Test support (no reference counterpart): the float64 square root, the division by a constant d and (y != NULL) the
 * division x[i] / y[i] exactly as the ray kernels evaluate them (lib/libprop.py:383, :448, :124, :694 are what they stand
 * for), element by element on x [n] -> out_sqrt [n], out_div [n], out_quot [n], so that the parity tests can hold them bit
 * for bit to numpy's sqrt(x), x / d and x / y over the domains csrc/real.h states, special values included. */
int msgw_probe_arith(msgw_ctx *ctx, int64_t n, const double *x, double d, double *out_sqrt, double *out_div,
                     const double *y, double *out_quot);

/* Copy the evolving slots back (blocking). Any pointer may be NULL. */
int msgw_download_rays(msgw_ctx *ctx, int64_t n, double *dens, double *rr, double *mm);
int msgw_download_column(msgw_ctx *ctx, double *uu, double *vv);

/* Snapshots (no reference counterpart; serve the lazy host copies of the Python mirror): a
 * stream-ordered device copy of the evolving slots -- dens, rr, mm, the uu, vv columns and, with HPROP on, lam, phi,
 * kk, ll and, with an N(z) column, drr, dmm.  msgw_step does not wait for its kernels, so a caller that may never look at a state need not copy it to
 * the host; if it does look later, after the resident state has moved on, the snapshot still has it.
 * Download: any pointer may be NULL.  A snapshot must be destroyed before its context. */
typedef struct msgw_snapshot msgw_snapshot;
#define MSGW_SLOT_DENS 0
#define MSGW_SLOT_RR   1
#define MSGW_SLOT_MM   2
#define MSGW_SLOT_UU   3
#define MSGW_SLOT_VV   4
#define MSGW_SLOT_LAM  5   /* 5..8: HPROP on */
#define MSGW_SLOT_PHI  6
#define MSGW_SLOT_KK   7
#define MSGW_SLOT_LL   8
#define MSGW_SLOT_DRR  9   /* 9, 10: N(z) column */
#define MSGW_SLOT_DMM  10
#define MSGW_SLOT_COUNT 11
int msgw_snapshot_create(msgw_ctx *ctx, msgw_snapshot **out);
int msgw_snapshot_download(msgw_ctx *ctx, msgw_snapshot *snap, int slot, double *out);
int msgw_snapshot_destroy(msgw_ctx *ctx, msgw_snapshot *snap);

/* Wait for all queued work of this context. */
int msgw_sync(msgw_ctx *ctx);

/* Multi-GPU (no reference counterpart: the reference is single-process).
 * Rays are sharded across ranks; the per-stage flux profile (2 x (ngrid-2)
 * float64) is summed over the ranks before the mean-flow update: inside the
 * persistent kernel by peer writes into HIP-IPC-mapped HBM (xGMI), with a
 * node-shared host segment and an RCCL all-reduce chain as fallbacks
 * (msgw_counters_t.transport says which).
 * msgw_comm_unique_id fills a 128-byte ncclUniqueId on rank 0; every rank then
 * calls msgw_comm_init with the same id. */
int msgw_comm_unique_id(void *id128);
int msgw_comm_init(msgw_ctx *ctx, const void *id128, int rank, int nranks);

/* Tuning knobs: workgroups per CU of the ray-stage kernel (default 4) and RK3
 * steps per captured hipGraph (default 4; 0 = always eager). */
int msgw_set_tuning(msgw_ctx *ctx, int blocks_per_cu, int graph_steps);

int msgw_counters(msgw_ctx *ctx, msgw_counters_t *out);

#ifdef __cplusplus
}
#endif
#endif /* MSGWAM_HIP_H */
