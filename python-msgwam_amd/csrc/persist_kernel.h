// Persistent RK3 kernel (single-GPU coupled path): ONE launch for `nsteps` RK3 steps = 3*nsteps
// passes; all workgroups stay resident and synchronise through global counters.
//
// Lagged deposit.  The mean-flow update of pass q needs F_{q-1} = wave_projection(state_{q-1}).
// Instead of depositing a pass's INPUT state (and stalling the next pass on the reduction), every
// pass deposits the state it has just PRODUCED: pass q publishes F_{q+1}, a deposit-only pre-pass
// publishes F_0.  The reduction of F_{q+1} (workgroup rows -> 32 group sums -> one final row, three
// ticket levels, fixed order) then has the whole of pass q+1 to complete before pass q+2 reads it, so
// the synchronisation chain is off the critical path and workgroups may drift up to one pass apart
// (no phase-locked load/compute bursts).  Same arithmetic, same values: cg_rr of the new state is
// evaluated from exactly the kk, ll, mm the next pass loads.
//
// Per pass a workgroup: issues its first tile's ray loads; waits (bounded spin of ONE lane on ONE
// counter) until F_{q-1} is final; advances its own LDS replica of the column (uu, vv, q_uu, q_vv
// never touch global memory inside the launch); runs its tiles; publishes its row of F_{q+1}.
// Because a workgroup can be one pass ahead of the slowest, rows, tickets and counters are
// double-buffered by the parity of the flux index.
//
// Cross-workgroup data is re-published at the same addresses inside ONE launch, so every load
// and store of it is an 8-byte agent-scope atomic (sc1): coherent by the memory model, no
// reliance on L1/L2 state.  Order: stores -> every storing wave s_waitcnt vmcnt(0) -> barrier ->
// relaxed agent fetch_add; consumer: relaxed poll -> barrier -> sc1 loads only
// (cdna_hip_programming.md Guideline 16, the all-sc1 form: no L1-invalidating acquire, which
// costs microseconds per workgroup at 4 workgroups per CU).
//
// Every wait is bounded (wall clock); on time-out a status word is raised and all workgroups
// leave.  The host sizes the grid from the occupancy query (all workgroups must be resident).
#pragma once
#include "ray_kernels.h"

namespace msgw {

constexpr int PERSIST_GROUPS = 32;       // most groups (= group sums added in the prologue)

struct PersistArgs {
    StageArgs s;                  // rays, constants, static column tables; grp_size/row_stride
    int nsteps;
    int ngroups;                  // workgroups [g*grp_size, ...) form group g
    double *grp_part2;            // [2][workgroups][row_stride]   workgroup rows, by flux parity
    double *grp_rows2;            // [2][PERSIST_GROUPS][2*(ng-2)] group sums, by flux parity
    double *flux2;                // [2][2*(ng-2)]                 final flux row, by flux parity
    unsigned int *grp_cnt2;       // [2][64] arrival tickets of the groups   (zero at launch)
    unsigned int *done2;          // [2] completed groups of a flux          (zero at launch)
    unsigned int *ready;          // fluxes whose final row is published     (zero at launch)
    int *status;                  // 0 ok, 1 a wait timed out
    unsigned long long timeout_ticks;   // wall_clock64 ticks (100 MHz)
    ColIn cin;                    // canonical column at entry
    ColOut cout;                  // canonical column at exit (workgroup 0)
    double *dudz, *dvdz, *slu, *slv;    // derived tables at exit (workgroup 0)
#ifdef MSGW_STAMP
    unsigned long long *pstamps;  // diagnostic build: [workgroups][PSTAMP_PASSES][4] wall-clock stamps
#endif
};
#ifdef MSGW_STAMP
constexpr int PSTAMP_PASSES = 16;
#define PSTAMP(q, k) do { if (threadIdx.x == 0 && p.pstamps && (q) < (unsigned)PSTAMP_PASSES) \
    p.pstamps[((size_t)blockIdx.x * PSTAMP_PASSES + (q)) * 4 + (k)] = wall_clock64(); } while (0)
#else
#define PSTAMP(q, k) do { } while (0)
#endif

typedef unsigned long long u64_t;

__device__ __forceinline__ double ld_agent(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64_t *>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<u64_t *>(p), (u64_t)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// Wait until *ready >= target stages (one lane polls, bounded), then release the workgroup.
__device__ __forceinline__ bool persist_wait(const PersistArgs p, unsigned int target, int *s_flag, int tid)
{
    if (tid == 0) {
        int ok = 1;
        if (__hip_atomic_load(p.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(p.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(8);
                if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                    wall_clock64() - t0 > p.timeout_ticks) {
                    __hip_atomic_store(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        // No agent-scope acquire (buffer_inv sc1 costs microseconds per workgroup at 4 workgroups/CU):
        // every handed-off byte is read with an sc1 (agent-scope atomic) load that bypasses L1, so
        // only the compiler must be kept from hoisting those loads above the poll.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

// Publish this workgroup's row of flux f, ticket, group reduction by the last arriver, final row
// by the reducer of the flux's last group (all cross-workgroup accesses are agent-scope atomics).
// (Dedicated "service" workgroups that poll the tickets and do the reductions instead were measured:
// publishing got slower and the period rose from 26.5 to 31 us per pass, so the last arriver reduces.)
__device__ __forceinline__ void persist_publish(const PersistArgs p, const double *rows, int ncp, int *s_flag,
                                                int tid, unsigned int f)
{
    const StageArgs a = p.s;
    const int ncols = 2 * ncp;
    const int b = blockIdx.x, nb = gridDim.x;
    const int g = b / a.grp_size, r0 = g * a.grp_size, r1 = min(nb, r0 + a.grp_size);
    const unsigned int par = f & 1u;
    double *part = p.grp_part2 + (size_t)par * nb * a.row_stride;
    unsigned int *ticket = p.grp_cnt2 + par * 64 + g;
    __syncthreads();                                          // all waves' rows complete in LDS
    double *mine = part + (size_t)b * a.row_stride;
    for (int col = tid; col < ncols; col += BLOCK) {
        double acc = rows[col];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) acc = acc + rows[w * ncols + col];
        st_agent(mine + col, acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains
    __syncthreads();
    if (tid == 0) {
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler-only: rows are read with sc1 loads
        *s_flag = (t == (unsigned int)(r1 - r0 - 1)) ? 1 : 0;
    }
    __syncthreads();
    if (!*s_flag) return;
    // second level: last arriver of the group adds the group's rows in row order
    double *grow = p.grp_rows2 + ((size_t)par * PERSIST_GROUPS + g) * ncols;
    if (tid < ncols) {
        const double *src = part + tid;
        double acc = 0.0;
        for (int r = r0; r < r1; r += 32) {                   // row order, 32 loads in flight
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = ld_agent(src + (size_t)min(r + u, r1 - 1) * a.row_stride);
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = acc + ((r + u < r1) ? v[u] : 0.0);
        }
        st_agent(grow + tid, acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm for flux f+2
        const unsigned int t2 = __hip_atomic_fetch_add(p.done2 + par, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        *s_flag = (t2 == (unsigned int)p.ngroups - 1u) ? 1 : 0;   // last group of this flux?
    }
    __syncthreads();
    if (!*s_flag) return;
    // third level: add the group sums (group order) into ONE final row, so that every workgroup
    // reads 2*(ng-2) values instead of ngroups times that
    if (tid < ncols) {
        const double *src = p.grp_rows2 + (size_t)par * PERSIST_GROUPS * ncols + tid;
        double tot = 0.0;
        for (int r = 0; r < p.ngroups; r += 32) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = ld_agent(src + (size_t)min(r + u, p.ngroups - 1) * ncols);
#pragma unroll
            for (int u = 0; u < 32; ++u) tot = tot + ((r + u < p.ngroups) ? v[u] : 0.0);
        }
        st_agent(p.flux2 + (size_t)par * ncols + tid, tot);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(p.done2 + par, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for flux f+2
        __hip_atomic_fetch_add(p.ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct PersistLds {
    double4 *sh; double2 *rho2; double *xg, *gs, *rows;
    double *cu, *cv, *cqu, *cqv, *crho, *cpg;     // column replica + static rhobar, pg [2][nc]
    double *F, *u, *v, *du, *dv;                  // scratch (aliases rows)
    int *flag;
};

// column_q = RK stage `pstage` of column_{q-1} with the final row of flux F_{q-1}
__device__ __forceinline__ void persist_column(const PersistArgs p, const PersistLds L, unsigned int q,
                                               int pstage, int tid)
{
    const StageArgs a = p.s;
    const int ng = a.ng, ni = ng - 2, nc = ng - 1, ncp = ng - 2, ncols = 2 * ncp;
    if (tid < ncols) {
        const int pp = tid / ncp, c = tid - pp * ncp;
        L.F[pp * ng + 1 + c] = ld_agent(p.flux2 + (size_t)((q - 1) & 1) * ncols + tid);   // pm_flux[:, 1:-1] (:654)
    }
    __syncthreads();
    column_flux_ends(tid, ng, L.F);
    __syncthreads();
    if (tid < nc) {
        double du, dv, un, vn, qu, qv;
        column_tendency(tid, ng, a.f0, a.dzg, 0, L.F, L.crho[tid], L.cpg[tid], L.cpg[nc + tid], L.cu[tid],
                        L.cv[tid], du, dv);
        column_rk(pstage, a.dt, du, dv, L.cu[tid], L.cv[tid], L.cqu[tid], L.cqv[tid], un, vn, qu, qv);
        L.cu[tid] = un; L.cv[tid] = vn; L.cqu[tid] = qu; L.cqv[tid] = qv;
    }
    __syncthreads();
    column_shear(tid, BLOCK, ng, a.dzg, L.cu, L.cv, L.du, L.dv);
    __syncthreads();
    for (int i = tid; i < ni; i += BLOCK) {
        const bool in = i < ni - 1;
        L.sh[i] = make_double4(L.du[i], in ? column_slope(L.du, L.xg, i) : 0.0,
                               L.dv[i], in ? column_slope(L.dv, L.xg, i) : 0.0);
    }
    __syncthreads();
}

template <int STAGE, bool SAT, bool FVEC, bool DIRECT>
__device__ __forceinline__ bool persist_stage(const PersistArgs p, const PersistLds L, unsigned int q,
                                              long long start, long long end, int tid, int wave, int lane)
{
    const StageArgs a = p.s;
    const int ncp = a.ng - 2;
    // Opaque copies: without them the three inlined stage bodies share (CSE) every per-array tile
    // address and keep ~60 VGPRs of 64-bit addresses alive across the whole step.
    asm volatile("" : "+s"(start));
    asm volatile("" : "+v"(tid));
    PSTAMP(q, 0);
    TileRegs cur;
    // wave 0 polls and fences (its acquire waits for its own outstanding loads), so it loads after
    if (wave != 0 || q == 0) load_tile<STAGE, SAT, FVEC, true, DIRECT>(cur, a, start, tid, end);
    if (q > 0) {
        if (!persist_wait(p, q, L.flag, tid)) return false;
        if (wave == 0) load_tile<STAGE, SAT, FVEC, true, DIRECT>(cur, a, start, tid, end);
        persist_column(p, L, q, (STAGE + 2) % 3, tid);
    }
    PSTAMP(q, 1);
    for (int i = tid; i < WAVES * 2 * ncp; i += BLOCK) L.rows[i] = 0.0;
    __syncthreads();
    int wmin = INT_MAX, wmax = INT_MIN;
    const StageLds SL{L.sh, L.rho2, L.xg, L.gs, L.rows};
    process_tiles<STAGE, SAT, FVEC, true, DIRECT, 2, true>(a, SL, cur, start, end, tid, wave, lane, wmin, wmax);
    PSTAMP(q, 2);
    persist_publish(p, L.rows, ncp, L.flag, tid, q + 1u);     // this pass produced state_{q+1}: publish F_{q+1}
    PSTAMP(q, 3);
    return true;
}

template <bool SAT, bool FVEC, bool DIRECT>
__global__ void __launch_bounds__(BLOCK) k_rk3_persist(const PersistArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const StageArgs a = p.s;
    const int ng = a.ng, ni = ng - 2, nc = ng - 1, ncp = ng - 2;
    PersistLds L;
    L.sh = reinterpret_cast<double4 *>(lds);
    L.rho2 = reinterpret_cast<double2 *>(lds + 4 * ni);
    L.xg = lds + 4 * ni + 2 * nc;
    L.gs = L.xg + ni;
    L.rows = L.gs + nc;
    double *colrep = L.rows + WAVES * 2 * ncp;
    L.cu = colrep; L.cv = L.cu + nc; L.cqu = L.cv + nc; L.cqv = L.cqu + nc; L.crho = L.cqv + nc; L.cpg = L.crho + nc;
    L.flag = reinterpret_cast<int *>(L.cpg + 2 * nc);
    L.F = L.rows; L.u = L.F + 2 * ng; L.v = L.u + nc; L.du = L.v + nc; L.dv = L.du + ni;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long start = (long long)blockIdx.x * a.rays_per_block;
    const long long end = min(a.n, start + a.rays_per_block);

    for (int i = tid; i < ni; i += BLOCK) L.xg[i] = a.c.xg[i];
    for (int i = tid; i < nc; i += BLOCK) {
        L.gs[i] = a.c.grids[i];
        L.rho2[i] = make_double2(a.c.rhobar[i], (i < nc - 1) ? a.c.slrho[i] : 0.0);
        L.cu[i] = p.cin.uu[i]; L.cv[i] = p.cin.vv[i]; L.cqu[i] = 0.0; L.cqv[i] = 0.0;
        L.crho[i] = a.c.rhobar[i]; L.cpg[i] = a.pg[i]; L.cpg[nc + i] = a.pg[nc + i];
    }
    __syncthreads();
    column_shear(tid, BLOCK, ng, a.dzg, L.cu, L.cv, L.du, L.dv);
    __syncthreads();
    for (int i = tid; i < ni; i += BLOCK) {
        const bool in = i < ni - 1;
        L.sh[i] = make_double4(L.du[i], in ? column_slope(L.du, L.xg, i) : 0.0,
                               L.dv[i], in ? column_slope(L.dv, L.xg, i) : 0.0);
    }
    __syncthreads();

    // deposit-only pre-pass: F_0 = wave_projection(state_0)
    {
        for (int i = tid; i < WAVES * 2 * ncp; i += BLOCK) L.rows[i] = 0.0;
        __syncthreads();
        const StageLds SL{L.sh, L.rho2, L.xg, L.gs, L.rows};
        deposit_pass<FVEC, 2>(a, SL, start, end, tid, wave, lane);
        persist_publish(p, L.rows, ncp, L.flag, tid, 0u);
    }
    unsigned int q = 0;
    for (int step = 0; step < p.nsteps; ++step) {
        if (!persist_stage<0, SAT, FVEC, DIRECT>(p, L, q, start, end, tid, wave, lane)) return;
        ++q;
        if (!persist_stage<1, SAT, FVEC, DIRECT>(p, L, q, start, end, tid, wave, lane)) return;
        ++q;
        if (!persist_stage<2, SAT, FVEC, DIRECT>(p, L, q, start, end, tid, wave, lane)) return;
        ++q;
    }
    if (blockIdx.x != 0) return;
    // workgroup 0 applies the last update (column_q needs F_{q-1}) and writes the column back;
    // F_q, published by the last pass, is not used (the next call starts with its own pre-pass)
    if (!persist_wait(p, q, L.flag, tid)) return;
    persist_column(p, L, q, 2, tid);
    for (int i = tid; i < nc; i += BLOCK) {
        p.cout.uu[i] = L.cu[i]; p.cout.vv[i] = L.cv[i]; p.cout.q_uu[i] = L.cqu[i]; p.cout.q_vv[i] = L.cqv[i];
    }
    for (int i = tid; i < ni; i += BLOCK) {
        const double4 t = L.sh[i];
        p.dudz[i] = t.x; p.dvdz[i] = t.z;
        if (i < ni - 1) { p.slu[i] = t.y; p.slv[i] = t.w; }
    }
}

}   // namespace msgw
