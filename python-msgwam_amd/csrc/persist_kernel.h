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
//
// Several GPUs (one process each, rays sharded, column replicated).  The sum over the ranks is a step of the column
// workgroup (persist_column_wg): it adds the reducers' group sums to this rank's row, writes that row into the slot it
// owns in EVERY rank's exchange buffer (device-resident transport: peer writes over xGMI into HIP-IPC-mapped HBM,
// system-scope stores, release, then its sequence number; fallback: a host-resident segment reached over PCIe), waits
// (bounded) until every rank's sequence number in its OWN buffer has reached this flux, and adds the rank rows in rank
// order -- the same arithmetic on every rank, so the replicated columns stay bitwise identical.  No launch, no
// collective kernel, no cross-stream event per RK stage.  (Round 1 used a separate exchange workgroup between reducers
// and column workgroup: one more hand-off on the reduce chain, 46.7 vs 45.0 us per step; it survives as the fallback
// for grids without reducer workgroups, persist_exchange.)
// Round 3 tried to shorten the reduce chain by letting every reducer own a slice of the column (rows in slice-major
// layout, reducers advance their winds and publish their part of the table, the rank sum reducer by reducer with
// tagged 8-byte words instead of release + flag): a pass was released 5.7 us after the last row instead of 9.2 and the
// 1-rank exchange cost +4.8 % instead of +13 %, but config 3 ran 37.4 us per step against 33.7 (all workgroups then
// wait for every release, phase-locked) -- kept on the branch r3-column-slice-reducers, DESIGN.md 6.
#pragma once
#include "ray_kernels.h"

namespace msgw {

constexpr unsigned int PERSIST_OPT_PRIO = 1u;   // workgroups that trail by a pass run it at raised wave priority
__device__ __forceinline__ void setprio_rt(unsigned int v)    // s_setprio takes an immediate
{
    if (v == 0u) __builtin_amdgcn_s_setprio(0);
    else if (v == 1u) __builtin_amdgcn_s_setprio(1);
    else if (v == 2u) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}
// BALANCE priorities in opts: bits 4-5 released on arrival, 6-7 had to wait, 8-9 prefetched (persist_publish)
constexpr unsigned int PERSIST_OPT_BALANCE = 4u;    // a workgroup that finds its pass released on arrival raises its wave priority (persist_stage)
constexpr unsigned int PERSIST_OPT_PREFETCH = 2u;   // early poll + table prefetch at the pass boundary (persist_publish)
constexpr unsigned int PERSIST_OPT_LEANPOLL = 8u;   // wait loops look at the status word / clock every 16th poll only
constexpr unsigned int PERSIST_OPT_NAP1 = 1u << 12, PERSIST_OPT_NAP8 = 1u << 13;   // nap between two polls of a release: s_sleep 1 / 8 instead of 2
constexpr int PERSIST_GROUPS = 32;       // most groups (= group sums added in the prologue)
constexpr int PD_ROW = 64;               // (unused since the column workgroup does the sum over the ranks itself)
constexpr int PD_LOCAL = 96;             // ready[PD_LOCAL]: several ranks: fluxes whose rank row is complete
constexpr int TICKET_STRIDE = 32;        // unsigned ints between two tickets: pollers and arrivers of different
                                         // groups never share a cache line

// Node-level exchange of the rank rows.  Two transports, same protocol (sequence-numbered rows, rank-order sum):
//  * direct (default when it passes its self-test): every rank owns a buffer in ITS OWN HBM that all ranks of the
//    node have mapped through HIP IPC.  A rank writes its row into the slot it owns in EVERY rank's buffer (one
//    posted write per peer over xGMI, nothing is read remotely), then its sequence number likewise; it polls and
//    sums its LOCAL buffer only.
//  * host segment (fallback): one POSIX shared-memory segment registered with every GPU (fine-grained host memory,
//    reached over PCIe); rows and sequence numbers are written to and polled in the segment.
// Read by the exchange workgroup only.
struct XchArgs {
    int nranks, rank;
    int stride;                   // doubles per rank row (a multiple of 8)
    int direct;                   // 1: device-resident transport (peer_rows / peer_flags are valid)
    double *rows;                 // [2][nranks][2 * stride] words: rank rows as tagged granules (two 8-byte words per
                                  // float64, see st_tagged), by parity of the sequence number (local view)
    unsigned long long *flags;    // [nranks][8]  (unused since the rows validate themselves; kept in the buffer layout)
    double *const *peer_rows;     // direct: [nranks] the `rows` of every rank's buffer as mapped in this process
    unsigned long long *const *peer_flags;   // direct: [nranks] the `flags` of every rank's buffer
    unsigned long long seq;       // sequence number of this launch's flux 0, minus 1
    unsigned long long timeout_ticks;
};

template <typename T>
struct PersistArgsT {
    StageArgsT<T> s;              // rays, constants, static column tables; grp_size/row_stride
    int nsteps;
    int ngroups;                  // workgroups [g*grp_size, ...) form group g
    double *grp_part2;            // [2][workgroups][row_stride]   workgroup rows, by flux parity
    double *grp_rows2;            // [2][PERSIST_GROUPS][2*(ng-2)] group sums, by flux parity
    double *flux2;                // [2][2*(ng-2)]                 final flux row, by flux parity
    unsigned int *grp_cnt2;       // [2][PERSIST_GROUPS][TICKET_STRIDE] arrival tickets of the groups, one
                                  // 128-byte line each (zero at launch)
    unsigned int *done2;          // [2] completed groups of a flux          (zero at launch)
    unsigned int *ready;          // passes that may start: the column has been advanced with that many fluxes
                                  // (zero at launch); ready[PD_ROW] counts the fluxes whose final row is in flux2
                                  // several ranks: ready[PD_LOCAL] counts the fluxes whose LOCAL row is complete and
                                  // flux2 + 2 * 2*(ng-2) holds [2][2*(ng-2)] this rank's rows, by flux parity
    int nworkers;                 // workgroups that own rays
    int nservice;                 // 0, or ngroups reducer workgroups without rays after the workers, followed by
                                  // ONE column workgroup (then the exchange workgroup, when xch is set)
    double *shtab;                // [2][ng-2] double4: shear table of a pass, by flux parity (column workgroup)
    unsigned int opts;            // PERSIST_OPT_*
    int *status;                  // 0 ok, 1 a wait timed out
    int *status_host;             // the same word in host-mapped memory (written once when a wait times out; read by the
                                  // host after the launch without a device-to-host copy), or nullptr
    double *fcarry;               // [2*(ng-2)] this rank's flux row of the FINAL state of the launch (written by the column
                                  // workgroup), = F_0 of the next launch when nothing touches the state in between
    int carry_in;                 // 1: F_0 is in fcarry (left by the previous launch): no deposit-only pre-pass, the reducers
                                  // start at flux 1 (reducer workgroups only)
    unsigned long long timeout_ticks;   // wall_clock64 ticks (100 MHz)
    const XchArgs *xch;           // several ranks: the node-level exchange runs (one extra workgroup) with this transport
                                  // (device memory, written once when the communicator is set up); else nullptr
    unsigned long long xch_seq;   // sequence number of this launch's flux 0, minus 1
    ColIn cin;                    // canonical column at entry
    ColOut cout;                  // canonical column at exit (workgroup 0)
    double *dudz, *dvdz, *slu, *slv;    // derived tables at exit (workgroup 0)
#ifdef MSGW_STAMP
    unsigned long long *pstamps;  // diagnostic build: [workgroups][PSTAMP_PASSES][4] wall-clock stamps
#endif
};
typedef PersistArgsT<double> PersistArgs;
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

__device__ __forceinline__ double ld_sys(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const u64_t *>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ void st_sys(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<u64_t *>(p), (u64_t)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- tagged granules.  A float64 is handed over as two naturally aligned 8-byte words {low half | tag << 32},
// {high half | tag << 32}, each written by ONE 8-byte store: a word is either the old or the new one, never torn, so
// the reader knows from the tags alone whether it holds the value it waits for.  No ordering between the words,
// between values, or against a flag is needed (MI355X_MICROARCH.md, hand-off price list: "granule").
typedef unsigned int u32_t;
template <int SCOPE>
__device__ __forceinline__ void st_tagged(u64_t *p, double v, u32_t tag)
{
    const u64_t b = (u64_t)__double_as_longlong(v), t = (u64_t)tag << 32;
    __hip_atomic_store(p, (b & 0xffffffffull) | t, __ATOMIC_RELAXED, SCOPE);
    __hip_atomic_store(p + 1, (b >> 32) | t, __ATOMIC_RELAXED, SCOPE);
}
template <int SCOPE>
__device__ __forceinline__ void ld_words(const u64_t *p, u64_t &a, u64_t &b)
{
    a = __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE);
    b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, SCOPE);
}
__device__ __forceinline__ bool words_ok(u64_t a, u64_t b, u32_t tag) { return (u32_t)(a >> 32) == tag && (u32_t)(b >> 32) == tag; }
__device__ __forceinline__ double words_value(u64_t a, u64_t b) { return __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32))); }
// one value, read until it carries `tag` (bounded; false after a time-out or when another wait has raised `status`)
template <int SCOPE>
__device__ __forceinline__ bool ld_tagged_wait(const u64_t *p, u32_t tag, u64_t a, u64_t b, const int *status,
                                               u64_t timeout_ticks, double &v)
{
    if (!words_ok(a, b, tag)) {
        const u64_t t0 = wall_clock64();
        do {
            __builtin_amdgcn_s_sleep(1);
            ld_words<SCOPE>(p, a, b);
            if (words_ok(a, b, tag)) break;
            if ((status && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ||
                wall_clock64() - t0 > timeout_ticks)
                return false;
        } while (true);
    }
    v = words_value(a, b);
    return true;
}
// n values src[2i], src[2i+1] -> sink(i, value): every thread fetches its own entries (two in flight) until they carry
// `tag`.  Returns this THREAD's success; the caller agrees on it across the workgroup (__syncthreads_or).
template <int SCOPE, typename Sink>
__device__ __forceinline__ bool sweep_tagged(const u64_t *src, int n, u32_t tag, int tid, const int *status,
                                             u64_t timeout_ticks, Sink sink)
{
    for (int i = tid; i < n; i += 2 * BLOCK) {                // (one trip for the default column's table: 396 values)
        const int i2 = i + BLOCK;
        const bool h2 = i2 < n;
        u64_t a0, b0, a1, b1;
        ld_words<SCOPE>(src + 2 * (size_t)i, a0, b0);
        ld_words<SCOPE>(src + 2 * (size_t)(h2 ? i2 : i), a1, b1);
        double v0, v1;
        if (!ld_tagged_wait<SCOPE>(src + 2 * (size_t)i, tag, a0, b0, status, timeout_ticks, v0)) return false;
        sink(i, v0);
        if (h2) {
            if (!ld_tagged_wait<SCOPE>(src + 2 * (size_t)i2, tag, a1, b1, status, timeout_ticks, v1)) return false;
            sink(i2, v1);
        }
    }
    return true;
}

// Node-level sum of one row per rank (see XchArgs).  `mine(col)` is this rank's value of column `col`, `sink(col, tot)`
// receives the sum over the ranks, added in rank order (thread tid handles columns tid, tid + BLOCK, ...: columns taller
// than one workgroup); returns false after a time-out (workgroup-uniform).  Rows travel as tagged granules (tag = the low
// half of the sequence number): a rank writes its row into the slot it owns in every rank's buffer -- one posted write
// per word and peer, nothing is read remotely -- and polls the rows of all ranks in its LOCAL buffer.  Round 2 published a
// row with release fence + sequence flag and read it after a flag poll: three more serial steps per flux on a reduce
// chain that has one pass of slack (1-rank communicator: +13 % per step).
// Slots alternate with the parity of `seq`: a rank rewrites a slot for seq + 2 only after it has finished the sum of
// seq + 1, which needed every rank's row of seq + 1, which that rank wrote after reading this slot's seq.
// `nitems` items k; item k is column colmap(k) of the rank rows (several workgroups may share a buffer, each summing
// its own columns; two of them may write the same column -- with the same value).
template <typename ColMap, typename Mine, typename Sink>
__device__ __forceinline__ bool xch_allsum(const XchArgs x, u64_t seq, int *status, int nitems, int tid, int *s_flag,
                                           ColMap colmap, Mine mine, Sink sink)
{
    constexpr int SYS = __HIP_MEMORY_SCOPE_SYSTEM;
    if (!x.direct) return xch_allsum_flags(x, seq, status, nitems, tid, s_flag, mine, sink);   // (identity column map there)
    const u32_t tag = (u32_t)seq;
    const size_t slot_off = 2 * (size_t)(seq & 1ull) * x.nranks * x.stride;           // in words
    u64_t *rows = reinterpret_cast<u64_t *>(x.rows);
    for (int k = tid; k < nitems; k += BLOCK) {
        const double m = mine(k);
        const size_t off = slot_off + 2 * ((size_t)x.rank * x.stride + colmap(k));
        if (x.direct) {                                       // my row into the slot I own in every rank's buffer
            for (int j = 0; j < x.nranks; ++j) st_tagged<SYS>(reinterpret_cast<u64_t *>(x.peer_rows[j]) + off, m, tag);
        } else {
            st_tagged<SYS>(rows + off, m, tag);
        }
    }
    bool ok = true;
    for (int k = tid; k < nitems && ok; k += BLOCK) {
        const int col = colmap(k);
        double tot = 0.0;
        for (int r0 = 0; r0 < x.nranks && ok; r0 += 8) {      // eight ranks' words in flight
            u64_t a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = min(r0 + u, x.nranks - 1);
                ld_words<SYS>(rows + slot_off + 2 * ((size_t)r * x.stride + col), a[u], b[u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (r0 + u < x.nranks && ok) {
                    double v;
                    ok = ld_tagged_wait<SYS>(rows + slot_off + 2 * ((size_t)(r0 + u) * x.stride + col), tag, a[u], b[u], status,
                                             x.timeout_ticks, v);
                    if (ok) tot = tot + v;                    // rank order
                }
            }
        }
        if (ok) sink(k, tot);
    }
    if (__syncthreads_or(ok ? 0 : 1)) {
        if (tid == 0 && status) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    return true;
}

// Wave 0: every lane r < nranks polls rank r's sequence number until it reaches `seq` (bounded).
__device__ __forceinline__ bool xch_wait_all(const u64_t *flags, int nranks, u64_t seq, int *status,
                                             u64_t timeout_ticks, int lane)
{
    bool ok = true;
    if (lane < nranks) {
        const u64_t *fl = flags + 8 * lane;
        if (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            const u64_t t0 = wall_clock64();
            while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
                __builtin_amdgcn_s_sleep(4);
                if ((status && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ||
                    wall_clock64() - t0 > timeout_ticks) {
                    ok = false;
                    break;
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // compiler-only: rows are read with system-scope loads
    return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
}

// Round 2's protocol, kept for the HOST-SEGMENT transport (rows and flags in host memory, every poll a PCIe round trip:
// there a waiting rank should poll one 8-byte flag per rank, not the rows).  Node-level sum of one row per rank (see XchArgs).  `mine(col)` is this rank's value of column `col`, `sink(col, tot)`
// receives the sum over the ranks (thread tid handles columns tid, tid + BLOCK, ...: columns taller than one workgroup);
// returns false after a time-out.  s_flag: an LDS word not used by other hand-offs.
template <typename Mine, typename Sink>
__device__ __forceinline__ bool xch_allsum_flags(const XchArgs x, u64_t seq, int *status, int ncols, int tid, int *s_flag,
                                                 Mine mine, Sink sink)
{
    const size_t slot_off = (size_t)(seq & 1ull) * x.nranks * x.stride;
    for (int col = tid; col < ncols; col += BLOCK) {
        const double m = mine(col);
        if (x.direct) {                                       // my row into the slot I own in every rank's buffer
            for (int j = 0; j < x.nranks; ++j) st_sys(x.peer_rows[j] + slot_off + (size_t)x.rank * x.stride + col, m);
        } else {
            st_sys(x.rows + slot_off + (size_t)x.rank * x.stride + col, m);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");             // system scope: every storing wave's row is out
    __syncthreads();
    if (x.direct) {
        if (tid < x.nranks) __hip_atomic_store(x.peer_flags[tid] + 8 * x.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        if (tid == 0) __hip_atomic_store(x.flags + 8 * x.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tid < 64) {
        const bool ok = xch_wait_all(x.flags, x.nranks, seq, status, x.timeout_ticks, tid);
        if (tid == 0) {
            if (!ok && status) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_flag = ok ? 1 : 0;
        }
    }
    __syncthreads();
    if (!*s_flag) return false;
    for (int col = tid; col < ncols; col += BLOCK) {
        double tot = 0.0;
        for (int r = 0; r < x.nranks; ++r) tot = tot + ld_sys(x.rows + slot_off + (size_t)r * x.stride + col);   // rank order
        sink(col, tot);
    }
    return true;
}

// Wait until *counter >= target (one lane polls, bounded), then release the workgroup.  `seen` is
// a value of the counter that lane 0 has already loaded (see persist_stage: the first poll is issued
// BEFORE the tile loads, so its result does not wait for them -- memory returns in order).
template <typename T>
__device__ __forceinline__ bool persist_wait_seen(const PersistArgsT<T> p, unsigned int target, unsigned int seen,
                                                  int *s_flag, int tid, const unsigned int *counter)
{
    if (tid == 0) {
        int ok = 1;
        if (seen > target) ok = 3;                             // bit 1: the next flux is final too, i.e. this workgroup trails
        if (seen < target) {
            ok |= 4;                                           // bit 2: this workgroup had to wait for the release
            // (the status word and the clock are looked at every 16th poll only: as a load of its own per poll the status
            // check doubles the round trips of the loop, i.e. the time a released pass goes unnoticed)
            const unsigned long long t0 = wall_clock64();
            unsigned int polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                // nap between two polls: s_sleep 2 (measured 8 / 4 / 2 / 1: config 3 33.40 / 33.35 / 33.13 / see DESIGN.md 6)
                if (p.opts & PERSIST_OPT_NAP1) __builtin_amdgcn_s_sleep(1);        // (MSGW_NAP=1 | 8: diagnostic)
                else if (p.opts & PERSIST_OPT_NAP8) __builtin_amdgcn_s_sleep(8);
                else __builtin_amdgcn_s_sleep(2);
                if ((p.opts & PERSIST_OPT_LEANPOLL) && (++polls & 15u) != 0u) continue;
                if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                    wall_clock64() - t0 > p.timeout_ticks) {
                    __hip_atomic_store(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (p.status_host) __hip_atomic_store(p.status_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
    const int r = *s_flag;
    // A workgroup that finds the NEXT flux final as well is a pass behind the others, and those are
    // waiting for it: it runs the coming pass at raised wave priority (with the last arriver reducing:
    // 77.8 -> 72.7 us per step; no gain with reducer workgroups, so only used without them).
    if (p.opts & PERSIST_OPT_PRIO) {
        if (r & 2) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
    }
    // BALANCE: of the two workgroups of a CU the older one wins the issue arbitration, finishes its tiles first and then
    // idles until the younger one -- whose loop is the period -- has caught up (tools/persist_timeline.py: tiles 6.6 vs
    // 8.5 us).  A workgroup that did not have to wait is the laggard of its CU and runs the pass at raised priority.
    if ((p.opts & PERSIST_OPT_BALANCE) && (int)blockIdx.x < p.nworkers) {   // (ray workgroups only)
        if (r & 4) setprio_rt((p.opts >> 6) & 3u);
        else setprio_rt((p.opts >> 4) & 3u);
    }
    return (r & 3) != 0;
}

template <typename T>
__device__ __forceinline__ bool persist_wait(const PersistArgsT<T> p, unsigned int target, int *s_flag, int tid,
                                             const unsigned int *counter = nullptr)
{
    if (!counter) counter = p.ready;
    unsigned int seen = 0;
    if (tid == 0) seen = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return persist_wait_seen(p, target, seen, s_flag, tid, counter);
}

// Second level: add the rows of group g (flux f) in row order into the group's row.
template <typename T>
__device__ __forceinline__ void persist_reduce_group(const PersistArgsT<T> p, int g, unsigned int f, int ncols, int tid)
{
    const StageArgsT<T> a = p.s;
    const int nb = p.nworkers;
    const int r0 = g * a.grp_size, r1 = min(nb, r0 + a.grp_size);
    const unsigned int par = f & 1u;
    const double *part = p.grp_part2 + (size_t)par * nb * a.row_stride;
    double *grow = p.grp_rows2 + ((size_t)par * PERSIST_GROUPS + g) * ncols;
    for (int col = tid; col < ncols; col += BLOCK) {          // (one trip up to 130 levels)
        const double *src = part + col;
        double acc = 0.0;
        for (int r = r0; r < r1; r += 32) {                   // row order, 32 loads in flight
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = ld_agent(src + (size_t)min(r + u, r1 - 1) * a.row_stride);
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = acc + ((r + u < r1) ? v[u] : 0.0);
        }
        st_agent(grow + col, acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// Third level: the sum of the group sums of flux f, in group order; `sink(col, tot)` receives column `col`.
template <typename T, typename Sink>
__device__ __forceinline__ void persist_sum_groups(const PersistArgsT<T> p, unsigned int f, int ncols, int tid, Sink sink)
{
    for (int col = tid; col < ncols; col += BLOCK) {
        double tot = 0.0;
        const double *src = p.grp_rows2 + (size_t)(f & 1u) * PERSIST_GROUPS * ncols + col;
        for (int r = 0; r < p.ngroups; r += 32) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = ld_agent(src + (size_t)min(r + u, p.ngroups - 1) * ncols);
#pragma unroll
            for (int u = 0; u < 32; ++u) tot = tot + ((r + u < p.ngroups) ? v[u] : 0.0);
        }
        sink(col, tot);
    }
}

// Without reducer workgroups: the last arriver of the flux's last group forms ONE final row (so that every
// workgroup reads 2*(ng-2) values instead of ngroups times that) and announces it.
template <typename T>
__device__ __forceinline__ void persist_reduce_final(const PersistArgsT<T> p, unsigned int f, int ncols, int tid)
{
    const unsigned int par = f & 1u;
    // several ranks: this is only the rank's row; the exchange workgroup turns it into the final one
    double *dst = (p.xch ? p.flux2 + 2 * ncols : p.flux2) + (size_t)par * ncols;
    persist_sum_groups(p, f, ncols, tid, [&](int col, double tot) { st_agent(dst + col, tot); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(p.xch ? p.ready + PD_LOCAL : p.ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Publish this workgroup's row of flux f, take a ticket; the LAST arriver of a group adds the group's
// rows, the reducer of the flux's last group adds the group sums (all cross-workgroup accesses are
// agent-scope atomics).  s_flag words [1], [2]: the two hand-off decisions ([0] is the wait's), so
// that a slow wave still reading one decision can never see the next one.
// With reducer workgroups (p.nservice, the default when they fit) the workgroup only takes its
// ticket.  Also measured: reductions as duties of the oldest, mostly idle ray workgroups, done inside
// their wait loop -- the reduce chain then reacts only when a reducer happens to be waiting, picks up
// ~10 us of polling latency and gates everybody (80 -> 88.6 us per step); and phase-staggering the
// dispatch rounds by up to half a pass at launch (no effect: the offsets relax within a few passes).
// PREFETCH (register-resident flavours with a column workgroup): the hand-off at a pass boundary used to be three
// serial L2 round trips and five barriers on the slowest workgroup, the one everybody waits for -- drain the row
// stores, poll `ready`, load the next pass's shear table.  For exactly that workgroup the table of the NEXT pass has
// long been published when it gets here (it trails; the lagged deposit gives the reduce chain a whole pass), so lane 0
// polls `ready` before the workgroup's last resident tile (`polled`, the latency hides behind that tile) and, when the
// next pass (index f) is already released, the table loads are issued together with the row stores: one round trip,
// two barriers, and the next pass starts on its tiles at once (returns true: table staged in `sh_dst`, wave rows
// zeroed).  The control dependency poll -> flag -> barrier -> table loads is kept, and the table slot of pass f is
// not rewritten before this workgroup has published in pass f (it is rewritten for pass f + 2, which needs that row).
template <typename T>
__device__ __forceinline__ int persist_publish(const PersistArgsT<T> p, double *rows, int ncp, int *s_flag,
                                               int tid, unsigned int f, bool try_pre = false, unsigned int polled = 0u,
                                               T *sh_dst = nullptr)
{
    const StageArgsT<T> a = p.s;
    const int ncols = 2 * ncp;
    const int b = blockIdx.x, nb = p.nworkers;
    const int g = b / a.grp_size, r0 = g * a.grp_size, r1 = min(nb, r0 + a.grp_size);
    const unsigned int par = f & 1u;
    double *part = p.grp_part2 + (size_t)par * nb * a.row_stride;
    unsigned int *ticket = p.grp_cnt2 + ((size_t)par * PERSIST_GROUPS + g) * TICKET_STRIDE;
    if (try_pre && tid == 0) s_flag[3] = (polled >= f) ? 1 : 0;   // pass f waits for ready >= f (persist_stage)
    __syncthreads();                                          // all waves' rows complete in LDS
    const bool pre = try_pre && s_flag[3] != 0;               // workgroup-uniform
    double *mine = part + (size_t)b * a.row_stride;
    for (int col = tid; col < ncols; col += BLOCK) {
        double acc = rows[col];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) acc = acc + rows[w * ncols + col];
        st_agent(mine + col, acc);
        if (try_pre) {                                        // a thread zeroes exactly the entries it has just read
#pragma unroll
            for (int w = 0; w < WAVES; ++w) rows[w * ncols + col] = 0.0;
        }
    }
    // second chance (the early poll came too soon): lane 0 looks again, the answer arrives behind the row stores at no
    // cost; a released pass then starts with its table load instead of a poll round trip (return value 2)
    unsigned int polled2 = 0u;
    if (try_pre && !pre && tid == 0) polled2 = __hip_atomic_load(p.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (pre) {                                                // table of pass f: slot (f - 1) & 1 (as in persist_stage)
        const int n4 = 4 * (a.ng - 2);
        const double *tab = p.shtab + (size_t)((f - 1u) & 1u) * n4;
        for (int i = tid; i < n4; i += 2 * BLOCK) {           // two loads in flight per trip (one trip up to 130 levels)
            const int i2 = i + BLOCK;
            const bool h2 = i2 < n4;
            const double t0 = ld_agent(tab + i), t1 = ld_agent(tab + (h2 ? i2 : i));
            sh_dst[i] = (T)t0;
            if (h2) sh_dst[i2] = (T)t1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains
    if (try_pre && !pre && tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler-only (the table is read with sc1 loads)
        s_flag[6] = (polled2 >= f) ? 1 : 0;                    // (its own word: s_flag[3] may still be being read)
    }
    __syncthreads();
    if (p.nservice) {                                         // the group's reducer workgroup takes it from here
        if (tid == 0) __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return pre ? 1 : ((try_pre && s_flag[6] != 0) ? 2 : 0);
    }
    if (tid == 0) {
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler-only: rows are read with sc1 loads
        s_flag[1] = (t % (unsigned int)(r1 - r0) == (unsigned int)(r1 - r0 - 1)) ? 1 : 0;   // (cumulative ticket: last of this flux)
    }
    __syncthreads();
    if (!s_flag[1]) return 0;
    persist_reduce_group(p, g, f, ncols, tid);
    if (tid == 0) {
        const unsigned int t2 = __hip_atomic_fetch_add(p.done2 + par, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        s_flag[2] = (t2 % (unsigned int)p.ngroups == (unsigned int)p.ngroups - 1u) ? 1 : 0;   // last group of this flux? (cumulative)
    }
    __syncthreads();
    if (!s_flag[2]) return 0;
    persist_reduce_final(p, f, ncols, tid);
    return 0;
}

// Reducer workgroup of group g (owns no rays; p.nservice of them run beside the workers): for every
// flux of the launch wait until the group's rows have all arrived and add them.  This takes ~6.5 us of serial L2 round
// trips per pass off the LAST ARRIVER, which is the workgroup everybody else is waiting for
// (tools/persist_timeline.py).  Each reducer polls its own ticket on its own cache line.
template <typename T>
__device__ __forceinline__ void persist_service(const PersistArgsT<T> p, int g, int *s_flag, int tid)
{
    const int ncols = 2 * (p.s.ng - 2);
    const unsigned int nflux = 3u * (unsigned int)p.nsteps + 1u;
    const unsigned int gsize = (unsigned int)(min(p.nworkers, (g + 1) * p.s.grp_size) - g * p.s.grp_size);
    // (with F_0 carried over from the previous launch nobody publishes flux 0: the fluxes of parity 0 then start at 2)
    const unsigned int skip0 = p.carry_in ? 1u : 0u;
    for (unsigned int f = skip0; f < nflux; ++f) {
        const unsigned int par = f & 1u;
        const unsigned int kth = (f >> 1) + 1u - (par == 0u ? skip0 : 0u);   // flux f is the kth of its parity in this launch
        unsigned int *ticket = p.grp_cnt2 + ((size_t)par * PERSIST_GROUPS + g) * TICKET_STRIDE;
        if (tid == 0) {
            int ok = 1;
            const unsigned long long t0 = wall_clock64();
            unsigned int polls = 0;
            while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gsize * kth) {
                __builtin_amdgcn_s_sleep(2);
                if ((p.opts & PERSIST_OPT_LEANPOLL) && (++polls & 15u) != 0u) continue;
                if (wall_clock64() - t0 > p.timeout_ticks ||
                    __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    __hip_atomic_store(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (p.status_host) __hip_atomic_store(p.status_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    ok = 0;
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler-only: rows are read with sc1 loads
            s_flag[par] = ok;
        }
        __syncthreads();
        if (!s_flag[par]) return;
        persist_reduce_group(p, g, f, ncols, tid);
        // (tickets and counters are CUMULATIVE over the launch -- flux f is the (f >> 1)-th of its parity -- so nothing is
        // ever re-armed: round 2 reset them with a relaxed store next to a relaxed add, ordered only by the pass in between)
        if (tid == 0) __hip_atomic_fetch_add(p.done2 + par, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the column / exchange workgroup
    }
}

// Column / exchange workgroup: wait until all group sums of flux f are there (cumulative counter) and add them
// (`sink(col, tot)`, into LDS).  Returns false after a time-out.
template <typename T, typename Sink>
__device__ __forceinline__ bool persist_take_groups(const PersistArgsT<T> p, unsigned int f, int ncols, int *s_flag,
                                                    int tid, Sink sink)
{
    const unsigned int par = f & 1u;
    const unsigned int kth = (f >> 1) + 1u - ((par == 0u && p.carry_in) ? 1u : 0u);   // (cumulative; see persist_service)
    if (!persist_wait(p, (unsigned int)p.ngroups * kth, s_flag, tid, p.done2 + par)) return false;
    persist_sum_groups(p, f, ncols, tid, sink);
    return true;
}

// The exchange workgroup (several ranks only; owns no rays): for every flux of the launch, wait for
// the rank's row, add the rows of all ranks (xch_allsum), publish the final row.
template <typename T>
__device__ __forceinline__ void persist_exchange(const PersistArgsT<T> p, int *s_flag, int tid)
{
    const int ncols = 2 * (p.s.ng - 2);
    const unsigned int nflux = 3u * (unsigned int)p.nsteps;      // the flux of the final state is not needed by anybody
    XchArgs x = *p.xch;
    x.seq = p.xch_seq;
    const double *flux_local = p.flux2 + 2 * ncols;
    double *mine = reinterpret_cast<double *>(s_flag + 16);    // the rank's row, staged in LDS behind the flag words
    for (unsigned int f = 0; f < nflux; ++f) {
        const unsigned int par = f & 1u;
        // (no reducer workgroups: the last arriver has formed the rank's row)
        if (!persist_wait(p, f + 1u, s_flag + par, tid, p.ready + PD_LOCAL)) return;
        for (int col = tid; col < ncols; col += BLOCK) mine[col] = ld_agent(flux_local + (size_t)par * ncols + col);
        // (a thread only ever reads back the columns it staged itself: no barrier needed in between)
        double *dst = p.flux2 + (size_t)par * ncols;
        if (!xch_allsum(x, x.seq + f + 1ull, p.status, ncols, tid, s_flag + 2 + par, [](int k) { return k; },
                        [&](int col) { return mine[col]; }, [&](int col, double tot) { st_agent(dst + col, tot); })) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T>
struct PersistLds {
    typename Real<T>::quad_t *sh; typename Real<T>::pair_t *rho2; T *xg, *gs;
    double *xgd, *rows;
    double4 *shd;                                 // the shear table in float64 (== sh when T = double)
    double *cu, *cv, *cqu, *cqv, *crho, *cpg;     // column replica + static rhobar, pg [2][nc]
    double *F, *u, *v, *du, *dv;                  // scratch (aliases rows)
    int *flag;
};

// shear table {dudz, slope, dvdz, slope} from the shear columns in LDS: float64 into L.shd (what the column
// workgroup publishes and writes back) and, for float32 rays, the packed float32 copy the rays interpolate in
template <typename T>
__device__ __forceinline__ void persist_pack_tables(const PersistLds<T> L, int ni, int tid)
{
    for (int i = tid; i < ni; i += BLOCK) {
        const bool in = i < ni - 1;
        const double4 t = make_double4(L.du[i], in ? column_slope(L.du, L.xgd, i) : 0.0,
                                       L.dv[i], in ? column_slope(L.dv, L.xgd, i) : 0.0);
        L.shd[i] = t;
        if constexpr (!std::is_same<T, double>::value) L.sh[i] = Real<T>::quad((T)t.x, (T)t.y, (T)t.z, (T)t.w);
    }
}

// column_q = RK stage `pstage` of column_{q-1} with the final row of flux F_{q-1}: `in_lds` = the caller has already
// put the row into L.F (pm_flux[:, 1:-1] at [p*ng + 1 + c]), else it is loaded from flux2
template <typename T>
__device__ __forceinline__ void persist_column(const PersistArgsT<T> p, const PersistLds<T> L, unsigned int q,
                                               int pstage, int tid, bool in_lds = false)
{
    const StageArgsT<T> a = p.s;
    const int ng = a.ng, ni = ng - 2, nc = ng - 1, ncp = ng - 2, ncols = 2 * ncp;
    if (!in_lds)
        for (int col = tid; col < ncols; col += BLOCK) {
            const int pp = col / ncp, c = col - pp * ncp;
            L.F[pp * ng + 1 + c] = ld_agent(p.flux2 + (size_t)((q - 1) & 1) * ncols + col);   // pm_flux[:, 1:-1] (:654)
        }
    __syncthreads();
    column_flux_ends(tid, ng, L.F);
    __syncthreads();
    for (int j = tid; j < nc; j += BLOCK) {
        double du, dv, un, vn, qu, qv;
        column_tendency(j, ng, a.f0, a.dzg, 0, L.F, L.crho[j], L.cpg[j], L.cpg[nc + j], L.cu[j], L.cv[j], du, dv);
        column_rk(pstage, a.dtc, du, dv, L.cu[j], L.cv[j], L.cqu[j], L.cqv[j], un, vn, qu, qv);
        L.cu[j] = un; L.cv[j] = vn; L.cqu[j] = qu; L.cqv[j] = qv;
    }
    __syncthreads();
    column_shear(tid, BLOCK, ng, a.dzg, L.cu, L.cv, L.du, L.dv);
    __syncthreads();
    persist_pack_tables(L, ni, tid);
    __syncthreads();
}

// The column workgroup (owns no rays; runs beside the reducer workgroups): the ONLY place where the
// mean flow advances.  For every flux of the launch: wait for its final row, apply the RK stage to
// the column replica in LDS, derive the shear table the rays interpolate in, publish the table and
// release the pass.  The ray workgroups then load 3.2 KB and pass one barrier instead of each
// repeating the update (a poll, a row load, five barriers: 3.7 us on the critical path per pass).
template <typename T>
__device__ __forceinline__ void persist_column_wg(const PersistArgsT<T> p, const PersistLds<T> L, int tid)
{
    const int ng = p.s.ng, ni = ng - 2, nc = ng - 1;
    const unsigned int nflux = 3u * (unsigned int)p.nsteps;      // the flux of the final state is unused
    const int ncols = 2 * (ng - 2);
    const int ncp = ng - 2;
    auto slot = [&](int col) { const int pp = col / ncp; return pp * ng + 1 + (col - pp * ncp); };   // pm_flux[:, 1:-1] (:654)
    XchArgs x{};
    if (p.xch) { x = *p.xch; x.seq = p.xch_seq; }
    for (unsigned int f = 0; f < nflux; ++f) {
        // pass q = f+1 is RK stage q % 3, column stage (q+2) % 3 = f % 3.  Add the reducers' group sums right here
        // (into L.F); several ranks: that is this rank's row, and the sum over the ranks follows at once (a
        // separate exchange workgroup cost one more hand-off per flux on the critical reduce chain: 46.7 vs 45.0 us
        // per step with a 1-rank communicator)
        if (f == 0u && p.carry_in) {                           // F_0 = the final flux of the previous launch
            for (int col = tid; col < ncols; col += BLOCK) L.F[slot(col)] = p.fcarry[col];
        } else if (!persist_take_groups(p, f, ncols, L.flag, tid, [&](int col, double tot) { L.F[slot(col)] = tot; })) return;
        if (p.xch && !xch_allsum(x, x.seq + f + 1ull, p.status, ncols, tid, L.flag + 4 + (f & 1u), [](int k) { return k; },
                                 [&](int col) { return L.F[slot(col)]; },
                                 [&](int col, double tot) { L.F[slot(col)] = tot; })) return;
        persist_column(p, L, f + 1u, (int)(f % 3u), tid, true);
        double *tab = p.shtab + (size_t)(f & 1u) * 4 * ni;
        const double *src = reinterpret_cast<const double *>(L.shd);
        for (int i = tid; i < 4 * ni; i += BLOCK) st_agent(tab + i, src[i]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // canonical column and derived tables at exit
    for (int i = tid; i < nc; i += BLOCK) {
        p.cout.uu[i] = L.cu[i]; p.cout.vv[i] = L.cv[i]; p.cout.q_uu[i] = L.cqu[i]; p.cout.q_vv[i] = L.cqv[i];
    }
    for (int i = tid; i < ni; i += BLOCK) {
        const double4 t = L.shd[i];
        p.dudz[i] = t.x; p.dvdz[i] = t.z;
        if (i < ni - 1) { p.slu[i] = t.y; p.slv[i] = t.w; }
    }
    // The flux of the FINAL state (published by the last pass, reduced by the reducers like any other): this rank's row
    // of it is F_0 of the next launch, which then needs no deposit-only pre-pass -- one streaming pass over all rays and
    // one trip down the reduce chain less per msgw_step call (bench.py's driver arguments time 20-step calls)
    if (p.fcarry)
        (void)persist_take_groups(p, nflux, ncols, L.flag, tid, [&](int col, double tot) { p.fcarry[col] = tot; });
}

template <typename T, int STAGE, bool SAT, bool FVEC, bool DIRECT, int NRES, bool RL>
__device__ __forceinline__ bool persist_stage(const PersistArgsT<T> p, const PersistLds<T> L, unsigned int q,
                                              long long start, long long end, int tid, int wave, int lane,
                                              TileRegs<T> (&res)[NRES > 0 ? NRES : 1], int &pre)
{
    const StageArgsT<T> a = p.s;
    const int ncp = a.ng - 2;
    // Opaque copies: without them the three inlined stage bodies share (CSE) every per-array tile
    // address and keep ~60 VGPRs of 64-bit addresses alive across the whole step.
    asm volatile("" : "+s"(start));
    asm volatile("" : "+v"(tid));
    PSTAMP(q, 0);
    TileRegs<T> cur;
    // Lane 0's first poll of `ready` is issued before its tile loads: returns are in order, so the
    // poll's result is there after one round trip while the tile's 11 loads are still in flight.
    // `pre` (workgroup-uniform): the previous pass's publish found this pass released already, staged its table and
    // zeroed the wave rows (persist_publish, PREFETCH) -- straight to the tiles
    unsigned int seen = 0;
    // pre == 2: the publish already saw this pass released (no poll), rows zeroed; pre == 0 with try_pre: rows zeroed too
    const bool try_pre = NRES > 0 && p.nservice != 0 && (p.opts & PERSIST_OPT_PREFETCH) != 0;
    if (pre == 0 && q > 0 && tid == 0) seen = __hip_atomic_load(p.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // `start` is the first STREAMED ray (the NRES resident tiles before it never leave the registers); a
    // workgroup may have no streamed tile at all (workgroup-uniform test, the arrays are padded by one tile only)
    constexpr bool CGMEM = NRES > 0 && NRES <= CGMEM_MAX_NRES && !SAT;   // see process_tiles
    if (NRES == 0 || start < end) load_tile<T, STAGE, SAT, FVEC, true, DIRECT, CGMEM>(cur, a, start, tid, end);
    if (pre != 1) {
        if (q > 0) {
            if (pre == 0) {
                if (!persist_wait_seen(p, q, seen, L.flag, tid, p.ready)) return false;
            } else if (p.opts & PERSIST_OPT_BALANCE) setprio_rt((p.opts >> 4) & 3u);   // released on arrival
            if (p.nservice) {                                  // the column workgroup has published this pass's table
                const double *tab = p.shtab + (size_t)((q - 1u) & 1u) * 4 * (a.ng - 2);
                T *dst = reinterpret_cast<T *>(L.sh);
                for (int i = tid; i < 4 * (a.ng - 2); i += BLOCK) dst[i] = (T)ld_agent(tab + i);
                __syncthreads();
            } else {
                persist_column(p, L, q, (STAGE + 2) % 3, tid);
            }
        }
        PSTAMP(q, 1);
        if (!try_pre) {                                        // (else the previous publish has zeroed them)
            for (int i = tid; i < WAVES * 2 * ncp; i += BLOCK) L.rows[i] = 0.0;
            __syncthreads();
        }
    } else {
        if (p.opts & PERSIST_OPT_BALANCE) setprio_rt((p.opts >> 8) & 3u);
        PSTAMP(q, 1);
    }
    int wmin = INT_MAX, wmax = INT_MIN;
    const StageLds<T> SL{L.sh, L.rho2, L.xg, L.gs, L.rows};
    // PREFETCH (persist_publish): lane 0 polls `ready` before the workgroup's last resident tile
    unsigned int polled = 0u;
    process_tiles<T, STAGE, SAT, FVEC, true, DIRECT, true, NRES, RL>(a, SL, cur, start, end, tid, wave, lane, wmin, wmax, &res,
                                                                     try_pre ? p.ready : nullptr, &polled);
    PSTAMP(q, 2);
    // this pass produced state_{q+1}: publish F_{q+1}
    pre = persist_publish(p, L.rows, ncp, L.flag, tid, q + 1u, try_pre, polled, reinterpret_cast<T *>(L.sh));
    PSTAMP(q, 3);
    return true;
}

// NRES > 0: the first NRES tiles of every ray workgroup are RESIDENT IN REGISTERS for the whole launch
// (loaded once, written back once): at 2 workgroups per CU a lane has 256 VGPRs, enough for two tiles'
// state (9 arrays x 2 rays per lane each) beside the working set, and the HBM traffic of a pass drops by
// NRES / tiles_per_block.  Same tile order, same arithmetic: the deposit order is still ray order.
// RL: the MSGW_RELAUNCH extension (BASELINE config 5) as a compile-time variant, so that the reference-parity
// kernels carry none of its registers.
// LEAN (float64, tall columns, reducer workgroups on; chosen by the host when it lets more workgroups share a CU): the
// column workgroup's replica lives in the LDS the ray workgroups use for their wave rows and density table (it needs
// neither), rhobar and the pressure gradient are read from global memory: 128 instead of 184 B of LDS per level.  A
// compile-time variant: as a run-time option its pointer selection cost the default column 1.4 % (register allocation).
template <typename T, bool SAT, bool FVEC, bool DIRECT, int NRES = 0, bool RL = false, bool LEAN = false>
__global__ void __launch_bounds__(BLOCK, NRES > 0 ? 2 : 4) k_rk3_persist(const PersistArgsT<T> p)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int TILE = Real<T>::TILE;
    constexpr int RPT = Real<T>::RPT;
    const StageArgsT<T> a = p.s;
    const int ng = a.ng, ni = ng - 2, nc = ng - 1, ncp = ng - 2;
    const StageCarve<T> C(lds, ng);
    PersistLds<T> L;
    L.sh = C.sh; L.rho2 = C.rho2; L.xg = C.xg; L.gs = C.gs; L.xgd = C.xgd; L.rows = C.rows;
    static_assert(!LEAN || std::is_same<T, double>::value, "the lean LDS layout is float64 only");
    if constexpr (LEAN) {
        // wave-row region [8 ncp]: F [2 ng], then cu, cv, cqu, cqv [4 nc]; density-table region [2 nc]: du, dv [2 ni]
        L.cu = L.rows + 2 * ng; L.cv = L.cu + nc; L.cqu = L.cv + nc; L.cqv = L.cqu + nc;
        L.crho = const_cast<double *>(a.c.rhobar); L.cpg = const_cast<double *>(a.pg);
        L.flag = reinterpret_cast<int *>(L.rows + WAVES * 2 * ncp);
    } else {
        double *colrep = L.rows + WAVES * 2 * ncp;
        L.cu = colrep; L.cv = L.cu + nc; L.cqu = L.cv + nc; L.cqv = L.cqu + nc; L.crho = L.cqv + nc; L.cpg = L.crho + nc;
        L.flag = reinterpret_cast<int *>(L.cpg + 2 * nc);
    }
    if constexpr (std::is_same<T, double>::value) L.shd = reinterpret_cast<double4 *>(L.sh);
    else                                                       // 32 bytes of flags, then [ni] double4 (32-B aligned)
        L.shd = reinterpret_cast<double4 *>((reinterpret_cast<uintptr_t>(L.flag + 8) + 31) & ~(uintptr_t)31);
    L.F = C.F; L.u = C.u; L.v = C.v; L.du = C.du; L.dv = C.dv;
    if constexpr (LEAN) { L.du = reinterpret_cast<double *>(C.rho2); L.dv = L.du + ni; }

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#ifdef MSGW_STAMP
    const unsigned int hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
    const unsigned int xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
    if (tid == 0 && p.pstamps) {                               // where does this workgroup run?
        p.pstamps[((size_t)blockIdx.x * PSTAMP_PASSES + (PSTAMP_PASSES - 1)) * 4 + 0] = hw_id;
        p.pstamps[((size_t)blockIdx.x * PSTAMP_PASSES + (PSTAMP_PASSES - 1)) * 4 + 1] = xcc_id;
    }
#endif
    const int role = (int)blockIdx.x - p.nworkers;             // >= 0: a workgroup without rays
    if (role >= 0) {
        __builtin_amdgcn_s_setprio(3);                         // they only ever poll and reduce: react at once
        if (role < p.nservice) { persist_service(p, role, reinterpret_cast<int *>(lds), tid); return; }
        if (!p.nservice) { persist_exchange(p, reinterpret_cast<int *>(lds), tid); return; }   // (several ranks, no reducers)
    }                                                          // role == nservice > 0: the column (+ exchange) workgroup, below
    const long long start = (long long)blockIdx.x * a.rays_per_block;
    const long long end = min(a.n, start + a.rays_per_block);
    // (Wave priority by dispatch round, youngest highest, was measured: it inverts which workgroup of
    // a CU finishes last -- the oldest instead of the youngest -- and leaves the period unchanged: the
    // CU's drain time plus the last workgroup's hand-off is what counts.)

    stage_xg(C, a.c, ni, tid);
    if constexpr (!LEAN) stage_rho(C, a.c, nc, tid, true);
    for (int i = tid; i < nc; i += BLOCK) {
        L.cu[i] = p.cin.uu[i]; L.cv[i] = p.cin.vv[i]; L.cqu[i] = 0.0; L.cqv[i] = 0.0;
        if constexpr (!LEAN) { L.crho[i] = a.c.rhobar[i]; L.cpg[i] = a.pg[i]; L.cpg[nc + i] = a.pg[nc + i]; }
    }
    __syncthreads();
    column_shear(tid, BLOCK, ng, a.dzg, L.cu, L.cv, L.du, L.dv);
    __syncthreads();
    persist_pack_tables(L, ni, tid);
    __syncthreads();
    if (role >= 0) {
        persist_column_wg(p, L, tid);
        return;
    }
    if constexpr (LEAN) stage_rho(C, a.c, nc, tid, true);     // (after the tables: du, dv lived in the density table's LDS)

    // deposit-only pre-pass: F_0 = wave_projection(state_0) -- unless the previous launch left it behind (carry_in: the
    // cg_rr of the streamed tiles of the two-resident-tile flavours is in memory too, stored by that launch's last pass)
    {
        for (int i = tid; i < WAVES * 2 * ncp; i += BLOCK) L.rows[i] = 0.0;
        __syncthreads();
        if (!p.carry_in) {
            const StageLds<T> SL{L.sh, L.rho2, L.xg, L.gs, L.rows};
            // (with resident tiles: also leaves cg_rr of the initial state of the streamed tiles in memory)
            deposit_pass<T, FVEC, (NRES > 0 && NRES <= CGMEM_MAX_NRES && !SAT)>(a, SL, start, end, tid, wave, lane, start + (long long)NRES * TILE);
            persist_publish(p, L.rows, ncp, L.flag, tid, 0u);
        }
        // flavours with the pass-boundary prefetch: every later publish leaves the wave rows zeroed for the next pass
        if (NRES > 0 && p.nservice != 0 && (p.opts & PERSIST_OPT_PREFETCH) != 0) {
            for (int i = tid; i < WAVES * 2 * ncp; i += BLOCK) L.rows[i] = 0.0;
            __syncthreads();
        }
    }
    // resident tiles: everything a stage may read (stage 2 of the DIRECT variant reads the most); lanes
    // beyond the workgroup's rays hold inert values and are never deposited or stored
    TileRegs<T> res[NRES > 0 ? NRES : 1];
    if constexpr (NRES > 0) {
#pragma unroll
        for (int i = 0; i < NRES; ++i) {
            const long long base = start + (long long)i * TILE;
            if (base < end) {                                  // workgroup-uniform
                load_tile<T, 2, SAT, FVEC, true, DIRECT>(res[i], a, base, tid, end);
#pragma unroll
                for (int r = 0; r < RPT; ++r) {                // cg_rr of the initial state (carried from then on)
                    const T f = FVEC ? res[i].ff[r] : a.f_uni;
                    T kh2, m2, vk2, om;
                    dispersion(res[i].kk[r], res[i].ll[r], res[i].mm[r], f * f, a.bvf2, kh2, m2, vk2, om, res[i].cg[r]);
                }
            } else {
                TileRegs<T> z{};
#pragma unroll
                for (int r = 0; r < RPT; ++r) { z.mm[r] = T(1); z.kk[r] = T(1); z.drr[r] = T(1); z.pvf[r] = T(1); z.v[r] = false; }
                res[i] = z;
            }
        }
    }
    const long long sstart = start + (long long)NRES * TILE;    // first streamed ray
    unsigned int q = 0;
    int pre = 0;
    for (int step = 0; step < p.nsteps; ++step) {
        if (!persist_stage<T, 0, SAT, FVEC, DIRECT, NRES, RL>(p, L, q, sstart, end, tid, wave, lane, res, pre)) return;
        ++q;
        if (!persist_stage<T, 1, SAT, FVEC, DIRECT, NRES, RL>(p, L, q, sstart, end, tid, wave, lane, res, pre)) return;
        ++q;
        if (!persist_stage<T, 2, SAT, FVEC, DIRECT, NRES, RL>(p, L, q, sstart, end, tid, wave, lane, res, pre)) return;
        ++q;
    }
    if constexpr (NRES > 0) {                                  // write the resident tiles back
#pragma unroll
        for (int i = 0; i < NRES; ++i) {
            if (res[i].v[0]) {
                storev(a.r.rr(), res[i].off, res[i].rr);
                storev(a.r.mm(), res[i].off, res[i].mm);
                if (SAT || DIRECT || RL) storev(a.r.dens(), res[i].off, res[i].dens);
            }
        }
    }
    if (blockIdx.x != 0 || p.nservice) return;
    // (no column workgroup:) workgroup 0 applies the last update (column_q needs F_{q-1}) and writes the column back;
    // F_q, published by the last pass, is not used (the next call starts with its own pre-pass)
    if (!persist_wait(p, q, L.flag, tid)) return;
    persist_column(p, L, q, 2, tid);
    for (int i = tid; i < nc; i += BLOCK) {
        p.cout.uu[i] = L.cu[i]; p.cout.vv[i] = L.cv[i]; p.cout.q_uu[i] = L.cqu[i]; p.cout.q_vv[i] = L.cqv[i];
    }
    for (int i = tid; i < ni; i += BLOCK) {
        const double4 t = L.shd[i];
        p.dudz[i] = t.x; p.dvdz[i] = t.z;
        if (i < ni - 1) { p.slu[i] = t.y; p.slv[i] = t.w; }
    }
}

// arguments of the exchange self-test (k_xch_selftest, misc_kernels.h)
struct XchTestArgs {
    XchArgs x;                    // seq: the first round's sequence number minus 1
    int rounds;
    int *result;                  // 1 = every round summed to the expected value
};

}   // namespace msgw
