// C-ABI host side of the MI355X ray-propagation library (see include/msgwam_hip.h).
// One context per GPU: SoA ray state + column resident in HBM, one HIP stream,
// optional hipGraph of the RK3 step, optional RCCL communicator (dlopen'ed so
// that the single-GPU path has no RCCL dependency).
#include "msgwam_hip.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <type_traits>

#include "kernel_table.h"
#include "column_kernels.h"
#include "chain_kernels.h"
#include "persist_kernel.h"
#include "ray_kernels.h"

using namespace msgw;

namespace {

std::string g_create_error;

// ---- minimal RCCL surface (resolved with dlsym; mirrors rccl.h) -------------
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclInt32 = 2, ncclFloat64 = 8, ncclSum = 0, ncclMin = 3 };   // ncclDataType_t / ncclRedOp_t values in rccl.h
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string load_error;
    bool load()
    {
        if (handle) return true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) { load_error = std::string("dlopen librccl failed: ") + dlerror(); return false; }
        GetUniqueId = (decltype(GetUniqueId))dlsym(handle, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(handle, "ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        AllReduce = (decltype(AllReduce))dlsym(handle, "ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce) {
            load_error = "librccl is missing ncclGetUniqueId/ncclCommInitRank/ncclAllReduce";
            return false;
        }
        return true;
    }
} g_rccl;

}   // namespace

// Head of the node-level exchange segment (POSIX shared memory, one per communicator).  The host
// side uses the counters to agree on the outcome of the set-up; the GPUs use flags[] and rows[].
constexpr int XCH_MAX_RANKS = 64;                               // one polling lane per rank
struct XchHeader {
    unsigned int magic;           // written last by rank 0
    unsigned int nranks;
    unsigned int arrived[4];      // host barriers of the set-up
    unsigned int okay[4];         // ranks that reached the barrier without an error
    unsigned long long agree_cnt[XCH_MAX_RANKS];      // per-call agreement (xch_agree_step): call counter of a rank
    unsigned int agree_val[XCH_MAX_RANKS][2];         // ... and its vote, by parity of the counter
};
static_assert(sizeof(XchHeader) <= 4096, "header page");
struct XchPeerSlot {              // one per rank, at XCH_PEERS_OFF: what the device-resident transport needs
    char handle[64];              // hipIpcMemHandle_t of the rank's exchange buffer
    char pci[16];                 // PCI id of the rank's device ("dddd:bb:dd"): ranks that share a device
    unsigned int has_handle;
    unsigned int pad_[11];
};
static_assert(sizeof(XchPeerSlot) == 128, "slot size");
constexpr unsigned int XCH_MAGIC = 0x4d534759u;                 // "MSGY"
constexpr size_t XCH_PEERS_OFF = 4096;                          // [XCH_MAX_RANKS] XchPeerSlot
constexpr size_t XCH_FLAGS_OFF = 4096 + 128 * XCH_MAX_RANKS;    // [nranks][8] u64, one 64-byte line per rank
constexpr size_t XCH_SCRATCH_BYTES = 4096;                      // device scratch: [0..63] ints, [64..] XchArgs, [1024..] peer pointers
constexpr size_t XCH_SCRATCH_PEERS = 1024;
constexpr unsigned long long XCH_TIMEOUT_TICKS = 2000000000ull; // 20 s of wall clock (100 MHz): ranks start apart
constexpr int XCH_TEST_ROUNDS = 6;
constexpr size_t PDONE_WORDS = 128 + (size_t)2 * msgw::PERSIST_GROUPS * msgw::TICKET_STRIDE;   // counters of the persistent kernel

struct msgw_ctx {
    int device = 0;
    int ncu = 256;
    int64_t cap = 0, n = 0;
    int ng = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;

    // config
    bool have_config = false, have_column = false, have_rays = false;
    double bvf = 0, f0 = 0, kappa = 0;
    int sat_online = 0;

    // rays: float64 (default) or float32 (MSGW_DTYPE_F32) SoA arrays
    int f32 = 0;
    size_t esz = sizeof(double);     // bytes per element of the ray arrays
    int tile = Real<double>::TILE;   // rays per workgroup iteration (2 rays per lane)
    std::vector<void *> ray_bufs;    // further per-ray allocations (the HPROP arrays)
    char *slab = nullptr;            // ONE allocation for the A_COUNT per-ray arrays, array k at slab + k * pitch
    size_t pitch = 0;                // bytes between two arrays (a multiple of 256)
    void *dens = nullptr, *rr = nullptr, *mm = nullptr, *drr = nullptr, *kk = nullptr, *ll = nullptr,
         *dmm = nullptr, *vol = nullptr, *fray = nullptr, *pvf = nullptr, *q_rr = nullptr,
         *q_mm = nullptr, *q_dens = nullptr, *rr0 = nullptr, *mm0 = nullptr,
         *src_dens = nullptr, *src_rr = nullptr, *src_mm = nullptr,   // MSGW_RELAUNCH: values at upload
         *cgbuf = nullptr;                                            // cg_rr carried between passes (persistent kernel)
    double relaunch_frac = 1e-6;
    // HPROP_GLOBAL = True (horizontal propagation): lam, phi and their RK registers, registers of kk, ll
    int hprop = 0;
    bool have_hprop = false;
    void *lam = nullptr, *phi = nullptr, *q_lam = nullptr, *q_phi = nullptr, *q_kk = nullptr, *q_ll = nullptr;   // (ray type)
    bool fvec = false;
    double f_uni = 0;
    // EXTENSION: N as a column on grids (msgw_set_bvf_column): drr, dmm evolve too; a per-stage kernel of its own
    int nz = 0;
    bool have_nz = false;            // dkdl / area of the resident rays are in place
    double *bvfcol = nullptr;        // [ng-1] N on grids (float64 in both modes, like the rest of the column)
    void *nz_q_drr = nullptr, *nz_q_dmm = nullptr, *nz_dkdl = nullptr, *nz_area = nullptr, *nz_drr0 = nullptr;   // (ray type)

    // column
    double *colbuf = nullptr;
    double *grid = nullptr, *grids = nullptr, *rhobar = nullptr, *pg = nullptr, *uu = nullptr,
           *vv = nullptr, *q_uu = nullptr, *q_vv = nullptr, *dudz = nullptr, *dvdz = nullptr,
           *slu = nullptr, *slv = nullptr, *slrho = nullptr, *flux = nullptr, *out_du = nullptr,
           *out_dv = nullptr, *out_flux = nullptr;
    double *alt_uu = nullptr, *alt_vv = nullptr, *alt_q_uu = nullptr, *alt_q_vv = nullptr;   // 2nd column set
    double dzg = 0, dzs = 0, xg0 = 0, gs0 = 0, xg_last = 0, gs_last = 0, z_bot = 0, z_top = 0;

    // launch geometry + per-workgroup flux rows
    int blocks_per_cu = 4;
    int blocks = 0, tiles_per_block = 0;
    int64_t rays_per_block = 0;
    double *partial = nullptr;
    size_t partial_elems = 0;
    int *ranges = nullptr;
    int ranges_cap = 0;
    double *row2 = nullptr;          // [RED1_GROUPS][ncols] second-level flux rows
    size_t row2_elems = 0;
    // in-kernel group reduction of the RK-stage kernels
    double *grp_part = nullptr, *grp_rows = nullptr;
    unsigned int *grp_cnt = nullptr;
    size_t grp_part_elems = 0;
    int grp_size = 1, ngroups = 1, row_stride = 0;
    bool groupred = false;           // set while enqueueing fused stages
    bool lagchain = false;           // set while enqueueing the lagged launch chain (collectives overlap)
    hipStream_t stream2 = nullptr;   // reduce + all-reduce of the next flux run here, beside the next K1
    hipEvent_t ev_rows[2] = {nullptr, nullptr}, ev_flux[2] = {nullptr, nullptr};
    // persistent RK3 kernel (single rank, coupled)
    int persist = 1;                 // 0 disables (MSGW_PERSIST=0 or after a time-out)
    int service = 1;                 // reducer workgroups beside the workers (MSGW_SERVICE=0: last arriver reduces)
    int balance = 1;                 // laggard workgroups of a CU raise their wave priority (MSGW_BALANCE=0 | 1)
    int prefetch = 1;                // early poll + table prefetch at the pass boundary of the resident-tile flavours (MSGW_PREFETCH=0 | 1)
    int fixed_narrow_force = -1;         // MSGW_FIXED_NARROW=0 | 1 (diagnostic), read when the context is created
    int64_t fixed_narrow_max = 400000;   // fixed background: ray counts up to this run one ray per lane (launch_fixed)
    int regtiles = 4;                // most register-resident tiles per workgroup in the persistent kernel (MSGW_REGTILES=0 | 2 | 4)
    double *grp_rows2 = nullptr;     // [2][PERSIST_GROUPS][ncols]
    unsigned int *pdone = nullptr;   // PDONE_WORDS: [0] ready, [32..33] done2, [64] final rows, [96] rank rows, [128..] group tickets
    int *pstatus = nullptr;          // raised by a persistent launch whose bounded wait timed out; sticky until the next
                                     // msgw_upload_rays, read back at the next blocking call (check_status)
    bool status_armed = false;       // a persistent launch has been enqueued since the last check
    int *pstatus_host = nullptr;     // host-mapped copy of the status word (hipHostMalloc): read after a sync, no D2H copy
    int *pstatus_host_dev = nullptr; // ... as the kernel sees it
    double *fcarry = nullptr;        // [2*(ng-2)] flux row of the final state of the last persistent launch (persist_kernel.h)
    int carry = 1;                   // MSGW_CARRY=0 (diagnostic, read when the context is created): always run the pre-pass
    int coop = 1;                    // launch the persistent kernel with hipLaunchCooperativeKernel where the device has it (one rank per device; MSGW_COOP=0: plain launch)
    unsigned long long carry_key = 0;    // != 0: fcarry is F_0 of the resident state for a launch of this flavour key
    unsigned long long plan_key = 0;     // the cached launch plan (plan_persist costs three occupancy queries per call)
    std::vector<char> plan_blob;
    double *flux2 = nullptr;         // [2][ncols] final flux rows of the persistent kernel
    double *shtab = nullptr;         // [2][ng-2] double4 shear tables published by the column workgroup
    unsigned long long *pstamps = nullptr;   // diagnostic builds only
    double *grp_part2 = nullptr;     // [2][blocks][row_stride]
    size_t grp_part2_elems = 0;

    // graph
    int graph_steps = 0;
    hipGraphExec_t gexec = nullptr;
    hipGraph_t graph = nullptr;
    double g_dt = 0;
    unsigned g_flags = 0;
    int64_t g_n = -1;
    int g_steps = 0;

    // rccl
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    bool force_coll = false;
    // node-level in-kernel exchange of the persistent kernel: POSIX shared memory, registered with HIP
    struct XchHeader *xch_hdr = nullptr;      // host mapping of the whole segment
    size_t xch_bytes = 0;
    bool xch_registered = false;
    double *xch_rows = nullptr;               // device view of the rank rows
    unsigned long long *xch_flags = nullptr;  // device view of the sequence numbers
    int xch_stride = 0;
    unsigned long long xch_seq = 0;           // sequence number of the newest flux exchanged
    bool xch_ok = false;                      // agreed by all ranks after the self-test
    int *xch_scratch = nullptr;               // device scratch: self-test result / agreement buffer / XchArgs / peer pointers
    // device-resident transport: this rank's buffer in its own HBM + the other ranks' buffers mapped through HIP IPC
    void *xch_local = nullptr;
    std::vector<void *> xch_peer_open;        // mappings to close
    double *const *xch_peer_rows = nullptr;   // device arrays of the peers' row / flag pointers (in xch_scratch)
    unsigned long long *const *xch_peer_flags = nullptr;
    bool xch_direct = false;
    int tenants = 1;                          // ranks of this communicator that share this rank's physical device
    unsigned long long xch_calls = 0;         // per-call agreements done (xch_agree_step)
    char pci_id[16] = {0};

    // snapshots of the evolving slots (msgw_snapshot_*): released buffers are kept for reuse
    std::vector<std::pair<size_t, void *>> snap_pool;
    // counters / kernel timing
    msgw_counters_t cnt{};
    std::vector<hipEvent_t> kev;
    size_t kev_used = 0;
    bool time_next = false;          // attach begin/end events to the ray kernels being enqueued
};

namespace {

void xch_teardown(msgw_ctx *c);

int fail(msgw_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((c), MSGW_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)

size_t stage_lds_bytes(const msgw_ctx *c)
{
    const int ng = c->ng, ni = ng - 2, nc = ng - 1, ncp = ng - 2;
    // tables in the ray type (rounded up to 16 B), a float64 copy of grid[1:-1] for float32 rays, float64 wave rows
    const size_t tables = ((c->esz * (size_t)(5 * ni + 3 * nc) + 15) & ~(size_t)15) + (c->f32 ? sizeof(double) * ni : 0);
    return tables + sizeof(double) * (size_t)(WAVES * 2 * ncp) + sizeof(int) * 2 * WAVES + 16;
}
size_t proj_lds_bytes(int nG, int np)
{
    return sizeof(double) * (size_t)(nG + WAVES * np * (nG - 1)) + sizeof(int) * 2 * WAVES + 16;
}
size_t col_lds_bytes(int ng, int nseg, int ncols, int nblocks)
{
    return sizeof(double) * (size_t)(2 * ng + 2 * (ng - 1) + 2 * (ng - 2) + (size_t)nseg * ncols) +
           sizeof(int) * 2 * (size_t)nblocks + 16;
}
constexpr int RED1_GROUPS = FUSE_ROWS;      // first-level reduce workgroups
constexpr int RED1_MIN_ROWS = 128;   // below this the single-workgroup column kernel reads the rows itself

int pick_nseg(int ncols)
{
    int s = COL_BLOCK / (ncols > 0 ? ncols : 1);
    return s < 1 ? 1 : (s > 8 ? 8 : s);
}

int ensure_lds(msgw_ctx *c, const void *kernel, size_t bytes)
{
    if (!kernel)
        return fail(c, MSGW_ERR_UNSUP, "this kernel variant is not built (see kernel_table.h)");
    if (bytes > 160 * 1024)
        return fail(c, MSGW_ERR_UNSUP, "column too large for LDS staging (%zu B needed, ngrid=%d)", bytes, c->ng);
    if (bytes > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MSGW_OK;
}

// Launch geometry: every workgroup owns `rays_per_block` contiguous rays, a whole number of
// tiles (512 rays), at most ncu*blocks_per_cu workgroups.  Measured and dropped:
//  * a finer split of whole tiles that balances the workgroups exactly over the CUs (1009 x 992 rays instead of
//    977 x 1024 at 1e6 rays): no gain for the per-stage kernels, 4 % slower for the persistent kernel, whose
//    synchronisation cost grows with the number of workgroups (round 1);
//  * ranges of whole WAVE quanta (a quarter tile) with wavefronts skipping the empty part of a workgroup's last
//    tile, so that 1.25e6 float32 rays make 1024 workgroups of ~1.2 tiles instead of 611 of two: 59 -> 71 us per
//    step with all rays streamed, 113 -> 129 us for the launch chain (more workgroups to synchronise, and a lone
//    wavefront in a partly empty tile has nothing to hide its latency behind).  The same for the resident flavour
//    only, with the number of workgroups unchanged (494 ranges of 2.5 instead of 407 of 3 float32 tiles, idle
//    wavefronts skipping the tile body): no gain either (config 5 59.2 vs 58.5, 1.2e6 float64 rays 55.1 vs 55.0 us):
//    a pass lasts as long as a wavefront's serial chain over its tiles, not as the CU's summed work.
void split_rays(const msgw_ctx *c, int64_t n, int64_t maxb, int64_t *rays_per_block, int *tiles_per_block, int *blocks)
{
    const int64_t ntiles = (n + c->tile - 1) / c->tile;
    if (maxb < 1) maxb = 1;
    int64_t tpb = (ntiles + maxb - 1) / maxb;
    if (tpb < 1) tpb = 1;
    *tiles_per_block = (int)tpb;
    *rays_per_block = tpb * c->tile;
    *blocks = (int)std::max<int64_t>((ntiles + tpb - 1) / tpb, 1);
}
void geometry(msgw_ctx *c, int64_t n)
{
    split_rays(c, n, (int64_t)c->ncu * c->blocks_per_cu, &c->rays_per_block, &c->tiles_per_block, &c->blocks);
}

void drop_graph(msgw_ctx *c)
{
    if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
    c->g_n = -1;
}

int ensure_partial(msgw_ctx *c, int blocks, size_t row_elems)
{
    const size_t need = (size_t)blocks * row_elems;
    if (need > c->partial_elems) {
        drop_graph(c);                               // a captured graph holds the old pointer
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->partial) HIPCHK(c, hipFree(c->partial));
        c->partial = nullptr;
        HIPCHK(c, hipMalloc(&c->partial, need * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(c->partial, 0, need * sizeof(double), c->stream));
        c->partial_elems = need;
    }
    if (blocks > c->ranges_cap) {
        drop_graph(c);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->ranges) HIPCHK(c, hipFree(c->ranges));
        c->ranges = nullptr;
        HIPCHK(c, hipMalloc(&c->ranges, sizeof(int) * 2 * (size_t)blocks));
        HIPCHK(c, hipMemsetAsync(c->ranges, 0, sizeof(int) * 2 * (size_t)blocks, c->stream));
        c->ranges_cap = blocks;
    }
    const size_t need2 = (size_t)RED1_GROUPS * row_elems;
    if (need2 > c->row2_elems) {
        drop_graph(c);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->row2) HIPCHK(c, hipFree(c->row2));
        c->row2 = nullptr;
        HIPCHK(c, hipMalloc(&c->row2, need2 * sizeof(double)));
        c->row2_elems = need2;
    }
    return MSGW_OK;
}

// buffers of the in-kernel group reduction for the current launch geometry
int ensure_groups(msgw_ctx *c)
{
    const int ncols = 2 * (c->ng - 2);
    c->row_stride = ((ncols + 15) / 16) * 16;                  // 128-B multiple: no line shared by two rows
    c->grp_size = (c->blocks + FUSE_ROWS - 1) / FUSE_ROWS;
    c->ngroups = (c->blocks + c->grp_size - 1) / c->grp_size;
    const size_t need = (size_t)c->blocks * c->row_stride;
    if (need > c->grp_part_elems) {
        drop_graph(c);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->grp_part) HIPCHK(c, hipFree(c->grp_part));
        c->grp_part = nullptr;
        HIPCHK(c, hipMalloc(&c->grp_part, need * sizeof(double)));
        c->grp_part_elems = need;
    }
    const size_t need2 = (size_t)2 * 2048 * c->row_stride;      // persistent kernel: <= 2048 ray workgroups
    if (need2 > c->grp_part2_elems) {
        drop_graph(c);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->grp_part2) HIPCHK(c, hipFree(c->grp_part2));
        c->grp_part2 = nullptr;
        HIPCHK(c, hipMalloc(&c->grp_part2, need2 * sizeof(double)));
        c->grp_part2_elems = need2;
    }
    if (!c->grp_rows) {
        HIPCHK(c, hipMalloc(&c->grp_rows, sizeof(double) * (size_t)2 * FUSE_ROWS * 2 * (c->ng - 2)));   // x2: flux parity
        HIPCHK(c, hipMalloc(&c->grp_cnt, sizeof(unsigned int) * 64));
        HIPCHK(c, hipMalloc(&c->grp_rows2, sizeof(double) * (size_t)2 * PERSIST_GROUPS * 2 * (c->ng - 2)));
        HIPCHK(c, hipMalloc(&c->pdone, sizeof(unsigned int) * PDONE_WORDS));
        HIPCHK(c, hipMalloc(&c->pstatus, 128));
        HIPCHK(c, hipMemsetAsync(c->pstatus, 0, 128, c->stream));
        HIPCHK(c, hipMalloc(&c->fcarry, sizeof(double) * (size_t)2 * (c->ng - 2)));
        if (hipHostMalloc(reinterpret_cast<void **>(&c->pstatus_host), 128, hipHostMallocMapped) == hipSuccess) {
            std::memset(c->pstatus_host, 0, 128);
            if (hipHostGetDevicePointer(reinterpret_cast<void **>(&c->pstatus_host_dev), c->pstatus_host, 0) != hipSuccess) {
                (void)hipGetLastError(); (void)hipHostFree(c->pstatus_host);
                c->pstatus_host = nullptr; c->pstatus_host_dev = nullptr;
            }
        } else {
            (void)hipGetLastError(); c->pstatus_host = nullptr;
        }
        HIPCHK(c, hipMalloc(&c->flux2, sizeof(double) * (size_t)4 * 2 * (c->ng - 2)));   // [2] final + [2] this rank's
        HIPCHK(c, hipMalloc(&c->shtab, sizeof(double) * (size_t)2 * 4 * (c->ng - 2)));
    }
    HIPCHK(c, hipMemsetAsync(c->grp_cnt, 0, sizeof(unsigned int) * 64, c->stream));
    return MSGW_OK;
}

hipEvent_t *timing_events(msgw_ctx *c)
{
    if (c->kev_used + 2 > c->kev.size()) {
        const size_t old = c->kev.size();
        c->kev.resize(old + 512);
        for (size_t i = old; i < c->kev.size(); ++i)
            if (hipEventCreate(&c->kev[i]) != hipSuccess) { c->kev.resize(i); return nullptr; }
    }
    hipEvent_t *p = &c->kev[c->kev_used];
    c->kev_used += 2;
    return p;
}

// ---- kernel dispatch --------------------------------------------------------
// Kernels live in other translation units (kernel_table.h) and are launched through their entry point.
// Ray kernels go through hipExtLaunchKernel so that, when asked, a pair of events receives the kernel's
// own begin/end timestamps (no launch gap in the interval).
template <typename Args>
int launch_struct(msgw_ctx *c, const void *fn, unsigned grid, unsigned block, size_t lds, const Args &a,
                  hipStream_t stream = nullptr, bool timed = false)
{
    if (int rc = ensure_lds(c, fn, lds)) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        hipEvent_t *ev = timing_events(c);
        if (ev) { e0 = ev[0]; e1 = ev[1]; }
    }
    void *args[] = {const_cast<Args *>(&a)};
    HIPCHK(c, hipExtLaunchKernel(fn, dim3(grid), dim3(block), args, lds, stream ? stream : c->stream, e0, e1, 0));
    return MSGW_OK;
}
// kernels with a plain parameter list; the argument types must be the kernel's parameter types exactly
template <typename... A>
int launch_list(msgw_ctx *c, const void *fn, unsigned grid, unsigned block, A... a)
{
    if (!fn) return fail(c, MSGW_ERR_UNSUP, "this kernel variant is not built (see kernel_table.h)");
    void *args[] = {(void *)&a...};
    HIPCHK(c, hipLaunchKernel(fn, dim3(grid), dim3(block), args, 0, c->stream));
    return MSGW_OK;
}

template <typename T>
RayPtrsT<T> ray_ptrs(const msgw_ctx *c)
{
    return RayPtrsT<T>{c->slab, (unsigned int)(c->pitch >> 8)};
}

// the exact constant division (div_const) needs d's significand not to be all ones
int markstein_ok(double d)
{
    uint64_t b;
    std::memcpy(&b, &d, sizeof b);
    return ((b & 0xFFFFFFFFFFFFFull) != 0xFFFFFFFFFFFFFull) && std::isfinite(d) && d > 0;
}
int markstein_ok(float d)
{
    uint32_t b;
    std::memcpy(&b, &d, sizeof b);
    return ((b & 0x7FFFFFu) != 0x7FFFFFu) && std::isfinite(d) && d > 0;
}

// First level of the flux reduction (many workgroup rows -> RED1_GROUPS dense rows);
// rewrites `a` so that the column kernel reads the second-level rows.
int reduce_level1(msgw_ctx *c, ColArgs &a, bool force = false)
{
    if (a.nblocks < RED1_MIN_ROWS && !force) return MSGW_OK;
    Red1Args r{};
    r.nblocks = a.nblocks; r.npay = a.npay; r.ncp = a.ncp; r.ncols = a.npay * a.ncp;
    r.nseg = pick_nseg(r.ncols);
    r.partial = a.partial; r.ranges = a.ranges; r.out = c->row2;
    const int rows_per_group = (a.nblocks + RED1_GROUPS - 1) / RED1_GROUPS + 1;
    const size_t lds = sizeof(double) * (size_t)r.nseg * r.ncols + sizeof(int) * 2 * (size_t)rows_per_group + 16;
    const int groups = a.nblocks < RED1_GROUPS ? a.nblocks : RED1_GROUPS;
    if (int rc = launch_struct(c, flux_reduce1_kernel(), groups, COL_BLOCK, lds, r)) return rc;
    a.partial = c->row2; a.ranges = nullptr; a.nblocks = groups;
    return MSGW_OK;
}

template <typename T>
StageArgsT<T> make_stage_args(msgw_ctx *c, double dt, unsigned flags)
{
    StageArgsT<T> a{};
    a.n = c->n;
    a.ng = c->ng;
    a.tiles_per_block = c->tiles_per_block;
    a.rays_per_block = c->rays_per_block;
    a.dt = (T)dt;
    a.dtc = dt;
    a.bvf2 = (T)std::pow(c->bvf, 2.0);            // python `bvf ** 2` (lib/libprop.py:383)
    a.f_uni = (T)c->f_uni;
    a.f0sq = (T)std::pow(c->f0, 2.0);             // python `ff ** 2` with scalar ff (:601)
    a.same_f = (!c->fvec && (a.f_uni * a.f_uni == a.f0sq)) ? 1 : 0;
    a.sat_c = (T)(std::pow(c->kappa, 2.0) * .5);  // `kappa**2 * .5` (:601)
    a.sat_rr_div = (flags & MSGW_DIRECT_SAT_QUIRK) ? T(1) : (T)dt;
    a.xg0 = (T)c->xg0; a.inv_dzg = (T)(1.0 / c->dzg);
    a.gs0 = (T)c->gs0; a.inv_dzs = T(1) / (T)c->dzs;
    a.xg_last = (T)c->xg_last; a.gs_last = (T)c->gs_last;
    a.dzs = (T)c->dzs;
    a.mk_ok = markstein_ok(a.dzs);
    a.r = ray_ptrs<T>(c);
    a.relaunch = (flags & MSGW_RELAUNCH) ? 1 : 0;
    a.z_bot = (T)c->z_bot; a.z_top = (T)c->z_top; a.relaunch_frac = (T)c->relaunch_frac;
    a.c = ColPtrs{c->grid + 1, c->dudz, c->dvdz, c->slu, c->slv, c->grids, c->rhobar, c->slrho};
    a.partial = c->partial;
    a.ranges = c->ranges;
    a.col_pending = 0;
    a.pg = c->pg; a.f0 = c->f0; a.dzg = c->dzg;
    a.grp_size = c->grp_size; a.row_stride = c->row_stride;
    a.grp_part = c->grp_part; a.grp_rows = c->grp_rows; a.grp_cnt = c->grp_cnt;
    a.ngroups = c->ngroups; a.flux_out = c->flux;
    return a;
}

ColArgs make_col_args(msgw_ctx *c, double dt, unsigned flags)
{
    ColArgs a{};
    a.ng = c->ng;
    a.nblocks = c->blocks;
    a.npay = 2;
    a.ncp = c->ng - 2;
    a.nseg = pick_nseg(a.npay * a.ncp);
    a.fixed_background = (flags & MSGW_FIXED_BACKGROUND) ? 1 : 0;
    a.dt = dt; a.f0 = c->f0; a.dzg = c->dzg;
    a.partial = c->partial; a.ranges = c->ranges; a.flux = c->flux;
    a.rhobar = c->rhobar; a.pg = c->pg;
    a.in = ColIn{c->uu, c->vv, c->q_uu, c->q_vv};
    a.out = ColOut{c->uu, c->vv, c->q_uu, c->q_vv};
    a.xg = c->grid + 1;
    a.dudz = c->dudz; a.dvdz = c->dvdz; a.slu = c->slu; a.slv = c->slv;
    a.out_du = c->out_du; a.out_dv = c->out_dv; a.out_flux = c->out_flux;
    return a;
}

// mode: 0 plain, 1 online saturation, 2 direct (driver) saturation.  The relaunch extension is a compile-time
// variant of stage 2 only, so that the default kernels carry none of its registers.
template <typename T>
int launch_stage(msgw_ctx *c, int stage, const StageArgsT<T> &a, int mode, bool timed)
{
    const bool sat = mode == 1, direct = mode == 2 && stage != 1, rl = stage == 2 && a.relaunch;
    int form = FORM_PLAIN;
    if (c->ng - 2 > 128) form = FORM_TALL;             // tall columns: per-level sums stay in LDS (NH = 0)
    else if (c->lagchain) form = FORM_LAG;             // lagged chain: deposit of the produced state, group rows by parity
    else if (c->groupred) form = FORM_GROUP;           // fused chain: first-level flux reduction inside the kernel
    c->cnt.launch_grid = c->cnt.launch_ray_workgroups = c->blocks; c->cnt.launch_reducers = 0; c->cnt.fixed_narrow = 0;
    return launch_struct(c, stage_kernel<T>(stage, sat, c->fvec, true, direct, form, rl), c->blocks, BLOCK,
                         stage_lds_bytes(c), a, nullptr, timed);
}

template <typename T>
int launch_probe(msgw_ctx *c, const StageArgsT<T> &a, bool sat, bool deposit)
{
    const int form = (c->ng - 2 > 128) ? FORM_TALL : FORM_PLAIN;
    return launch_struct(c, stage_kernel<T>(3, sat, c->fvec, deposit, false, form, false), c->blocks, BLOCK,
                         stage_lds_bytes(c), a);
}

// Fixed background (k_ray_step_fixed): independent rays, bound by the issue of dependent float64 chains.  Below
// `fixed_narrow_max` rays the NARROW form runs -- one wavefront per workgroup, one ray per lane -- so that 1e5 rays
// (BASELINE config 2) put 1563 wavefronts on all 1024 SIMDs instead of 784 two-ray wavefronts on 784 of them.
// MSGW_FIXED_NARROW=0 | 1 forces a form (diagnostic).
template <typename T>
int launch_fixed(msgw_ctx *c, StageArgsT<T> a, int mode, bool timed)
{
    const bool narrow = c->fixed_narrow_force >= 0 ? c->fixed_narrow_force != 0 : c->n <= c->fixed_narrow_max;
    // tables only (the kernel deposits nothing): [sh][rho2][xg][gs] in the ray type (+ the float64 grid copy stage_xg
    // writes for float32 rays)
    const int ni = c->ng - 2, nc = c->ng - 1;
    const size_t lds = ((c->esz * (size_t)(5 * ni + 3 * nc) + 15) & ~(size_t)15) + (c->f32 ? sizeof(double) * ni : 0) + 16;
    unsigned grid = (unsigned)c->blocks, block = BLOCK;
    if (narrow) {
        const int64_t ntiles = (c->n + 63) / 64;
        const int64_t maxb = (int64_t)c->ncu * 4 * 8;            // 8 wavefronts per SIMD at most
        int64_t tpb = (ntiles + maxb - 1) / maxb;
        if (tpb < 1) tpb = 1;
        a.tiles_per_block = (int)tpb;
        a.rays_per_block = tpb * 64;
        grid = (unsigned)std::max<int64_t>((ntiles + tpb - 1) / tpb, 1);
        block = 64;
    }
    c->cnt.fixed_narrow = narrow ? 1 : 0;
    c->cnt.launch_grid = c->cnt.launch_ray_workgroups = (int)grid;
    c->cnt.launch_reducers = 0;
    return launch_struct(c, fixed_kernel<T>(mode == 1, c->fvec, mode == 2, narrow), grid, block, lds, a, nullptr, timed);
}

int launch_column(msgw_ctx *c, int stage, int mode, const ColArgs &a, hipStream_t stream = nullptr)
{
    const size_t lds = col_lds_bytes(a.ng, a.nseg, a.npay * a.ncp, a.nblocks);
    return launch_struct(c, column_kernel(stage, mode), 1, COL_BLOCK, lds, a, stream);
}

int allreduce_flux(msgw_ctx *c, double *buf = nullptr, hipStream_t stream = nullptr)
{
    const size_t count = (size_t)2 * (c->ng - 2);
    if (!buf) buf = c->flux;
    if (!c->comm)
        return fail(c, MSGW_ERR_RCCL, "this step needs the RCCL all-reduce chain, but the communicator was set up "
                    "without RCCL (MSGW_EXCHANGE_ONLY=1)");
    ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclFloat64, ncclSum, c->comm, stream ? stream : c->stream);
    if (r != 0)
        return fail(c, MSGW_ERR_RCCL, "ncclAllReduce failed: %s",
                    g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return MSGW_OK;
}

// flux finalise (+ all-reduce over the ranks) + mean-flow RK stage
int column_stage(msgw_ctx *c, int stage, const ColArgs &a0)
{
    ColArgs a = a0;
    if (int rc = reduce_level1(c, a)) return rc;
    if (c->nranks > 1 || (c->force_coll && c->comm)) {
        if (int rc = launch_column(c, stage, COL_REDUCE, a)) return rc;
        if (int rc = allreduce_flux(c)) return rc;
        return launch_column(c, stage, COL_UPDATE, a);
    }
    return launch_column(c, stage, COL_REDUCE | COL_UPDATE, a);
}

size_t persist_lds_bytes(const msgw_ctx *c, bool lean = false)
{
    if (lean) return stage_lds_bytes(c) + 64;                  // the LEAN layout (persist_kernel.h): only the flag words on top

    // + column replica (7 x (ng-1) float64) + flags; float32 rays: + the float64 shear table of the column arithmetic
    return stage_lds_bytes(c) + sizeof(double) * (size_t)7 * (c->ng - 1) + 32 +
           (c->f32 ? sizeof(double) * 4 * (size_t)(c->ng - 2) + 32 : 0);
}

// A raised status word means a bounded wait of a persistent launch timed out (a workgroup was not resident, the GPU
// is shared, or a rank died): the step was abandoned half-way and the ray state is invalid.  The launch itself is not
// waited for (calls are stream-ordered); the word is read at the next blocking call.
int check_status(msgw_ctx *c)
{
    if (!c->status_armed || !c->pstatus) return MSGW_OK;
    int status = 0;
    if (c->pstatus_host) {                                     // the kernel also raises the word in host-mapped memory
        HIPCHK(c, hipStreamSynchronize(c->stream));
        status = *reinterpret_cast<volatile int *>(c->pstatus_host);
    } else {
        HIPCHK(c, hipMemcpyAsync(&status, c->pstatus, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->status_armed = false;
    if (status == 0) return MSGW_OK;
    c->persist = 0;                                            // the per-stage kernels are used from now on
    c->have_rays = false;
    c->carry_key = 0;
    if (c->nranks > 1 || c->force_coll)
        return fail(c, MSGW_ERR_HIP, "persistent RK3 kernel timed out in the node-level flux exchange (a rank "
                    "died, or the ranks did not call msgw_step alike); state is invalid, upload the rays again");
    return fail(c, MSGW_ERR_HIP, "persistent RK3 kernel timed out waiting for other workgroups "
                "(not all of them resident: is the GPU shared?); state is invalid, upload the rays again "
                "(the per-stage kernels will be used from now on)");
}

// Geometry of one persistent launch (nres: register-resident tiles per workgroup).  *fits = false when this
// flavour does not apply to the resident rays (then nothing is changed).
struct PersistPlan {
    int nres = 0, blocks = 0, tiles_per_block = 0, grp_size = 1, ngroups = 1, nservice = 0, grid = 0;
    int64_t rays_per_block = 0;
    const void *fn = nullptr;
    size_t lds = 0;
};
template <typename T>
int plan_persist(msgw_ctx *c, int nres, int mode, bool rl, bool multi, PersistPlan &pl, bool *fits)
{
    *fits = false;
    pl = PersistPlan{};
    pl.nres = nres;
    pl.fn = persist_kernel<T>(mode == 1, c->fvec, mode == 2, nres, rl);
    if (!pl.fn && nres > 2) return MSGW_OK;                    // this variant has no four-tile flavour
    // the resident-tile flavours assume the exact constant division by the grid spacing (process_tiles: mk_ok)
    if (nres > 0 && !markstein_ok((T)c->dzs)) return MSGW_OK;
    pl.lds = persist_lds_bytes(c);
    if (int rc = ensure_lds(c, pl.fn, pl.lds)) return rc;
    int per_cu = 0;
    HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pl.fn, BLOCK, pl.lds));
    // tall columns: the lean-LDS variant of the kernel (float64, two or four resident tiles, reducer workgroups on)
    // when it lets more workgroups share a CU -- never for the default column, whose occupancy the registers decide
    bool lean = false;
    if (!c->f32 && c->service && nres > 0) {
        const void *lf = persist_kernel<T>(mode == 1, c->fvec, mode == 2, nres, rl, true);
        const size_t ll = persist_lds_bytes(c, true);
        int per_cu_lean = 0;
        if (lf && ensure_lds(c, lf, ll) == MSGW_OK &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_lean, lf, BLOCK, ll) == hipSuccess && per_cu_lean > per_cu) {
            lean = true; pl.fn = lf; pl.lds = ll; per_cu = per_cu_lean;
        }
    }
    // all workgroups of the grid must be co-resident.  The occupancy query is the device's capacity for THIS
    // process; ranks that share one device (rehearsals on a 1-GPU box) split it evenly (c->tenants, agreed on
    // when the communicator was set up).  A foreign tenant cannot be seen: the bounded waits turn that case
    // into an error instead of a hang.
    const long long slots = (long long)per_cu * c->ncu / (c->tenants > 1 ? c->tenants : 1);
    int blocks = c->blocks;
    pl.tiles_per_block = c->tiles_per_block;
    pl.rays_per_block = c->rays_per_block;
    if (nres > 0) {
        // own geometry: as many ray workgroups as fit beside 16 reducers, the column and the exchange
        // workgroup at this kernel's occupancy (2 per CU)
        const long long maxb = slots - (16 + 2);
        if (maxb < 1) return MSGW_OK;
        split_rays(c, c->n, maxb, &pl.rays_per_block, &pl.tiles_per_block, &blocks);
        // measured with 2 resident tiles: +10 % at 2e6 rays (8 tiles per workgroup), +3 % at 4e6 and 8e6 (16, 32),
        // -10 % at 16e6 (64), where 4 workgroups per CU with all rays streamed are better
        if (pl.tiles_per_block > 32) return MSGW_OK;
    } else if (blocks + PERSIST_GROUPS + 2 > slots) {
        // tall columns: the LDS footprint leaves fewer than the default 4 workgroups per CU
        const long long maxb = slots - (PERSIST_GROUPS + 2);
        if (maxb < 1) return MSGW_OK;
        split_rays(c, c->n, maxb, &pl.rays_per_block, &pl.tiles_per_block, &blocks);
    }
    pl.blocks = blocks;
    const int max_groups = nres > 0 ? 16 : PERSIST_GROUPS;
    pl.grp_size = (blocks + max_groups - 1) / max_groups;
    pl.ngroups = (blocks + pl.grp_size - 1) / pl.grp_size;
    // reducer workgroups (one per group) + the column workgroup when they fit beside the ray workgroups,
    // else the last arriver reduces
    pl.nservice = (c->service && blocks + pl.ngroups + 1 <= slots) ? pl.ngroups : 0;
    // + reducer workgroups + the column workgroup (which also sums over the ranks); without them, several ranks:
    // + one exchange workgroup
    pl.grid = blocks + pl.nservice + (pl.nservice ? 1 : 0) + ((multi && !pl.nservice) ? 1 : 0);
    *fits = slots >= pl.grid && pl.grid <= 2048;               // every workgroup co-resident
    if (lean && !pl.nservice) *fits = false;                   // (the lean layout has no room for per-workgroup column replicas)
    return MSGW_OK;
}

bool xch_agree_step(msgw_ctx *c, bool mine, bool *all);

// All `count` steps in ONE persistent launch.  *used = false when the path does not apply
// (then nothing was enqueued and the caller takes the per-stage path).
template <typename T>
int run_persistent(msgw_ctx *c, double dt, unsigned flags, int count, bool time_kernels, bool *used)
{
    *used = false;
    // any column whose tables, wave rows and replica fit the LDS of a CU (ngrid <= ~870; up to 130 levels a thread of
    // the reducer / column / exchange workgroups owns one column entry, beyond that it strides)
    const bool can_fuse = persist_lds_bytes(c) <= 160 * 1024;
    const bool multi = c->nranks > 1 || c->force_coll;
    if ((flags & MSGW_FIXED_BACKGROUND) || count <= 0) return MSGW_OK;
    const int mode = c->sat_online ? 1 : ((flags & (MSGW_DIRECT_SAT | MSGW_DIRECT_SAT_QUIRK)) ? 2 : 0);
    const bool rl = (flags & MSGW_RELAUNCH) != 0;
    bool want = c->persist && can_fuse && (!multi || c->xch_ok);
    PersistPlan pl;
    bool fits = false;
    // the plan of the previous call when nothing it depends on has changed (each plan_persist is up to two occupancy
    // queries and a function-attribute call: ~15 us per msgw_step call, the size of a 20-step call's launch latency)
    const unsigned long long pkey = 0x9e3779b97f4a7c15ull * (unsigned long long)(c->n + 1) ^
        ((unsigned long long)mode | (unsigned long long)rl << 2 | (unsigned long long)multi << 3 | (unsigned long long)c->fvec << 4 |
         (unsigned long long)c->f32 << 5 | (unsigned long long)c->regtiles << 6 | (unsigned long long)c->service << 10 |
         (unsigned long long)c->tenants << 11 | (unsigned long long)c->blocks_per_cu << 20 | (unsigned long long)c->ng << 32 | 1ull << 63);
    const bool cached = want && c->plan_key == pkey && c->plan_blob.size() == sizeof(PersistPlan);
    if (cached) { std::memcpy(&pl, c->plan_blob.data(), sizeof pl); fits = true; }
    if (want && !cached && c->regtiles > 3)                    // first choice: four register-resident tiles,
        if (int rc = plan_persist<T>(c, 4, mode, rl, multi, pl, &fits)) return rc;
    if (want && !cached && c->regtiles > 2 && !fits)           // three (variants that do not have four),
        if (int rc = plan_persist<T>(c, 3, mode, rl, multi, pl, &fits)) return rc;
    if (want && !cached && c->regtiles && !fits)               // then two,
        if (int rc = plan_persist<T>(c, 2, mode, rl, multi, pl, &fits)) return rc;
    if (want && !cached && !fits)
        if (int rc = plan_persist<T>(c, 0, mode, rl, multi, pl, &fits)) return rc;
    if (want && fits && !cached) {
        c->plan_blob.resize(sizeof(PersistPlan));
        std::memcpy(c->plan_blob.data(), &pl, sizeof pl);
        c->plan_key = pkey;
    }
    want = want && fits;
    // several ranks: either every rank launches the persistent kernel or none does (a rank that declined while
    // its peers spin in the exchange would leave them to their time-out)
    if (multi && c->nranks > 1 && c->xch_ok) {
        bool all = false;
        if (!xch_agree_step(c, want, &all))
            return fail(c, MSGW_ERR_HIP, "the ranks did not agree on the kernel path within the time limit (a rank "
                        "died, or the ranks did not call msgw_step alike)");
        want = all;
    }
    if (!want) return MSGW_OK;                                 // per-stage path

    PersistArgsT<T> pa{};
    pa.s = make_stage_args<T>(c, dt, flags);
    pa.s.tiles_per_block = pl.tiles_per_block;
    pa.s.rays_per_block = pl.rays_per_block;
    pa.s.grp_size = pl.grp_size;
    pa.nsteps = count;
    pa.ngroups = pl.ngroups;
    pa.nworkers = pl.blocks;
    pa.nservice = pl.nservice;
    pa.opts = pl.nservice ? 0u : PERSIST_OPT_PRIO;
    if (const char *e = std::getenv("MSGW_PRIO")) pa.opts = std::atoi(e) ? PERSIST_OPT_PRIO : 0u;
    if (c->prefetch) pa.opts |= PERSIST_OPT_PREFETCH;
    { const char *e = std::getenv("MSGW_LEANPOLL"); if (!e || std::atoi(e)) pa.opts |= PERSIST_OPT_LEANPOLL; }
    if (const char *e = std::getenv("MSGW_NAP")) pa.opts |= std::atoi(e) == 1 ? PERSIST_OPT_NAP1 : (std::atoi(e) == 8 ? PERSIST_OPT_NAP8 : 0u);
    if (c->balance && pl.nres > 0 && pl.nservice) {
        // MSGW_BALANCE=<abc> (diagnostic): priorities of a workgroup released on arrival / that had to wait / prefetched
        const int b = c->balance == 1 ? 202 : c->balance;
        pa.opts |= PERSIST_OPT_BALANCE | ((unsigned)((b / 100) % 10 & 3) << 4) | ((unsigned)((b / 10) % 10 & 3) << 6) |
                   ((unsigned)(b % 10 & 3) << 8);
    }
    pa.grp_rows2 = c->grp_rows2;
    pa.grp_part2 = c->grp_part2;
    pa.flux2 = c->flux2;
    pa.shtab = c->shtab;
    pa.ready = c->pdone;
    pa.status = c->pstatus;
    pa.status_host = c->pstatus_host_dev;
    // F_0 carried over from the previous launch: same ray state (nothing uploaded, no step through another path since),
    // same kernel flavour (the streamed tiles' cg_rr in memory belongs to it), reducer workgroups on
    const unsigned long long ckey = 1ull | ((unsigned long long)pl.nres << 1) | ((unsigned long long)mode << 5) |
                                    ((unsigned long long)rl << 8) | ((unsigned long long)c->fvec << 9) |
                                    ((unsigned long long)c->f32 << 10) | ((unsigned long long)(flags & 0xffu) << 16);
    pa.fcarry = (pl.nservice && c->carry) ? c->fcarry : nullptr;
    pa.carry_in = (pa.fcarry && c->carry_key == ckey) ? 1 : 0;
    c->carry_key = pa.fcarry ? ckey : 0;
    c->cnt.carried_flux = pa.carry_in;
    pa.done2 = c->pdone + 32;
    pa.grp_cnt2 = c->pdone + 128;
#ifdef MSGW_STAMP
    if (!c->pstamps) HIPCHK(c, hipMalloc(&c->pstamps, sizeof(unsigned long long) * 4096 * PSTAMP_PASSES * 4));
    HIPCHK(c, hipMemsetAsync(c->pstamps, 0, sizeof(unsigned long long) * 4096 * PSTAMP_PASSES * 4, c->stream));
    pa.pstamps = c->pstamps;
#endif
    pa.timeout_ticks = 20000000ull;                            // 0.2 s of wall clock per wait
    if (multi) {                                               // the other ranks' launches may start much later
        pa.xch = reinterpret_cast<const XchArgs *>(c->xch_scratch + 16);   // written once by xch_setup
        pa.xch_seq = c->xch_seq;
        pa.timeout_ticks = XCH_TIMEOUT_TICKS + 100000000ull;
    }
    pa.cin = ColIn{c->uu, c->vv, c->q_uu, c->q_vv};
    pa.cout = ColOut{c->uu, c->vv, c->q_uu, c->q_vv};
    pa.dudz = c->dudz; pa.dvdz = c->dvdz; pa.slu = c->slu; pa.slv = c->slv;
    HIPCHK(c, hipMemsetAsync(c->pdone, 0, sizeof(unsigned int) * PDONE_WORDS, c->stream));
    bool launched = false;
    if (c->coop && !multi) {
        // hipLaunchCooperativeKernel: the RUNTIME vouches for co-residency of the whole grid (it refuses a grid that
        // does not fit) instead of this file's reading of the occupancy query.  Measured: no cost (33.1 vs 33.0 us per
        // step at config 3, 34.1 vs 34.4 with 20-step calls).  Kept to one rank per device: cooperative launches of
        // several processes on one device take turns, which the in-kernel exchange between ranks that share a device
        // (the 1-GPU rehearsals) could not survive; on distinct devices the peers' grids are separate launches anyway.
        if (int rc = ensure_lds(c, pl.fn, pl.lds)) return rc;
        void *args[] = {&pa};
        // (MSGW_TIME_KERNELS: hipLaunchCooperativeKernel takes no events, so the pair is recorded on the stream right
        // around the launch -- nothing else is enqueued in between)
        hipEvent_t *ev = time_kernels ? timing_events(c) : nullptr;
        if (ev) HIPCHK(c, hipEventRecord(ev[0], c->stream));
        const hipError_t e = hipLaunchCooperativeKernel(pl.fn, dim3(pl.grid), dim3(BLOCK), args, (unsigned int)pl.lds, c->stream);
        if (e == hipSuccess && ev) HIPCHK(c, hipEventRecord(ev[1], c->stream));
        if (e != hipSuccess && ev) c->kev_used -= 2;
        if (e == hipSuccess) launched = true;
        else {                                                 // refused: this call takes the launch chain, later ones the plain launch
            (void)hipGetLastError();
            c->coop = 0;
            c->cnt.coop_refused += 1;
            return MSGW_OK;
        }
    }
    if (!launched)
        if (int rc = launch_struct(c, pl.fn, pl.grid, BLOCK, pl.lds, pa, nullptr, time_kernels)) return rc;
    c->cnt.cooperative = launched ? 1 : 0;
    *used = true;
    c->status_armed = true;
    c->cnt.persist_resident_tiles = pl.nres;
    c->cnt.persist_steps = count;
    c->cnt.launch_grid = pl.grid; c->cnt.launch_ray_workgroups = pl.blocks; c->cnt.launch_reducers = pl.nservice;
    if (multi) c->xch_seq += 3ull * (unsigned long long)count;   // fluxes 0 .. 3*count-1 were exchanged (strictly alternating slots)
    return MSGW_OK;
}

// Lagged launch chain for multi-GPU runs.  Pass q deposits the state it PRODUCES, i.e. publishes the
// group rows of F_{q+1}; their reduction to one row and the RCCL all-reduce run on a second stream
// WHILE the next ray-stage kernel executes (it needs F_q, reduced one pass earlier), so the
// collective is off the critical path.  Per pass on stream A:  wait(F_{q-1} all-reduced) ->
// k_ray_stage<LAG> -> record;  on stream B:  wait(row of F_{q+1}) ->
// ncclAllReduce -> record (the ray-stage kernel itself reduces its rows to one row).  Group rows and
// flux rows are double-buffered by flux parity.
template <typename T>
int enqueue_steps_lagged(msgw_ctx *c, double dt, unsigned flags, int count, bool time_kernels)
{
    const int mode = c->sat_online ? 1 : ((flags & (MSGW_DIRECT_SAT | MSGW_DIRECT_SAT_QUIRK)) ? 2 : 0);
    const int ncols = 2 * (c->ng - 2);
    // several ranks must sum their rows: without a communicator (MSGW_EXCHANGE_ONLY=1) only the persistent kernel
    // can do that, and it has declined this step (relaunch off its fast path, MSGW_PERSIST=0, a grid that is not
    // co-resident ...): every rank would silently advance its column with its local flux only
    if (c->nranks > 1 && !c->comm)
        return fail(c, MSGW_ERR_RCCL, "this multi-rank step cannot take the persistent kernel and needs the RCCL "
                    "all-reduce chain, but the communicator was set up without RCCL (MSGW_EXCHANGE_ONLY=1)");
    if (!c->stream2) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_rows[i], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_flux[i], hipEventDisableTiming));
        }
    }
    StageArgsT<T> sa = make_stage_args<T>(c, dt, flags);
    double *rows_par[2] = {c->grp_rows, c->grp_rows + (size_t)FUSE_ROWS * ncols};
    double *flux_par[2] = {c->flux2, c->flux2 + ncols};
    const ColIn in_set[2] = {ColIn{c->uu, c->vv, c->q_uu, c->q_vv}, ColIn{c->alt_uu, c->alt_vv, c->alt_q_uu, c->alt_q_vv}};
    const ColOut out_set[2] = {ColOut{c->uu, c->vv, c->q_uu, c->q_vv}, ColOut{c->alt_uu, c->alt_vv, c->alt_q_uu, c->alt_q_vv}};
    // reduce the group rows of flux f to one row and all-reduce it over the ranks, on stream B
    auto reduce_on_b = [&](int f) -> int {
        const int par = f & 1;
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_rows[par], 0));
        if (c->comm)
            if (int rc = allreduce_flux(c, flux_par[par], c->stream2)) return rc;
        HIPCHK(c, hipEventRecord(c->ev_flux[par], c->stream2));
        return MSGW_OK;
    };
    c->lagchain = true;
    struct Guard { msgw_ctx *c; ~Guard() { c->lagchain = false; } } guard{c};
    // pre-pass: F_0
    sa.grp_rows = rows_par[0];
    sa.flux_out = flux_par[0];
    if (int rc = launch_struct(c, deposit_only_kernel<T>(c->fvec), c->blocks, BLOCK, stage_lds_bytes(c), sa, nullptr, time_kernels))
        return rc;
    HIPCHK(c, hipEventRecord(c->ev_rows[0], c->stream));
    if (int rc = reduce_on_b(0)) return rc;
    int cur = 0;
    const int npass = 3 * count;
    for (int q = 0; q < npass; ++q) {
        const int s = q % 3;
        sa.col_pending = q >= 1 ? 1 : 0;
        sa.col_stage = (s + 2) % 3;                      // RK stage of the column update that yields column_q
        sa.col_rows = flux_par[(q + 1) & 1];             // == parity of q-1: the all-reduced row of F_{q-1}
        sa.cin = in_set[cur];
        sa.cout = out_set[cur ^ 1];
        sa.grp_rows = rows_par[(q + 1) & 1];             // this pass publishes F_{q+1} ...
        sa.flux_out = flux_par[(q + 1) & 1];             // ... reduced in-kernel to one row
        if (q >= 1) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_flux[(q - 1) & 1], 0));
        if (int rc = launch_stage<T>(c, s, sa, mode, time_kernels)) return rc;
        if (q >= 1) cur ^= 1;
        HIPCHK(c, hipEventRecord(c->ev_rows[(q + 1) & 1], c->stream));
        if (q + 1 <= npass - 1)                          // F_{q+1} is consumed by pass q+2 <= npass (the final update)
            if (int rc2 = reduce_on_b(q + 1)) return rc2;
    }
    // column_{npass} = update of column_{npass-1} with F_{npass-1}
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_flux[(npass - 1) & 1], 0));
    ColArgs fin = make_col_args(c, dt, flags);
    fin.flux = flux_par[(npass - 1) & 1];
    fin.in = in_set[cur];
    fin.out = out_set[0];
    return launch_column(c, 2, COL_UPDATE, fin);
}

// Enqueue `count` RK3 steps (lib/libprop.py:693-698) on the context's stream.
// Coupled mode chains the stages so that each stage's mean-flow update is applied in the
// PROLOGUE of the next ray-stage kernel (all workgroups redo the tiny column update in LDS,
// workgroup 0 publishes it to the other column set); only the very last update of the batch
// runs as a standalone k_column, which always lands in the canonical set (c->uu, ...).
//   per stage:  ONE launch of k_ray_stage (prologue: previous stage's column update from the one
//   final flux row; tail: in-kernel reduction of its own rows to the next final row)
template <typename T>
int enqueue_steps(msgw_ctx *c, double dt, unsigned flags, int count, bool time_kernels)
{
    if (count <= 0) return MSGW_OK;
    const int mode = c->sat_online ? 1 : ((flags & (MSGW_DIRECT_SAT | MSGW_DIRECT_SAT_QUIRK)) ? 2 : 0);
    StageArgsT<T> sa = make_stage_args<T>(c, dt, flags);
    if (flags & MSGW_FIXED_BACKGROUND) {                       // independent rays: all steps in ONE launch
        sa.fixed_steps = count;
        c->cnt.persist_steps = count;
        return launch_fixed<T>(c, sa, mode, time_kernels);
    }
    const ColIn in_set[2] = {ColIn{c->uu, c->vv, c->q_uu, c->q_vv}, ColIn{c->alt_uu, c->alt_vv, c->alt_q_uu, c->alt_q_vv}};
    const ColOut out_set[2] = {ColOut{c->uu, c->vv, c->q_uu, c->q_vv}, ColOut{c->alt_uu, c->alt_vv, c->alt_q_uu, c->alt_q_vv}};
    const bool can_fuse = (2 * (c->ng - 2) <= BLOCK) && (c->ng - 1 <= BLOCK);   // one thread per column/level
    if (!can_fuse) {                   // tall columns: standalone column kernel after every stage
        const ColArgs ca = make_col_args(c, dt, flags);
        for (int step = 0; step < count; ++step)
            for (int s = 0; s < 3; ++s) {
                if (int rc = launch_stage<T>(c, s, sa, mode, time_kernels)) return rc;
                if (int rc = column_stage(c, s, ca)) return rc;
            }
        return MSGW_OK;
    }
    if (c->nranks > 1 || (c->force_coll && c->comm))   // collectives: overlap them with the next pass
        return enqueue_steps_lagged<T>(c, dt, flags, count, time_kernels);
    int cur = 0;                       // set that holds the column BEFORE the pending update
    bool pending = false;
    int pend_stage = 0;
    sa.col_rows = c->flux;             // every launch reduces its rows to this one row (in-kernel, three
    sa.flux_out = c->flux;             // ticket levels); it is rewritten only after all prologues have read it
    c->groupred = true;
    struct Guard { msgw_ctx *c; ~Guard() { c->groupred = false; } } guard{c};
    for (int step = 0; step < count; ++step) {
        for (int s = 0; s < 3; ++s) {
            sa.col_pending = pending ? 1 : 0;
            sa.col_stage = pend_stage;
            sa.cin = in_set[cur];
            sa.cout = out_set[cur ^ 1];
            if (int rc = launch_stage<T>(c, s, sa, mode, time_kernels)) return rc;
            if (pending) cur ^= 1;     // workgroup 0 has published the updated column there
            pending = true;
            pend_stage = s;
        }
    }
    // the last update of the batch: standalone, from set `cur` into the canonical set
    ColArgs fin = make_col_args(c, dt, flags);
    fin.in = in_set[cur];
    fin.out = out_set[0];
    return launch_column(c, 2, COL_UPDATE, fin);
}

int ready(msgw_ctx *c)
{
    if (!c) return MSGW_ERR_ARG;
    if (!c->have_config) return fail(c, MSGW_ERR_ARG, "msgw_set_config has not been called");
    if (!c->have_column) return fail(c, MSGW_ERR_ARG, "msgw_set_column has not been called");
    if (!c->have_rays) return fail(c, MSGW_ERR_ARG, "msgw_upload_rays has not been called");
    if (c->hprop && !c->have_hprop) return fail(c, MSGW_ERR_ARG, "HPROP is on: msgw_upload_hprop (lam, phi) has not been called");
    if (c->nz && !c->have_nz) return fail(c, MSGW_ERR_ARG, "an N(z) column is set: call msgw_upload_rays after msgw_set_bvf_column");
    return MSGW_OK;
}

// ---- HPROP_GLOBAL = True and / or the N(z) column extension: the general per-stage kernel (chain_kernels.h) + the
// standalone column kernel.  Online saturation is a compile-time variant; the driver's direct saturation and the
// relaunch extension are run-time branches of the stage-2 kernel.
template <typename T>
ChainArgsT<T> make_chain_args(msgw_ctx *c, double dt, unsigned flags)
{
    ChainArgsT<T> h{};
    h.s = make_stage_args<T>(c, dt, flags);
    auto t = [](void *v) { return static_cast<T *>(v); };
    h.lam = t(c->lam); h.phi = t(c->phi); h.kk = t(c->kk); h.ll = t(c->ll);
    h.q_lam = t(c->q_lam); h.q_phi = t(c->q_phi); h.q_kk = t(c->q_kk); h.q_ll = t(c->q_ll);
    h.uu = c->uu; h.vv = c->vv;
    const double rot = 7.2921e-5;                                  // ROT_EARTH  (lib/libprop.py:4)
    h.rad_earth = (T)6378e3;                                       // RAD_EARTH  (lib/libprop.py:3)
    h.two_rot = (T)(2 * rot);
    h.df2c = (T)(8 * (rot * rot));                                 // 8 * ROT_EARTH**2 (:489)
    h.drr = t(c->drr); h.dmm = t(c->dmm);
    h.q_drr = t(c->nz_q_drr); h.q_dmm = t(c->nz_q_dmm); h.drr0 = t(c->nz_drr0);
    h.dkdl = t(c->nz_dkdl); h.area = t(c->nz_area); h.bvf = c->bvfcol;
    h.direct = (flags & (MSGW_DIRECT_SAT | MSGW_DIRECT_SAT_QUIRK)) ? 1 : 0;
    return h;
}

template <typename T>
int launch_chain_stage(msgw_ctx *c, int stage, const ChainArgsT<T> &h)
{
    return launch_struct(c, chain_kernel<T>(stage, c->sat_online != 0, c->hprop != 0, c->nz != 0), c->blocks, BLOCK,
                         chain_lds_bytes<T>(c->ng, c->hprop != 0, c->nz != 0), h);
}

// The stage kernels of this chain reduce their flux rows inside the launch (flush_rows_group: one row in c->flux) when a
// thread of the last reducer can own a column entry; the column kernel then only updates.
bool chain_group_reduce(const msgw_ctx *c)
{
    static const bool off = [] { const char *e = std::getenv("MSGW_CHAIN_GROUPRED"); return e && std::atoi(e) == 0; }();
    return !off && 2 * (c->ng - 2) <= BLOCK;
}
int column_from_row(msgw_ctx *c, int stage, const ColArgs &a)
{
    if (c->nranks > 1 || (c->force_coll && c->comm))
        if (int rc = allreduce_flux(c)) return rc;
    return launch_column(c, stage, COL_UPDATE, a);
}

template <typename T>
int enqueue_steps_chain(msgw_ctx *c, double dt, unsigned flags, int count)
{
    ChainArgsT<T> h = make_chain_args<T>(c, dt, flags);
    h.group_reduce = chain_group_reduce(c) ? 1 : 0;
    const ColArgs ca = make_col_args(c, dt, flags);
    for (int step = 0; step < count; ++step)
        for (int s = 0; s < 3; ++s) {
            if (int rc = launch_chain_stage<T>(c, s, h)) return rc;
            if (int rc = h.group_reduce ? column_from_row(c, s, ca) : column_stage(c, s, ca)) return rc;
        }
    return MSGW_OK;
}

// SURVEY 8d accounting, words per ray-step of this chain = 3 L + 7 E: every stage reads L per-ray arrays (dens, rr, kk,
// ll, mm; lam, phi or the per-ray f; drr and vol, or drr, dmm and dkk*dll with an N(z) column; the phase-volume factor
// with online saturation) and writes the E evolving slots; their RK registers are written by stages 0, 1 and read by
// stages 1, 2.  HPROP: 69, N(z): 55, N(z) + online saturation: 65, HPROP + N(z): 86, with online saturation: 96.
// (Rounds 1-2 quoted 71 / 63 / 75: HPROP's two flux words included, N(z) with RK-register traffic in all three stages.)
double chain_words_per_step(const msgw_ctx *c)
{
    const int E = 2 + (c->sat_online ? 1 : 0) + (c->hprop ? 4 : 0) + (c->nz ? 2 : 0);
    const int L = 5 + (c->hprop ? 2 : 1) + (c->nz ? 3 : 2) + (c->sat_online ? 1 : 0);
    return 3.0 * L + 7.0 * E;
}

// float32 state <-> the float64 arrays of the C ABI: staged through a float64 device buffer and converted there
int upload_array(msgw_ctx *c, void *dst, const double *src, int64_t n, double *stage)
{
    if (!c->f32) {
        HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        return MSGW_OK;
    }
    HIPCHK(c, hipMemcpyAsync(stage, src, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return launch_list(c, convert_kernel_d2f(), (unsigned)((n + 255) / 256), 256, (long long)n, (const double *)stage,
                       static_cast<float *>(dst));
}
int download_array(msgw_ctx *c, double *dst, const void *src, int64_t n, double *stage)
{
    if (!c->f32) {
        HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        return MSGW_OK;
    }
    if (int rc = launch_list(c, convert_kernel_f2d(), (unsigned)((n + 255) / 256), 256, (long long)n,
                             static_cast<const float *>(src), stage))
        return rc;
    HIPCHK(c, hipMemcpyAsync(dst, stage, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return MSGW_OK;
}
// a float64 staging buffer of `count` arrays of n rays (float32 contexts only; nullptr otherwise)
struct Staging {
    double *p = nullptr;
    ~Staging() { if (p) (void)hipFree(p); }
};

template <typename T>
int fill_padding(msgw_ctx *c, int64_t n, int64_t n_pad)
{
    auto t = [](void *v) { return static_cast<T *>(v); };
    struct { void *p; double v; } pad[] = {
        {c->dens, 0.0}, {c->rr, 0.0}, {c->mm, 1.0}, {c->drr, 1.0}, {c->kk, 1.0}, {c->ll, 0.0},
        {c->dmm, 0.0}, {c->vol, 0.0}, {c->fray, 0.0}, {c->pvf, 1.0}, {c->q_rr, 0.0}, {c->q_mm, 0.0},
        {c->q_dens, 0.0}, {c->rr0, 0.0}, {c->mm0, 1.0}, {c->src_dens, 0.0}, {c->src_rr, 0.0}, {c->src_mm, 1.0},
        {c->cgbuf, 0.0}};
    for (auto &x : pad)
        if (int rc = launch_list(c, fill_kernel<T>(), (unsigned)((n_pad - n + 255) / 256), 256, t(x.p), (long long)n,
                                 (long long)n_pad, (T)x.v))
            return rc;
    return MSGW_OK;
}

template <typename T>
int step_impl(msgw_ctx *c, double dt, int nsteps, unsigned gflags, bool eager, bool time_kernels)
{
    int done = 0;
    {
        bool used = false;
        c->cnt.persist_steps = 0;
        c->cnt.persist_resident_tiles = 0;
        if (int rc = run_persistent<T>(c, dt, gflags, nsteps, time_kernels, &used)) return rc;
        if (used) done = nsteps;
        else c->carry_key = 0;                                 // another path advances the state: no flux carried over
    }
    if (done < nsteps && !eager && nsteps >= c->graph_steps) {
        if (!c->gexec || c->g_dt != dt || c->g_flags != gflags || c->g_n != c->n || c->g_steps != c->graph_steps) {
            drop_graph(c);
            bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            int rc = MSGW_OK;
            if (ok) {
                rc = enqueue_steps<T>(c, dt, gflags, c->graph_steps, false);
                hipError_t e = hipStreamEndCapture(c->stream, &c->graph);
                ok = (rc == MSGW_OK) && (e == hipSuccess) && c->graph;
            }
            if (ok) ok = hipGraphInstantiate(&c->gexec, c->graph, nullptr, nullptr, 0) == hipSuccess;
            if (!ok) {                       // capture unsupported (e.g. a collective): stay eager
                (void)hipGetLastError();
                drop_graph(c);
                c->graph_steps = 0;
            } else {
                c->g_dt = dt; c->g_flags = gflags; c->g_n = c->n; c->g_steps = c->graph_steps;
            }
        }
        while (c->gexec && nsteps - done >= c->g_steps) {
            HIPCHK(c, hipGraphLaunch(c->gexec, c->stream));
            done += c->g_steps;
        }
    }
    if (done < nsteps)
        if (int rc = enqueue_steps<T>(c, dt, gflags, nsteps - done, time_kernels)) return rc;
    return MSGW_OK;
}

}   // namespace

namespace {

// shared tail of the two projection entry points: launch the projection kernel `fn` on `a`, reduce, copy out
template <typename PA>
int run_projection(msgw_ctx *c, PA &a, const void *fn, int tile, int np, const double *G, int nG, double *out)
{
    const int ncp = nG - 1;
    double *dG = nullptr, *dflux = nullptr;
    HIPCHK(c, hipMalloc(&dG, sizeof(double) * (size_t)nG));
    if (hipMalloc(&dflux, sizeof(double) * (size_t)np * ncp) != hipSuccess) {
        (void)hipFree(dG);
        return fail(c, MSGW_ERR_HIP, "hipMalloc failed in projection");
    }
    int rc = MSGW_OK;
    do {
        if (hipMemcpyAsync(dG, G, sizeof(double) * (size_t)nG, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(c, MSGW_ERR_HIP, "H2D of G failed"); break; }
        const int64_t ntiles = (a.n + tile - 1) / tile;
        const int64_t maxb = (int64_t)c->ncu * c->blocks_per_cu;
        int64_t tpb = (ntiles + maxb - 1) / maxb; if (tpb < 1) tpb = 1;
        int blocks = (int)((ntiles + tpb - 1) / tpb); if (blocks < 1) blocks = 1;
        a.tiles_per_block = (int)tpb;
        if ((rc = ensure_partial(c, blocks, (size_t)np * ncp))) break;
        a.G = dG; a.partial = c->partial; a.ranges = c->ranges;
        if ((rc = launch_struct(c, fn, blocks, BLOCK, proj_lds_bytes(nG, np), a))) break;
        ColArgs ca{};
        ca.ng = 4; ca.nblocks = blocks; ca.npay = np; ca.ncp = ncp; ca.nseg = pick_nseg(np * ncp);
        ca.partial = c->partial; ca.ranges = c->ranges; ca.flux = dflux;
        if ((rc = reduce_level1(c, ca))) break;
        if ((rc = launch_column(c, 4, COL_REDUCE, ca))) break;
        std::vector<double> tmp;
        double *dst = out;
        if (a.boundary) { tmp.resize((size_t)np * ncp); dst = tmp.data(); }   // var 3/4: rows of nG values, last one 0
        if (hipMemcpyAsync(dst, dflux, sizeof(double) * (size_t)np * ncp, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = fail(c, MSGW_ERR_HIP, "D2H of projection failed"); break; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(c, MSGW_ERR_HIP, "sync failed in projection"); break; }
        if (a.boundary)
            for (int p = 0; p < np; ++p) {
                std::memcpy(out + (size_t)p * nG, tmp.data() + (size_t)p * ncp, sizeof(double) * (size_t)ncp);
                out[(size_t)p * nG + ncp] = 0.0;
            }
    } while (0);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(dG);
    (void)hipFree(dflux);
    return rc;
}

template <typename T>
int project_resident(msgw_ctx *c, int var, const double *G, int nG, double *out)
{
    ProjArgsT<T> a{};
    // var 3 / 4 (:199-219): the payloads of var 1 / var 0, summed at the interfaces
    a.n = c->n; a.nG = nG; a.var = (var == 3) ? 1 : (var == 4 ? 0 : var); a.boundary = var >= 3;
    a.bvf2 = (T)std::pow(c->bvf, 2.0); a.f_uni = (T)c->f_uni; a.dz = (T)(G[1] - G[0]);
    a.cdz = T(1) / a.dz; a.mk_ok = markstein_ok(a.dz);
    a.r = ray_ptrs<T>(c);
    if (c->hprop && c->have_hprop) {                           // HPROP: the Coriolis parameter of the current latitude
        a.phi = static_cast<const T *>(c->phi);
        a.two_rot = (T)(2 * 7.2921e-5);                        // 2 * ROT_EARTH (lib/libprop.py:4, :382)
    }
    if (c->nz) {                                               // N(z) column: N at the ray centre rr, volume from the current dmm
        a.dkdl = c->have_nz ? static_cast<const T *>(c->nz_dkdl) : nullptr;
        a.bvfcol = c->bvfcol; a.grids = c->grids; a.nc = c->ng - 1;
        a.gs0 = c->gs0; a.gs_last = c->gs_last; a.inv_dzs = 1.0 / c->dzs;
    }
    const int np = (var == 0 || var == 4) ? 2 : 1;
    return run_projection(c, a, project_kernel<T>(np, c->fvec), Real<T>::TILE, np, G, nG, out);
}

}   // namespace

// ============================================================================ C ABI
extern "C" {

int msgw_abi_version(void) { return MSGW_ABI_VERSION; }

const char *msgw_last_error(const msgw_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int msgw_create_ex(msgw_ctx **out, int device, int64_t nray_cap, int ngrid, unsigned create_flags)
{
    if (!out) return fail(nullptr, MSGW_ERR_ARG, "out is NULL");
    *out = nullptr;
    // ngrid >= 5: the scratch of the fused column update (6*ngrid - 6 doubles) aliases the per-wave flux rows
    // (8*(ngrid - 2) doubles) in LDS
    if (nray_cap < 1 || ngrid < 5) return fail(nullptr, MSGW_ERR_ARG, "need nray_cap >= 1 and ngrid >= 5");
    if (nray_cap > (1ll << 29))
        return fail(nullptr, MSGW_ERR_ARG, "at most 2^29 rays per context (32-bit byte offsets into the SoA arrays)");
    if (create_flags & ~MSGW_DTYPE_F32) return fail(nullptr, MSGW_ERR_ARG, "unknown create flag");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, MSGW_ERR_NOGPU, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, MSGW_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return fail(nullptr, MSGW_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MSGW_ERR_NOGPU, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    msgw_ctx *c = new msgw_ctx();
    c->device = device;
    c->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    snprintf(c->pci_id, sizeof c->pci_id, "%04x:%02x:%02x", prop.pciDomainID, prop.pciBusID, prop.pciDeviceID);
    c->cap = nray_cap;
    c->ng = ngrid;
    c->f32 = (create_flags & MSGW_DTYPE_F32) ? 1 : 0;
    c->esz = c->f32 ? sizeof(float) : sizeof(double);
    c->tile = c->f32 ? Real<float>::TILE : Real<double>::TILE;
#define CR(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int rc_ = fail(nullptr, MSGW_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); msgw_destroy(c); return rc_; } } while (0)
    CR(hipSetDevice(device));
    CR(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CR(hipEventCreate(&c->ev0));
    CR(hipEventCreate(&c->ev1));
    void **rp[A_COUNT] = {&c->dens, &c->rr, &c->mm, &c->drr, &c->kk, &c->ll, &c->dmm, &c->vol, &c->fray,
                          &c->pvf, &c->q_rr, &c->q_mm, &c->q_dens, &c->rr0, &c->mm0, &c->src_dens, &c->src_rr, &c->src_mm,
                          &c->cgbuf};                            // in the order of enum RayArray
    const size_t padded = (((size_t)nray_cap + c->tile - 1) / c->tile + 1) * c->tile;   // whole tiles + one: unconditional vector access
    // one slab, array k at slab + k * pitch (see RayPtrsT); the pitch gets an odd number of 256-B lines on top so
    // that the same tile of different arrays does not start on the same memory channel
    c->pitch = ((padded * c->esz + 4095) / 4096) * 4096 + 256;
    if (const char *e = std::getenv("MSGW_PITCH_SKEW")) c->pitch += 256 * (size_t)std::atoi(e);
    CR(hipMalloc(&c->slab, c->pitch * A_COUNT));
    for (int k = 0; k < A_COUNT; ++k) *rp[k] = c->slab + (size_t)k * c->pitch;
    // column: one allocation carved into equally sized slots of ng doubles (+ 2 for the flux output)
    const size_t slot = (size_t)ngrid + 2;
    const int nslots = 24;
    CR(hipMalloc(&c->colbuf, slot * nslots * sizeof(double) * 2));
    // on c->stream, NOT the null stream: c->stream is non-blocking, so a null-stream fill (asynchronous to the
    // host for device memory) could land after msgw_set_column's uploads and wipe them
    CR(hipMemsetAsync(c->colbuf, 0, slot * nslots * sizeof(double) * 2, c->stream));
    CR(hipMemsetAsync(c->slab, 0, c->pitch * A_COUNT, c->stream));
    CR(hipStreamSynchronize(c->stream));
    double *b = c->colbuf;
    c->grid = b; b += slot; c->grids = b; b += slot; c->rhobar = b; b += slot;
    c->pg = b; b += 2 * slot; c->uu = b; b += slot; c->vv = b; b += slot;
    c->q_uu = b; b += slot; c->q_vv = b; b += slot; c->dudz = b; b += slot; c->dvdz = b; b += slot;
    c->slu = b; b += slot; c->slv = b; b += slot; c->slrho = b; b += slot;
    c->flux = b; b += 2 * slot; c->out_du = b; b += slot; c->out_dv = b; b += slot;
    c->out_flux = b; b += 2 * slot;
    c->alt_uu = b; b += slot; c->alt_vv = b; b += slot; c->alt_q_uu = b; b += slot; c->alt_q_vv = b; b += slot;
#undef CR
    c->cnt.ngrid = ngrid;
    c->cnt.nranks = 1;
    c->cnt.elem_bytes = (int32_t)c->esz;
    if (const char *e = std::getenv("MSGW_PERSIST")) c->persist = std::atoi(e) ? 1 : 0;
    if (const char *e = std::getenv("MSGW_FIXED_NARROW")) c->fixed_narrow_force = std::atoi(e) ? 1 : 0;
    if (const char *e = std::getenv("MSGW_CARRY")) c->carry = std::atoi(e) ? 1 : 0;
    {
        int has = 0;
        if (hipDeviceGetAttribute(&has, hipDeviceAttributeCooperativeLaunch, c->device) != hipSuccess) { has = 0; (void)hipGetLastError(); }
        c->coop = has ? 1 : 0;
        // under a rocprofiler-sdk tool (rocprofv3 exports ROCP_TOOL_LIBRARIES) the process dies at exit() with a SIGSEGV
        // in the runtime's teardown once a cooperative launch has happened (observed with rocprofv3 --kernel-trace and
        // --pmc on ROCm 7.2, after all output files are written): the plain launch of the same kernel there
        const char *pre = std::getenv("LD_PRELOAD");
        if (std::getenv("ROCP_TOOL_LIBRARIES") || std::getenv("ROCPROFILER_LIBRARY_CTOR") ||
            (pre && (std::strstr(pre, "rocprofiler") || std::strstr(pre, "roctracer"))))
            c->coop = 0;
    }
    if (const char *e = std::getenv("MSGW_COOP")) c->coop = (c->coop && std::atoi(e)) ? 1 : 0;
    if (const char *e = std::getenv("MSGW_SERVICE")) c->service = std::atoi(e) ? 1 : 0;
    if (const char *e = std::getenv("MSGW_BALANCE")) c->balance = std::atoi(e);
    if (const char *e = std::getenv("MSGW_PREFETCH")) c->prefetch = std::atoi(e) ? 1 : 0;
    if (const char *e = std::getenv("MSGW_REGTILES")) c->regtiles = std::atoi(e) >= 4 ? 4 : (std::atoi(e) == 3 ? 3 : (std::atoi(e) ? 2 : 0));
    *out = c;
    return MSGW_OK;
}

int msgw_create(msgw_ctx **out, int device, int64_t nray_cap, int ngrid)
{
    return msgw_create_ex(out, device, nray_cap, ngrid, 0u);
}

int msgw_destroy(msgw_ctx *c)
{
    if (!c) return MSGW_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_graph(c);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    xch_teardown(c);
    for (hipEvent_t e : c->kev) (void)hipEventDestroy(e);
    for (void *p : c->ray_bufs) (void)hipFree(p);
    for (auto &sp : c->snap_pool) (void)hipFree(sp.second);
    if (c->slab) (void)hipFree(c->slab);
    if (c->colbuf) (void)hipFree(c->colbuf);
    if (c->partial) (void)hipFree(c->partial);
    if (c->ranges) (void)hipFree(c->ranges);
    if (c->row2) (void)hipFree(c->row2);
    if (c->grp_part) (void)hipFree(c->grp_part);
    if (c->grp_rows) (void)hipFree(c->grp_rows);
    if (c->grp_cnt) (void)hipFree(c->grp_cnt);
    if (c->grp_rows2) (void)hipFree(c->grp_rows2);
    if (c->pdone) (void)hipFree(c->pdone);
    if (c->pstatus) (void)hipFree(c->pstatus);
    if (c->pstatus_host) (void)hipHostFree(c->pstatus_host);
    if (c->fcarry) (void)hipFree(c->fcarry);
    if (c->flux2) (void)hipFree(c->flux2);
    if (c->shtab) (void)hipFree(c->shtab);
    if (c->grp_part2) (void)hipFree(c->grp_part2);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_rows[i]) (void)hipEventDestroy(c->ev_rows[i]);
        if (c->ev_flux[i]) (void)hipEventDestroy(c->ev_flux[i]);
    }
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return MSGW_OK;
}

int msgw_set_config(msgw_ctx *c, double bvf, double f0, double kappa, int saturate_online, int hprop)
{
    if (!c) return MSGW_ERR_ARG;
    c->carry_key = 0;                                          // (the flux of the resident state depends on bvf and f)
    HIPCHK(c, hipSetDevice(c->device));
    if (hprop && !c->lam) {                                    // the six extra ray arrays of the spherical branch
        const size_t padded = (((size_t)c->cap + c->tile - 1) / c->tile + 1) * c->tile;
        void **hp[] = {&c->lam, &c->phi, &c->q_lam, &c->q_phi, &c->q_kk, &c->q_ll};
        for (void **p : hp) {
            HIPCHK(c, hipMalloc(p, padded * c->esz));
            c->ray_bufs.push_back(*p);
            HIPCHK(c, hipMemsetAsync(*p, 0, padded * c->esz, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->bvf = bvf; c->f0 = f0; c->kappa = kappa; c->sat_online = saturate_online ? 1 : 0;
    c->hprop = hprop ? 1 : 0;
    c->have_config = true;
    drop_graph(c);
    return MSGW_OK;
}

int msgw_set_column(msgw_ctx *c, const double *grid, const double *grids, const double *rhobar,
                    const double *pgrad, const double *uu, const double *vv)
{
    if (!c || !grid || !grids || !rhobar || !pgrad || !uu || !vv) return fail(c, MSGW_ERR_ARG, "NULL column pointer");
    HIPCHK(c, hipSetDevice(c->device));
    const int ng = c->ng, nc = ng - 1;
    const size_t B = sizeof(double);
    HIPCHK(c, hipMemcpyAsync(c->grid, grid, ng * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->grids, grids, nc * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->rhobar, rhobar, nc * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->pg, pgrad, 2 * nc * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->uu, uu, nc * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->vv, vv, nc * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));      // host buffers are the caller's
    c->dzg = grid[1] - grid[0];                      // np.diff(grid[:2])[0]  (lib/libprop.py:349, :662)
    c->dzs = grids[1] - grids[0];                    // the same on grids     (:123 with G = grids)
    c->plan_key = 0;                                 // (the launch plan looks at the grid spacing)
    c->xg0 = grid[1];
    c->gs0 = grids[0];
    c->z_bot = grid[0]; c->z_top = grid[ng - 1];     // MSGW_RELAUNCH: the column's extent
    c->xg_last = grid[ng - 2];                       // last point of grid[1:-1]
    c->gs_last = grids[nc - 1];
    if (!(c->dzg > 0) || !(c->dzs > 0)) return fail(c, MSGW_ERR_ARG, "grid must be increasing");
    if (int rc = launch_list(c, rho_slopes_kernel(), (unsigned)((nc + 255) / 256), 256, (int)nc, (const double *)c->grids,
                             (const double *)c->rhobar, c->slrho))
        return rc;
    ColArgs a = make_col_args(c, 0.0, 0);
    a.nblocks = 0; a.nseg = 1;
    if (int rc = launch_column(c, 4, COL_UPDATE, a)) return rc;
    c->have_column = true;
    return MSGW_OK;
}

int msgw_upload_rays(msgw_ctx *c, int64_t n, const double *dens, const double *rr, const double *drr,
                     const double *kk, const double *ll, const double *mm, const double *dmm,
                     const double *fray, const double *dkk, const double *dll, const double *area)
{
    if (!c) return MSGW_ERR_ARG;
    if (n < 1 || n > c->cap) return fail(c, MSGW_ERR_ARG, "n=%lld outside [1, cap=%lld]", (long long)n, (long long)c->cap);
    if (!dens || !rr || !drr || !kk || !ll || !mm || !dmm || !fray || !dkk || !dll || !area)
        return fail(c, MSGW_ERR_ARG, "NULL ray pointer");
    HIPCHK(c, hipSetDevice(c->device));
    // float64 staging of what k_prepare reads in float64 (dkk, dll, area, drr, dmm) -- and, for a float32 state, of
    // every array on its way to the conversion kernel
    Staging st;
    const size_t nstage = c->f32 ? 6 : 3;
    HIPCHK(c, hipMalloc(&st.p, sizeof(double) * (size_t)n * nstage));
    double *s_dkk = st.p, *s_dll = st.p + n, *s_area = st.p + 2 * n;
    const size_t B = (size_t)n * sizeof(double);
    HIPCHK(c, hipMemcpyAsync(s_dkk, dkk, B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s_dll, dll, B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(s_area, area, B, hipMemcpyHostToDevice, c->stream));
    const double *d_drr, *d_dmm;
    if (!c->f32) {
        struct { void *d; const double *h; } cp[] = {
            {c->dens, dens}, {c->rr, rr}, {c->drr, drr}, {c->kk, kk}, {c->ll, ll}, {c->mm, mm}, {c->dmm, dmm}, {c->fray, fray}};
        for (auto &x : cp) HIPCHK(c, hipMemcpyAsync(x.d, x.h, B, hipMemcpyHostToDevice, c->stream));
        d_drr = static_cast<const double *>(c->drr);
        d_dmm = static_cast<const double *>(c->dmm);
    } else {
        double *s_drr = st.p + 3 * n, *s_dmm = st.p + 4 * n, *s_tmp = st.p + 5 * n;
        if (int rc = upload_array(c, c->drr, drr, n, s_drr)) return rc;
        if (int rc = upload_array(c, c->dmm, dmm, n, s_dmm)) return rc;
        struct { void *d; const double *h; } cp[] = {
            {c->dens, dens}, {c->rr, rr}, {c->kk, kk}, {c->ll, ll}, {c->mm, mm}, {c->fray, fray}};
        for (auto &x : cp)
            if (int rc = upload_array(c, x.d, x.h, n, s_tmp)) return rc;     // stream-ordered reuse of s_tmp
        d_drr = s_drr;
        d_dmm = s_dmm;
    }
    // the source a recycled slot returns to (MSGW_RELAUNCH) is the state at upload
    const size_t BE = (size_t)n * c->esz;
    HIPCHK(c, hipMemcpyAsync(c->src_dens, c->dens, BE, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->src_rr, c->rr, BE, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->src_mm, c->mm, BE, hipMemcpyDeviceToDevice, c->stream));
    const unsigned pg = (unsigned)((n + 255) / 256);
    int rc;
    if (c->f32)
        rc = launch_list(c, prepare_kernel<float>(), pg, 256, (long long)n, (const double *)s_dkk, (const double *)s_dll,
                         (const double *)s_area, d_drr, d_dmm, static_cast<float *>(c->vol), static_cast<float *>(c->pvf));
    else
        rc = launch_list(c, prepare_kernel<double>(), pg, 256, (long long)n, (const double *)s_dkk, (const double *)s_dll,
                         (const double *)s_area, d_drr, d_dmm, static_cast<double *>(c->vol), static_cast<double *>(c->pvf));
    if (rc) return rc;
    c->have_nz = false;
    if (c->nz) {                                               // N(z) column: dkk*dll and rr_mm_area stay on the device
        const int rc4 = c->f32
            ? launch_list(c, nz_prepare_kernel<float>(), pg, 256, (long long)n, (const double *)s_dkk, (const double *)s_dll,
                          (const double *)s_area, static_cast<float *>(c->nz_dkdl), static_cast<float *>(c->nz_area))
            : launch_list(c, nz_prepare_kernel<double>(), pg, 256, (long long)n, (const double *)s_dkk, (const double *)s_dll,
                          (const double *)s_area, static_cast<double *>(c->nz_dkdl), static_cast<double *>(c->nz_area));
        if (rc4) return rc4;
        c->have_nz = true;
    }
    // inert padding up to a whole tile (finite, never deposited: validity is index < n)
    const long long n_pad = ((n + c->tile - 1) / c->tile + 1) * c->tile;   // a workgroup's last tile may overhang by < TILE
    if (n_pad > n)
        if (int rc2 = c->f32 ? fill_padding<float>(c, n, n_pad) : fill_padding<double>(c, n, n_pad)) return rc2;
    if (c->pstatus) HIPCHK(c, hipMemsetAsync(c->pstatus, 0, 128, c->stream));   // a fresh state: forget an old time-out
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->pstatus_host) *reinterpret_cast<volatile int *>(c->pstatus_host) = 0;
    c->status_armed = false;
    c->carry_key = 0;                                          // a new ray state: no flux carried over
    // a single latitude for all rays (the 1-D column case, raytracer.py:87) keeps f in a scalar
    bool uni = true;
    for (int64_t i = 1; i < n && uni; ++i) uni = (std::memcmp(&fray[i], &fray[0], sizeof(double)) == 0);
    c->fvec = !uni;
    c->f_uni = fray[0];
    c->n = n;
    geometry(c, n);
    if (int rc3 = ensure_partial(c, c->blocks, (size_t)2 * (c->ng - 2))) return rc3;
    if (int rc3 = ensure_groups(c)) return rc3;
    c->have_rays = true;
    c->have_hprop = false;                                     // lam, phi belong to the previous set of rays
    c->cnt.nray = n;
    c->cnt.blocks = c->blocks;
    drop_graph(c);
    return MSGW_OK;
}

int msgw_upload_hprop(msgw_ctx *c, int64_t n, const double *lam, const double *phi)
{
    if (!c || !lam || !phi) return fail(c, MSGW_ERR_ARG, "NULL argument");
    if (!c->hprop || !c->lam) return fail(c, MSGW_ERR_ARG, "HPROP is off (msgw_set_config(..., hprop = 1) first)");
    if (!c->have_rays || n != c->n) return fail(c, MSGW_ERR_ARG, "call msgw_upload_rays first, with the same n");
    HIPCHK(c, hipSetDevice(c->device));
    Staging st;
    if (c->f32) HIPCHK(c, hipMalloc(&st.p, sizeof(double) * (size_t)n * 2));
    if (int rc = upload_array(c, c->lam, lam, n, st.p)) return rc;
    if (int rc = upload_array(c, c->phi, phi, n, st.p ? st.p + n : nullptr)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_hprop = true;
    return MSGW_OK;
}

int msgw_download_hprop(msgw_ctx *c, int64_t n, int tendencies, double *lam, double *phi, double *kk, double *ll)
{
    if (!c) return MSGW_ERR_ARG;
    if (!c->hprop || !c->have_hprop || n != c->n) return fail(c, MSGW_ERR_ARG, "no HPROP state of that size on the device");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_status(c)) return rc;
    const void *src[4] = {tendencies ? c->q_lam : c->lam, tendencies ? c->q_phi : c->phi,
                          tendencies ? (const void *)c->q_kk : c->kk, tendencies ? (const void *)c->q_ll : c->ll};
    double *dst[4] = {lam, phi, kk, ll};
    Staging st;
    if (c->f32) HIPCHK(c, hipMalloc(&st.p, sizeof(double) * (size_t)n * 4));
    for (int i = 0; i < 4; ++i)
        if (dst[i]) if (int rc = download_array(c, dst[i], src[i], n, st.p ? st.p + (size_t)i * n : nullptr)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSGW_OK;
}

int msgw_set_bvf_column(msgw_ctx *c, const double *bvf)
{
    if (!c) return MSGW_ERR_ARG;
    c->carry_key = 0;
    HIPCHK(c, hipSetDevice(c->device));
    if (!bvf) {                                                // back to the scalar of msgw_set_config
        c->nz = 0;
        return MSGW_OK;
    }
    const size_t nc = (size_t)c->ng - 1;
    if (!c->bvfcol) {
        // allocate into locals and commit only when everything is there: a failed call leaves the context as it was
        const size_t padded = (((size_t)c->cap + c->tile - 1) / c->tile + 1) * c->tile;
        double *col = nullptr;
        void *ray[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        bool ok = hipMalloc(&col, nc * sizeof(double)) == hipSuccess;
        for (int i = 0; i < 5 && ok; ++i)
            ok = hipMalloc(&ray[i], padded * c->esz) == hipSuccess &&
                 hipMemsetAsync(ray[i], 0, padded * c->esz, c->stream) == hipSuccess;
        if (!ok) {
            (void)hipStreamSynchronize(c->stream);
            for (void *p : ray) if (p) (void)hipFree(p);
            if (col) (void)hipFree(col);
            (void)hipGetLastError();
            return fail(c, MSGW_ERR_HIP, "hipMalloc of the N(z) column buffers failed");
        }
        c->nz_q_drr = ray[0]; c->nz_q_dmm = ray[1]; c->nz_dkdl = ray[2]; c->nz_area = ray[3]; c->nz_drr0 = ray[4];
        for (void *p : ray) c->ray_bufs.push_back(p);
        c->bvfcol = col;
        c->ray_bufs.push_back(col);
    }
    HIPCHK(c, hipMemcpyAsync(c->bvfcol, bvf, nc * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!c->nz) c->have_nz = false;                            // dkk*dll / rr_mm_area come with the next msgw_upload_rays
    c->nz = 1;
    drop_graph(c);
    return MSGW_OK;
}

int msgw_download_extents(msgw_ctx *c, int64_t n, int tendencies, double *drr, double *dmm)
{
    if (!c || !c->have_rays) return fail(c, MSGW_ERR_ARG, "no rays resident");
    if (n != c->n) return fail(c, MSGW_ERR_ARG, "n=%lld but %lld rays are resident", (long long)n, (long long)c->n);
    if (tendencies && !c->nz) return fail(c, MSGW_ERR_ARG, "drr, dmm have tendencies only with an N(z) column");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_status(c)) return rc;
    Staging st;
    if (c->f32) HIPCHK(c, hipMalloc(&st.p, sizeof(double) * (size_t)n * 2));
    const void *s0 = tendencies ? (const void *)c->nz_q_drr : c->drr, *s1 = tendencies ? (const void *)c->nz_q_dmm : c->dmm;
    if (drr) if (int rc = download_array(c, drr, s0, n, st.p)) return rc;
    if (dmm) if (int rc = download_array(c, dmm, s1, n, st.p ? st.p + n : nullptr)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSGW_OK;
}

int msgw_set_relaunch(msgw_ctx *c, double frac)
{
    if (!c) return MSGW_ERR_ARG;
    if (!(frac >= 0.0) || !(frac < 1.0)) return fail(c, MSGW_ERR_ARG, "relaunch fraction must be in [0, 1)");
    c->relaunch_frac = frac;
    drop_graph(c);
    return MSGW_OK;
}

int msgw_set_relaunch_source(msgw_ctx *c, int64_t n, const double *dens, const double *rr, const double *mm)
{
    if (!c || !dens || !rr || !mm) return fail(c, MSGW_ERR_ARG, "NULL argument");
    if (!c->have_rays || n != c->n) return fail(c, MSGW_ERR_ARG, "call msgw_upload_rays first, with the same n");
    HIPCHK(c, hipSetDevice(c->device));
    Staging st;
    if (c->f32) HIPCHK(c, hipMalloc(&st.p, sizeof(double) * (size_t)n * 3));
    if (int rc = upload_array(c, c->src_dens, dens, n, st.p)) return rc;
    if (int rc = upload_array(c, c->src_rr, rr, n, st.p ? st.p + n : nullptr)) return rc;
    if (int rc = upload_array(c, c->src_mm, mm, n, st.p ? st.p + 2 * n : nullptr)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));                // host buffers are the caller's
    return MSGW_OK;
}

int msgw_set_tuning(msgw_ctx *c, int blocks_per_cu, int graph_steps)
{
    if (!c) return MSGW_ERR_ARG;
    if (blocks_per_cu < 1 || blocks_per_cu > 64 || graph_steps < 0 || graph_steps > 64)
        return fail(c, MSGW_ERR_ARG, "blocks_per_cu in [1,64], graph_steps in [0,64]");
    c->blocks_per_cu = blocks_per_cu;
    c->graph_steps = graph_steps;
    c->plan_key = 0;
    drop_graph(c);
    if (c->have_rays) {
        geometry(c, c->n);
        if (int rc = ensure_partial(c, c->blocks, (size_t)2 * (c->ng - 2))) return rc;
        if (int rc = ensure_groups(c)) return rc;
        c->cnt.blocks = c->blocks;
    }
    return MSGW_OK;
}

int msgw_step(msgw_ctx *c, double dt, int nsteps, unsigned flags)
{
    if (int rc = ready(c)) return rc;
    if (nsteps < 0) return fail(c, MSGW_ERR_ARG, "nsteps < 0");
    if ((flags & (MSGW_DIRECT_SAT | MSGW_DIRECT_SAT_QUIRK)) && c->sat_online)
        flags &= ~(MSGW_DIRECT_SAT | MSGW_DIRECT_SAT_QUIRK);      // raytracer.py:182: only if not online
    HIPCHK(c, hipSetDevice(c->device));
    const bool time_kernels = (flags & MSGW_TIME_KERNELS) != 0;
    // With a real collective in the chain launches stay eager: hipGraph capture of ncclAllReduce
    // across ranks cannot be exercised on the 1-GPU development boxes, so it is not relied upon.
    // (measured with a 1-rank communicator: replaying the captured two-stream chain is slower than launching it
    // eagerly, 146 vs 120 us per step)
    const bool eager = time_kernels || (flags & MSGW_NO_GRAPH) || c->graph_steps == 0 || c->nranks > 1 || c->force_coll ||
                       (flags & MSGW_FIXED_BACKGROUND);      // fixed background: one launch for all steps anyway
    const unsigned gflags = flags & ~(MSGW_NO_GRAPH | MSGW_TIME_KERNELS);
    if (time_kernels) c->kev_used = 0;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (c->hprop || c->nz) {                                   // HPROP_GLOBAL = True / N(z) column: the general per-stage chain
        c->cnt.persist_steps = 0;
        c->carry_key = 0;
        if (int rc = c->f32 ? enqueue_steps_chain<float>(c, dt, gflags, nsteps) : enqueue_steps_chain<double>(c, dt, gflags, nsteps))
            return rc;
        HIPCHK(c, hipEventRecord(c->ev1, c->stream));
        c->cnt.ray_steps_total += c->n * (int64_t)nsteps;
        c->cnt.algorithmic_bytes_total += chain_words_per_step(c) * (double)c->esz * (double)c->n * nsteps;
        return MSGW_OK;
    }
    if (int rc = c->f32 ? step_impl<float>(c, dt, nsteps, gflags, eager, time_kernels)
                        : step_impl<double>(c, dt, nsteps, gflags, eager, time_kernels))
        return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    c->cnt.ray_steps_total += c->n * (int64_t)nsteps;
    // SURVEY 8d words per ray-step: fixed background 6, coupled 35, coupled + online saturation 45
    c->cnt.algorithmic_bytes_total += ((flags & MSGW_FIXED_BACKGROUND) ? 6.0 : (c->sat_online ? 45.0 : 35.0)) *
                                      (double)c->esz * (double)c->n * nsteps;
    c->cnt.graph_steps = c->gexec ? c->g_steps : 0;
    if (time_kernels) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (size_t i = 0; i + 1 < c->kev_used; i += 2) {
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, c->kev[i], c->kev[i + 1]));
            c->cnt.ray_kernel_ms_sum += ms;
            c->cnt.ray_kernel_launches += 1;
        }
        return check_status(c);
    }
    return MSGW_OK;
}

int msgw_rhs(msgw_ctx *c, double dt, unsigned flags, double *st_dens, double *st_rr, double *st_mm,
             double *st_uu, double *st_vv, double *pm_flux)
{
    if (int rc = ready(c)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_status(c)) return rc;
    if (c->hprop || c->nz) {
        if (int rc = c->f32 ? launch_chain_stage<float>(c, 3, make_chain_args<float>(c, dt, flags))
                            : launch_chain_stage<double>(c, 3, make_chain_args<double>(c, dt, flags)))
            return rc;
    } else if (c->f32) {
        if (int rc = launch_probe<float>(c, make_stage_args<float>(c, dt, flags), c->sat_online != 0, true)) return rc;
    } else {
        if (int rc = launch_probe<double>(c, make_stage_args<double>(c, dt, flags), c->sat_online != 0, true)) return rc;
    }
    const ColArgs ca = make_col_args(c, dt, flags);
    if (int rc = column_stage(c, 3, ca)) return rc;
    const size_t nc = c->ng - 1;
    Staging st;
    if (c->f32) HIPCHK(c, hipMalloc(&st.p, sizeof(double) * (size_t)c->n * 3));
    if (st_dens) if (int rc = download_array(c, st_dens, c->q_dens, c->n, st.p)) return rc;
    if (st_rr) if (int rc = download_array(c, st_rr, c->q_rr, c->n, st.p ? st.p + c->n : nullptr)) return rc;
    if (st_mm) if (int rc = download_array(c, st_mm, c->q_mm, c->n, st.p ? st.p + 2 * c->n : nullptr)) return rc;
    if (st_uu) HIPCHK(c, hipMemcpyAsync(st_uu, c->out_du, nc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (st_vv) HIPCHK(c, hipMemcpyAsync(st_vv, c->out_dv, nc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (pm_flux) HIPCHK(c, hipMemcpyAsync(pm_flux, c->out_flux, 2 * (size_t)c->ng * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSGW_OK;
}

int msgw_project(msgw_ctx *c, int var, const double *G, int nG, double *out)
{
    if (int rc = ready(c)) return rc;
    if (var < 0 || var > 4) return fail(c, MSGW_ERR_ARG, "wave_projection var=%d: must be 0 .. 4", var);
    if (!G || !out || nG < 3) return fail(c, MSGW_ERR_ARG, "bad projection grid");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_status(c)) return rc;
    return c->f32 ? project_resident<float>(c, var, G, nG, out) : project_resident<double>(c, var, G, nG, out);
}

int msgw_project_arrays(msgw_ctx *c, int64_t n, int var, double bvf, const double *dens,
                        const double *rr_low, const double *rr_up, const double *kk, const double *ll,
                        const double *mm_low, const double *mm_up, const double *dkk, const double *dll,
                        const double *dmm, const double *fray, const double *G, int nG, double *out)
{
    if (!c) return MSGW_ERR_ARG;
    if (var < 0 || var > 4) return fail(c, MSGW_ERR_ARG, "wave_projection var=%d: must be 0 .. 4", var);
    if (n < 1 || !G || !out || nG < 3) return fail(c, MSGW_ERR_ARG, "bad projection arguments");
    const double *h[11] = {dens, rr_low, rr_up, kk, ll, mm_low, mm_up, dkk, dll, dmm, fray};
    for (const double *p : h) if (!p) return fail(c, MSGW_ERR_ARG, "NULL array");
    if (std::isnan(bvf) && !(c->nz && c->bvfcol && c->have_column))
        return fail(c, MSGW_ERR_ARG, "bvf = NaN asks for the context's N(z) column, but none is set "
                    "(msgw_set_bvf_column, msgw_set_column)");
    HIPCHK(c, hipSetDevice(c->device));
    const int tile = Real<double>::TILE;                       // caller arrays are float64 whatever the context holds
    const size_t padded = (((size_t)n + tile - 1) / tile) * tile;
    double *buf = nullptr;
    HIPCHK(c, hipMalloc(&buf, sizeof(double) * padded * 11));
    int rc = MSGW_OK;
    if (hipMemsetAsync(buf, 0, sizeof(double) * padded * 11, c->stream) != hipSuccess)
        rc = fail(c, MSGW_ERR_HIP, "memset failed in msgw_project_arrays");
    const double *d[11];
    for (int i = 0; i < 11 && rc == MSGW_OK; ++i) {
        d[i] = buf + padded * i;
        if (hipMemcpyAsync(buf + padded * i, h[i], sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream) != hipSuccess)
            rc = fail(c, MSGW_ERR_HIP, "H2D failed in msgw_project_arrays");
    }
    if (rc == MSGW_OK) {
        ProjArgs a{};
        a.n = n; a.nG = nG; a.var = (var == 3) ? 1 : (var == 4 ? 0 : var); a.boundary = var >= 3;
        a.bvf2 = std::pow(bvf, 2.0); a.f_uni = 0.0; a.dz = G[1] - G[0];
        a.cdz = 1.0 / a.dz; a.mk_ok = markstein_ok(a.dz);
        if (std::isnan(bvf)) {                                 // N from the context's column, at .5 * (rr_low + rr_up)
            a.bvfcol = c->bvfcol; a.grids = c->grids; a.nc = c->ng - 1;
            a.gs0 = c->gs0; a.gs_last = c->gs_last; a.inv_dzs = 1.0 / c->dzs;
        }
        a.e = ProjExplicit{d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10]};
        const int np = (var == 0 || var == 4) ? 2 : 1;
        rc = run_projection(c, a, project_arrays_kernel(np), tile, np, G, nG, out);
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(buf);
    return rc;
}

int msgw_saturation(msgw_ctx *c, int64_t n, double dt, int direct, const double *dens,
                    const double *rr_center, const double *rr_center_st, const double *drr,
                    const double *drr_st, const double *kk, const double *ll, const double *mm_center,
                    const double *mm_center_st, const double *dkk, const double *dll,
                    const double *area, double *out)
{
    if (!c) return MSGW_ERR_ARG;
    if (!c->have_config || !c->have_column) return fail(c, MSGW_ERR_ARG, "msgw_saturation needs set_config and set_column first");
    if (n < 1 || !out) return fail(c, MSGW_ERR_ARG, "bad saturation arguments");
    const double *h[12] = {dens, rr_center, rr_center_st, drr, drr_st, kk, ll, mm_center, mm_center_st, dkk, dll, area};
    for (const double *p : h) if (!p) return fail(c, MSGW_ERR_ARG, "NULL array");
    HIPCHK(c, hipSetDevice(c->device));
    double *buf = nullptr;
    HIPCHK(c, hipMalloc(&buf, sizeof(double) * (size_t)n * 13));
    int rc = MSGW_OK;
    for (int i = 0; i < 12 && rc == MSGW_OK; ++i)
        if (hipMemcpyAsync(buf + (size_t)n * i, h[i], sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream) != hipSuccess)
            rc = fail(c, MSGW_ERR_HIP, "H2D failed in msgw_saturation");
    if (rc == MSGW_OK) {
        SatArgs a{};
        a.n = n; a.nc = c->ng - 1; a.direct = direct ? 1 : 0; a.dt = dt;
        a.bvf2 = std::pow(c->bvf, 2.0); a.f0sq = std::pow(c->f0, 2.0); a.sat_c = std::pow(c->kappa, 2.0) * .5;
        a.gs0 = c->gs0; a.gs_last = c->gs_last; a.inv_dzs = 1.0 / c->dzs;
        const double *b = buf;
        a.dens = b; a.rr = b + n; a.rr_st = b + 2 * n; a.drr = b + 3 * n; a.drr_st = b + 4 * n;
        a.kk = b + 5 * n; a.ll = b + 6 * n; a.mm = b + 7 * n; a.mm_st = b + 8 * n;
        a.dkk = b + 9 * n; a.dll = b + 10 * n; a.area = b + 11 * n;
        a.grids = c->grids; a.rhobar = c->rhobar; a.slrho = c->slrho;
        a.bvfcol = c->nz ? c->bvfcol : nullptr;
        a.out = buf + (size_t)n * 12;
        rc = launch_struct(c, saturation_kernel(), (unsigned)((n + BLOCK - 1) / BLOCK), BLOCK, 0, a);
        if (rc == MSGW_OK && hipMemcpyAsync(out, a.out, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
            rc = fail(c, MSGW_ERR_HIP, "D2H failed in msgw_saturation");
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == MSGW_OK) rc = fail(c, MSGW_ERR_HIP, "sync failed in msgw_saturation");
    (void)hipFree(buf);
    return rc;
}

int msgw_probe_arith(msgw_ctx *c, int64_t n, const double *x, double d, double *out_sqrt, double *out_div,
                     const double *y, double *out_quot)
{
    if (!c || !x || !out_sqrt || !out_div || n < 1 || (y && !out_quot)) return fail(c, MSGW_ERR_ARG, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    double *buf = nullptr;
    const size_t B = sizeof(double) * (size_t)n;
    HIPCHK(c, hipMalloc(&buf, B * 5));
    int rc = MSGW_OK;
    if (hipMemcpyAsync(buf, x, B, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        (y && hipMemcpyAsync(buf + 3 * n, y, B, hipMemcpyHostToDevice, c->stream) != hipSuccess))
        rc = fail(c, MSGW_ERR_HIP, "H2D failed in msgw_probe_arith");
    if (rc == MSGW_OK)
        rc = launch_list(c, probe_arith_kernel(), (unsigned)((n + 255) / 256), 256, (long long)n, (const double *)buf, d, 1.0 / d,
                         markstein_ok(d), buf + n, buf + 2 * n, (const double *)(y ? buf + 3 * n : nullptr), buf + 4 * n);
    if (rc == MSGW_OK && (hipMemcpyAsync(out_sqrt, buf + n, B, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                          hipMemcpyAsync(out_div, buf + 2 * n, B, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                          (y && hipMemcpyAsync(out_quot, buf + 4 * n, B, hipMemcpyDeviceToHost, c->stream) != hipSuccess)))
        rc = fail(c, MSGW_ERR_HIP, "D2H failed in msgw_probe_arith");
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == MSGW_OK) rc = fail(c, MSGW_ERR_HIP, "sync failed in msgw_probe_arith");
    (void)hipFree(buf);
    return rc;
}

int msgw_download_rays(msgw_ctx *c, int64_t n, double *dens, double *rr, double *mm)
{
    if (!c || !c->have_rays) return fail(c, MSGW_ERR_ARG, "no rays resident");
    if (n != c->n) return fail(c, MSGW_ERR_ARG, "n=%lld but %lld rays are resident", (long long)n, (long long)c->n);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_status(c)) return rc;
    Staging st;
    if (c->f32) HIPCHK(c, hipMalloc(&st.p, sizeof(double) * (size_t)n * 3));
    if (dens) if (int rc = download_array(c, dens, c->dens, n, st.p)) return rc;
    if (rr) if (int rc = download_array(c, rr, c->rr, n, st.p ? st.p + n : nullptr)) return rc;
    if (mm) if (int rc = download_array(c, mm, c->mm, n, st.p ? st.p + 2 * n : nullptr)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSGW_OK;
}

int msgw_download_column(msgw_ctx *c, double *uu, double *vv)
{
    if (!c || !c->have_column) return fail(c, MSGW_ERR_ARG, "no column resident");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_status(c)) return rc;
    const size_t B = (size_t)(c->ng - 1) * sizeof(double);
    if (uu) HIPCHK(c, hipMemcpyAsync(uu, c->uu, B, hipMemcpyDeviceToHost, c->stream));
    if (vv) HIPCHK(c, hipMemcpyAsync(vv, c->vv, B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSGW_OK;
}

// ---- snapshots: a stream-ordered device copy of the evolving slots, for host mirrors that hand out results lazily
// (msgwam_amd/libprop.py: the arrays RK3 returns are downloaded on first access; a state that is advanced before it
// was read stays readable through its snapshot).  ~24 MB of device copies per 1e6 float64 rays.
struct msgw_snapshot {
    void *buf = nullptr;
    size_t bytes = 0;
    int64_t n = 0;
    size_t off[MSGW_SLOT_COUNT] = {0};     // byte offset of each slot in buf
    size_t len[MSGW_SLOT_COUNT] = {0};     // bytes (0: not part of this snapshot)
    int ray_typed[MSGW_SLOT_COUNT] = {0};  // stored in the ray type (float32 contexts: converted on download)
};

int msgw_snapshot_create(msgw_ctx *c, msgw_snapshot **out)
{
    if (!c || !out) return MSGW_ERR_ARG;
    *out = nullptr;
    if (!c->have_rays || !c->have_column) return fail(c, MSGW_ERR_ARG, "nothing resident to snapshot");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t ray = (size_t)c->n * c->esz;
    const size_t col = (size_t)(c->ng - 1) * sizeof(double);
    const bool hp = c->hprop && c->have_hprop, nz = c->nz && c->have_nz;
    msgw_snapshot *s = new msgw_snapshot();
    const void *src[MSGW_SLOT_COUNT] = {c->dens, c->rr, c->mm, c->uu, c->vv, hp ? c->lam : nullptr, hp ? c->phi : nullptr,
                                        hp ? c->kk : nullptr, hp ? c->ll : nullptr, nz ? c->drr : nullptr, nz ? c->dmm : nullptr};
    const size_t len[MSGW_SLOT_COUNT] = {ray, ray, ray, col, col, ray, ray, ray, ray, ray, ray};
    size_t bytes = 0;
    for (int k = 0; k < MSGW_SLOT_COUNT; ++k)
        if (src[k]) { s->off[k] = bytes; s->len[k] = len[k]; s->ray_typed[k] = len[k] == ray && k != MSGW_SLOT_UU && k != MSGW_SLOT_VV; bytes += (len[k] + 255) / 256 * 256; }
    for (size_t i = 0; i < c->snap_pool.size(); ++i)
        if (c->snap_pool[i].first >= bytes) {
            s->buf = c->snap_pool[i].second; s->bytes = c->snap_pool[i].first;
            c->snap_pool.erase(c->snap_pool.begin() + i);
            break;
        }
    if (!s->buf) {
        if (hipMalloc(&s->buf, bytes) != hipSuccess) { delete s; return fail(c, MSGW_ERR_HIP, "hipMalloc of a snapshot failed"); }
        s->bytes = bytes;
    }
    s->n = c->n;
    for (int k = 0; k < MSGW_SLOT_COUNT; ++k)
        if (s->len[k] &&
            hipMemcpyAsync(static_cast<char *>(s->buf) + s->off[k], src[k], s->len[k], hipMemcpyDeviceToDevice, c->stream) != hipSuccess) {
            (void)msgw_snapshot_destroy(c, s);                 // back to the pool (or freed): nothing leaks on the error path
            return fail(c, MSGW_ERR_HIP, "device copy of a snapshot failed");
        }
    *out = s;
    return MSGW_OK;
}

int msgw_snapshot_download(msgw_ctx *c, msgw_snapshot *s, int slot, double *out)
{
    if (!c || !s || !s->buf || !out) return MSGW_ERR_ARG;
    if (slot < 0 || slot >= MSGW_SLOT_COUNT || !s->len[slot]) return fail(c, MSGW_ERR_ARG, "slot %d is not part of this snapshot", slot);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_status(c)) return rc;
    const char *src = static_cast<const char *>(s->buf) + s->off[slot];
    if (s->ray_typed[slot] && c->f32) {
        Staging st;
        HIPCHK(c, hipMalloc(&st.p, sizeof(double) * (size_t)s->n));
        if (int rc = download_array(c, out, src, s->n, st.p)) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return MSGW_OK;
    }
    HIPCHK(c, hipMemcpyAsync(out, src, s->len[slot], hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSGW_OK;
}

int msgw_snapshot_destroy(msgw_ctx *c, msgw_snapshot *s)
{
    if (!s) return MSGW_OK;
    if (c && s->buf) {
        // reuse is stream-ordered: the next snapshot's copies are enqueued behind everything that read this buffer
        if (c->snap_pool.size() < 8) c->snap_pool.emplace_back(s->bytes, s->buf);
        else { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); (void)hipFree(s->buf); }
    }
    delete s;
    return MSGW_OK;
}

int msgw_sync(msgw_ctx *c)
{
    if (!c) return MSGW_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_status(c);
}

int msgw_comm_unique_id(void *id128)
{
    if (!id128) return fail(nullptr, MSGW_ERR_ARG, "id128 is NULL");
    if (!g_rccl.load()) return fail(nullptr, MSGW_ERR_RCCL, "%s", g_rccl.load_error.c_str());
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != 0) return fail(nullptr, MSGW_ERR_RCCL, "ncclGetUniqueId failed (%d)", r);
    std::memcpy(id128, &id, sizeof id);
    return MSGW_OK;
}

namespace {

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void xch_teardown(msgw_ctx *c)
{
    for (void *p : c->xch_peer_open) (void)hipIpcCloseMemHandle(p);
    c->xch_peer_open.clear();
    if (c->xch_local) { (void)hipFree(c->xch_local); c->xch_local = nullptr; }
    if (c->xch_registered) { (void)hipHostUnregister(c->xch_hdr); c->xch_registered = false; }
    if (c->xch_hdr) { munmap(c->xch_hdr, c->xch_bytes); c->xch_hdr = nullptr; }
    if (c->xch_scratch) { (void)hipFree(c->xch_scratch); c->xch_scratch = nullptr; }
    c->xch_rows = nullptr; c->xch_flags = nullptr; c->xch_peer_rows = nullptr; c->xch_peer_flags = nullptr;
    c->xch_ok = false; c->xch_bytes = 0; c->xch_direct = false; c->tenants = 1;
}

// Host barrier `k` of the set-up: every rank reports `ok`; returns true when all ranks arrived in
// time and all of them reported ok.
bool xch_barrier(msgw_ctx *c, int k, bool ok, double timeout_s)
{
    XchHeader *h = c->xch_hdr;
    if (ok) __atomic_fetch_add(&h->okay[k], 1u, __ATOMIC_ACQ_REL);
    __atomic_fetch_add(&h->arrived[k], 1u, __ATOMIC_ACQ_REL);
    const double t0 = now_s();
    while (__atomic_load_n(&h->arrived[k], __ATOMIC_ACQUIRE) < (unsigned int)c->nranks) {
        if (now_s() - t0 > timeout_s) return false;
        usleep(200);
    }
    return __atomic_load_n(&h->okay[k], __ATOMIC_ACQUIRE) == (unsigned int)c->nranks;
}

// Per-call agreement of the ranks of one node (host side, through the segment header): every rank publishes
// `mine`; *all = every rank said yes.  Values are double-buffered by the parity of the call counter: a rank can
// only start call k+2 after every rank has finished reading call k.  False = a rank did not show up in time.
bool xch_agree_step(msgw_ctx *c, bool mine, bool *all)
{
    XchHeader *h = c->xch_hdr;
    *all = mine;
    if (!h || c->nranks <= 1) return true;
    const unsigned long long k = ++c->xch_calls;
    __atomic_store_n(&h->agree_val[c->rank][k & 1], mine ? 1u : 0u, __ATOMIC_RELAXED);
    __atomic_store_n(&h->agree_cnt[c->rank], k, __ATOMIC_RELEASE);
    const double t0 = now_s();
    bool yes = true;
    for (int r = 0; r < c->nranks; ++r) {
        unsigned spins = 0;
        while (__atomic_load_n(&h->agree_cnt[r], __ATOMIC_ACQUIRE) < k) {
            if ((++spins & 1023u) == 0) {
                if (now_s() - t0 > 60.0) return false;
                usleep(50);
            }
        }
        yes = yes && __atomic_load_n(&h->agree_val[r][k & 1], __ATOMIC_RELAXED) != 0u;
    }
    *all = yes;
    return true;
}

// When there is an RCCL communicator the outcome is also agreed through it (covers ranks on
// other nodes, which can never join this node's segment).
bool xch_agree_rccl(msgw_ctx *c, bool ok)
{
    if (!c->comm) return ok;
    int v = ok ? 1 : 0;
    if (hipMemcpyAsync(c->xch_scratch + 1, &v, sizeof v, hipMemcpyHostToDevice, c->stream) != hipSuccess) return false;
    if (g_rccl.AllReduce(c->xch_scratch + 1, c->xch_scratch + 1, 1, ncclInt32, ncclMin, c->comm, c->stream) != 0) return false;
    if (hipMemcpyAsync(&v, c->xch_scratch + 1, sizeof v, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return false;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return false;
    return v == 1;
}

// `rounds` node-level sums of known rows through the very code path of the persistent kernel, with the transport
// described by `x` (sequence numbers x.seq + 1 ...).
bool xch_selftest(msgw_ctx *c, const XchArgs &x)
{
    XchTestArgs t{};
    t.x = x; t.rounds = XCH_TEST_ROUNDS; t.result = c->xch_scratch;
    int res = 0;
    if (hipMemsetAsync(c->xch_scratch, 0, 64, c->stream) != hipSuccess) return false;
    if (launch_struct(c, xch_selftest_kernel(), 1, BLOCK, 0, t) != MSGW_OK) return false;
    return hipMemcpyAsync(&res, c->xch_scratch, sizeof res, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
           hipStreamSynchronize(c->stream) == hipSuccess && res == 1;
}

// Create/join the communicator's segment, set up the transports, run the self-tests, agree.  Never
// fatal: on any failure xch_ok stays false and multi-rank steps use the all-reduce launch chain.
//  1. POSIX shared-memory segment (bootstrap + host barriers + per-call agreement; also the fallback transport,
//     registered with the GPU as fine-grained host memory);
//  2. device-resident transport: every rank allocates its exchange buffer in its own HBM, publishes the HIP IPC
//     handle in the segment and maps the buffers of all other ranks (xGMI peer writes; two ranks that share one GPU
//     in a rehearsal map each other's buffers in the same HBM);
//  3. the ranks count how many of them share each physical device (`tenants`): they split its residency slots.
void xch_setup(msgw_ctx *c, const void *id128, std::string &why, int want_direct)
{
    xch_teardown(c);
    c->xch_seq = 0;
    c->xch_calls = 0;
    if (c->nranks > XCH_MAX_RANKS) { why = "more than 64 ranks (one polling lane per rank)"; return; }   // the same on every rank
    if (hipMalloc(&c->xch_scratch, XCH_SCRATCH_BYTES) != hipSuccess) { (void)hipGetLastError(); why = "hipMalloc"; return; }
    // the name is derived from the communicator's unique id (FNV-1a), identical on all ranks
    unsigned long long hsh = 1469598103934665603ull;
    for (int i = 0; i < 128; ++i) hsh = (hsh ^ ((const unsigned char *)id128)[i]) * 1099511628211ull;
    char name[64];
    snprintf(name, sizeof name, "/msgw-%016llx", hsh);
    const int ncols = 2 * (c->ng - 2);
    const int stride = ((ncols > 64 ? ncols : 64) + 7) / 8 * 8;
    const size_t flags_bytes = ((size_t)c->nranks * 64 + 4095) / 4096 * 4096;
    const size_t rows_off = XCH_FLAGS_OFF + flags_bytes;
    const size_t rows_bytes = sizeof(double) * 2 * 2 * (size_t)c->nranks * stride;   // [2 slots][ranks][stride] x two 8-byte words per float64 (tagged granules)
    const size_t bytes = (rows_off + rows_bytes + 4095) / 4096 * 4096;
    const double join_timeout = 60.0;
    int fd = -1;
    bool ok = true;
    if (c->rank == 0) {
        shm_unlink(name);                                      // a stale segment of a crashed run
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { ok = false; why = "shm_open/ftruncate"; }
    } else {
        const double t0 = now_s();
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == bytes) break;
            if (fd >= 0) { close(fd); fd = -1; }
            if (now_s() - t0 > join_timeout) { ok = false; why = "segment of rank 0 not found (another node?)"; break; }
            usleep(1000);
        }
    }
    void *m = MAP_FAILED;
    if (ok) m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (fd >= 0) close(fd);
    if (ok && m == MAP_FAILED) { ok = false; why = "mmap"; }
    if (!ok) {
        if (c->rank == 0) shm_unlink(name);
        (void)xch_agree_rccl(c, false);
        return;
    }
    c->xch_hdr = static_cast<XchHeader *>(m);
    c->xch_bytes = bytes;
    XchHeader *h = c->xch_hdr;
    if (c->rank == 0) {                                        // ftruncate zero-filled the segment
        h->nranks = (unsigned int)c->nranks;
        __atomic_store_n(&h->magic, XCH_MAGIC, __ATOMIC_RELEASE);
    } else {
        const double t0 = now_s();
        while (__atomic_load_n(&h->magic, __ATOMIC_ACQUIRE) != XCH_MAGIC && now_s() - t0 < join_timeout) usleep(200);
        if (h->magic != XCH_MAGIC || h->nranks != (unsigned int)c->nranks) { ok = false; why = "segment header mismatch"; }
    }
    // map the segment into this rank's GPU (fine-grained: host memory is never cached by the GPU)
    void *dev = nullptr;
    if (ok) {
        if (hipHostRegister(m, bytes, hipHostRegisterMapped) == hipSuccess) c->xch_registered = true;
        else { (void)hipGetLastError(); ok = false; why = "hipHostRegister"; }
    }
    if (ok && hipHostGetDevicePointer(&dev, m, 0) != hipSuccess) { (void)hipGetLastError(); ok = false; why = "hipHostGetDevicePointer"; }
    // device-resident transport, part 1: my buffer and its IPC handle
    XchPeerSlot *slots = reinterpret_cast<XchPeerSlot *>(static_cast<char *>(m) + XCH_PEERS_OFF);
    const size_t local_bytes = flags_bytes + rows_bytes;
    bool direct = ok && want_direct != 0;
    if (ok) {
        std::memset(&slots[c->rank], 0, sizeof(XchPeerSlot));
        std::memcpy(slots[c->rank].pci, c->pci_id, sizeof c->pci_id);
        if (direct) {
            // fine-grained device memory: peers' writes must become visible to a running kernel
            if (hipExtMallocWithFlags(&c->xch_local, local_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
                (void)hipGetLastError();
                c->xch_local = nullptr;
                direct = false;
            }
            if (direct && hipMemset(c->xch_local, 0, local_bytes) != hipSuccess) direct = false;
            hipIpcMemHandle_t hd;
            if (direct && hipIpcGetMemHandle(&hd, c->xch_local) == hipSuccess) {
                static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(slots[0].handle), "IPC handle does not fit its slot");
                std::memcpy(slots[c->rank].handle, &hd, sizeof hd);
                slots[c->rank].has_handle = 1;
            } else if (direct) {
                (void)hipGetLastError();
                direct = false;
            }
        }
    }
    ok = xch_barrier(c, 0, ok, join_timeout) && ok;            // everybody has opened the segment ...
    if (c->rank == 0) shm_unlink(name);                        // ... so its name can go
    if (ok) {
        // ranks that share this rank's physical device split its residency slots
        int tenants = 0;
        for (int r = 0; r < c->nranks; ++r)
            if (std::memcmp(slots[r].pci, c->pci_id, sizeof c->pci_id) == 0) ++tenants;
        c->tenants = tenants > 0 ? tenants : 1;
        c->xch_stride = stride;
        XchArgs x{};
        x.nranks = c->nranks; x.rank = c->rank; x.stride = stride; x.timeout_ticks = XCH_TIMEOUT_TICKS;
        // device-resident transport, part 2: map every other rank's buffer, self-test, agree
        for (int r = 0; r < c->nranks && direct; ++r) direct = slots[r].has_handle != 0;
        std::vector<double *> prow(c->nranks, nullptr);
        std::vector<unsigned long long *> pflag(c->nranks, nullptr);
        for (int r = 0; r < c->nranks && direct; ++r) {
            void *p = c->xch_local;
            if (r != c->rank) {
                hipIpcMemHandle_t hd;
                std::memcpy(&hd, slots[r].handle, sizeof hd);
                if (hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); direct = false; break; }
                c->xch_peer_open.push_back(p);
            }
            pflag[r] = reinterpret_cast<unsigned long long *>(p);
            prow[r] = reinterpret_cast<double *>(static_cast<char *>(p) + flags_bytes);
        }
        double **d_prow = reinterpret_cast<double **>(reinterpret_cast<char *>(c->xch_scratch) + XCH_SCRATCH_PEERS);
        unsigned long long **d_pflag = reinterpret_cast<unsigned long long **>(d_prow + XCH_MAX_RANKS);
        if (direct)
            direct = hipMemcpy(d_prow, prow.data(), sizeof(double *) * c->nranks, hipMemcpyHostToDevice) == hipSuccess &&
                     hipMemcpy(d_pflag, pflag.data(), sizeof(void *) * c->nranks, hipMemcpyHostToDevice) == hipSuccess;
        bool d_ok = direct;
        if (xch_barrier(c, 1, direct, join_timeout)) {         // every rank has mapped every buffer
            x.direct = 1; x.rows = prow[c->rank]; x.flags = pflag[c->rank];
            x.peer_rows = d_prow; x.peer_flags = d_pflag; x.seq = 0;
            d_ok = xch_selftest(c, x);
        } else {
            d_ok = false;
        }
        d_ok = xch_barrier(c, 2, d_ok, join_timeout) && d_ok;
        d_ok = xch_agree_rccl(c, d_ok);
        if (d_ok) {
            c->xch_direct = true;
            c->xch_rows = x.rows; c->xch_flags = x.flags; c->xch_peer_rows = d_prow; c->xch_peer_flags = d_pflag;
            c->xch_seq = XCH_TEST_ROUNDS;                      // the self-test used sequence numbers 1..rounds
        } else {
            // fallback transport: rows and sequence numbers in the host segment
            x = XchArgs{};
            x.nranks = c->nranks; x.rank = c->rank; x.stride = stride; x.timeout_ticks = XCH_TIMEOUT_TICKS;
            x.rows = reinterpret_cast<double *>(static_cast<char *>(dev) + rows_off);
            x.flags = reinterpret_cast<unsigned long long *>(static_cast<char *>(dev) + XCH_FLAGS_OFF);
            x.seq = 0;
            ok = xch_selftest(c, x);
            if (!ok) why = "self-test (rows of the other ranks not seen)";
            ok = xch_barrier(c, 3, ok, join_timeout) && ok;
            ok = xch_agree_rccl(c, ok);
            if (ok) {
                c->xch_rows = x.rows; c->xch_flags = x.flags;
                c->xch_seq = XCH_TEST_ROUNDS;
            }
        }
    } else {
        (void)xch_agree_rccl(c, false);
    }
    if (!ok && why.empty()) why = "another rank failed";
    if (ok) {                                                  // what the persistent kernel's exchange workgroup reads
        XchArgs x{};                                           // (everything but the per-launch sequence number)
        x.nranks = c->nranks; x.rank = c->rank; x.stride = c->xch_stride; x.direct = c->xch_direct ? 1 : 0;
        x.rows = c->xch_rows; x.flags = c->xch_flags;
        x.peer_rows = c->xch_peer_rows; x.peer_flags = c->xch_peer_flags;
        x.timeout_ticks = XCH_TIMEOUT_TICKS;
        ok = hipMemcpy(c->xch_scratch + 16, &x, sizeof x, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) why = "hipMemcpy of the transport description";
    }
    c->xch_ok = ok;
    if (!ok) xch_teardown(c);
}

}   // namespace

int msgw_comm_init(msgw_ctx *c, const void *id128, int rank, int nranks)
{
    if (!c || !id128) return fail(c, MSGW_ERR_ARG, "NULL argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, MSGW_ERR_ARG, "bad rank %d / %d", rank, nranks);
    // MSGW_EXCHANGE_ONLY=1: no RCCL communicator at all (single node, every coupled step must then be
    // able to take the persistent kernel); also the way two ranks can share ONE GPU in tests, which
    // RCCL refuses.  MSGW_EXCHANGE=0: never use the in-kernel exchange (always the all-reduce chain).
    // MSGW_XCH_TRANSPORT=shm: skip the device-resident transport (host segment only).
    const char *eo = std::getenv("MSGW_EXCHANGE_ONLY");
    const bool exchange_only = eo && std::atoi(eo) != 0;
    const char *ex = std::getenv("MSGW_EXCHANGE");
    const bool want_exchange = exchange_only || !(ex && std::atoi(ex) == 0);
    const char *tr = std::getenv("MSGW_XCH_TRANSPORT");
    const int want_direct = !(tr && std::strcmp(tr, "shm") == 0);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_status(c)) return rc;
    if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    if (!exchange_only) {
        if (!g_rccl.load()) return fail(c, MSGW_ERR_RCCL, "%s", g_rccl.load_error.c_str());
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof id);
        ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
        if (r != 0)
            return fail(c, MSGW_ERR_RCCL, "ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    }
    c->rank = rank;
    c->nranks = nranks;
    c->cnt.nranks = nranks;
    c->plan_key = 0;
    // MSGW_FORCE_COLLECTIVE=1: run the all-reduce chain even for a 1-rank communicator, so that the
    // multi-GPU code path (reduce -> ncclAllReduce -> 1-row prologue) can be tested on a 1-GPU box
    if (const char *e = std::getenv("MSGW_FORCE_COLLECTIVE")) c->force_coll = std::atoi(e) != 0;
    std::string why;
    if (want_exchange && (nranks > 1 || c->force_coll)) xch_setup(c, id128, why, want_direct);
    else xch_teardown(c);
    c->cnt.exchange = c->xch_ok ? 1 : 0;
    c->cnt.transport = c->xch_ok ? (c->xch_direct ? MSGW_TRANSPORT_DEVICE_IPC : MSGW_TRANSPORT_HOST_SHM)
                                 : ((nranks > 1 || c->force_coll) ? MSGW_TRANSPORT_RCCL : MSGW_TRANSPORT_NONE);
    c->cnt.tenants = c->tenants;
    if (exchange_only && !c->xch_ok && nranks > 1)
        return fail(c, MSGW_ERR_RCCL, "MSGW_EXCHANGE_ONLY=1 but the node-level exchange could not be set up: %s", why.c_str());
    if (!c->xch_ok && !why.empty() && std::getenv("MSGW_VERBOSE"))
        fprintf(stderr, "msgwam_hip: rank %d: in-kernel exchange unavailable (%s); using the RCCL launch chain\n", rank, why.c_str());
    drop_graph(c);
    return MSGW_OK;
}

#ifdef MSGW_STAMP
// diagnostic build only: copy out the persistent kernel's [workgroups][PSTAMP_PASSES][4] stamps
int msgw_debug_stamps(msgw_ctx *c, unsigned long long *out, int nblocks)
{
    if (!c || !c->pstamps) return MSGW_ERR_ARG;
    HIPCHK(c, hipMemcpy(out, c->pstamps, sizeof(unsigned long long) * (size_t)nblocks * PSTAMP_PASSES * 4, hipMemcpyDeviceToHost));
    return MSGW_OK;
}
#endif

int msgw_counters(msgw_ctx *c, msgw_counters_t *out)
{
    if (!c || !out) return MSGW_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->cnt.ray_steps_total > 0) {
        HIPCHK(c, hipEventSynchronize(c->ev1));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        c->cnt.last_step_ms = ms;
    }
    if (int rc = check_status(c)) return rc;
    *out = c->cnt;
    return MSGW_OK;
}

}   // extern "C"
