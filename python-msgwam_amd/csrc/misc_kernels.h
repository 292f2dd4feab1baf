// Kernels that exist once (float64 only, no template on the ray type): lprop.saturation on caller arrays and
// the self-test of the node-level exchange.  Included by ONE translation unit (kern_misc.hip).
#pragma once
#include "column_kernels.h"
#include "persist_kernel.h"
#include "ray_kernels.h"

namespace msgw {

// ------------------------------------------------------------------ lprop.saturation on caller arrays (:561-615)
__global__ void __launch_bounds__(BLOCK) k_saturation(const SatArgs a)
{
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const double rr_f = a.rr[i] + a.rr_st[i] * a.dt;                 // :591
    const double drr_f = a.drr[i] + a.drr_st[i] * a.dt;              // :592
    const double mm_f = a.mm[i] + a.mm_st[i] * a.dt;                 // :593
    const double dmm_f = a.area[i] / drr_f;                          // :594
    const Bracket<double> br = interp_locate(rr_f, a.grids, a.nc, a.gs0, a.gs_last, a.gs0, a.inv_dzs);
    const double rho_f = interp_eval(rr_f, br, a.rhobar[br.j], a.slrho[min(br.j, a.nc - 2)]);   // :595
    const double kh2 = a.kk[i] * a.kk[i] + a.ll[i] * a.ll[i];
    const double m2 = a.mm[i] * a.mm[i];
    double n2_c = a.bvf2, n2_f = a.bvf2;
    if (a.bvfcol) {                                                  // EXTENSION: N(z) column
        const double N_c = interp_global(a.rr[i], a.grids, a.bvfcol, a.nc, a.gs0, a.gs_last, a.inv_dzs);
        const double N_f = interp_global(rr_f, a.grids, a.bvfcol, a.nc, a.gs0, a.gs_last, a.inv_dzs);
        n2_c = N_c * N_c; n2_f = N_f * N_f;
    }
    const double omh = sqrt((n2_c * kh2 + a.f0sq * m2) / (kh2 + m2));            // :597 (N at rr_center)
    const double pv = a.dkk[i] * a.dll[i] * dmm_f;                   // :599
    const double maxd = sat_cap(a.sat_c, rho_f, omh, n2_f, mm_f, a.f0sq);        // :601 (NN at rr_final)
    const double d = a.dens[i];
    const bool hit = maxd < d * pv;                                  // :604
    if (a.direct) a.out[i] = hit ? maxd : d;                         // :606-610
    else a.out[i] = hit ? (maxd - d) / a.dt : 0.0;                   // :612-615
}

__global__ void __launch_bounds__(COL_BLOCK) k_flux_reduce1(const Red1Args a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *s_seg = lds;                                              // [nseg][ncols]
    int *s_rng = reinterpret_cast<int *>(s_seg + (size_t)a.nseg * a.ncols);
    const int tid = threadIdx.x;
    const int r0 = (int)((long long)blockIdx.x * a.nblocks / gridDim.x);
    const int r1 = (int)((long long)(blockIdx.x + 1) * a.nblocks / gridDim.x);
    const int nr = r1 - r0;
    for (int i = tid; i < 2 * nr; i += COL_BLOCK) s_rng[i] = a.ranges[2 * r0 + i];
    __syncthreads();
    for (int idx = tid; idx < a.nseg * a.ncols; idx += COL_BLOCK) {
        const int seg = idx / a.ncols, col = idx - seg * a.ncols;
        const int c = col % a.ncp;
        const int b0 = (int)((long long)seg * nr / a.nseg), b1 = (int)((long long)(seg + 1) * nr / a.nseg);
        const double *src = a.partial + (size_t)r0 * a.ncols + col;
        double acc = 0.0;
        for (int b = b0; b < b1; b += 16) {                  // 16 loads in flight per thread
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[(size_t)min(b + u, b1 - 1) * a.ncols];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int bb = min(b + u, b1 - 1);
                const bool in = (b + u < b1) && (c >= s_rng[2 * bb]) && (c < s_rng[2 * bb + 1]);
                acc = acc + (in ? v[u] : 0.0);
            }
        }
        s_seg[idx] = acc;
    }
    __syncthreads();
    for (int col = tid; col < a.ncols; col += COL_BLOCK) {
        double tot = s_seg[col];
        for (int s = 1; s < a.nseg; ++s) tot = tot + s_seg[s * a.ncols + col];
        a.out[(size_t)blockIdx.x * a.ncols + col] = tot;
    }
}

// slopes of np.interp(., grids, rhobar) (lib/libprop.py:595), once per column upload
__global__ void k_rho_slopes(int nc, const double *grids, const double *rhobar, double *slrho)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nc - 1) slrho[j] = (rhobar[j + 1] - rhobar[j]) / (grids[j + 1] - grids[j]);
}


// dkk*dll (:137, :599) and rr_mm_area (:594) per ray, once per upload
template <typename T>
__global__ void k_nz_prepare(long long n, const double *dkk, const double *dll, const double *area, T *dkdl, T *area_out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dkdl[i] = (T)(dkk[i] * dll[i]);
    area_out[i] = (T)area[i];
}


// Arithmetic probe (msgw_probe_arith): the float64 square root and the exact constant division AS THE RAY KERNELS
// EVALUATE THEM (sqrt_, div_const), element by element on caller arrays, so that a test can hold them bit for bit to
// numpy over the whole exponent range (zeros, denormals, infinities, NaNs and negative arguments included).
__global__ void k_probe_arith(long long n, const double *x, double d, double c, int ok, double *out_sqrt, double *out_div,
                              const double *y, double *out_quot)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_sqrt[i] = sqrt_(x[i]);
    out_div[i] = div_const(x[i], d, c, ok);
    if (y) out_quot[i] = div_(x[i], y[i]);
}

// Self-test of the node-level exchange, run once by every rank when the communicator is set up:
// `rounds` node-level sums of known rows through the very code path of the persistent kernel.  A rank
// that cannot see the others' rows (or sees them out of order) reports 0 and the host side falls
// back to the all-reduce launch chain on every rank.

__device__ __forceinline__ double xch_test_value(int rank, int round, int col)
{
    return (double)(rank + 1) * 1048576.0 + (double)round * 1024.0 + (double)col + 0.5;
}

__global__ void __launch_bounds__(BLOCK) k_xch_selftest(const XchTestArgs t)
{
    __shared__ int s_flag[4];
    const int tid = threadIdx.x;
    const XchArgs x = t.x;
    const int ncols = x.stride;                                // the whole row (taller than one workgroup: strided)
    int good = 1;
    for (int round = 1; round <= t.rounds; ++round) {
        if (!xch_allsum(x, x.seq + (u64_t)round, nullptr, ncols, tid, s_flag + (round & 1), [](int k) { return k; },
                        [&](int col) { return xch_test_value(x.rank, round, col); },
                        [&](int col, double tot) {
                            double want = 0.0;
                            for (int r = 0; r < x.nranks; ++r) want = want + xch_test_value(r, round, col);
                            if (tot != want) good = 0;
                        })) {
            good = 0;
            break;
        }
    }
    const int all_good = __syncthreads_and(good);
    if (tid == 0) *t.result = all_good ? 1 : 0;
}

}   // namespace msgw
