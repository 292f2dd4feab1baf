// Ablation microbenchmark (developer tool, not product): times, on 1e6 synthetic rays,
//  stream : 7 x 16-B loads + 4 x 16-B stores per lane, no math (the access pattern of stage 0)
//  nodep  : k_ray_stage<1,...,DEPOSIT=false>
//  full   : k_ray_stage<1,...,DEPOSIT=true>
// build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Ipython-msgwam_amd/csrc -Iinclude tools/kbench.hip -o gpurun_out/kbench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
#include "ray_kernels.h"
#include "column_kernels.h"
using namespace msgw;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NR, int NW>
__global__ void __launch_bounds__(BLOCK) k_stream(long long n, int tpb, double *const *in, double *const *out)
{
    const long long tile0 = (long long)blockIdx.x * tpb;
    for (int t = 0; t < tpb; ++t) {
        const long long base = (tile0 + t) * (long long)TILE;
        if (base >= n) break;
        const long long i0 = base + 2 * threadIdx.x;
        if (i0 + 1 >= n) continue;
        double2 acc = make_double2(0, 0);
#pragma unroll
        for (int k = 0; k < NR; ++k) { const double2 v = *reinterpret_cast<const double2 *>(in[k] + i0); acc.x += v.x; acc.y += v.y; }
#pragma unroll
        for (int k = 0; k < NW; ++k) *reinterpret_cast<double2 *>(out[k] + i0) = make_double2(acc.x + k, acc.y - k);
    }
}

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int ng = 101, reps = 200;
    const int bpc = argc > 2 ? atoi(argv[2]) : 4;
    hipStream_t st; CK(hipStreamCreate(&st));
    std::vector<double *> buf(16);
    for (auto &p : buf) { CK(hipMalloc(&p, (n + 2 * TILE) * sizeof(double))); }
    // synthetic z-major spectrum-like state
    std::vector<double> h(n);
    auto up = [&](double *d, auto f) { for (long long i = 0; i < n; ++i) h[i] = f(i); return hipMemcpy(d, h.data(), n * sizeof(double), hipMemcpyHostToDevice); };
    double *dens = buf[0], *rr = buf[1], *mm = buf[2], *drr = buf[3], *kk = buf[4], *ll = buf[5], *dmm = buf[6], *vol = buf[7],
           *fray = buf[8], *pvf = buf[9], *q_rr = buf[10], *q_mm = buf[11], *q_dens = buf[12], *rr0 = buf[13], *mm0 = buf[14];
    CK(up(dens, [&](long long i) { return 1e9 + i % 7; }));
    CK(up(rr, [&](long long i) { return 75.0 + 15000.0 * (double)i / n; }));
    CK(up(mm, [&](long long i) { return -1.2566e-3 * (0.7 + 0.6 * ((i / 4) % 2500) / 2500.0); }));
    CK(up(drr, [&](long long) { return 150.0; }));
    CK(up(kk, [&](long long i) { return 1.2566e-4 * sin(0.785 + 1.5708 * (i % 4)); }));
    CK(up(ll, [&](long long i) { return 1.2566e-4 * cos(0.785 + 1.5708 * (i % 4)); }));
    CK(up(dmm, [&](long long) { return 3e-7; }));
    CK(up(vol, [&](long long) { return 3e-15; }));
    CK(up(fray, [&](long long) { return 0.0; }));
    CK(up(pvf, [&](long long) { return 3e-15; }));
    for (int k = 10; k < 15; ++k) CK(hipMemset(buf[k], 0, n * sizeof(double)));
    // column
    std::vector<double> grid(ng), grids(ng - 1), z(ng, 0.0), lin(ng);
    for (int i = 0; i < ng; ++i) grid[i] = 1000.0 * i;
    for (int i = 0; i < ng - 1; ++i) grids[i] = 500.0 + 1000.0 * i;
    for (int i = 0; i < ng; ++i) lin[i] = 1e-4 * sin(i * 0.3);
    double *col; CK(hipMalloc(&col, 16 * ng * sizeof(double))); CK(hipMemset(col, 0, 16 * ng * sizeof(double)));
    CK(hipMemcpy(col, grid.data(), ng * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(col + ng, grids.data(), (ng - 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(col + 2 * ng, lin.data(), ng * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(col + 3 * ng, lin.data(), ng * 8, hipMemcpyHostToDevice));
    const long long ntiles = (n + TILE - 1) / TILE;
    long long maxb = 256LL * bpc, tpb = (ntiles + maxb - 1) / maxb; if (tpb < 1) tpb = 1;
    const int blocks = (int)((ntiles + tpb - 1) / tpb);
    double *partial; int *ranges;
    CK(hipMalloc(&partial, (size_t)blocks * 2 * (ng - 2) * 8)); CK(hipMalloc(&ranges, blocks * 8));
    StageArgs a{};
    a.n = n; a.ng = ng; a.tiles_per_block = (int)tpb; a.rays_per_block = tpb * TILE; a.dt = 120.0; a.bvf2 = 1e-4; a.f_uni = 0; a.f0sq = 0; a.same_f = 1;
    a.sat_c = .5; a.sat_rr_div = 120.0; a.xg0 = 1000.0; a.inv_dzg = 1e-3; a.gs0 = 500.0; a.xg_last = 99000.0; a.gs_last = 99500.0; a.inv_dzs = 1.0 / 1000.0; a.dzs = 1000.0; a.mk_ok = 1;
    a.r = RayPtrs{dens, rr, mm, drr, kk, ll, dmm, vol, fray, pvf, q_rr, q_mm, q_dens, rr0, mm0};
    a.c = ColPtrs{col + 1, col + 2 * ng, col + 3 * ng, col + 4 * ng, col + 5 * ng, col + ng, col + 6 * ng, col + 7 * ng};
    a.partial = partial; a.ranges = ranges;
    const size_t lds = sizeof(double) * (size_t)(5 * (ng - 2) + 3 * (ng - 1) + WAVES * 2 * (ng - 2)) + 64;
    double **din, **dout; CK(hipMalloc(&din, 16 * sizeof(double *))); CK(hipMalloc(&dout, 16 * sizeof(double *)));
    double *hin[9] = {rr, mm, kk, ll, dens, drr, vol, q_rr, q_mm}, *hout[4] = {buf[13], buf[14], buf[12], buf[9]};
    CK(hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice)); CK(hipMemcpy(dout, hout, sizeof hout, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto &&launch, double bytes) {
        for (int i = 0; i < 10; ++i) launch();
        hipStreamSynchronize(st);
        std::vector<float> ts;
        for (int i = 0; i < reps; ++i) {
            hipEventRecord(e0, st); launch(); hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        // back-to-back batch for the amortised time
        hipEventRecord(e0, st); for (int i = 0; i < reps; ++i) launch(); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s blocks %5d  single med %7.2f us  back-to-back %7.2f us/launch  -> %6.0f GB/s\n", name, blocks, ts[reps / 2] * 1e3, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
    };
    printf("n = %lld rays, tiles/block %lld\n", n, tpb);
    timeit("stream 7r+4w", [&] { hipLaunchKernelGGL((k_stream<7, 4>), dim3(blocks), dim3(BLOCK), 0, st, n, (int)tpb, din, dout); }, n * 88.0);
    timeit("stream 9r+4w", [&] { hipLaunchKernelGGL((k_stream<9, 4>), dim3(blocks), dim3(BLOCK), 0, st, n, (int)tpb, din, dout); }, n * 104.0);
    timeit("stream 9r+2w", [&] { hipLaunchKernelGGL((k_stream<9, 2>), dim3(blocks), dim3(BLOCK), 0, st, n, (int)tpb, din, dout); }, n * 88.0);
    timeit("stage1 no deposit", [&] { hipLaunchKernelGGL((k_ray_stage<1, false, false, false, false>), dim3(blocks), dim3(BLOCK), lds, st, a); }, n * 88.0);
    timeit("stage1 full", [&] { hipLaunchKernelGGL((k_ray_stage<1, false, false, true, false>), dim3(blocks), dim3(BLOCK), lds, st, a); }, n * 104.0);
#ifdef MSGW_STAMP
    {   // phase timeline of one launch of the full stage-1 kernel (wall_clock64 = 100 MHz)
        unsigned long long *dst; CK(hipMalloc(&dst, (size_t)blocks * 8 * sizeof(unsigned long long)));
        CK(hipMemset(dst, 0, (size_t)blocks * 8 * sizeof(unsigned long long)));
        StageArgs b = a; b.stamps = dst;
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL((k_ray_stage<1, false, false, true, false>), dim3(blocks), dim3(BLOCK), lds, st, b); }
        hipStreamSynchronize(st);
        std::vector<unsigned long long> hs((size_t)blocks * 8);
        CK(hipMemcpy(hs.data(), dst, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull; for (int bI = 0; bI < blocks; ++bI) t0 = std::min(t0, hs[bI * 8]);
#ifdef MSGW_DBG_LEVELS
        unsigned long long hl = 0, ht = 0, hw = 0;
        CK(hipMemcpyFromSymbol(&hl, HIP_SYMBOL(g_dbg_levels), 8)); CK(hipMemcpyFromSymbol(&ht, HIP_SYMBOL(g_dbg_tiles), 8));
        CK(hipMemcpyFromSymbol(&hw, HIP_SYMBOL(g_dbg_wide), 8));
        printf("  deposit: %llu wave-tiles, %llu level iterations (%.2f per wave-tile), %llu wide (atomic path)\n", ht, hl, (double)hl / ht, hw);
#endif
        const char *nm[7] = {"entry", "col staged", "t0 physics", "t0 deposit", "t1 physics", "t1 deposit", "end"};
        for (int k = 0; k < 7; ++k) {
            std::vector<double> v; for (int bI = 0; bI < blocks; ++bI) if (hs[bI * 8 + k]) v.push_back((hs[bI * 8 + k] - t0) * 0.01);
            if (v.empty()) continue; std::sort(v.begin(), v.end());
            printf("  stamp %-11s  min %6.2f  p10 %6.2f  med %6.2f  p90 %6.2f  max %6.2f us\n", nm[k], v[0], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
        }
    }
#endif
    timeit("stage0 full", [&] { hipLaunchKernelGGL((k_ray_stage<0, false, false, true, false>), dim3(blocks), dim3(BLOCK), lds, st, a); }, n * 88.0);
    timeit("stage2 full", [&] { hipLaunchKernelGGL((k_ray_stage<2, false, false, true, false>), dim3(blocks), dim3(BLOCK), lds, st, a); }, n * 88.0);
    return 0;
}
