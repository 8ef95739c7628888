// Micro-benchmark: how fast can MI355X move the bytes of one flux launch when nothing else is in the way?
// A tile-shaped streaming kernel: workgroup t reads `rd` KiB of contiguous doubles (8 B per lane per load, the
// shape of the incidence-row streams; or 16 B per lane) and writes `wr` KiB, data resident in the Infinity
// Cache between back-to-back launches (the bench's situation) or not (--cold: a 512 MiB buffer walked through).
// Prints us per launch and TB/s for several grid sizes, occupancies (static LDS per workgroup) and loads in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_ceiling.hip -o /tmp/stream_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>

template <int LDS_BYTES, int INFLIGHT, bool WIDE>
__global__ void __launch_bounds__(256) k_stream(const double *__restrict__ in, double *__restrict__ out, int rd_per_thread,
                                                int wr_per_thread, long in_stride_wg, long out_stride_wg)
{
    __shared__ double lds[LDS_BYTES / 8 > 0 ? LDS_BYTES / 8 : 1];
    const double *src = in + long(blockIdx.x) * in_stride_wg;
    double acc = 0.0;
    if (WIDE) {
        const double2 *s2 = reinterpret_cast<const double2 *>(src) + threadIdx.x;
        for (int k = 0; k < rd_per_thread / 2; k += INFLIGHT) {
            double2 v[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) v[u] = s2[(k + u < rd_per_thread / 2 ? k + u : 0) * 256];
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) acc += v[u].x + v[u].y;
        }
    } else {
        const double *s = src + threadIdx.x;
        for (int k = 0; k < rd_per_thread; k += INFLIGHT) {
            double v[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) v[u] = s[(k + u < rd_per_thread ? k + u : 0) * 256];
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) acc += v[u];
        }
    }
    if (LDS_BYTES > 0 && acc == 1.2345e300) { lds[threadIdx.x] = acc; __syncthreads(); acc = lds[(threadIdx.x + 1) & 255]; }
    if (acc == 2.3456e300) out[0] = acc;                                  // keeps the loads alive when nothing is written
    double *dst = out + long(blockIdx.x) * out_stride_wg + threadIdx.x;
    for (int k = 0; k < wr_per_thread; k++) dst[k * 256] = acc + k;
}

// The same stream with the reads as LDS-DMA (round 4: `global_load_lds_dwordx4`, 16 bytes per lane = 1 KiB per wave-instruction
// straight into LDS, no destination registers): eight transfers in flight per wave into an 8 KiB ring of its own, the data never
// read back — what the memory side of a tile costs when nothing passes through the register file.
template <int LDS_BYTES>
__global__ void __launch_bounds__(256) k_stream_dma(const double *__restrict__ in, double *__restrict__ out, int rd_per_thread,
                                                    int wr_per_thread, long in_stride_wg, long out_stride_wg)
{
    __shared__ double lds[LDS_BYTES / 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char *src = reinterpret_cast<const char *>(in + long(blockIdx.x) * in_stride_wg) + wave * 1024 + lane * 16;
    auto *ring = (__attribute__((address_space(3))) char *)(lds) + wave * 8192;
    const int n_xfer = rd_per_thread / 2;                   // 16 bytes per lane and transfer; a workgroup's four waves interleave 1 KiB pieces
    for (int k = 0; k < n_xfer; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = k + u < n_xfer ? k + u : 0;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + long(j) * 4096),
                                             (__attribute__((address_space(3))) void *)(ring + u * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    double acc = lds[(threadIdx.x * 7) & 1023];            // (one LDS word, so the ring is not dead)
    if (acc == 2.3456e300) out[0] = acc;
    double *dst = out + long(blockIdx.x) * out_stride_wg + threadIdx.x;
    for (int k = 0; k < wr_per_thread; k++) dst[k * 256] = acc + k;
}

template <int LDS_BYTES>
double run_dma(int grid, int rd_per_thread, int wr_per_thread, int reps)
{
    const long in_wg = long(rd_per_thread) * 256, out_wg = long(wr_per_thread) * 256;
    const long in_n = in_wg * grid + 4096, out_n = out_wg * grid + 256;
    double *in, *out;
    hipMalloc(&in, in_n * 8); hipMalloc(&out, out_n * 8);
    hipMemset(in, 0, in_n * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++)
        hipLaunchKernelGGL((k_stream_dma<LDS_BYTES>), dim3(grid), dim3(256), 0, 0, in, out, rd_per_thread, wr_per_thread, in_wg, out_wg);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL((k_stream_dma<LDS_BYTES>), dim3(grid), dim3(256), 0, 0, in, out, rd_per_thread, wr_per_thread, in_wg, out_wg);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(in); hipFree(out);
    return ms * 1e3 / reps;
}

template <int LDS_BYTES, int INFLIGHT, bool WIDE>
double run(int grid, int rd_per_thread, int wr_per_thread, int reps, bool cold)
{
    const long in_wg = long(rd_per_thread) * 256, out_wg = long(wr_per_thread) * 256;
    const long in_n = in_wg * grid, out_n = out_wg * grid + 256;
    const long pool = cold ? (long(768) << 20) / 8 : in_n;           // doubles
    double *in, *out;
    hipMalloc(&in, pool * 8); hipMalloc(&out, out_n * 8);
    hipMemset(in, 0, pool * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const long slots = cold ? pool / in_n : 1;
    for (int i = 0; i < 5; i++)
        hipLaunchKernelGGL((k_stream<LDS_BYTES, INFLIGHT, WIDE>), dim3(grid), dim3(256), 0, 0, in, out, rd_per_thread, wr_per_thread, in_wg, out_wg);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL((k_stream<LDS_BYTES, INFLIGHT, WIDE>), dim3(grid), dim3(256), 0, 0, in + (i % slots) * in_n, out, rd_per_thread, wr_per_thread, in_wg, out_wg);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(in); hipFree(out);
    return ms * 1e3 / reps;
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "--dma") {
        // the order-free kernel's tile: ~47 KB read (24 x 2 KiB) + 10 KB written, register loads (8 B and 16 B per lane) against LDS-DMA
        printf("cache-resident data; 24 x 2 KiB read + 5 x 2 KiB written per workgroup; us per launch (TB/s of rd+wr bytes)\n");
        printf("grid    registers 8B x8    registers 16B x4   LDS-DMA 16B x8 (40 KB LDS)   LDS-DMA 16B x8 (33 KB LDS)\n");
        for (int g : {768, 1024, 1175, 2350}) {
            const double bytes = double(24 + 5) * 2048.0 * g;
            const double t0 = run<40960, 8, false>(g, 24, 5, 300, false), t1 = run<40960, 4, true>(g, 24, 5, 300, false),
                         t2 = run_dma<40960>(g, 24, 5, 300), t3 = run_dma<33792>(g, 24, 5, 300);
            printf("%5d  %6.2f (%5.2f)     %6.2f (%5.2f)     %6.2f (%5.2f)              %6.2f (%5.2f)\n", g, t0, bytes / t0 / 1e6, t1, bytes / t1 / 1e6, t2, bytes / t2 / 1e6, t3, bytes / t3 / 1e6);
        }
        return 0;
    }
    const bool cold = argc > 1 && std::string(argv[1]) == "--cold";
    // per-workgroup bytes of the flux kernel today: ~63 KB read (51 rows + 10 own state + ids), 10 KB written; the
    // edge-once layout: ~38 KB read, 10 KB written.  1 double per thread = 2 KiB per workgroup.
    struct Case { const char *name; int rd, wr; } cases[] = {
        {"today 64K rd + 10K wr", 32, 5}, {"edge-once 38K rd + 10K wr", 19, 5}, {"read only 64K", 32, 0}, {"time_step-like 22K rd + 20K wr", 11, 10}};
    const int grids[] = {768, 1175, 2350, 9399};
    printf("%s data; us per launch (TB/s of rd+wr bytes)\n", cold ? "cold (768 MiB pool)" : "cache-resident");
    for (auto &c : cases) {
        printf("== %s per workgroup ==\n", c.name);
        printf("grid    lds53K,8B,x4      lds53K,8B,x8      lds40K,8B,x8      lds24K,8B,x8      lds0,8B,x8        lds40K,16B,x4\n");
        for (int g : grids) {
            const double bytes = double(c.rd + c.wr) * 2048.0 * g;
            const int reps = g > 5000 ? 50 : 200;
            const double t0 = run<53760, 4, false>(g, c.rd, c.wr, reps, cold), t1 = run<53760, 8, false>(g, c.rd, c.wr, reps, cold),
                         t2 = run<40960, 8, false>(g, c.rd, c.wr, reps, cold), t3 = run<24576, 8, false>(g, c.rd, c.wr, reps, cold),
                         t4 = run<0, 8, false>(g, c.rd, c.wr, reps, cold), t5 = run<40960, 4, true>(g, c.rd & ~1, c.wr, reps, cold);
            printf("%5d  %6.2f (%5.2f)   %6.2f (%5.2f)   %6.2f (%5.2f)   %6.2f (%5.2f)   %6.2f (%5.2f)   %6.2f (%5.2f)\n", g, t0, bytes / t0 / 1e6,
                   t1, bytes / t1 / 1e6, t2, bytes / t2 / 1e6, t3, bytes / t3 / 1e6, t4, bytes / t4 / 1e6, t5, bytes / t5 / 1e6);
        }
    }
    return 0;
}
