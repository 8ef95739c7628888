// Micro-benchmark: what does an (almost) empty launch cost on MI355X as a function of grid size,
// static LDS per workgroup and VGPR budget?  (DESIGN.md §5 "launch ramp".)  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_BYTES> __global__ void __launch_bounds__(256) k_empty(double *out, int never)
{
    __shared__ double lds[LDS_BYTES / 8 > 0 ? LDS_BYTES / 8 : 1];
    if (never) { lds[threadIdx.x] = out[threadIdx.x]; __syncthreads(); out[blockIdx.x] = lds[(threadIdx.x + 1) & 255]; }
}
template <int LDS_BYTES> double run(int grid, int reps)
{
    double *d; hipMalloc(&d, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_empty<LDS_BYTES>, dim3(grid), dim3(256), 0, 0, d, 0);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_empty<LDS_BYTES>, dim3(grid), dim3(256), 0, 0, d, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); hipFree(d);
    return ms * 1e3 / reps;
}
int main()
{
    const int grids[] = {256, 768, 1175, 2350, 4700};
    printf("grid   lds=0     lds=24K   lds=40K   lds=53.7K  (us per launch, 500 back-to-back)\n");
    for (int g : grids)
        printf("%5d  %7.2f   %7.2f   %7.2f   %7.2f\n", g, run<0>(g, 500), run<24576>(g, 500), run<40960>(g, 500), run<53760>(g, 500));
    return 0;
}
