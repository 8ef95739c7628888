// Does hipExtAnyOrderLaunch let the next kernel of a stream start before the previous one has ended (gfx950, ROCm 7.2)?
// Two one-workgroup kernels that each spin for ~100 us, in ONE stream: in order they take ~200 us, overlapped ~100 us.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/any_order.hip -o /tmp/any_order && /tmp/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void k_spin(unsigned long long ticks, unsigned long long *out)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (out) *out = wall_clock64();
}
int main()
{
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned long long *d; (void)hipMalloc(&d, 16);
    const unsigned long long ticks = 10000;                    // 100 us at 100 MHz
    for (int flags = 0; flags < 2; flags++) {
        for (int rep = 0; rep < 3; rep++) {
            (void)hipStreamSynchronize(st);
            const auto t0 = std::chrono::steady_clock::now();
            hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, 0, ticks, d);
            hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, ticks, d + 1);
            (void)hipStreamSynchronize(st);
            std::printf("second launch %s: %.1f us for two 100-us kernels\n", flags ? "with hipExtAnyOrderLaunch" : "in order",
                        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
    }
    return 0;
}
