// What a process that only wakes the device costs on this box, start to gone: the floor under the drop-in binary's wall time.
//   hipcc --offload-arch=gfx950 -O2 tools/hip_process_floor.hip -o /tmp/hip_process_floor
//   python3 -c 'import subprocess,time; ...'  (tools/driver_at_scale.py times it with DRIVER_FLOOR=/tmp/hip_process_floor)
// argv[1] = megabytes to allocate, fill from the host and free again (0: none).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_touch(double *p, size_t n) { size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; if (i < n) p[i] += 1.0; }
int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 0;
    if (hipSetDevice(0) != hipSuccess || hipFree(nullptr) != hipSuccess) { std::fprintf(stderr, "no device\n"); return 1; }
    if (mb) {
        const size_t n = mb * (1u << 20) / sizeof(double);
        std::vector<double> h(n, 1.0);
        double *d = nullptr;
        if (hipMalloc(&d, n * sizeof(double)) != hipSuccess) return 1;
        (void)hipMemcpy(d, h.data(), n * sizeof(double), hipMemcpyHostToDevice);
        k_touch<<<dim3((n + 255) / 256), dim3(256)>>>(d, n);
        (void)hipDeviceSynchronize();
        (void)hipFree(d);
    }
    return 0;
}
