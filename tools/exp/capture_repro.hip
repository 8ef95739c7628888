// Which multi-stream capture patterns does hipStreamEndCapture survive on this ROCm?  (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); fflush(stdout); exit(1); } } while (0)
__global__ void k(double *p) { p[threadIdx.x] += 1.0; }
int main(int argc, char **argv)
{
    const int pattern = argc > 1 ? atoi(argv[1]) : 0;
    double *a, *b, *ra, *rb;
    CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096)); CK(hipMalloc(&ra, 4096)); CK(hipMalloc(&rb, 4096));
    hipStream_t s0, s1, c0, c1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&c0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&c1, hipStreamNonBlocking));
    hipEvent_t fork, join1, p0, p1, a0, a1;
    for (hipEvent_t *e : {&fork, &join1, &p0, &p1, &a0, &a1}) CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    hipGraph_t g; hipGraphExec_t ex;
    printf("pattern %d: begin\n", pattern); fflush(stdout);
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
    CK(hipEventRecord(fork, s0));
    if (pattern >= 1) CK(hipStreamWaitEvent(s1, fork, 0));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s0, a);
    if (pattern >= 1) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, b);
    if (pattern >= 2) {              // comm streams: each waits the OTHER main stream's event, copies, records; each main waits its comm
        CK(hipEventRecord(p0, s0)); CK(hipEventRecord(p1, s1));
        CK(hipStreamWaitEvent(c1, p0, 0)); CK(hipMemcpyAsync(rb, a, 512, hipMemcpyDeviceToDevice, c1)); CK(hipEventRecord(a1, c1));
        CK(hipStreamWaitEvent(c0, p1, 0)); CK(hipMemcpyAsync(ra, b, 512, hipMemcpyDeviceToDevice, c0)); CK(hipEventRecord(a0, c0));
        if (pattern >= 3) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s0, a); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, b); }
        CK(hipStreamWaitEvent(s0, a0, 0)); CK(hipStreamWaitEvent(s1, a1, 0));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s0, ra); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, rb);
    }
    if (pattern >= 4) {              // a comm stream that waits on BOTH mains before its first node, 8-byte copies, then three more rounds
        hipEvent_t r0, r1, g0, g1;
        for (hipEvent_t *e : {&r0, &r1, &g0, &g1}) CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        CK(hipEventRecord(r0, s0)); CK(hipEventRecord(r1, s1));
        CK(hipStreamWaitEvent(c0, r0, 0)); CK(hipMemcpyAsync(ra, a, 8, hipMemcpyDeviceToDevice, c0));
        CK(hipStreamWaitEvent(c0, r1, 0)); CK(hipMemcpyAsync(ra + 1, b, 8, hipMemcpyDeviceToDevice, c0)); CK(hipEventRecord(g0, c0));
        CK(hipStreamWaitEvent(s0, g0, 0));
        CK(hipStreamWaitEvent(c1, r0, 0)); CK(hipMemcpyAsync(rb, a, 8, hipMemcpyDeviceToDevice, c1));
        CK(hipStreamWaitEvent(c1, r1, 0)); CK(hipMemcpyAsync(rb + 1, b, 8, hipMemcpyDeviceToDevice, c1)); CK(hipEventRecord(g1, c1));
        CK(hipStreamWaitEvent(s1, g1, 0));
        if (pattern >= 5)
            for (int round = 0; round < 3; round++) {
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s0, a); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, b);
                CK(hipEventRecord(p0, s0)); CK(hipEventRecord(p1, s1));
                CK(hipStreamWaitEvent(c1, p0, 0)); CK(hipMemcpyAsync(rb, a, 512, hipMemcpyDeviceToDevice, c1)); CK(hipEventRecord(a1, c1));
                CK(hipStreamWaitEvent(c0, p1, 0)); CK(hipMemcpyAsync(ra, b, 512, hipMemcpyDeviceToDevice, c0)); CK(hipEventRecord(a0, c0));
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s0, a); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, b);
                CK(hipStreamWaitEvent(s0, a0, 0)); CK(hipStreamWaitEvent(s1, a1, 0));
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s0, ra); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, rb);
            }
    }
    if (pattern >= 1) { CK(hipEventRecord(join1, s1)); CK(hipStreamWaitEvent(s0, join1, 0)); }
    printf("pattern %d: end capture\n", pattern); fflush(stdout);
    CK(hipStreamEndCapture(s0, &g));
    printf("pattern %d: instantiate\n", pattern); fflush(stdout);
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ex, s0));
    CK(hipStreamSynchronize(s0));
    printf("pattern %d: ok\n", pattern);
    return 0;
}
