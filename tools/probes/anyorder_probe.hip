// Probe: does hipExtAnyOrderLaunch let a kernel start before its same-stream predecessor has finished on gfx950?
// K1 spins ~100 us and stamps its end; K2 stamps its start.  Overlap <=> K2.start < K1.end.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k1(unsigned long long* t, int us) {
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = wall_clock64();
}
__global__ void k2(unsigned long long* t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64();
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16); unsigned long long h[2];
    hipStream_t s; hipStreamCreate(&s);
    for (int flags = 0; flags < 2; ++flags) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(d, 0, 16, s); hipStreamSynchronize(s);
            int us = 100;
            hipExtLaunchKernelGGL(k1, dim3(64), dim3(256), 0, s, nullptr, nullptr, 0, d, us);
            hipExtLaunchKernelGGL(k2, dim3(64), dim3(256), 0, s, nullptr, nullptr, flags, d);
            hipStreamSynchronize(s);
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("flags=%d rep=%d: k2.start - k1.end = %.2f us (100 MHz clock)\n", flags, rep, ((double)h[1] - (double)h[0]) / 100.0);
        }
    }
    return 0;
}
