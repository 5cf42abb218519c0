// concurrency.hip -- do kernels of DIFFERENT HIP streams run beside each other on this box?   (round 6)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/concurrency.hip -o tools/ubench/bin/concurrency && tools/ubench/bin/concurrency
// A kernel of 256 workgroups x 256 threads (one per CU) that spins for ~200 us is launched 24 times: on one stream, and
// dealt round-robin over 2 / 3 / 4 streams.  The GPU has room for ~8 such kernels at once (8+ workgroups of this size per
// CU), so with concurrent queues the multi-stream times are 1/2, 1/3, 1/4 of the one-stream time; a box whose queues are
// served one after the other prints the same time in every row.  PMVO's iterations rotate over three streams so that the
// front end of iteration i+1 runs beside the search of iteration i.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(long long cycles, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
    }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

int main() {
    int *sink;
    hipMalloc(&sink, 4);
    std::vector<hipStream_t> st(4);
    for (auto &s : st) hipStreamCreate(&s);
    const long long cyc = 20000;   // wall_clock64 ticks at 100 MHz: 200 us
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st[0], cyc, sink);
    hipDeviceSynchronize();
    for (int ns : {1, 2, 3, 4}) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 24; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st[i % ns], cyc, sink);
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%d stream(s): 24 x 200 us kernels in %.2f ms\n", ns, ms);
    }
    return 0;
}
