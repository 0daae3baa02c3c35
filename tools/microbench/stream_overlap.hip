// Microbenchmark: what does it cost to run a small latency-bound kernel on a second stream beside a chip-filling one,
// with the event waits that make it safe?  Pattern of one forward substep (times = what the kernels take alone):
//   one stream :  dense1(40us) -> sparse(15us) -> dense2(40us)
//   two streams:  A: dense1 -> [E1] -> dense2_clean(37us) -> wait E2 -> dense2_dirty(3us)
//                 B: wait E1 -> sparse -> [E2]
// Kernels burn a fixed number of clock cycles per workgroup (no memory traffic), `dense` with 8192 workgroups, `sparse` with 200.
// Build: hipcc --offload-arch=gfx950 -O3 stream_overlap.hip -o stream_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void burn(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    int acc = 0;
    while (wall_clock64() - t0 < cycles) acc += 1;
    if (acc == -1) *sink = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    int* sink;
    CK(hipMalloc(&sink, 4));
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hipEvent_t e1, e2, t0, t1;
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    // wall_clock64 ticks at 100 MHz on gfx9: 10 ns per tick
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const double tick_us = 1e3 / (double)rate_khz;
    printf("wall clock %d kHz (%.3f us per tick)\n", rate_khz, tick_us);
    // a dense kernel of 8192 workgroups on 256 CUs x 8 resident = 4 rounds: per-workgroup burn = total / 4
    auto cyc = [&](double us) { return (long long)(us / tick_us); };
    const int ND = 8192, NS = 200, REP = 200;
    const long long d_wg = cyc(40.0 / 4.0), s_wg = cyc(15.0), dirty_wg = cyc(3.0);
    for (int variant = 0; variant < 4; ++variant) {
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, A));
            for (int r = 0; r < REP; ++r) {
                if (variant == 0) {                   // one stream, three kernels
                    hipLaunchKernelGGL(burn, dim3(ND), dim3(256), 0, A, d_wg, sink);
                    hipLaunchKernelGGL(burn, dim3(NS), dim3(256), 0, A, s_wg, sink);
                    hipLaunchKernelGGL(burn, dim3(ND), dim3(256), 0, A, d_wg, sink);
                } else if (variant == 1) {            // one stream, the sparse kernel removed (the floor)
                    hipLaunchKernelGGL(burn, dim3(ND), dim3(256), 0, A, d_wg, sink);
                    hipLaunchKernelGGL(burn, dim3(ND), dim3(256), 0, A, d_wg, sink);
                } else if (variant == 2) {            // two streams, events, dirty pass of 600 workgroups
                    hipLaunchKernelGGL(burn, dim3(ND), dim3(256), 0, A, d_wg, sink);
                    CK(hipEventRecord(e1, A));
                    CK(hipStreamWaitEvent(B, e1, 0));
                    hipLaunchKernelGGL(burn, dim3(NS), dim3(256), 0, B, s_wg, sink);
                    CK(hipEventRecord(e2, B));
                    hipLaunchKernelGGL(burn, dim3(ND - 600), dim3(256), 0, A, d_wg, sink);
                    CK(hipStreamWaitEvent(A, e2, 0));
                    hipLaunchKernelGGL(burn, dim3(600), dim3(256), 0, A, dirty_wg, sink);
                } else {                              // two streams, events, but the dirty pass is a full-size launch of early exits
                    hipLaunchKernelGGL(burn, dim3(ND), dim3(256), 0, A, d_wg, sink);
                    CK(hipEventRecord(e1, A));
                    CK(hipStreamWaitEvent(B, e1, 0));
                    hipLaunchKernelGGL(burn, dim3(NS), dim3(256), 0, B, s_wg, sink);
                    CK(hipEventRecord(e2, B));
                    hipLaunchKernelGGL(burn, dim3(ND - 600), dim3(256), 0, A, d_wg, sink);
                    CK(hipStreamWaitEvent(A, e2, 0));
                    hipLaunchKernelGGL(burn, dim3(ND), dim3(256), 0, A, 0LL, sink);
                    hipLaunchKernelGGL(burn, dim3(600), dim3(256), 0, A, dirty_wg, sink);
                }
            }
            CK(hipEventRecord(t1, A));
            CK(hipEventSynchronize(t1));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (pass == 1) {
                const char* names[] = {"one stream: dense, sparse, dense", "one stream: dense, dense (floor)", "two streams + events, compact dirty pass",
                                       "two streams + events, full-size early-exit pass + dirty pass"};
                printf("%-64s %7.1f us per iteration\n", names[variant], ms * 1e3 / REP);
            }
        }
    }
    return 0;
}
