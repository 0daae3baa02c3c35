// Does a kernel launched with hipExtAnyOrderLaunch start while the previous kernel of the SAME stream is still running (gfx950, ROCm 7.2)?
// And does it still see what the kernel before that one wrote (is its start-of-kernel acquire kept)?
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder/probe.hip -o /tmp/anyorder_probe && /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_fill(unsigned* buf, size_t n, unsigned tag) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = tag + (unsigned)i;
}
// spins for `ticks` of the 100 MHz wall clock; stamps [start, end]
__global__ void k_spin(unsigned long long* stamp, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(20);
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[1] = wall_clock64();
}
__global__ void k_check(const unsigned* buf, size_t n, unsigned tag, unsigned long long* stamp, unsigned* bad) {
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[2] = wall_clock64();
    unsigned b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += buf[i] != tag + (unsigned)i;
    if (b) atomicAdd(bad, b);
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[3] = wall_clock64();
}

int main() {
    const size_t n = 1 << 20;     // 4 MB: stays in the L2s between iterations
    unsigned* buf; unsigned long long* stamp; unsigned* bad;
    CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&stamp, 64)); CK(hipMalloc(&bad, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int flags = 0; flags <= 1; ++flags) {
        int overlapped = 0; unsigned total_bad = 0; double gap = 0;
        const int iters = 50;
        for (int it = 0; it < iters; ++it) {
            CK(hipMemsetAsync(bad, 0, 4, s)); CK(hipMemsetAsync(stamp, 0, 64, s));
            const unsigned tag = 0x1000u * (it + 1) + flags;
            hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, buf, n, tag);
            hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, stamp, 5000ull);          // 50 us
            hipExtLaunchKernelGGL(k_check, dim3(1024), dim3(256), 0, s, nullptr, nullptr, flags, (const unsigned*)buf, n, tag, stamp, bad);
            CK(hipGetLastError());
            unsigned long long h[4]; unsigned hb;
            CK(hipMemcpyAsync(h, stamp, 32, hipMemcpyDeviceToHost, s)); CK(hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            overlapped += h[2] < h[1];
            gap += ((double)h[2] - (double)h[1]) * 0.01;
            total_bad += hb;
        }
        printf("flags=%d: check kernel started before the spin kernel ended in %d of %d runs; mean (check start - spin end) = %+.1f us; stale words %u\n",
               flags, overlapped, iters, gap / iters, total_bad);
    }
    return 0;
}
