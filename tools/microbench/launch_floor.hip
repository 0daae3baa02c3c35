// What a launch costs before any wave has anything to do, by the resources its waves are allocated: launch-to-launch time of kernels that do nothing,
// 256 workgroups of 256 threads each, back to back in one stream.
//   k_thin      no registers to speak of, no LDS, 16 bytes of arguments
//   k_args      the same with a 1,536-byte by-value argument (DevSim's size), one word of it read
//   k_fat       256 VGPRs + 32 AGPRs claimed, 43 KB of LDS, the 1,536-byte argument: k_contact_grad's allocation (one workgroup per CU)
// Build + run: hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>

struct Big { int w[384]; };
__global__ void k_thin(int* out, int n) { if (n == 12345) out[0] = 1; }
__global__ void k_args(Big b, int* out) { if (b.w[7] == 12345) out[0] = 1; }
__global__ __launch_bounds__(256) void k_fat(Big b, int* out) {
    __shared__ int lds[43 * 256];
    if (b.w[7] == 12345) { lds[threadIdx.x] = 1; out[0] = lds[(threadIdx.x + 1) & 255]; }
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a31, v255" ::: "v255", "a31");
}

// a kernel that lasts `ticks` of the constant 100 MHz clock in every wave: with the host far ahead of the device, launch-to-launch time minus the kernel's own is the
// device-side gap between two dependent launches
__global__ void k_busy(int* out, unsigned ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks == 12345u) out[0] = 1;
}

template <class F> static float time_us(F launch, int reps = 400) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 40; ++i) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

// the same launches recorded once into a graph of 120 kernel nodes (a chain: every node depends on the one before) and replayed
template <class F> static float graph_us(F launch) {
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipGraph_t g;
    hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 120; ++i) launch(st);
    (void)hipStreamEndCapture(st, &g);
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) return -1.f;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    (void)hipEventRecord(a, st);
    for (int i = 0; i < 10; ++i) (void)hipGraphLaunch(ge, st);
    (void)hipEventRecord(b, st);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / 1200;
}

int main() {
    int* out;
    (void)hipMalloc(&out, 64);
    Big big = {};
    for (unsigned ticks : {1000u, 2000u}) {                 // 10 us and 20 us kernels: the host enqueues a launch in ~3 us and is ahead
        const float ts = time_us([&] { hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, 0, out, ticks); }, 200);
        const float tg = graph_us([&](hipStream_t st) { hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, st, out, ticks); });
        printf("kernels of %u us, 256 workgroups, chained: %.2f us launch to launch in a stream, %.2f us as nodes of a graph\n", ticks / 100, ts, tg);
    }
    printf("graph of 120 chained nodes, 256 workgroups: thin %.2f us, fat %.2f us per node\n",
           graph_us([&](hipStream_t st) { hipLaunchKernelGGL(k_thin, dim3(256), dim3(256), 0, st, out, 0); }),
           graph_us([&](hipStream_t st) { hipLaunchKernelGGL(k_fat, dim3(256), dim3(256), 0, st, big, out); }));
    for (int wgs : {1, 256, 1024}) {
        const float t0 = time_us([&] { hipLaunchKernelGGL(k_thin, dim3(wgs), dim3(256), 0, 0, out, 0); });
        const float t1 = time_us([&] { hipLaunchKernelGGL(k_args, dim3(wgs), dim3(256), 0, 0, big, out); });
        const float t2 = time_us([&] { hipLaunchKernelGGL(k_fat, dim3(wgs), dim3(256), 0, 0, big, out); });
        printf("%4d workgroups: thin %.2f us, 1.5 KB of arguments %.2f us, + 256 VGPRs + 32 AGPRs + 43 KB LDS %.2f us per launch\n", wgs, t0, t1, t2);
    }
    return 0;
}
