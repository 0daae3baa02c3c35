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

int main() {
    int* out;
    (void)hipMalloc(&out, 64);
    Big big = {};
    for (int wgs : {1, 256, 1024}) {
        const float t0 = time_us([&] { hipLaunchKernelGGL(k_thin, dim3(wgs), dim3(256), 0, 0, out, 0); });
        const float t1 = time_us([&] { hipLaunchKernelGGL(k_args, dim3(wgs), dim3(256), 0, 0, big, out); });
        const float t2 = time_us([&] { hipLaunchKernelGGL(k_fat, dim3(wgs), dim3(256), 0, 0, big, out); });
        printf("%4d workgroups: thin %.2f us, 1.5 KB of arguments %.2f us, + 256 VGPRs + 32 AGPRs + 43 KB LDS %.2f us per launch\n", wgs, t0, t1, t2);
    }
    return 0;
}
