// Microbenchmark: streaming one particle frame (read 3 scalars, write 15 per particle - the traffic of k_g2p) with
//   SoA rows      addr(c, p) = c * Npad + p
//   AoSoA tiles   addr(c, p) = (p / 64) * (NC * 64) + c * 64 + (p % 64)
// Build: hipcc --offload-arch=gfx950 -O3 frame_layout.hip -o frame_layout
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int NC = 24;
// TILED: 0 = SoA rows, 1 = AoSoA tiles of 64 particles, 2 = AoSoA tiles of 4096 particles (16 KB per component)
template <int TILED> __device__ __forceinline__ size_t at(int c, int p, int Npad) {
    if (TILED == 1) return (size_t)(p >> 6) * (NC * 64) + (size_t)c * 64 + (p & 63);
    if (TILED == 2) return (size_t)(p >> 12) * (NC * 4096) + (size_t)c * 4096 + (p & 4095);
    if (TILED == 3) return (size_t)(p >> 8) * (NC * 256) + (size_t)c * 256 + (p & 255);
    if (TILED == 4) return (size_t)(p >> 10) * (NC * 1024) + (size_t)c * 1024 + (p & 1023);
    return (size_t)c * Npad + p;
}
template <int TILED, int NR, int NW>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ dst, int N, int Npad, int work) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < NR; ++c) acc += src[at<TILED>(c, p, Npad)];
    for (int i = 0; i < work; ++i) acc = acc * 1.0001f + 0.5f;          // stand-in for the stencil arithmetic
#pragma unroll
    for (int c = 0; c < NW; ++c) dst[at<TILED>(c + 3, p, Npad)] = acc + c;
}

// SoA rows, but the NW outputs of the workgroup go through LDS and every wave writes whole 1 KB row segments
template <int NR, int NW>
__global__ __launch_bounds__(256) void k_staged(const float* __restrict__ src, float* __restrict__ dst, int N, int Npad, int work) {
    __shared__ float st[NW][256];
    const int p = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (p < N) {
#pragma unroll
        for (int c = 0; c < NR; ++c) acc += src[(size_t)c * Npad + p];
    }
    for (int i = 0; i < work; ++i) acc = acc * 1.0001f + 0.5f;
#pragma unroll
    for (int c = 0; c < NW; ++c) st[c][threadIdx.x] = acc + c;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < NW; r += 4)
#pragma unroll
        for (int seg = 0; seg < 4; ++seg) {
            const int q = blockIdx.x * 256 + seg * 64 + lane;
            if (q < N) dst[(size_t)(r + 3) * Npad + q] = st[r][seg * 64 + lane];
        }
}
template <int NR, int NW> void run_staged(const char* name, float* a, float* b, int N, int Npad, int work) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_staged<NR, NW>), dim3((N + 255) / 256), dim3(256), 0, 0, a, b, N, Npad, work);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k_staged<NR, NW>), dim3((N + 255) / 256), dim3(256), 0, 0, a, b, N, Npad, work);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)N * 4 * (NR + NW);
    printf("%-34s read %2d write %2d work %4d : %7.1f us  %6.2f TB/s\n", name, NR, NW, work, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
}

template <int TILED, int NR, int NW> void run(const char* name, float* a, float* b, int N, int Npad, int work) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<TILED, NR, NW>), dim3((N + 255) / 256), dim3(256), 0, 0, a, b, N, Npad, work);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<TILED, NR, NW>), dim3((N + 255) / 256), dim3(256), 0, 0, a, b, N, Npad, work);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)N * 4 * (NR + NW);
    printf("%-34s read %2d write %2d work %4d : %7.1f us  %6.2f TB/s\n", name, NR, NW, work, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int N = 1 << 20;
    float *a, *b;
    (void)hipMalloc(&a, (size_t)NC * (N + (1 << 20)) * 4);
    (void)hipMalloc(&b, (size_t)NC * (N + (1 << 20)) * 4);
    (void)hipMemset(a, 0, (size_t)NC * (N + (1 << 20)) * 4);
    // row-stride skews (scalars) for the SoA layout: does the g2p-like slowdown depend on the stride between rows?
    for (int skew : {0, 4160}) {
        char name[64];
        snprintf(name, sizeof name, "SoA skew %6d (g2p-like)", skew);
        run<0, 3, 15>(name, a, b, N, N + skew, 256);
    }
    run<1, 3, 15>("AoSoA 64   (g2p-like)", a, b, N, N, 256);
    run<2, 3, 15>("AoSoA 4096 (g2p-like)", a, b, N, N, 256);
    run<3, 3, 15>("AoSoA 256  (g2p-like)", a, b, N, N, 256);
    run<4, 3, 15>("AoSoA 1024 (g2p-like)", a, b, N, N, 256);
    run<0, 24, 9>("SoA        (p2g-like)", a, b, N, N + 4160, 512);
    run<2, 24, 9>("AoSoA 4096 (p2g-like)", a, b, N, N, 512);
    run<4, 24, 9>("AoSoA 1024 (p2g-like)", a, b, N, N, 512);
    run<0, 18, 3>("SoA        (g2p_grad-like)", a, b, N, N + 4160, 256);
    run<2, 18, 3>("AoSoA 4096 (g2p_grad-like)", a, b, N, N, 256);
    run<4, 18, 3>("AoSoA 1024 (g2p_grad-like)", a, b, N, N, 256);
    run_staged<3, 15>("SoA staged 1 KB runs (g2p-like)", a, b, N, N + 4160, 256);
    run_staged<21, 21>("SoA staged 1 KB runs (p2g_grad)", a, b, N, N + 4160, 256);
    for (int skew : {0, 4160, 4128, 32 * 33}) {
        char name[64];
        snprintf(name, sizeof name, "SoA skew %6d (p2g_grad-like)", skew);
        run<0, 21, 21>(name, a, b, N, N + skew, 256);
    }
    run<1, 21, 21>("AoSoA 64   (p2g_grad-like)", a, b, N, N, 256);
    run<2, 21, 21>("AoSoA 4096 (p2g_grad-like)", a, b, N, N, 256);
    run<3, 21, 21>("AoSoA 256  (p2g_grad-like)", a, b, N, N, 256);
    run<4, 21, 21>("AoSoA 1024 (p2g_grad-like)", a, b, N, N, 256);
    return 0;
}
