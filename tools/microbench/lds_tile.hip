// Microbenchmark: LDS f64 atomic scatter of a 27-node stencil into a 6^3 tile, by tile strides and lane->cell order.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_tile.hip -o lds_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>

template <class T, int SX, int SY, int SZ, int WORDS>
__global__ __launch_bounds__(256) void k(const int* cells, T* out, int iters) {
    __shared__ T buf[WORDS];
    for (int i = threadIdx.x; i < WORDS; i += 256) buf[i] = T(0);
    __syncthreads();
    const int c = cells[blockIdx.x * 256 + threadIdx.x];
    const int x = c >> 4, y = (c >> 2) & 3, z = c & 3;
    const int base = x * SX + y * SY + z * SZ;
    T v = T(1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int kk = 0; kk < 3; ++kk)
                    __hip_atomic_fetch_add(&buf[base + i * SX + j * SY + kk * SZ], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x];
}

template <class T, int SX, int SY, int SZ, int WORDS> void run(const char* name, const int* dcells) {
    T* d;
    (void)hipMalloc(&d, 1024 * 256 * sizeof(T));
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int ITERS = 64;
    hipLaunchKernelGGL((k<T, SX, SY, SZ, WORDS>), dim3(1024), dim3(256), 0, 0, dcells, d, 4);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<T, SX, SY, SZ, WORDS>), dim3(1024), dim3(256), 0, 0, dcells, d, ITERS);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    double waveinstr = 1024.0 * 4 * ITERS * 27;
    double cyc = ms * 1e-3 * 2.4e9 / (waveinstr / 256);
    printf("%-44s %8.3f ms  %6.1f cycles/wave-instr/CU\n", name, ms, cyc);
    (void)hipFree(d);
}

int main() {
    const int n = 1024 * 256;
    std::vector<int> ident(n), perm(n), gaps(n);
    std::mt19937 rng(1);
    for (int w = 0; w < n / 64; ++w) {
        std::vector<int> p(64);
        for (int i = 0; i < 64; ++i) p[i] = i;
        for (int i = 0; i < 64; ++i) ident[w * 64 + i] = i;
        std::shuffle(p.begin(), p.end(), rng);
        for (int i = 0; i < 64; ++i) perm[w * 64 + i] = p[i];
        // "sorted with gaps and a bin boundary": ascending cells, each kept with prob 0.8, wrapping into the next bin
        int c = rng() % 64, i = 0;
        while (i < 64) { if (rng() % 10 < 8) gaps[w * 64 + i++] = c; c = (c + 1) % 64; }
    }
    int *di, *dp, *dg;
    (void)hipMalloc(&di, n * 4); (void)hipMalloc(&dp, n * 4); (void)hipMalloc(&dg, n * 4);
    (void)hipMemcpy(di, ident.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dp, perm.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dg, gaps.data(), n * 4, hipMemcpyHostToDevice);
    run<double, 36, 6, 1, 216>("f64 strides 36,6,1  lane=random cell", dp);
    run<double, 36, 6, 1, 216>("f64 strides 36,6,1  lane=cell", di);
    run<double, 36, 6, 1, 216>("f64 strides 36,6,1  sorted+gaps", dg);
    run<double, 80, 4, 9, 466>("f64 strides 80,4,9  lane=random cell", dp);
    run<double, 80, 4, 9, 466>("f64 strides 80,4,9  lane=cell", di);
    run<double, 80, 4, 9, 466>("f64 strides 80,4,9  sorted+gaps", dg);
    run<double, 80, 36, 1, 586>("f64 strides 80,36,1 lane=cell", di);
    run<double, 80, 36, 1, 586>("f64 strides 80,36,1 sorted+gaps", dg);
    run<unsigned, 36, 6, 1, 216>("u32 strides 36,6,1  lane=random cell", dp);
    run<unsigned, 36, 6, 1, 216>("u32 strides 36,6,1  lane=cell", di);
    run<unsigned, 36, 6, 1, 216>("u32 strides 36,6,1  sorted+gaps", dg);
    run<unsigned, 52, 16, 11, 396>("u32 strides 52,16,11 lane=random cell", dp);
    run<unsigned, 52, 16, 11, 396>("u32 strides 52,16,11 lane=cell", di);
    run<unsigned, 52, 16, 11, 396>("u32 strides 52,16,11 sorted+gaps (2 bins)", dg);
    run<int, 52, 16, 11, 396>("i32 strides 52,16,11 lane=random cell", dp);
    return 0;
}
