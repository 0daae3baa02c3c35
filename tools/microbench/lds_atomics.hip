// Microbenchmark: LDS atomic throughput on gfx950 by flavour and conflict pattern.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomics.hip -o lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int WORDS = 4096;
constexpr int ITERS = 512;

template <class T, int MODE>
__global__ __launch_bounds__(256) void k(T* out, int stride, int iters) {
    __shared__ T buf[WORDS];
    for (int i = threadIdx.x; i < WORDS; i += 256) buf[i] = T(0);
    __syncthreads();
    const int lane = threadIdx.x;
    int idx = (lane * stride) % WORDS;
    T v = T(1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int a = (idx + u * 67) % WORDS;
            if (MODE == 3) { if ((lane & 7) == 0) __hip_atomic_fetch_add(&buf[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            else if (MODE == 4) { if ((lane & 1) == 0) __hip_atomic_fetch_add(&buf[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            else if (MODE == 0) __hip_atomic_fetch_add(&buf[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 1) buf[a] = v;                      // plain store
            else if (MODE == 2) v += buf[a];                     // plain load
        }
        idx = (idx + 131) % WORDS;
    }
    __syncthreads();
    if (MODE == 2) buf[lane] = v;
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x];
}

template <class T, int MODE> void run(const char* name, int stride) {
    T* d;
    hipMalloc(&d, 1024 * 256 * sizeof(T));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<T, MODE>), dim3(1024), dim3(256), 0, 0, d, stride, 16);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<T, MODE>), dim3(1024), dim3(256), 0, 0, d, stride, ITERS);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double waveinstr = 1024.0 * 4 * ITERS * 8;            // wave-level LDS instructions
    double per_cu = waveinstr / 256;
    double cyc = ms * 1e-3 * 2.4e9 / per_cu;               // LDS cycles per wave-instruction per CU (at 2.4 GHz)
    printf("%-28s stride %3d : %8.3f ms  %7.1f cycles/wave-instr/CU  %7.1f Glane-ops/s\n", name, stride, ms, cyc, waveinstr * 64 / ms / 1e6);
    hipFree(d);
}

int main() {
    for (int stride : {1, 0, 32}) {     // 1: conflict-free, 0: all lanes same address, 32: same bank different address
        run<float, 0>("ds_add_f32", stride);
        run<double, 0>("ds_add_f64", stride);
        run<unsigned, 0>("ds_add_u32", stride);
        run<unsigned long long, 0>("ds_add_u64", stride);
        run<int, 0>("ds_add_i32", stride);
        run<double, 3>("ds_add_f64 1/8 lanes", stride);
        run<double, 4>("ds_add_f64 1/2 lanes", stride);
        run<float, 3>("ds_add_f32 1/8 lanes", stride);
        run<float, 1>("ds_write_b32", stride);
        run<float, 2>("ds_read_b32", stride);
    }
    return 0;
}
