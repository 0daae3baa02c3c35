// Microbenchmark: do VALU work and LDS f64 atomics of different waves on one CU overlap?
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_overlap.hip -o lds_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float valu_work(float x, int n) {
    float a = x, b = x * 0.5f, c = x * 0.25f, d = x + 1.0f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { a = a * 1.0001f + b; b = b * 0.9999f + c; c = c * 1.0002f + d; d = d * 0.9998f + a; }
    }
    return a + b + c + d;
}
__device__ __forceinline__ void lds_work(double* buf, int lane, int n) {
    int idx = lane;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            __hip_atomic_fetch_add(&buf[(idx + u * 67) & 1023], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        idx = (idx + 131) & 1023;
    }
}

// mode 0: V only, 1: L only, 2: every wave V then L, 3: even (wave+block) waves V then L, odd waves L then V
__global__ __launch_bounds__(256) void k(float* out, int mode, int nv, int nl) {
    __shared__ double buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = 0.0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    const bool flip = mode == 3 && ((wave + blockIdx.x) & 1);
    if (mode == 0) r = valu_work((float)threadIdx.x, nv);
    else if (mode == 1) lds_work(buf, threadIdx.x, nl);
    else if (!flip) { r = valu_work((float)threadIdx.x, nv); lds_work(buf, threadIdx.x, nl); }
    else { lds_work(buf, threadIdx.x, nl); r = valu_work((float)threadIdx.x, nv); }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = r + (float)buf[threadIdx.x];
}

int main() {
    float* d;
    (void)hipMalloc(&d, 4096 * 256 * sizeof(float));
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int nv = 256, nl = 64;      // 8192 dependent-ish FMAs x4 chains, 512 atomics per wave
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, mode, 8, 8);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, mode, nv, nl);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        const char* names[] = {"VALU only", "LDS atomics only", "every wave: VALU then LDS", "half the waves LDS first"};
        printf("%-28s %8.3f ms\n", names[mode], ms);
    }
    return 0;
}
