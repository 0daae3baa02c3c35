// What a launch pays for global atomics that many workgroups aim at the SAME words - the tails of the two hit-list kernels (k_contact_hits: 6 doubles of
// ext_f per primitive from every workgroup; k_contact_grad: 13 doubles of prim_state.grad per primitive from every workgroup, and f32 adds of neighbouring hits
// onto the same grid nodes).  A launch ends when its last atomic has been performed, whether a wave waited for it or not.
//   k_same     G workgroups, lanes 0..W-1 of each add one double to word [lane] of ONE record            (what the kernels did through round 5)
//   k_spread   the same, workgroup g aiming at record g % K of K records                                  (partial sums, folded by the reader)
//   k_f32      G workgroups x 216 lanes add 4 floats each to node records; every run of `share` consecutive workgroups aims at the same 216 nodes
//   k_f32_lanes  the same adds, the four words of a node in four neighbouring lanes of one instruction
// Build + run: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_chain atomic_chain.hip && ./atomic_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty() {}
__global__ void k_same(double* dst, int words, int K) {
    if ((int)threadIdx.x < words) unsafeAtomicAdd(dst + (size_t)(blockIdx.x % K) * 64 + threadIdx.x, 1.0);
}
__global__ void k_f32(float* dst, int share) {
    if (threadIdx.x < 216) {
        float* rec = dst + ((size_t)(blockIdx.x / share) * 216 + threadIdx.x) * 4;
        unsafeAtomicAdd(rec + 0, 1.f); unsafeAtomicAdd(rec + 1, 1.f); unsafeAtomicAdd(rec + 2, 1.f); unsafeAtomicAdd(rec + 3, 1.f);
    }
}

// the same adds with the four words of a node in four neighbouring LANES of one instruction (a line of 8 nodes gets ONE request per workgroup, not four)
__global__ void k_f32_lanes(float* dst, int share) {
    for (int q = threadIdx.x; q < 216 * 4; q += blockDim.x) unsafeAtomicAdd(dst + (size_t)(blockIdx.x / share) * 216 * 4 + q, 1.f);
}

template <class F> static float time_us(F launch, int reps = 200) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    double* d64; float* d32;
    hipMalloc(&d64, 4096 * 64 * sizeof(double)); hipMemset(d64, 0, 4096 * 64 * sizeof(double));
    hipMalloc(&d32, (size_t)2048 * 216 * 4 * sizeof(float)); hipMemset(d32, 0, (size_t)2048 * 216 * 4 * sizeof(float));
    const float empty = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0); });
    printf("back-to-back launches of an empty kernel, 256 workgroups: %.2f us each (subtract from the rows below)\n", empty);
    for (int words : {6, 13, 26})
        for (int G : {1, 16, 64, 256, 512})
            for (int K : {1, 8, 32}) {
                if (K > G) continue;
                const float t = time_us([&] { hipLaunchKernelGGL(k_same, dim3(G), dim3(256), 0, 0, d64, words, K); });
                printf("f64 adds: %2d words x %3d workgroups onto %2d record(s): %6.2f us per launch  (%5.1f ns per add in the longest chain of %d)\n", words, G, K, t,
                       (t - empty) * 1e3f / (G / K), G / K);
            }
    for (int G : {64, 256})
        for (int share : {1, 4, 16, 64, 256}) {
            const float t = time_us([&] { hipLaunchKernelGGL(k_f32, dim3(G), dim3(256), 0, 0, d32, share); });
            const float t2 = time_us([&] { hipLaunchKernelGGL(k_f32_lanes, dim3(G), dim3(256), 0, 0, d32, share); });
            printf("f32 adds: %3d workgroups x 216 nodes x 4 words, %2d workgroups per node set: %6.2f us per launch; a node's words in neighbouring lanes: %6.2f us\n", G, share, t, t2);
        }
    return 0;
}
