// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS code's access shapes: streaming reads of a known byte count
//   k_read16   16 B per lane (global_load_dwordx4): the shape MI355X_MICROARCH.md states its "FETCH_SIZE reports 1/2" rule for;
//   k_read4    4 B per lane, 24 SoA rows of one frame read by the same thread (256 B per wave instruction): the particle kernels' row reads;
//   k_write4   4 B per lane SoA row stores, k_write16 16 B per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; run each pass under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE
// (profiles/scripts/r05_fetch_calib.sh) and divide the counter (KB) by the bytes printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_read16(const float4* src, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) *sink = acc;
}
__global__ void k_read4(const float* src, size_t npad, int rows, float* sink) {      // one particle per thread, `rows` component rows
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npad) return;
    float acc = 0.f;
    for (int c = 0; c < rows; ++c) acc += src[(size_t)c * npad + p];
    if (acc == 12345.678f) *sink = acc;
}
__global__ void k_write4(float* dst, size_t npad, int rows) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npad) return;
    for (int c = 0; c < rows; ++c) dst[(size_t)c * npad + p] = (float)c;
}
__global__ void k_write16(float4* dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const size_t npad = 1 << 20;
    const int rows = 24, reps = 20;
    const size_t frames = 12;                                     // 12 frames x 96 MiB: a working set far beyond the 256 MiB Infinity Cache, each byte read once
    const size_t bytes = frames * rows * npad * sizeof(float);
    float *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const float4*)buf, bytes / 16, sink);
        for (size_t f = 0; f < frames; ++f) hipLaunchKernelGGL(k_read4, dim3(npad / 256), dim3(256), 0, 0, (const float*)(buf + f * rows * npad), npad, rows, sink);
        for (size_t f = 0; f < frames; ++f) hipLaunchKernelGGL(k_write4, dim3(npad / 256), dim3(256), 0, 0, buf + f * rows * npad, npad, rows);
        hipLaunchKernelGGL(k_write16, dim3(4096), dim3(256), 0, 0, (float4*)buf, bytes / 16);
    }
    hipDeviceSynchronize();
    printf("bytes per launch: k_read16 %zu  k_read4 %zu  k_write4 %zu  k_write16 %zu\n", bytes, (size_t)rows * npad * 4, (size_t)rows * npad * 4, bytes);
    return 0;
}
