// Which XCD does workgroup b of a 1-D grid run on?  (xcd_chunk() in smac_kernels.hpp assumes that
// blocks b and b+8 share an XCD.)  Reads HW_REG_XCC_ID in every workgroup and prints, per residue b % 8, the set of XCC ids seen.
//   hipcc --offload-arch=gfx950 -O2 -o tools/microbench/xcc_probe tools/microbench/xcc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_probe(int* out, int spin) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    // keep the block resident for a while so that the grid really spreads over the chip
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(id & 0xf) | (a == 123.f ? 16 : 0);
}

int main() {
    for (int nblocks : {256, 2048, 4688, 16384}) {
        for (int threads : {256}) {
            int* d;
            hipMalloc(&d, nblocks * sizeof(int));
            hipLaunchKernelGGL(k_probe, dim3(nblocks), dim3(threads), 0, 0, d, 2000);
            std::vector<int> h(nblocks);
            hipMemcpy(h.data(), d, nblocks * sizeof(int), hipMemcpyDeviceToHost);
            int sets[8] = {0};
            int mismatch = 0;
            for (int b = 0; b < nblocks; ++b) {
                sets[b & 7] |= 1 << (h[b] & 15);
                if ((h[b] & 15) != (h[b & 7] & 15)) ++mismatch;
            }
            printf("grid %d x %d: ", nblocks, threads);
            for (int r = 0; r < 8; ++r) printf("b%%8=%d->{mask 0x%x} ", r, sets[r]);
            printf(" blocks not on the XCD of block (b%%8): %d\n", mismatch);
            hipFree(d);
        }
    }
    return 0;
}
