// Register / LDS probe of the hot f32 kernels: compiles in ~20 s without the rest of the library (tools/regprobe/run.sh prints VGPRs, scratch, occupancy, LDS).
#include "../../softmac_amd/csrc/smac_kernels.hpp"
namespace smac {
template __global__ void k_p2g<float, true, false>(DevSim<float>, int);
template __global__ void k_g2p<float, true>(DevSim<float>, int);
template __global__ void k_g2p_grad<float, false>(DevSim<float>, int);
template __global__ void k_p2g_grad<float, false, true>(DevSim<float>, int);
template __global__ void k_p2g_g2p_grad<float, false>(DevSim<float>, int);
}
