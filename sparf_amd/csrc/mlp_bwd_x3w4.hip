// Data-gradient kernels of the bf16x3 mode, 4 waves / 128-row workgroup tiles: the geometry api.hip picks for launches whose row count
// leaves the 256-row kernel's last round of tiles mostly empty (small passes: a 512-ray step, i.e. a 4096-ray batch strong-scaled over
// 8 GPUs).  Same arithmetic, same tile-block areas, bit-identical results (tests/test_hip_gpu.py); 12-17 % slower per row than the
// 8-wave kernel at full occupancy (mlp_bwd_impl.h), which is why it is not the only one.
#include "mlp_bwd_impl.h"

namespace sparf {

int launch_mlp_bwd_x3w4(bool pose, const MlpBwdArgs& a, int grid, hipStream_t stream) {
    if (pose) hipLaunchKernelGGL((mlp_bwd_kernel<PREC_X3, true, PolicyX3DgradW4>), dim3(grid), dim3(PolicyX3DgradW4::NWAVES * 64), 0, stream, a);
    else hipLaunchKernelGGL((mlp_bwd_kernel<PREC_X3, false, PolicyX3DgradW4>), dim3(grid), dim3(PolicyX3DgradW4::NWAVES * 64), 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf
