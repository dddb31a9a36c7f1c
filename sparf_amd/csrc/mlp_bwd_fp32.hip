// Data-gradient kernels of the fp32 mode (the code is mlp_bwd_impl.h; dispatch: mlp_bwd.hip).
#include "mlp_bwd_impl.h"

namespace sparf {

int launch_mlp_bwd_fp32(bool pose, const MlpBwdArgs& a, int grid, hipStream_t stream) {
    if (pose) hipLaunchKernelGGL((mlp_bwd_kernel<PREC_FP32, true>), dim3(grid), dim3(Policy<PREC_FP32>::NWAVES * 64), 0, stream, a);
    else hipLaunchKernelGGL((mlp_bwd_kernel<PREC_FP32, false>), dim3(grid), dim3(Policy<PREC_FP32>::NWAVES * 64), 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf
