// Data-gradient kernels over 8-bit save / gradient areas (layout.h AREA_Q8; the code is mlp_bwd_impl.h): the bf16-operand modes.
#include "mlp_bwd_impl.h"

namespace sparf {

int launch_mlp_bwd_q8(int prec, bool pose, const MlpBwdArgs& a, int grid, hipStream_t stream) {
    if (a.rows <= 0) return 0;
    if (prec == PREC_BF16) {
        if (pose) hipLaunchKernelGGL((mlp_bwd_kernel<PREC_BF16, true, Policy<PREC_BF16>, true>), dim3(grid), dim3(Policy<PREC_BF16>::NWAVES * 64), 0, stream, a);
        else hipLaunchKernelGGL((mlp_bwd_kernel<PREC_BF16, false, Policy<PREC_BF16>, true>), dim3(grid), dim3(Policy<PREC_BF16>::NWAVES * 64), 0, stream, a);
    } else if (prec == PREC_X3) {
        // (weights head + tail, propagated gradient in bf16, 256-row tiles: mlp_bwd.hip)
        if (pose) hipLaunchKernelGGL((mlp_bwd_kernel<PREC_X3, true, PolicyX3Dgrad, true>), dim3(grid), dim3(PolicyX3Dgrad::NWAVES * 64), 0, stream, a);
        else hipLaunchKernelGGL((mlp_bwd_kernel<PREC_X3, false, PolicyX3Dgrad, true>), dim3(grid), dim3(PolicyX3Dgrad::NWAVES * 64), 0, stream, a);
    } else return 1;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf
