// Data-gradient kernels: the launch dispatch, and the bf16 kernels over plane save / gradient areas (the code is mlp_bwd_impl.h; the other
// precisions are their own translation units -- mlp_bwd_fp32.hip, mlp_bwd_x3.hip, mlp_bwd_x3w4.hip, mlp_bwd_q8.hip -- so that they compile in
// parallel: as one unit the six plane kernels took eleven minutes).
#include "mlp_bwd_impl.h"

namespace sparf {

int launch_mlp_bwd_q8(int prec, bool pose, const MlpBwdArgs& a, int grid, hipStream_t stream);     // mlp_bwd_q8.hip
int launch_mlp_bwd_fp32(bool pose, const MlpBwdArgs& a, int grid, hipStream_t stream);             // mlp_bwd_fp32.hip
int launch_mlp_bwd_x3(bool pose, const MlpBwdArgs& a, int grid, hipStream_t stream);               // mlp_bwd_x3.hip   (8 waves, 256-row tiles)
int launch_mlp_bwd_x3w4(bool pose, const MlpBwdArgs& a, int grid, hipStream_t stream);             // mlp_bwd_x3w4.hip (4 waves, 128-row tiles)

// waves: geometry of the bf16x3 kernel (8 | 4; kernels.h) -- ignored by the other precisions and by the 8-bit-area kernels
int launch_mlp_bwd(int prec, bool pose, bool q8, const MlpBwdArgs& a, int grid, hipStream_t stream, int waves) {
    if (a.rows <= 0) return 0;
    if (q8) return launch_mlp_bwd_q8(prec, pose, a, grid, stream);
    if (prec == PREC_FP32) return launch_mlp_bwd_fp32(pose, a, grid, stream);
    if (prec == PREC_X3) {
#if SP_X3_DGRAD_WAVES
        waves = SP_X3_DGRAD_WAVES;
#endif
        return waves == 4 ? launch_mlp_bwd_x3w4(pose, a, grid, stream) : launch_mlp_bwd_x3(pose, a, grid, stream);
    }
    if (prec != PREC_BF16) return 1;
    if (pose) hipLaunchKernelGGL((mlp_bwd_kernel<PREC_BF16, true>), dim3(grid), dim3(Policy<PREC_BF16>::NWAVES * 64), 0, stream, a);
    else hipLaunchKernelGGL((mlp_bwd_kernel<PREC_BF16, false>), dim3(grid), dim3(Policy<PREC_BF16>::NWAVES * 64), 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf

