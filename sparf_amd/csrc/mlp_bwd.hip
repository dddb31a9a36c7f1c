// Data-gradient kernels over plane save / gradient areas (the code is mlp_bwd_impl.h) and the launch dispatch.
#include "mlp_bwd_impl.h"

namespace sparf {

int launch_mlp_bwd_q8(int prec, bool pose, const MlpBwdArgs& a, int grid, hipStream_t stream);     // mlp_bwd_q8.hip

int launch_mlp_bwd(int prec, bool pose, bool q8, const MlpBwdArgs& a, int grid, hipStream_t stream) {
    if (a.rows <= 0) return 0;
    if (q8) return launch_mlp_bwd_q8(prec, pose, a, grid, stream);
#define SP_LAUNCH(PR, PO) \
    hipLaunchKernelGGL((mlp_bwd_kernel<PR, PO>), dim3(grid), dim3(Policy<PR>::NWAVES * 64), 0, stream, a)
    if (prec == PREC_BF16) { if (pose) SP_LAUNCH(PREC_BF16, true); else SP_LAUNCH(PREC_BF16, false); }
    else if (prec == PREC_FP32) { if (pose) SP_LAUNCH(PREC_FP32, true); else SP_LAUNCH(PREC_FP32, false); }
    else if (prec == PREC_X3) {
#ifdef SP_X3_DGRAD_FULL
        if (pose) SP_LAUNCH(PREC_X3, true); else SP_LAUNCH(PREC_X3, false);
#else
        // weights head + tail, propagated gradient in bf16 (mlp_dev.h PolicyX3Dgrad): the caller sized the grid
        // for 128-row tiles, the kernel strides over 256-row tiles
        if (pose) hipLaunchKernelGGL((mlp_bwd_kernel<PREC_X3, true, PolicyX3Dgrad>), dim3(grid), dim3(PolicyX3Dgrad::NWAVES * 64), 0, stream, a);
        else hipLaunchKernelGGL((mlp_bwd_kernel<PREC_X3, false, PolicyX3Dgrad>), dim3(grid), dim3(PolicyX3Dgrad::NWAVES * 64), 0, stream, a);
#endif
    }
    else return 1;
#undef SP_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf

#ifdef SP_PROF
extern "C" int sparf_debug_prof_bwd(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sparf::g_prof_bwd), 10 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
