// Data-gradient kernels of the bf16x3 mode, 8 waves / 256-row workgroup tiles (the code is mlp_bwd_impl.h; dispatch: mlp_bwd.hip):
// weights head + tail, propagated gradient in bf16 (mlp_dev.h PolicyX3DgradT).  -DSP_X3_DGRAD_FULL: the full head + tail backward instead.
#include "mlp_bwd_impl.h"

namespace sparf {

int launch_mlp_bwd_x3(bool pose, const MlpBwdArgs& a, int grid, hipStream_t stream) {
#ifdef SP_X3_DGRAD_FULL
    if (pose) hipLaunchKernelGGL((mlp_bwd_kernel<PREC_X3, true>), dim3(grid), dim3(Policy<PREC_X3>::NWAVES * 64), 0, stream, a);
    else hipLaunchKernelGGL((mlp_bwd_kernel<PREC_X3, false>), dim3(grid), dim3(Policy<PREC_X3>::NWAVES * 64), 0, stream, a);
#else
    // (the caller sizes the grid by CU count; the kernel strides over its own 256-row tiles)
    if (pose) hipLaunchKernelGGL((mlp_bwd_kernel<PREC_X3, true, PolicyX3Dgrad>), dim3(grid), dim3(PolicyX3Dgrad::NWAVES * 64), 0, stream, a);
    else hipLaunchKernelGGL((mlp_bwd_kernel<PREC_X3, false, PolicyX3Dgrad>), dim3(grid), dim3(PolicyX3Dgrad::NWAVES * 64), 0, stream, a);
#endif
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf

#ifdef SP_PROF      // wave-time accounting of THIS unit's kernels (each translation unit has its own g_prof_bwd): tools/kernel_bench.py bf16x3
extern "C" int sparf_debug_prof_bwd(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sparf::g_prof_bwd), 10 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
