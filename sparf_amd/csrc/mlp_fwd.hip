// Fused NeRF MLP forward: dispatch to the per-precision translation units (mlp_fwd_{bf16,fp32,x3}.hip, all
// instantiating mlp_fwd_impl.h).
#include "kernels.h"
#include "layout.h"

namespace sparf {

int launch_mlp_fwd_bf16(bool save, const MlpFwdArgs& a, int grid, hipStream_t stream);
int launch_mlp_fwd_fp32(bool save, const MlpFwdArgs& a, int grid, hipStream_t stream);
int launch_mlp_fwd_x3(bool save, const MlpFwdArgs& a, int grid, hipStream_t stream);

int launch_mlp_fwd(int prec, bool save, const MlpFwdArgs& a, int grid, hipStream_t stream) {
    if (prec == PREC_BF16) return launch_mlp_fwd_bf16(save, a, grid, stream);
    if (prec == PREC_FP32) return launch_mlp_fwd_fp32(save, a, grid, stream);
    if (prec == PREC_X3) return launch_mlp_fwd_x3(save, a, grid, stream);
    return 1;
}

}  // namespace sparf
