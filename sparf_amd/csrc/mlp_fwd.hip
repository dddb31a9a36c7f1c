// Fused NeRF MLP forward: dispatch to the per-kernel translation units (mlp_fwd_{bf16,fp32,x3}_{train,infer}.hip, all
// instantiating mlp_fwd_impl.h: one kernel per unit, compiled in parallel -- the fp32 kernels take minutes each).
#include "kernels.h"
#include "layout.h"

namespace sparf {

#define SP_DECL(n) int launch_mlp_fwd_##n##_train(const MlpFwdArgs&, int, hipStream_t); int launch_mlp_fwd_##n##_infer(const MlpFwdArgs&, int, hipStream_t);
SP_DECL(bf16) SP_DECL(fp32) SP_DECL(x3)
#undef SP_DECL
int launch_mlp_fwd_bf16_train_q8(const MlpFwdArgs&, int, hipStream_t);
int launch_mlp_fwd_x3_train_q8(const MlpFwdArgs&, int, hipStream_t);

// save: FWD_INFER (nothing saved), FWD_SAVE_PLANES, FWD_SAVE_Q8 (kernels.h)
int launch_mlp_fwd(int prec, int save, const MlpFwdArgs& a, int grid, hipStream_t stream) {
    if (save == FWD_SAVE_Q8) {
        if (prec == PREC_BF16) return launch_mlp_fwd_bf16_train_q8(a, grid, stream);
        if (prec == PREC_X3) return launch_mlp_fwd_x3_train_q8(a, grid, stream);
        return 1;
    }
    if (prec == PREC_BF16) return save ? launch_mlp_fwd_bf16_train(a, grid, stream) : launch_mlp_fwd_bf16_infer(a, grid, stream);
    if (prec == PREC_FP32) return save ? launch_mlp_fwd_fp32_train(a, grid, stream) : launch_mlp_fwd_fp32_infer(a, grid, stream);
    if (prec == PREC_X3) return save ? launch_mlp_fwd_x3_train(a, grid, stream) : launch_mlp_fwd_x3_infer(a, grid, stream);
    return 1;
}

}  // namespace sparf
