// Fused NeRF MLP forward, fp32 training (activation-saving) kernel; the code is mlp_fwd_impl.h.
#define SP_FWD_PREC sparf::PREC_FP32
#define SP_FWD_SAVE true
#define SP_FWD_LAUNCHER launch_mlp_fwd_fp32_train
#define SP_FWD_PROF_EXPORT 0
#include "mlp_fwd_impl.h"
