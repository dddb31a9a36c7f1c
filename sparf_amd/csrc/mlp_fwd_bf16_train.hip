// Fused NeRF MLP forward, bf16 training (activation-saving) kernel; the code is mlp_fwd_impl.h.
#define SP_FWD_PREC sparf::PREC_BF16
#define SP_FWD_SAVE true
#define SP_FWD_LAUNCHER launch_mlp_fwd_bf16_train
#define SP_FWD_PROF_EXPORT 0
#include "mlp_fwd_impl.h"
