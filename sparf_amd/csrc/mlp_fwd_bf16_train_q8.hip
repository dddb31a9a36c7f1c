// Fused NeRF MLP forward, bf16 training kernel with 8-bit saves (layout.h AREA_Q8); the code is mlp_fwd_impl.h.
#define SP_FWD_PREC sparf::PREC_BF16
#define SP_FWD_SAVE 2
#define SP_FWD_LAUNCHER launch_mlp_fwd_bf16_train_q8
#define SP_FWD_PROF_EXPORT 0
#include "mlp_fwd_impl.h"
