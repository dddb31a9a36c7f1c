// Fused NeRF MLP forward, x3 training (activation-saving) kernel; the code is mlp_fwd_impl.h.
#define SP_FWD_PREC sparf::PREC_X3
#define SP_FWD_SAVE true
#define SP_FWD_LAUNCHER launch_mlp_fwd_x3_train
#define SP_FWD_PROF_EXPORT 1
#include "mlp_fwd_impl.h"
