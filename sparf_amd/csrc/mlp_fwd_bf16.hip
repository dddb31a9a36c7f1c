// Fused NeRF MLP forward, bf16 instantiations (training + inference kernels); the code is mlp_fwd_impl.h.
#define SP_FWD_PREC sparf::PREC_BF16
#define SP_FWD_LAUNCHER launch_mlp_fwd_bf16
#define SP_FWD_PROF_EXPORT 1
#include "mlp_fwd_impl.h"
