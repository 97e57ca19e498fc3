// K2 placeholder: strided-prefill chunk attention (q_len > 1).  Filled in by the MFMA kernel.
#include "ekv_common.h"
#include "ekv_kernels.h"

bool ekv_attn_chunk_supported(int, int, int) { return false; }
hipError_t ekv_launch_attn_chunk(const EkvAttnArgs&, int, int, hipStream_t) { return hipErrorInvalidValue; }
