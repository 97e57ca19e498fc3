// strided-prefill chunk kernels for head_dim = 32
#define EKV_D 32
#include "ekv_attn_chunk.inc"
