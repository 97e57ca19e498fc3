// strided-prefill chunk kernels for head_dim = 64
#define EKV_D 64
#include "ekv_attn_chunk.inc"
