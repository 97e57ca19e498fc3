// strided-prefill chunk kernels for head_dim = 128
#define EKV_D 128
#include "ekv_attn_chunk.inc"
