// strided-prefill chunk kernels for head_dim = 96, EKV_CHUNK_MODE = 1 (see ekv_attn_chunk.inc)
#define EKV_D 96
#define EKV_CHUNK_MODE 1
#include "ekv_attn_chunk.inc"
