// strided-prefill chunk kernels for head_dim = 32, EKV_CHUNK_MODE = 0 (see ekv_attn_chunk.inc)
#define EKV_D 32
#define EKV_CHUNK_MODE 0
#include "ekv_attn_chunk.inc"
