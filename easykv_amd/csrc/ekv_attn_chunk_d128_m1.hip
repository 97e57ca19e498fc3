// strided-prefill chunk kernels for head_dim = 128, EKV_CHUNK_MODE = 1 (see ekv_attn_chunk.inc)
#define EKV_D 128
#define EKV_CHUNK_MODE 1
#include "ekv_attn_chunk.inc"
