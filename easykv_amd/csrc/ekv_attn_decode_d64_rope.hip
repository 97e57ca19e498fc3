// decode kernels for head_dim = 64, rope-on-read (streaming)
#define EKV_D 64
#define EKV_ROPE true
#define EKV_ROPE_TAG rope
#include "ekv_attn_decode.inc"
