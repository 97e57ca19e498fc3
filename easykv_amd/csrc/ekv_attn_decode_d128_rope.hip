// decode kernels for head_dim = 128, rope-on-read (streaming)
#define EKV_D 128
#define EKV_ROPE true
#define EKV_ROPE_TAG rope
#include "ekv_attn_decode.inc"
