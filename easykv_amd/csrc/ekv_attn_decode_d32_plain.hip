// decode kernels for head_dim = 32, pre-rotated keys
#define EKV_D 32
#define EKV_ROPE false
#define EKV_ROPE_TAG plain
#include "ekv_attn_decode.inc"
