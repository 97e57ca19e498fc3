// decode kernels for head_dim = 64, pre-rotated keys
#define EKV_D 64
#define EKV_ROPE false
#define EKV_ROPE_TAG plain
#include "ekv_attn_decode.inc"
