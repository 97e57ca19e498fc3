// decode kernels for head_dim = 128, pre-rotated keys
#define EKV_D 128
#define EKV_ROPE false
#define EKV_ROPE_TAG plain
#include "ekv_attn_decode.inc"
