// decode kernels for head_dim = 96 (12 live lanes of a 16-lane row group)
#define EKV_D 96
#define EKV_ROPE false
#define EKV_ROPE_TAG plain
#include "ekv_attn_decode.inc"
