// decode kernels for head_dim = 96 (12 live lanes of a 16-lane row group), rope-on-read (streaming)
#define EKV_D 96
#define EKV_ROPE true
#define EKV_ROPE_TAG rope
#include "ekv_attn_decode.inc"
