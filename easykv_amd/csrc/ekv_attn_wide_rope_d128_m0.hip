// wide-query-block attention kernel (ekv_attn_wide.inc), head_dim 128, mode 0, RoPE-on-read
#define EKV_D 128
#define EKV_WIDE_MODE 0
#define EKV_WIDE_ROPE 1
#include "ekv_attn_wide.inc"
