// wide-query-block attention kernel (ekv_attn_wide.inc), head_dim 64, mode 1
#define EKV_D 64
#define EKV_WIDE_MODE 1
#include "ekv_attn_wide.inc"
