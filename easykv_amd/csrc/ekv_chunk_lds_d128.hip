// ekv_chunk_lds_kernel for head_dim 128
#define EKV_D 128
#include "ekv_chunk_lds.inc"
