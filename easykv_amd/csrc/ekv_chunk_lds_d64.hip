// ekv_chunk_lds_kernel for head_dim 64
#define EKV_D 64
#include "ekv_chunk_lds.inc"
