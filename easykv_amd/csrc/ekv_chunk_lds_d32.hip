// ekv_chunk_lds_kernel for head_dim 32
#define EKV_D 32
#include "ekv_chunk_lds.inc"
