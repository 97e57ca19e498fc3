// logits-resident scored chunk step (ekv_attn_resident.inc), head_dim 128
#include "ekv_attn_resident.inc"
