// generic scorer, 512 threads per workgroup
#define EKV_SS_NT 512
#include "ekv_score_select.inc"
