// generic scorer, 256 threads per workgroup
#define EKV_SS_NT 256
#include "ekv_score_select.inc"
