// generic scorer, 1024 threads per workgroup (one workgroup per CU: few (head, layer) pairs, or score rows so wide that LDS
// holds only one workgroup — its exact expf / IEEE-div sweep is VALU-bound and wants all 16 wave slots of the CU)
#define EKV_SS_NT 1024
#include "ekv_score_select.inc"
