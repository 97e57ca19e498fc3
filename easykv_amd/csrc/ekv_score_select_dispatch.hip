// Block-size dispatch of the generic scorer: 256-thread workgroups when the grid alone fills the chip (more workgroups
// resident per CU, cheaper barriers), 512 threads when there are few (head, layer) pairs.
#include <cstdlib>
#include "ekv_common.h"
#include "ekv_kernels.h"

size_t ekv_score_lds_bytes_nt256(const EkvScoreArgs&);
size_t ekv_score_lds_bytes_nt512(const EkvScoreArgs&);
hipError_t ekv_launch_score_select_nt256(const EkvScoreArgs&, int, hipStream_t);
hipError_t ekv_launch_score_select_nt512(const EkvScoreArgs&, int, hipStream_t);
hipError_t ekv_launch_tova_headmean_nt512(const EkvScoreArgs&, int, hipStream_t);
size_t ekv_score_lds_bytes_nt1024(const EkvScoreArgs&);
hipError_t ekv_launch_score_select_nt1024(const EkvScoreArgs&, int, hipStream_t);

size_t ekv_score_lds_bytes(const EkvScoreArgs& a) { return a.big_rows != nullptr ? ekv_score_lds_bytes_nt1024(a) : ekv_score_lds_bytes_nt512(a); }

// (mirrors ekv_score_lds_bytes_nt512 with all four arrays in LDS; decided at workspace-planning time, before the arguments exist)
bool ekv_score_rows_exceed_lds(int W, int rows) {
  return ekv_align((size_t)(4 * (size_t)W + 2 * (size_t)rows) * 4, 16) + 2 * 8 * 8 * 4 + 264 * 4 + 512 * 8 > 160 * 1024;
}

hipError_t ekv_launch_score_select(const EkvScoreArgs& a, int layer_count, hipStream_t s) {
  // 256 threads only while at least three such workgroups fit a CU's LDS; wide score rows (C4: W = 5098 -> 82 KB) leave
  // room for one workgroup per CU, which must then bring 512 threads
  if (a.big_rows != nullptr) return ekv_launch_score_select_nt1024(a, layer_count, s);   // rows in global scratch, keys in LDS
  static const int force = [] { const char* e = std::getenv("EKV_SS_NT"); return e ? std::atoi(e) : 0; }();   // (A/B knob: 256 / 512 / 1024)
  if (force == 256 && ekv_score_lds_bytes_nt256(a) <= 160 * 1024) return ekv_launch_score_select_nt256(a, layer_count, s);
  if (force == 512 && ekv_score_lds_bytes_nt512(a) <= 160 * 1024) return ekv_launch_score_select_nt512(a, layer_count, s);
  if (force == 1024 && ekv_score_lds_bytes_nt1024(a) <= 160 * 1024) return ekv_launch_score_select_nt1024(a, layer_count, s);
  const bool small_blocks = a.n_kv_heads * layer_count >= 768 && ekv_score_lds_bytes_nt256(a) <= 53 * 1024;
  if (small_blocks) return ekv_launch_score_select_nt256(a, layer_count, s);
  // one workgroup per CU either way (at most one (head, layer) pair per CU, or LDS rows too wide for two): give it all 16
  // wave slots — the logits sweep is VALU-bound on exact expf / IEEE div and 2 waves per SIMD do not fill the pipeline
  const bool one_per_cu = a.n_kv_heads * layer_count <= 256 || ekv_score_lds_bytes_nt512(a) > 80 * 1024;
  if (one_per_cu && ekv_score_lds_bytes_nt1024(a) <= 160 * 1024) return ekv_launch_score_select_nt1024(a, layer_count, s);
  return ekv_launch_score_select_nt512(a, layer_count, s);
}

hipError_t ekv_launch_tova_headmean(const EkvScoreArgs& a, int layer_count, hipStream_t s) {
  return ekv_launch_tova_headmean_nt512(a, layer_count, s);
}
