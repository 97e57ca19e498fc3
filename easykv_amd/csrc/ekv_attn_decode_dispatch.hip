// head_dim / rope dispatch over the per-(D, rope) decode objects.
#include <cstdlib>

#include "ekv_common.h"
#include "ekv_kernels.h"

hipError_t ekv_launch_attn_decode_d32_plain(const EkvAttnArgs&, int, int, hipStream_t);
hipError_t ekv_launch_decode_fused_d32_plain(const EkvAttnArgs&, const EkvScoreArgs&, int, int, int, hipStream_t);
size_t ekv_fused_lds_d32_plain(int, int, int, int);
hipError_t ekv_launch_attn_decode_d32_rope(const EkvAttnArgs&, int, int, hipStream_t);
hipError_t ekv_launch_decode_fused_d32_rope(const EkvAttnArgs&, const EkvScoreArgs&, int, int, int, hipStream_t);
size_t ekv_fused_lds_d32_rope(int, int, int, int);
hipError_t ekv_launch_attn_decode_d64_plain(const EkvAttnArgs&, int, int, hipStream_t);
hipError_t ekv_launch_decode_fused_d64_plain(const EkvAttnArgs&, const EkvScoreArgs&, int, int, int, hipStream_t);
size_t ekv_fused_lds_d64_plain(int, int, int, int);
hipError_t ekv_launch_attn_decode_d64_rope(const EkvAttnArgs&, int, int, hipStream_t);
hipError_t ekv_launch_decode_fused_d64_rope(const EkvAttnArgs&, const EkvScoreArgs&, int, int, int, hipStream_t);
size_t ekv_fused_lds_d64_rope(int, int, int, int);
hipError_t ekv_launch_attn_decode_d96_plain(const EkvAttnArgs&, int, int, hipStream_t);
hipError_t ekv_launch_decode_fused_d96_plain(const EkvAttnArgs&, const EkvScoreArgs&, int, int, int, hipStream_t);
size_t ekv_fused_lds_d96_plain(int, int, int, int);
hipError_t ekv_launch_attn_decode_d96_rope(const EkvAttnArgs&, int, int, hipStream_t);
hipError_t ekv_launch_decode_fused_d96_rope(const EkvAttnArgs&, const EkvScoreArgs&, int, int, int, hipStream_t);
size_t ekv_fused_lds_d96_rope(int, int, int, int);
hipError_t ekv_launch_attn_decode_d128_plain(const EkvAttnArgs&, int, int, hipStream_t);
hipError_t ekv_launch_decode_fused_d128_plain(const EkvAttnArgs&, const EkvScoreArgs&, int, int, int, hipStream_t);
size_t ekv_fused_lds_d128_plain(int, int, int, int);
hipError_t ekv_launch_attn_decode_d128_rope(const EkvAttnArgs&, int, int, hipStream_t);
hipError_t ekv_launch_decode_fused_d128_rope(const EkvAttnArgs&, const EkvScoreArgs&, int, int, int, hipStream_t);
size_t ekv_fused_lds_d128_rope(int, int, int, int);

// any GQA factor (repeat_kv, llama_patch.py:19-29): factors <= 8 on the build of the next power of two, wider ones in groups of 8
bool ekv_attn_decode_supported(int head_dim, int rep) {
  return (head_dim == 32 || head_dim == 64 || head_dim == 96 || head_dim == 128) && rep >= 1;
}

#define EKV_DISPATCH(fn, ...)                                                    \
  switch (head_dim) {                                                            \
    case 32: return rope ? fn##32_rope(__VA_ARGS__) : fn##32_plain(__VA_ARGS__);  \
    case 64: return rope ? fn##64_rope(__VA_ARGS__) : fn##64_plain(__VA_ARGS__);  \
    case 96: return rope ? fn##96_rope(__VA_ARGS__) : fn##96_plain(__VA_ARGS__);  \
    case 128: return rope ? fn##128_rope(__VA_ARGS__) : fn##128_plain(__VA_ARGS__); \
  }

hipError_t ekv_launch_attn_decode(const EkvAttnArgs& a, int head_dim, int layer_count, hipStream_t s) {
  const int rep = a.n_q_heads / a.n_kv_heads;
  const bool rope = a.rope_cos != nullptr;
  EKV_DISPATCH(ekv_launch_attn_decode_d, a, rep, layer_count, s)
  return hipErrorInvalidValue;
}

// The whole decode step in one launch: possible when a head is not split, at most one victim, and the row fits.
// nw = 4: up to four workgroups per CU (LDS <= 80 KB keeps >= 2); nw = 8: one or two workgroups per CU.
int ekv_decode_fused_nw(int n_heads_in_launch) {
  static const int force = [] { const char* e = std::getenv("EKV_FUSED_NW"); return e ? std::atoi(e) : 0; }();   // (A/B knob)
  if (force == 4 || force == 8) return force;
  return (n_heads_in_launch >= 256 && n_heads_in_launch <= 512) ? 8 : 4;
}

bool ekv_decode_fused_supported(int head_dim, int rep, int n_slots, int t_pad, int l_pad, int n_evict, int cap, int nw) {
  // (the slot map and the score rows are fetched 16 bytes at a time: rows must be 16-byte aligned)
  if (!ekv_attn_decode_supported(head_dim, rep) || rep > 8 || n_evict > 1 || n_slots > 256 * 24 || (cap & 3) != 0 || cap < 16) return false;
  size_t lds = 1 << 30;
  switch (head_dim) {
    case 32: lds = ekv_fused_lds_d32_plain(rep, t_pad, l_pad, nw); break;
    case 64: lds = ekv_fused_lds_d64_plain(rep, t_pad, l_pad, nw); break;
    case 96: lds = ekv_fused_lds_d96_plain(rep, t_pad, l_pad, nw); break;
    case 128: lds = ekv_fused_lds_d128_plain(rep, t_pad, l_pad, nw); break;
  }
  return lds <= (nw == 8 ? 150 : 80) * 1024;   // 80 KB still leaves two 4-wave workgroups per CU
}

hipError_t ekv_launch_decode_fused(const EkvAttnArgs& a, const EkvScoreArgs& sc, int head_dim, int layer_count, int nw,
                                   hipStream_t s) {
  const int rep = a.n_q_heads / a.n_kv_heads;
  const bool rope = a.rope_cos != nullptr;
  EKV_DISPATCH(ekv_launch_decode_fused_d, a, sc, rep, layer_count, nw, s)
  return hipErrorInvalidValue;
}
