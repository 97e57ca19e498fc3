// Split-path scorer for decode steps (q_len == 1, at most one victim): one workgroup per (KV head, layer).
// Folds the key-range-split partials of ekv_attn_decode_kernel into the fp16 output, pulls the exported logits and
// the score rows into LDS by LDS-DMA and runs the same scorer tail as the fused kernel (ekv_decode_tail.h).
// Used when layers are launched one at a time (heads must be split to fill the chip).
#ifdef EKV_TAIL_PROFILE
#define EKV_STAMP(i) do { if (threadIdx.x == 0) stamps[i] = __builtin_readcyclecounter(); } while (0)
#endif
#include "ekv_decode_tail.h"

namespace {

#ifndef EKV_SCORE_NW
#define EKV_SCORE_NW 8
#endif
constexpr int kSNW = EKV_SCORE_NW, kSNT = 64 * kSNW;   // waves / threads per scorer workgroup

template <int REP, int ITEMS>
__global__ void __launch_bounds__(kSNT) ekv_decode_score_kernel(const EkvScoreArgs sc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int h = blockIdx.x, ll = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int T = sc.n_slots, D = sc.head_dim, t_pad = sc.t_pad;
  const bool roco = sc.policy == EKV_POLICY_ROCO;
  const bool scored = roco || sc.policy == EKV_POLICY_H2O_HEAD || sc.policy == EKV_POLICY_TOVA;
  const int off = scored ? sc.score_off : 0;
  const int W = T - off;
  const int w_pad = (int)ekv_align((size_t)W, 256);
  float* s_logit = reinterpret_cast<float*>(smem);
  float* sS = s_logit + (size_t)REP * t_pad;
  float* sQ = sS + w_pad;
  float* sC = sQ + w_pad;
  RedN<kSNW> red;
  red.buf = reinterpret_cast<unsigned long long*>(sS + (size_t)(roco ? 3 : 1) * w_pad);
  red.phase = 0;
  red.lane = lane;
  red.wave = wave;
  const size_t head_row = ((size_t)(sc.layer_begin + ll) * sc.n_kv_heads + h) * sc.cap;
  // REP = the GQA factor rounded up to 1 / 2 / 4 / 8 (ekv_attn_decode.inc); the padding rows repeat the last real head's logits
  const int nrep = (REP == 1 || REP == 2) ? REP : sc.n_q_heads / sc.n_kv_heads;
  const size_t hq0 = (size_t)ll * sc.n_q_heads + (size_t)h * nrep;

#ifdef EKV_TAIL_PROFILE
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(sc.tova_row) + ((size_t)ll * sc.n_kv_heads + h) * 8;
#endif
  EKV_STAMP(0);
  if (scored) ekv_tail_prefetch_rows<kSNW>(sc, head_row, W, w_pad, roco, sS, sQ, sC);
  if (scored && sc.accumulate) {   // logits rows -> LDS (rows are 256-byte aligned in the workspace)
    const int full = t_pad / 256;
    for (int c = wave; c < full * REP; c += kSNW) {
      const int r = c / full, ch = c % full;
      const float* src = sc.logits + (hq0 + min(r, nrep - 1)) * t_pad + ch * 256 + lane * 4;
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(s_logit + (size_t)r * t_pad + ch * 256), 16, 0, 0);
    }
    for (int r = 0; r < REP; ++r)
      for (int j = full * 256 + tid; j < t_pad; j += kSNT) s_logit[(size_t)r * t_pad + j] = sc.logits[(hq0 + min(r, nrep - 1)) * t_pad + j];
  }

  // fold the key-range splits into the attention output
  const int PS = D + 2;
  for (int idx = tid; idx < (sc.skip_fold ? 0 : nrep * D); idx += kSNT) {
    const int r = idx / D, d = idx % D;
    sc.out[(hq0 + r) * D + d] = __float2half(ekv_fold_partials_auto(sc.partials + ((hq0 + r) * sc.n_split) * PS, sc.n_split, PS, d));
  }
  __syncthreads();   // LDS-DMA complete (vmcnt(0) before the barrier) and visible
  EKV_STAMP(1);
  uint32_t* s_hist = reinterpret_cast<uint32_t*>(red.buf + 2 * kSNW * 8);     // roco select scratch: histogram, candidate list
  unsigned long long* s_list = reinterpret_cast<unsigned long long*>(s_hist + 264);
  ekv_decode_tail<REP, ITEMS, kSNW>(sc, ll, h, head_row, T, off, W, s_logit, t_pad, sS, sQ, sC, red, s_hist, s_list, kSNT, nullptr, 0, 0, nrep);
}

// Partials of the key-range splits -> fp16 attention output, nothing else (rows = q_len * n_q_heads per layer).
__global__ void __launch_bounds__(128) ekv_fold_kernel(const EkvScoreArgs sc) {
  const int D = sc.head_dim, PS = D + 2;
  const size_t row = (size_t)blockIdx.y * sc.n_q_heads * sc.q_len + blockIdx.x;
  const float* p0 = sc.partials + row * sc.n_split * PS;
  // (ekv_step.out_*_stride: blockIdx.x = head * q_len + token)
  __half* orow = sc.out + (size_t)blockIdx.y * sc.n_q_heads * sc.q_len * D + (size_t)(blockIdx.x / sc.q_len) * sc.o_hs + (size_t)(blockIdx.x % sc.q_len) * sc.o_ts;
  for (int d = threadIdx.x; d < D; d += 128) orow[d] = __float2half(ekv_fold_partials_auto(p0, sc.n_split, PS, d));
}

size_t score_lds(int rep, int t_pad, int policy) {
  const size_t n_state = policy == EKV_POLICY_ROCO ? 3 : 1;
  return ((size_t)rep * t_pad + n_state * ekv_align((size_t)t_pad, 256)) * 4 + 2 * kSNW * 8 * 8 + 264 * 4 + kSNT * 8;
}

template <int REP, int ITEMS>
hipError_t launch_k(const EkvScoreArgs& sc, int layer_count, hipStream_t s) {
  const size_t lds = score_lds(REP, sc.t_pad, sc.policy);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ekv_decode_score_kernel<REP, ITEMS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((ekv_decode_score_kernel<REP, ITEMS>), dim3(sc.n_kv_heads, layer_count), dim3(kSNT), lds, s, sc);
  return hipGetLastError();
}

template <int REP>
hipError_t launch_rep(const EkvScoreArgs& sc, int layer_count, hipStream_t s) {
  // ITEMS = ceil(row width / threads) (a floor here sent T = 2049 to the 6144-wide build: 12 items per thread instead of 5)
  constexpr int I0 = (2304 + kSNT - 1) / kSNT, I1 = (6144 + kSNT - 1) / kSNT;
  return sc.n_slots <= kSNT * I0 ? launch_k<REP, I0>(sc, layer_count, s) : launch_k<REP, I1>(sc, layer_count, s);
}

}  // namespace

bool ekv_decode_score_supported(const EkvScoreArgs& sc) {
  const int rep = sc.n_q_heads / sc.n_kv_heads;
  if (sc.q_len != 1 || sc.n_evict > 1 || (sc.cap & 3) != 0 || sc.n_slots > 256 * 24) return false;
  if (rep < 1 || rep > 8) return false;      // (wider GQA factors: the generic scorer)
  return score_lds(rep <= 2 ? rep : (rep <= 4 ? 4 : 8), sc.t_pad, sc.policy) <= 150 * 1024;
}

hipError_t ekv_launch_fold(const EkvScoreArgs& sc, int layer_count, hipStream_t s) {
  hipLaunchKernelGGL(ekv_fold_kernel, dim3(sc.n_q_heads * sc.q_len, layer_count), dim3(128), 0, s, sc);
  return hipGetLastError();
}

hipError_t ekv_launch_decode_score(const EkvScoreArgs& sc, int layer_count, hipStream_t s) {
  switch (sc.n_q_heads / sc.n_kv_heads) {
    case 1: return launch_rep<1>(sc, layer_count, s);
    case 2: return launch_rep<2>(sc, layer_count, s);
    case 3: case 4: return launch_rep<4>(sc, layer_count, s);
    case 5: case 6: case 7: case 8: return launch_rep<8>(sc, layer_count, s);
    default: return hipErrorInvalidValue;
  }
}
