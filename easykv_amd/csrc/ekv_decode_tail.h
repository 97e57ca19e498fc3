// The scorer tail of a decode step (shared by the fused kernel and the split-path decode scorer).
//
// Inputs live in LDS: s_logit[REP][t_pad] (raw logits q.k/sqrt(D) of all T positions), sS/sQ/sC (the head's score
// rows, index j <-> position off + j).  Does: exact softmax -> GQA mean -> accumulate (easykv/easykv.py:271-300),
// victim selection for at most one victim (:310-347, :711-747), write-back of the rows compacted past the victim
// (:315-333) and the slot-map shift that recycles the victim's K/V row (:56-68).  256 threads.
#pragma once
#include "ekv_common.h"
#include "ekv_kernels.h"

#ifndef EKV_STAMP
#define EKV_STAMP(i) do { } while (0)
#endif

template <int NW>
struct RedN {  // block reductions for an NW-wave workgroup; scratch = 2 x NW x 8 x 8 bytes
  unsigned long long* buf;
  int phase, lane, wave;
  __device__ __forceinline__ unsigned long long* slot() { return buf + (phase++ & 1) * (NW * 8); }
  template <int N>
  __device__ __forceinline__ void max_n(float (&x)[N]) {
    float* r = reinterpret_cast<float*>(slot());
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float w = ekv_wave_max(x[i]);
      if (lane == 0) r[wave * 8 + i] = w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float y = r[i];
#pragma unroll
      for (int w = 1; w < NW; ++w) y = fmaxf(y, r[w * 8 + i]);
      x[i] = y;
    }
  }
  template <int N>
  __device__ __forceinline__ void sum_n(float (&x)[N]) {
    float* r = reinterpret_cast<float*>(slot());
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float w = ekv_wave_sum(x[i]);
      if (lane == 0) r[wave * 8 + i] = w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float y = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) y += r[w * 8 + i];
      x[i] = y;
    }
  }
  __device__ __forceinline__ int sum_int(int x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    int* r = reinterpret_cast<int*>(slot());
    if (lane == 0) r[wave] = x;
    __syncthreads();
    int y = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) y += r[w];
    return y;
  }
  __device__ __forceinline__ unsigned long long min_u64(unsigned long long x) {
    x = ekv_wave_min_u64(x);
    unsigned long long* r = slot();
    if (lane == 0) r[wave] = x;
    __syncthreads();
    unsigned long long y = r[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) y = r[w] < y ? r[w] : y;
    return y;
  }
};
using Red4 = RedN<4>;

// Score rows of (layer, head) -> LDS by asynchronous LDS-DMA (1 KiB per wave-instruction); the ragged end by plain
// loads.  The data is complete after the caller's next __syncthreads() (vmcnt(0) precedes the barrier).
template <int NW = 4>
__device__ __forceinline__ void ekv_tail_prefetch_rows(const EkvScoreArgs& sc, size_t head_row, int W, int w_pad, bool roco,
                                                       float* sS, float* sQ, float* sC) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_arr = roco ? 3 : 1;
  const int full = W / 256;
  for (int c = wave; c < full * n_arr; c += NW) {
    const int arr = c / full, ch = c % full;
    const float* src = (arr == 0 ? sc.score_sum : arr == 1 ? sc.score_sq : sc.score_cnt) + head_row + ch * 256 + lane * 4;
    float* dst = sS + (size_t)arr * w_pad + ch * 256;     // wave-uniform base; lane i lands at +16*i bytes
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
  for (int j = full * 256 + tid; j < W; j += 64 * NW) {
    sS[j] = sc.score_sum[head_row + j];
    if (roco) {
      sQ[j] = sc.score_sq[head_row + j];
      sC[j] = sc.score_cnt[head_row + j];
    }
  }
}

template <int REP, int ITEMS, int NW = 4>
__device__ __forceinline__ void ekv_decode_tail(const EkvScoreArgs& sc, int ll, int h, size_t head_row, int T, int off, int W,
                                                float* s_logit, int t_pad, float* sS, float* sQ, float* sC, RedN<NW>& red) {
  const int tid = threadIdx.x;
  constexpr int NT = 64 * NW;
  const bool roco = sc.policy == EKV_POLICY_ROCO;
  const bool scored = roco || sc.policy == EKV_POLICY_H2O_HEAD || sc.policy == EKV_POLICY_TOVA;
#ifdef EKV_TAIL_PROFILE
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(sc.tova_row) + ((size_t)ll * sc.n_kv_heads + h) * 8;
#endif
  EKV_STAMP(2);
  // ---- exact softmax of the row(s), GQA fold, accumulate into the LDS-resident rows (easykv/easykv.py:271-300) ----
  // (every thread only ever touches its own columns j = tid + NT*it, so no barrier is needed between the phases)
  if (scored && sc.accumulate) {
    float mx[REP], sm[REP];
#pragma unroll
    for (int r = 0; r < REP; ++r) mx[r] = EKV_NEG_INF, sm[r] = 0.f;
#pragma unroll 4
    for (int j = tid; j < T; j += NT) {
#pragma unroll
      for (int r = 0; r < REP; ++r) mx[r] = fmaxf(mx[r], s_logit[(size_t)r * t_pad + j]);
    }
    red.template max_n<REP>(mx);
#pragma unroll 4
    for (int j = tid; j < T; j += NT) {      // e = exp(x - max) once: it replaces the logit in LDS
#pragma unroll
      for (int r = 0; r < REP; ++r) {
        const float e = expf(s_logit[(size_t)r * t_pad + j] - mx[r]);
        s_logit[(size_t)r * t_pad + j] = e;
        sm[r] += e;
      }
    }
    red.template sum_n<REP>(sm);
    // off + j == a column this thread wrote itself only when off % NT == 0; otherwise wait for the other writers
    if ((off % NT) != 0) __syncthreads();
#pragma unroll 4
    for (int j = tid; j < W; j += NT) {
      float pb = 0.f;
#pragma unroll
      for (int r = 0; r < REP; ++r) pb += s_logit[(size_t)r * t_pad + off + j] / sm[r];
      if (REP > 1) pb = pb / (float)REP;
      if (sc.policy == EKV_POLICY_TOVA) {
        sS[j] = pb;
      } else {
        sS[j] += pb;
        if (roco) sQ[j] += pb * pb;
      }
    }
  }

  EKV_STAMP(3);
  // ---- victim (k <= 1) ----------------------------------------------------------------------------
  int victim = -1;
  if (sc.n_evict == 1) {
    if (sc.policy == EKV_POLICY_RANGE) {
      victim = sc.range_start;
    } else if (roco) {
      // std keys overwrite the (dead) first logit row at the thread's own columns
      uint32_t* kstd = reinterpret_cast<uint32_t*>(s_logit);
      if ((off % NT) != 0) __syncthreads();   // all e's consumed before their cells are reused
  #pragma unroll 4
    for (int j = tid; j < W; j += NT) {
        const float c = sC[j] + sc.count_add;
        sC[j] = c;
        const float mean = sS[j] / c;
        float sd = sqrtf(sQ[j] / c - mean * mean);
        if (j >= W - sc.roco_tail || j < sc.win_lo) sd = 1e9f;
        kstd[j] = ekv_fkey(sd);
      }
      // victim = argmin mean over F = {k1 smallest std}: walk candidates in increasing (mean, index) order,
      // take the first whose std rank is < k1 (three block reductions per try instead of a k-select)
      int excl[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) excl[i] = -1;
#pragma unroll
      for (int attempt = 0; attempt < 8; ++attempt) {
        if (victim >= 0) break;
        unsigned long long best = ~0ull;
    #pragma unroll 4
    for (int j = tid; j < W; j += NT) {
          bool dropped = false;
#pragma unroll
          for (int i = 0; i < 8; ++i) dropped |= (i < attempt && excl[i] == j);
          const unsigned long long x = ((unsigned long long)ekv_fkey(sS[j] / sC[j]) << 32) | (uint32_t)j;
          if (!dropped) best = x < best ? x : best;
        }
        const int cand = (int)(red.min_u64(best) & 0xFFFFFFFFu);
        const uint32_t sk = kstd[cand];          // written before the barrier inside min_u64
        int c = 0;
#pragma unroll 4
        for (int j = tid; j < W; j += NT) c += (kstd[j] < sk || (kstd[j] == sk && j < cand)) ? 1 : 0;
        if (red.sum_int(c) < sc.roco_k1) victim = cand;
        else excl[attempt] = cand;               // not feasible: drop it from the walk
      }
      if (victim < 0) {
        // fallback: explicit k1-select on the std keys (bitwise bisection), then argmin mean over the set
        uint32_t tau = 0;
        for (int bit = 31; bit >= 0; --bit) {
          const uint32_t t = tau | (1u << bit);
          int c = 0;
  #pragma unroll 4
        for (int j = tid; j < W; j += NT) c += kstd[j] < t ? 1 : 0;
          if (red.sum_int(c) < sc.roco_k1) tau = t;
        }
        int c_less = 0, c_eq = 0;
    #pragma unroll 4
    for (int j = tid; j < W; j += NT) {
          c_less += kstd[j] < tau ? 1 : 0;
          c_eq += kstd[j] == tau ? 1 : 0;
        }
        const int need = sc.roco_k1 - red.sum_int(c_less);
        int bound = W;
        if (red.sum_int(c_eq) != need) {  // ties at the threshold: lowest indices first
          int lo = 0, hi = W;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
    #pragma unroll 4
        for (int j = tid; j < W; j += NT) c += (kstd[j] == tau && j < mid) ? 1 : 0;
            if (red.sum_int(c) >= need) hi = mid; else lo = mid + 1;
          }
          bound = lo;
        }
        unsigned long long best = ~0ull;
    #pragma unroll 4
    for (int j = tid; j < W; j += NT) {
          const bool feas = kstd[j] < tau || (kstd[j] == tau && j < bound);
          const unsigned long long x = ((unsigned long long)ekv_fkey(sS[j] / sC[j]) << 32) | (uint32_t)j;
          if (feas) best = x < best ? x : best;
        }
        victim = (int)(red.min_u64(best) & 0xFFFFFFFFu);
      }
    } else if (scored) {  // h2o_head / tova: argmin of the accumulated score inside the candidate window
      unsigned long long best = ~0ull;
      for (int j = sc.win_lo + tid; j < W - sc.win_tail; j += NT) {
        const unsigned long long x = ((unsigned long long)ekv_fkey(sS[j]) << 32) | (uint32_t)j;
        best = x < best ? x : best;
      }
      victim = (int)(red.min_u64(best) & 0xFFFFFFFFu);
    }
  }

  EKV_STAMP(4);
  // ---- write back: score rows (compacted past the victim), evict id, slot map ----------------------------
  if (scored && (sc.accumulate || victim >= 0)) {
#pragma unroll 4
    for (int j = tid; j < W; j += NT) {
      if (j != victim) {
        const int d = j - ((victim >= 0 && j > victim) ? 1 : 0);
        sc.score_sum[head_row + d] = sS[j];
        if (roco) {
          sc.score_sq[head_row + d] = sQ[j];
          if (victim >= 0) sc.score_cnt[head_row + d] = sC[j];
        }
      }
    }
    if (victim >= 0 && tid == 0) {
      sc.score_sum[head_row + W - 1] = 0.f;
      if (roco) {
        sc.score_sq[head_row + W - 1] = 0.f;
        sc.score_cnt[head_row + W - 1] = 0.f;
      }
    }
  }
  if (victim >= 0) {
    if (sc.evict_ids != nullptr && tid == 0) sc.evict_ids[(size_t)ll * sc.n_kv_heads + h] = off + victim;
    // positions behind the victim move up by one; its row becomes the free tail, recycled by the next append.
    // read everything that moves, barrier, then write (the row belongs to this workgroup only)
    const int pv = off + victim;
    int32_t* map = sc.slot_of_pos + head_row;
    int moved[ITEMS + 1];
#pragma unroll
    for (int it = 0; it <= ITEMS; ++it) {
      const int p = pv + tid + it * NT;
      moved[it] = map[min(p, T - 1)];   // unconditional (clamped): predicated loads get serialised by hipcc
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it <= ITEMS; ++it) {
      const int p = pv + tid + it * NT;
      if (p < T) map[p == pv ? T - 1 : p - 1] = moved[it];
    }
  }
}
