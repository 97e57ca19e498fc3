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
  // block-wide minimum of two independent uint32 values in one barrier
  __device__ __forceinline__ void min2_u32(uint32_t& a, uint32_t& b) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a = min(a, (uint32_t)__shfl_xor((int)a, off, 64));
      b = min(b, (uint32_t)__shfl_xor((int)b, off, 64));
    }
    uint32_t* r = reinterpret_cast<uint32_t*>(slot());
    if (lane == 0) {
      r[wave * 2] = a;
      r[wave * 2 + 1] = b;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      a = min(a, r[w * 2]);
      b = min(b, r[w * 2 + 1]);
    }
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
// The ragged end of the score rows (the last W % 256 columns) by plain loads.  Separate from the DMA part so that a caller can
// issue the DMA early and keep these loads (whose LDS stores wait for them) out of the way of its own first loads.
template <int NW = 4>
__device__ __forceinline__ void ekv_tail_prefetch_ragged(const EkvScoreArgs& sc, size_t head_row, int W, bool roco, float* sS,
                                                         float* sQ, float* sC, bool with_cnt = true) {
  const int tid = threadIdx.x;
  for (int j = W / 256 * 256 + tid; j < W; j += 64 * NW) {
    sS[j] = sc.score_sum[head_row + j];
    if (roco) {
      sQ[j] = sc.score_sq[head_row + j];
      if (with_cnt) sC[j] = sc.score_cnt[head_row + j];
    }
  }
}

template <int NW = 4, bool RAGGED = true>
__device__ __forceinline__ void ekv_tail_prefetch_rows(const EkvScoreArgs& sc, size_t head_row, int W, int w_pad, bool roco,
                                                       float* sS, float* sQ, float* sC, bool with_cnt = true) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_arr = roco ? (with_cnt ? 3 : 2) : 1;
  const int full = W / 256;
  for (int c = wave; c < full * n_arr; c += NW) {
    const int arr = c / full, ch = c % full;
    const float* src = (arr == 0 ? sc.score_sum : arr == 1 ? sc.score_sq : sc.score_cnt) + head_row + ch * 256 + lane * 4;
    float* dst = sS + (size_t)arr * w_pad + ch * 256;     // wave-uniform base; lane i lands at +16*i bytes
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
  if (RAGGED) ekv_tail_prefetch_ragged<NW>(sc, head_row, W, roco, sS, sQ, sC, with_cnt);
}

// PHYS: s_logit is indexed by PHYSICAL row (the fused kernel streamed the rows in address order); the logit of position
// j sits at s_logit[slot_of_pos[j]].  The passes below still run in position order with the same thread <-> column
// mapping, so sums are formed in exactly the order of the position-indexed variant (bit-identical scores).
template <int REP, int ITEMS, int NW = 4, bool PHYS = false>
__device__ __forceinline__ void ekv_decode_tail(const EkvScoreArgs& sc, int ll, int h, size_t head_row, int T, int off, int W,
                                                float* s_logit, int t_pad, float* sS, float* sQ, float* sC, RedN<NW>& red,
                                                uint32_t* s_hist, unsigned long long* s_list, int list_cap,
                                                const float* part_max = nullptr, int n_part = 0, int part_stride = 0, int nrep_in = 0) {
  const int tid = threadIdx.x;
  constexpr int NT = 64 * NW;
  // GQA factors 3 / 5 / 6 / 7 run the REP = 4 / 8 build: the logit rows r >= nrep are copies of the last real head and stay out of
  // the mean over the group (easykv/easykv.py:188-196 averages the rep real heads)
  const int nrep = (REP == 1 || REP == 2 || nrep_in <= 0) ? REP : nrep_in;
  // cell[it] = physical row of position tid + it * NT, loaded once up front: PHYS reads the logits through it, and the
  // slot-map shift at the end writes straight from these registers (no dependent re-read + barrier at the very end)
  int cell[ITEMS];
  bool have_cells = false;
  if (!PHYS && sc.n_evict == 1) {
    const int32_t* map0 = sc.slot_of_pos + head_row;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) cell[it] = map0[min(tid + it * NT, T - 1)];
    have_cells = true;
  }
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
    // PHYS: cell of position j (this thread's columns j = tid + NT*it) and of position off + j, loaded together
    // (unconditional, clamped: predicated loads get serialised by hipcc)
    const int32_t* map_c = sc.slot_of_pos + head_row;
    int cell_off[PHYS ? ITEMS : 1];
    have_cells = PHYS;
    if (PHYS) {
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) cell[it] = map_c[min(tid + it * NT, T - 1)];
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) cell_off[it] = off == 0 ? cell[it] : map_c[min(off + tid + it * NT, T - 1)];
    }
    if (part_max != nullptr) {
      // the row maximum is already known: the streaming loop kept a running max per wave (same floats, max is exact)
#pragma unroll
      for (int r = 0; r < REP; ++r)
        for (int i = 0; i < n_part; ++i) mx[r] = fmaxf(mx[r], part_max[(size_t)(i * REP + r) * part_stride]);
    } else {
      if (PHYS) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
          if (tid + it * NT < T) {
#pragma unroll
            for (int r = 0; r < REP; ++r) mx[r] = fmaxf(mx[r], s_logit[(size_t)r * t_pad + cell[it]]);
          }
        }
      } else {
#pragma unroll 4
        for (int j = tid; j < T; j += NT) {
#pragma unroll
          for (int r = 0; r < REP; ++r) mx[r] = fmaxf(mx[r], s_logit[(size_t)r * t_pad + j]);
        }
      }
      red.template max_n<REP>(mx);
    }
    if (PHYS) {
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        if (tid + it * NT < T) {
#pragma unroll
          for (int r = 0; r < REP; ++r) {
            const float e = expf(s_logit[(size_t)r * t_pad + cell[it]] - mx[r]);
            s_logit[(size_t)r * t_pad + cell[it]] = e;
            sm[r] += e;
          }
        }
      }
    } else {
#pragma unroll 4
      for (int j = tid; j < T; j += NT) {      // e = exp(x - max) once: it replaces the logit in LDS
#pragma unroll
        for (int r = 0; r < REP; ++r) {
          const float e = expf(s_logit[(size_t)r * t_pad + j] - mx[r]);
          s_logit[(size_t)r * t_pad + j] = e;
          sm[r] += e;
        }
      }
    }
    red.template sum_n<REP>(sm);
    float inv_sm[REP];     // p = e * (1 / sum): see ekv_score_select.inc on why not a division per element
#pragma unroll
    for (int r = 0; r < REP; ++r) inv_sm[r] = 1.f / sm[r];
    // off + j == a column this thread wrote itself only when off % NT == 0; otherwise wait for the other writers
    if ((off % NT) != 0) __syncthreads();
    if (PHYS) {
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const int j = tid + it * NT;
        if (j < W) {
          float pb = 0.f;
#pragma unroll
          for (int r = 0; r < REP; ++r)
            if (r < nrep) pb += s_logit[(size_t)r * t_pad + cell_off[it]] * inv_sm[r];
          if (REP > 1) pb = pb / (float)nrep;
          if (sc.policy == EKV_POLICY_TOVA) {
            sS[j] = pb;
          } else {
            sS[j] += pb;
            if (roco) sQ[j] += pb * pb;
          }
        }
      }
    } else {
#pragma unroll 4
      for (int j = tid; j < W; j += NT) {
        float pb = 0.f;
#pragma unroll
        for (int r = 0; r < REP; ++r)
          if (r < nrep) pb += s_logit[(size_t)r * t_pad + off + j] * inv_sm[r];
        if (REP > 1) pb = pb / (float)nrep;
        if (sc.policy == EKV_POLICY_TOVA) {
          sS[j] = pb;
        } else {
          sS[j] += pb;
          if (roco) sQ[j] += pb * pb;
        }
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
      if (PHYS || (off % NT) != 0) __syncthreads();   // all e's consumed before their cells are reused
  #pragma unroll 4
    for (int j = tid; j < W; j += NT) {
        const float c = sC[j] + sc.count_add;
        sC[j] = c;
        const float mean = sS[j] / c;
        float sd = sqrtf(sQ[j] / c - mean * mean);
        if (j >= W - sc.roco_tail || j < sc.win_lo) sd = 1e9f;
        kstd[j] = ekv_fkey(sd);
      }
      // victim = argmin mean over F = {the k1 smallest std, ties to the lower index} (easykv/easykv.py:319-326).
      // F = {j : (key_j, j) < thr} where thr - 1 is the element of rank k1 - 1 in (key, index) order.  (An earlier version
      // walked the candidates in increasing mean and tested each one's std rank; in steady state that always fails: the
      // policy itself keeps the low-mean tokens that are NOT in F alive, so the bottom of the mean order fills up with them
      // and every step fell through to a 32-pass bisection, 65 us per workgroup after ~1000 steps.)
      // Exact select in five barriers: range of the real keys -> 256-bin histogram -> the bin holding rank k1 - 1 ->
      // its (<= list_cap) members ranked against each other.
      const uint32_t kSent = ekv_fkey(1e9f);   // the 1e9 sentinels and NaN (0xFFFFFFFF) sort above every real std
      unsigned long long thr = 0;              // exclusive threshold on (key << 32 | j); 0 = not found yet
      {
        uint32_t kmin = ~0u, nmax = ~0u;       // nmax = ~max
  #pragma unroll 4
        for (int j = tid; j < W; j += NT) {
          const uint32_t k = kstd[j];
          if (k < kSent) {
            kmin = min(kmin, k);
            nmax = min(nmax, ~k);
          }
        }
        if (tid < 256) s_hist[tid] = 0;
        if (tid < 8) s_hist[256 + tid] = 0;    // [256] members listed so far, [257] bin, [258] below, [260..261] result
        red.min2_u32(kmin, nmax);              // (its barrier also publishes the zeroed histogram)
        const uint32_t kmax = ~nmax;
        // Up to three levels: when the threshold bin holds more members than the list (broad score distributions: a dense cluster
        // of stds inside a range that a few outliers stretch — 563 members in the median head after 9 k steps on keys with log-normal
        // norms, and the 32-pass bisection below ran in 3 heads of 4), the histogram is repeated over THAT bin's key range with the
        // rank reduced by what lies below it.  The sentinel bin (1e9, NaN) cannot be refined by key: it falls through.
        uint32_t lo = kmin, hi = kmax, k_rem = (uint32_t)sc.roco_k1, below_acc = 0;
        for (int level = 0; level < 3 && thr == 0 && lo <= hi; ++level) {
          const uint32_t range = hi - lo;
          const int shift = range < 256u ? 0 : 24 - __clz(range);     // range >> shift < 256
          // 256 = not in this level's range (level 0: every key is, the sentinels in bin 255)
          auto bin_of = [&](uint32_t k) {
            if (level == 0) return k >= kSent ? 255u : min(255u, (k - lo) >> shift);
            return (k < lo || k > hi) ? 256u : min(255u, (k - lo) >> shift);
          };
          if (level > 0) {
            __syncthreads();                     // (everybody has read the previous level's header)
            if (tid < 256) s_hist[tid] = 0;
            if (tid < 8) s_hist[256 + tid] = 0;
            __syncthreads();
          }
  #pragma unroll 4
          for (int j = tid; j < W; j += NT) {
            const uint32_t bn = bin_of(kstd[j]);
            if (bn < 256u) atomicAdd(&s_hist[bn], 1u);
          }
          __syncthreads();
          if (tid < 64) {                      // wave 0: lane l owns bins 4l .. 4l+3; inclusive scan over the lanes
            const uint32_t c0 = s_hist[4 * tid], c1 = s_hist[4 * tid + 1], c2 = s_hist[4 * tid + 2], c3 = s_hist[4 * tid + 3];
            uint32_t incl = c0 + c1 + c2 + c3;
            const uint32_t mine = incl;
  #pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
              const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
              if (tid >= o) incl += up;
            }
            const uint32_t excl = incl - mine, k1 = k_rem;
            if (excl < k1 && k1 <= incl) {     // exactly one lane: the bin where the cumulative count reaches the rank
              uint32_t below = excl, b = 4 * tid;
              if (below + c0 < k1) { below += c0; ++b;
                if (below + c1 < k1) { below += c1; ++b;
                  if (below + c2 < k1) { below += c2; ++b; } } }
              s_hist[257] = b;
              s_hist[258] = below;
            }
          }
          __syncthreads();
          const uint32_t b_sel = s_hist[257], below = s_hist[258], in_bin = s_hist[b_sel];
          if ((int)in_bin <= list_cap && (int)in_bin <= NT) {
  #pragma unroll 4
            for (int j = tid; j < W; j += NT) {
              const uint32_t k = kstd[j];
              if (bin_of(k) == b_sel) s_list[atomicAdd(&s_hist[256], 1u)] = ((unsigned long long)k << 32) | (uint32_t)j;
            }
            __syncthreads();
            if (tid < (int)in_bin) {
              const unsigned long long e = s_list[tid];
              uint32_t rank = 0;
              for (int i = 0; i < (int)in_bin; ++i) rank += s_list[i] < e ? 1u : 0u;
              if (below_acc + below + rank == (uint32_t)sc.roco_k1 - 1u) *reinterpret_cast<unsigned long long*>(s_hist + 260) = e + 1ull;
            }
            __syncthreads();
            thr = *reinterpret_cast<const unsigned long long*>(s_hist + 260);
            break;
          }
          if ((level == 0 && b_sel == 255u) || shift == 0) break;      // sentinels / one key value: not a matter of key ranges
          below_acc += below;
          k_rem -= below;
          const uint32_t nlo = lo + (b_sel << shift);
          hi = b_sel == 255u ? hi : min(hi, nlo + ((1u << shift) - 1u));
          lo = nlo;
        }
      }
      if (thr == 0) {
        // fallback (threshold bin too crowded, e.g. all keys equal in the first steps; or no real key at all): explicit
        // k1-select on the std keys by bitwise bisection
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
        thr = ((unsigned long long)tau << 32) | (uint32_t)bound;
      }
      {
        unsigned long long best = ~0ull;
    #pragma unroll 4
    for (int j = tid; j < W; j += NT) {
          const bool feas = (((unsigned long long)kstd[j] << 32) | (uint32_t)j) < thr;
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
    if (have_cells) {
      // every thread still holds the map entries of its positions: nothing to read.  Scored policies have passed block
      // reductions since cell[] was loaded; 'recency' / 'random' have passed none, and a thread's store lands on the entry a
      // neighbouring wave loads as its own cell: make sure every load has happened
      if (sc.policy == EKV_POLICY_RANGE) __syncthreads();
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const int p = tid + it * NT;
        if (p >= pv && p < T) map[p == pv ? T - 1 : p - 1] = cell[it];
      }
    } else {
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
}

// ---------------------------------------------------------------------------------------------------------------------
// Slot-indexed score rows (ekv_step.phases & EKV_PHASE_SLOT_ROWS; fused decode step, plain keys, at most one victim).
//
// In the ordered layout above the score rows and the slot map are indexed by ORDER (age rank): an eviction shifts every entry
// behind the victim, so a decode step rewrites S, Q, C and half the slot map — 28 KB per head and step next to 24 KB read — and
// written bytes cost about twice what read bytes do in the middle of a read stream (measured, ekv_chunk_lds.inc).  Here the rows
// are indexed by PHYSICAL row, like K / V and like the logits the stream leaves in LDS:
//   S[row], Q[row]      accumulated scores (easykv/easykv.py:287-300), rewritten every step (they change)
//   C0[row]             count base, written once when the row is (re)used: count = C0[row] + g, g = the head's running sum of
//                       count_add (easykv.py:304 adds the same number to every entry: exact while counts are integers < 2^24)
//   birth[row]          order key, written once: the age rank of the ordered layout is the rank of `birth` among the live rows
//   state[head]         (g, next birth)
// An eviction moves nothing: the victim's row goes to the front of the free list (slot_of_pos[T - 1], one word) and its S / Q / C0 /
// birth die with it.  Ties go to the lower birth = the lower order index (same decisions as the ordered layout); windows that the
// ordered layout states as index ranges — the newest `tail` entries — are birth thresholds, exact by a counting check with a
// bisection fallback.  ekv_rows_to_slots / ekv_rows_to_order convert between the layouts (ekv_abi.hip).
// Thread t owns rows t, t + NT, ...: every pass touches its own columns only, so the passes need no barrier between them.
// S / Q are staged in LDS by LDS-DMA under the stream (like the ordered layout's rows); C0 and birth come straight into registers
// (`cB`, `cC`: loads issued by the caller when its stream ends, consumed after the softmax passes) — a third and fourth LDS row
// would cost the fourth workgroup per CU.
template <int REP, int ITEMS, int NW>
__device__ __forceinline__ void ekv_decode_tail_slot(const EkvScoreArgs& sc, int ll, int h, size_t head_row, int T, int E,
                                                     float* s_logit, int l_pad, float* sS, float* sQ, const float (&cC)[ITEMS],
                                                     const int32_t (&cB)[ITEMS], RedN<NW>& red, uint32_t* s_hist,
                                                     unsigned long long* s_list, int list_cap, const float* part_max, int n_part,
                                                     int part_stride, const uint32_t* s_deadw, int new_row, float g_old, float c_new,
                                                     int nb, int nrep_in = 0) {
  const int tid = threadIdx.x;
  constexpr int NT = 64 * NW;
  const int nrep = (REP == 1 || REP == 2 || nrep_in <= 0) ? REP : nrep_in;      // (GQA factor 3 on the REP = 4 build: see ekv_decode_tail)
  const bool roco = sc.policy == EKV_POLICY_ROCO;
  const float g_new = g_old + sc.count_add;
  // this thread's rows: j = tid + it * NT; bit `it` of livem = the row holds a live entry (the appended row included)
  unsigned livem = 0;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int j = tid + it * NT;
    const bool lv = j < E && (j == new_row || !((s_deadw[j >> 5] >> (j & 31)) & 1u));
    livem |= (lv ? 1u : 0u) << it;
  }
  auto lives = [&](int it) { return (livem >> it) & 1u; };
  // select scratch: histogram bins [0, 256), [256] listed so far, [257] bin, [258] below, [260..261] threshold, [262] victim row —
  // zeroed here, published by the softmax reduction's barrier (slot-layout steps always accumulate)
  if (tid < 256) s_hist[tid] = 0;
  if (tid < 8) s_hist[256 + tid] = 0;
  // ---- exact softmax over the live rows, GQA fold, accumulate (easykv/easykv.py:271-300); S / Q go back to HBM right away ----
  if (sc.accumulate) {
    float mx[REP], sm[REP];
#pragma unroll
    for (int r = 0; r < REP; ++r) mx[r] = EKV_NEG_INF, sm[r] = 0.f;
#pragma unroll
    for (int r = 0; r < REP; ++r)
      for (int i = 0; i < n_part; ++i) mx[r] = fmaxf(mx[r], part_max[(size_t)(i * REP + r) * part_stride]);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int j = tid + it * NT;
      if (j < E) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
          const float e = lives(it) ? expf(s_logit[(size_t)r * l_pad + j] - mx[r]) : 0.f;
          s_logit[(size_t)r * l_pad + j] = e;
          sm[r] += e;
        }
      }
    }
    red.template sum_n<REP>(sm);
    float inv_sm[REP];
#pragma unroll
    for (int r = 0; r < REP; ++r) inv_sm[r] = 1.f / sm[r];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int j = tid + it * NT;
      if (lives(it)) {
        float pb = 0.f;
#pragma unroll
        for (int r = 0; r < REP; ++r)
          if (r < nrep) pb += s_logit[(size_t)r * l_pad + j] * inv_sm[r];
        if (REP > 1) pb = pb / (float)nrep;
        const bool fresh = j == new_row || sc.policy == EKV_POLICY_TOVA;
        const float s_new = fresh ? pb : sS[j] + pb;
        sS[j] = s_new;
        sc.score_sum[head_row + j] = s_new;
        if (roco) {
          const float q_new = j == new_row ? pb * pb : sQ[j] + pb * pb;
          sQ[j] = q_new;
          sc.score_sq[head_row + j] = q_new;
        }
      }
    }
  }
  const size_t head = (size_t)(sc.layer_begin + ll) * sc.n_kv_heads + h;
  if (tid == 0) {      // the appended row's count base (c_new: what the ordered row holds behind its live entries, minus g) and birth
    sc.score_cnt[head_row + new_row] = c_new;
    sc.birth[head_row + new_row] = nb;
    sc.slot_state[4 * head] = g_new;
    reinterpret_cast<int32_t*>(sc.slot_state)[4 * head + 1] = nb + 1;
  }
  if (sc.n_evict != 1) return;

  // count base and birth of this thread's rows (the caller's loads are first waited for HERE, behind the softmax passes)
  float rC[ITEMS];
  int32_t rB[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int j = tid + it * NT;
    rC[it] = j == new_row ? c_new : cC[it];      // (a recycled row still holds its last owner's count base and birth)
    rB[it] = j == new_row ? nb : cB[it];
  }
  // ---- the newest `tail` entries are protected: birth > nb - tail when none of them was ever evicted (counting check) ----
  const int tail = roco ? sc.roco_tail : sc.win_tail;
  int b_prot = nb - tail;                       // protected <=> birth > b_prot
  if (tail > 0 && !sc.slot_tail_ok) {           // (slot_tail_ok: the caller vouches for it — EKV_PHASE_SLOT_TAIL_OK — and the check's reduction is saved)
    int c = 0;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) c += (lives(it) && rB[it] > b_prot) ? 1 : 0;
    if (red.sum_int(c) != min(tail, T)) {       // rare (the window grew since those entries were appended): exact threshold by bisection
      int lo = -1, hi = nb;                     // largest b with #{birth > b} >= min(tail, T)
      while (lo < hi) {
        const int mid = lo + (hi - lo + 1) / 2;
        int c2 = 0;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) c2 += (lives(it) && rB[it] > mid) ? 1 : 0;
        if (red.sum_int(c2) >= min(tail, T)) lo = mid; else hi = mid - 1;
      }
      b_prot = lo;
    }
  }
  unsigned long long best = ~0ull;              // (order-preserving key of the victim criterion) << 32 | birth
  if (roco) {
    const uint32_t kSent = ekv_fkey(1e9f);
    uint32_t kstd[ITEMS];                       // std keys; rows that are not live rank after everything (composite ~0)
    uint32_t* kmean = reinterpret_cast<uint32_t*>(s_logit);
    auto comp = [&](int it) { return lives(it) ? (((unsigned long long)kstd[it] << 32) | (uint32_t)rB[it]) : ~0ull; };
    uint32_t kmin = ~0u, nmax = ~0u;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int j = tid + it * NT;
      uint32_t key = ~0u;
      if (lives(it)) {
        const float c = rC[it] + g_new;
        const float mean = sS[j] / c;          // (IEEE divisions: a shared refined reciprocal without the v_div_scale / v_div_fixup wrapper
        float sd = sqrtf(sQ[j] / c - mean * mean);   //  is bit-identical only while numerator * 2^-24 stays normal — sums of p^2 do not)
        if (rB[it] > b_prot) sd = 1e9f;
        key = ekv_fkey(sd);
        kmean[j] = ekv_fkey(mean);               // for the arg-min over the feasible set below (own column of the dead e row)
        if (key < kSent) {
          kmin = min(kmin, key);
          nmax = min(nmax, ~key);
        }
      }
      kstd[it] = key;
    }
    unsigned long long thr = 0;
    // Warm start: the threshold key of the previous step (slot_state[head][2]).  A step adds one probability to every column, so the
    // k1-th smallest std moves by a fraction of a percent: count the keys below a narrow window around the hint and list the window's
    // members in the SAME pass — one reduction and one ranking barrier instead of range + histogram + scan + list + ranking.  Exact:
    // when the window misses (first step, a jump) the full select below runs.
    const uint32_t hint = reinterpret_cast<const uint32_t*>(sc.slot_state)[4 * head + 2];
    if (hint != 0u) {
#ifndef EKV_SLOT_WARM_WIN
#define EKV_SLOT_WARM_WIN 15
#endif
      constexpr uint32_t kWin = 1u << EKV_SLOT_WARM_WIN;      // +- 2^15 ulps = +- 0.4 % of the key: a handful of the ~2000 columns
      const uint32_t lo = hint > kWin ? hint - kWin : 0u, hi = hint < kSent - kWin ? hint + kWin : kSent - 1u;
      int below = 0;                           // (the list counter was zeroed at the top of the tail, behind the softmax reduction's barrier)
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        if (lives(it)) {
          below += kstd[it] < lo ? 1 : 0;
          if (kstd[it] >= lo && kstd[it] <= hi) {
            const uint32_t at = atomicAdd(&s_hist[256], 1u);
            if ((int)at < list_cap) s_list[at] = comp(it);
          }
        }
      }
      below = red.sum_int(below);
      const int in_win = (int)s_hist[256];
      if (below < sc.roco_k1 && sc.roco_k1 <= below + in_win && in_win <= list_cap && in_win <= NT) {
        if (tid < in_win) {
          const unsigned long long e = s_list[tid];
          uint32_t rank = 0;
          for (int i = 0; i < in_win; ++i) rank += s_list[i] < e ? 1u : 0u;
          if (below + (int)rank == sc.roco_k1 - 1) *reinterpret_cast<unsigned long long*>(s_hist + 260) = e + 1ull;
        }
        __syncthreads();
        thr = *reinterpret_cast<const unsigned long long*>(s_hist + 260);
      } else {
        __syncthreads();                       // (everybody has read the counter)
        if (tid == 0) s_hist[256] = 0;
      }
    }
    // Full select (no hint yet, or the window missed): 256 bins of equal key width over the range of the real keys, the bin holding
    // rank k1 - 1, its members ranked — and when that bin holds more members than the list (broad score distributions: a dense cluster
    // of stds inside a range a few outliers stretch), the histogram again over THAT bin's key range, up to three levels (see the
    // ordered tail).  (Bins cut at 64 sampled composites — ekv_pivot_threshold — were equally robust, 195 -> 189.5 us on keys with
    // log-normal norms, but cost i.i.d. keys 1-3 us while a fresh state settles: every miss paid the pivot ordering and seven-step
    // searches.)
    if (thr == 0) {
      if (tid < 4) s_hist[256 + tid] = 0;    // (bins [0, 256) are still zero from the top of the tail; [260..262]: the result and the victim row stay)
      red.min2_u32(kmin, nmax);              // (its barrier also publishes the zeroed counters)
      const uint32_t kmax = ~nmax;
      uint32_t lo = kmin, hi = kmax, k_rem = (uint32_t)sc.roco_k1, below_acc = 0;
      for (int level = 0; level < 3 && thr == 0 && lo <= hi; ++level) {
        const uint32_t range = hi - lo;
        const int shift = range < 256u ? 0 : 24 - __clz(range);
        auto bin_of = [&](uint32_t k) {      // 256 = not in this level's range (level 0: every live key is, the sentinels in bin 255)
          if (level == 0) return k >= kSent ? 255u : min(255u, (k - lo) >> shift);
          return (k < lo || k > hi) ? 256u : min(255u, (k - lo) >> shift);
        };
        if (level > 0) {
          __syncthreads();                   // (everybody has read the previous level's header)
          if (tid < 256) s_hist[tid] = 0;
          if (tid < 4) s_hist[256 + tid] = 0;
          __syncthreads();
        }
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
          const uint32_t bn = lives(it) ? bin_of(kstd[it]) : 256u;
          if (bn < 256u) atomicAdd(&s_hist[bn], 1u);
        }
        __syncthreads();
        if (tid < 64) {
          const uint32_t c0 = s_hist[4 * tid], c1 = s_hist[4 * tid + 1], c2 = s_hist[4 * tid + 2], c3 = s_hist[4 * tid + 3];
          uint32_t incl = c0 + c1 + c2 + c3;
          const uint32_t mine = incl;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
            if (tid >= o) incl += up;
          }
          const uint32_t excl = incl - mine, k1 = k_rem;
          if (excl < k1 && k1 <= incl) {
            uint32_t below = excl, bb = 4 * tid;
            if (below + c0 < k1) { below += c0; ++bb;
              if (below + c1 < k1) { below += c1; ++bb;
                if (below + c2 < k1) { below += c2; ++bb; } } }
            s_hist[257] = bb;
            s_hist[258] = below;
          }
        }
        __syncthreads();
        const uint32_t b_sel = s_hist[257], below = s_hist[258], in_bin = s_hist[b_sel];
        if ((int)in_bin <= list_cap && (int)in_bin <= NT) {
#pragma unroll
          for (int it = 0; it < ITEMS; ++it)
            if (lives(it) && bin_of(kstd[it]) == b_sel) s_list[atomicAdd(&s_hist[256], 1u)] = comp(it);
          __syncthreads();
          if (tid < (int)in_bin) {
            const unsigned long long e = s_list[tid];
            uint32_t rank = 0;
            for (int i = 0; i < (int)in_bin; ++i) rank += s_list[i] < e ? 1u : 0u;
            if (below_acc + below + rank == (uint32_t)sc.roco_k1 - 1u) *reinterpret_cast<unsigned long long*>(s_hist + 260) = e + 1ull;
          }
          __syncthreads();
          thr = *reinterpret_cast<const unsigned long long*>(s_hist + 260);
          break;
        }
        if ((level == 0 && b_sel == 255u) || shift == 0) break;
        below_acc += below;
        k_rem -= below;
        const uint32_t nlo = lo + (b_sel << shift);
        hi = b_sel == 255u ? hi : min(hi, nlo + ((1u << shift) - 1u));
        lo = nlo;
      }
    }
    if (thr == 0) {      // fallback: bitwise bisection on the keys, ties at the threshold to the lower births
      uint32_t tau = 0;
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t t = tau | (1u << bit);
        int c = 0;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) c += (lives(it) && kstd[it] < t) ? 1 : 0;
        if (red.sum_int(c) < sc.roco_k1) tau = t;
      }
      int c_less = 0, c_eq = 0;
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        c_less += (lives(it) && kstd[it] < tau) ? 1 : 0;
        c_eq += (lives(it) && kstd[it] == tau) ? 1 : 0;
      }
      const int need = sc.roco_k1 - red.sum_int(c_less);
      long long bound = (long long)nb + 1;      // exclusive bound on the birth of the tied keys that still belong to F
      if (red.sum_int(c_eq) != need) {
        long long lo = 0, hi = (long long)nb + 1;
        while (lo < hi) {
          const long long mid = (lo + hi) >> 1;
          int c = 0;
#pragma unroll
          for (int it = 0; it < ITEMS; ++it) c += (lives(it) && kstd[it] == tau && (long long)rB[it] < mid) ? 1 : 0;
          if (red.sum_int(c) >= need) hi = mid; else lo = mid + 1;
        }
        bound = lo;
      }
      thr = ((unsigned long long)tau << 32) | (uint32_t)bound;
    }
    if (tid == 0) {      // hint for the next step: the key of the k1-th smallest composite (0 = none: sentinel keys make no useful pivot)
      const uint32_t kt = (uint32_t)((thr - 1ull) >> 32);
      reinterpret_cast<uint32_t*>(sc.slot_state)[4 * head + 2] = (thr != 0 && kt < kSent && kt != 0u) ? kt : 0u;
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int j = tid + it * NT;
      if (lives(it) && comp(it) < thr) {
        const unsigned long long x = ((unsigned long long)kmean[j] << 32) | (uint32_t)rB[it];
        best = x < best ? x : best;
      }
    }
  } else {             // h2o_head / tova: argmin of the accumulated score outside the protected tail (win_lo == 0 in this layout)
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int j = tid + it * NT;
      if (lives(it) && rB[it] <= b_prot) {
        const unsigned long long x = ((unsigned long long)ekv_fkey(sS[j]) << 32) | (uint32_t)rB[it];
        best = x < best ? x : best;
      }
    }
  }
  best = red.min_u64(best);
  // the victim's row: births are unique, its owner publishes the row; its order index (what the reference reports) = the number of
  // live rows born before it
  const int b_v = (int)(best & 0xFFFFFFFFu);
  int n_before = 0;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    if (lives(it)) {
      if (rB[it] == b_v) s_hist[262] = (uint32_t)(tid + it * NT);
      n_before += rB[it] < b_v ? 1 : 0;
    }
  }
  if (sc.evict_ids != nullptr) n_before = red.sum_int(n_before); else __syncthreads();
  if (tid == 0) {
    if (sc.evict_ids != nullptr) sc.evict_ids[(size_t)ll * sc.n_kv_heads + h] = n_before;
    sc.slot_of_pos[head_row + T - 1] = (int32_t)s_hist[262];     // front of the free list: the next append recycles the victim's row
    sc.cnt_tail[head_row + T - 1] = 0.f;                         // (the ordered step leaves a zero count behind the live entries)
  }
}
