// K3 + K4: per (layer, KV head) — exact softmax of the exported logits, GQA fold, score accumulation,
// victim selection, and order-preserving compaction of the score rows and of the slot map.
//
// Replaces, per layer (SURVEY.md probe 7 shows per-layer == all-layers-at-once):
//   GQA fold              easykv/easykv.py:188-196, :271-276
//   accumulate            easykv/easykv.py:287-300 (decode), :443-457 (prefill), :693-707 (auto)
//   count advance         easykv/easykv.py:304, :460, :708
//   select                easykv/easykv.py:310-337 (decode), :462-490 (prefill), :711-740 (auto)
//   K/V compaction        easykv/easykv.py:56-82, :105-112   -> 4-byte slot-map compaction, rows never move
//   score-row compaction  easykv/easykv.py:315-333, :465-490
// It also folds the key-range split partials of the attention kernel into the fp16 output.
//
// Selection is exact and deterministic: "k smallest" is an MSB-first radix select (8 bits per pass) on
// order-preserving uint32 keys held in LDS (no sort); ties go to the lowest index, NaN ranks largest.
// The score arithmetic mirrors the reference op for op in fp32 (IEEE div/sqrt; the library is built
// with -ffp-contract=off so q/c - (s/c)^2 is not fused).
#include "ekv_common.h"
#include "ekv_kernels.h"

namespace {

constexpr int kNT = 512;
constexpr int kNWV = kNT / 64;

struct Blk {
  int tid, lane, wave;
  unsigned long long* red;  // 2 * kNWV entries
  int phase;
};

__device__ __forceinline__ float blk_max(Blk& b, float x) {
  x = ekv_wave_max(x);
  float* r = reinterpret_cast<float*>(b.red) + (b.phase & 1) * kNWV * 2;
  if (b.lane == 0) r[b.wave] = x;
  __syncthreads();
  float y = r[0];
  for (int i = 1; i < kNWV; ++i) y = fmaxf(y, r[i]);
  b.phase++;
  return y;
}
__device__ __forceinline__ float blk_sum(Blk& b, float x) {
  x = ekv_wave_sum(x);
  float* r = reinterpret_cast<float*>(b.red) + (b.phase & 1) * kNWV * 2;
  if (b.lane == 0) r[b.wave] = x;
  __syncthreads();
  float y = 0.f;
  for (int i = 0; i < kNWV; ++i) y += r[i];
  b.phase++;
  return y;
}
__device__ __forceinline__ int blk_sum_uniform(Blk& b, int wave_value) {  // value already wave-uniform
  int* r = reinterpret_cast<int*>(b.red) + (b.phase & 1) * kNWV * 2;
  if (b.lane == 0) r[b.wave] = wave_value;
  __syncthreads();
  int y = 0;
  for (int i = 0; i < kNWV; ++i) y += r[i];
  b.phase++;
  return y;
}
__device__ __forceinline__ unsigned long long blk_min_u64(Blk& b, unsigned long long x) {
  x = ekv_wave_min_u64(x);
  unsigned long long* r = b.red + (b.phase & 1) * kNWV;
  if (b.lane == 0) r[b.wave] = x;
  __syncthreads();
  unsigned long long y = r[0];
  for (int i = 1; i < kNWV; ++i) y = r[i] < y ? r[i] : y;
  b.phase++;
  return y;
}

// 8 values at once: one barrier for 8 reductions (scratch: 2 x kNWV x 8 floats)
__device__ __forceinline__ void blk_max8(Blk& b, float (&x)[8]) {
  float* r = reinterpret_cast<float*>(b.red) + (b.phase & 1) * kNWV * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float w = ekv_wave_max(x[i]);
    if (b.lane == 0) r[b.wave * 8 + i] = w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float y = r[i];
    for (int w = 1; w < kNWV; ++w) y = fmaxf(y, r[w * 8 + i]);
    x[i] = y;
  }
  b.phase++;
}
__device__ __forceinline__ void blk_sum8(Blk& b, float (&x)[8]) {
  float* r = reinterpret_cast<float*>(b.red) + (b.phase & 1) * kNWV * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float w = ekv_wave_sum(x[i]);
    if (b.lane == 0) r[b.wave * 8 + i] = w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float y = 0.f;
    for (int w = 0; w < kNWV; ++w) y += r[w * 8 + i];
    x[i] = y;
  }
  b.phase++;
}

// #{ j < n : pred(key[j], j) }, pred evaluated by every thread on a strided sweep
template <typename P>
__device__ __forceinline__ int blk_count(Blk& b, const uint32_t* key, int n, P pred) {
  int c = 0;
  const int n_round = (n + kNT - 1) / kNT * kNT;
  for (int j = b.tid; j < n_round; j += kNT) {
    const bool p = j < n && pred(key[j], j);
    c += __popcll(__ballot(p));
  }
  return blk_sum_uniform(b, c);
}

// k-th smallest key (1-indexed): MSB-first radix select, 8 bits per pass (4 passes).  Per pass: a 256-bin LDS
// histogram of the keys that still match the prefix (ds_add, worst case a few hundred cycles when every key falls in
// one bin), then wave 0 scans the bins and publishes (bin, keys below it).  Exact; no sort.
__device__ uint32_t blk_kth(Blk& b, const uint32_t* key, int n, int k, uint32_t* hist /* 256 + 2 words of LDS */) {
  uint32_t prefix = 0;
  int kk = k;   // rank still to find among the keys matching `prefix`
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = b.tid; i < 256; i += kNT) hist[i] = 0;
    __syncthreads();
    const uint32_t hi_mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int j = b.tid; j < n; j += kNT) {
      const uint32_t x = key[j];
      if ((x & hi_mask) == prefix) atomicAdd(&hist[(x >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (b.wave == 0) {
      const uint32_t c0 = hist[4 * b.lane], c1 = hist[4 * b.lane + 1], c2 = hist[4 * b.lane + 2], c3 = hist[4 * b.lane + 3];
      const uint32_t mine = c0 + c1 + c2 + c3;
      uint32_t incl = mine;
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o, 64);
        if (b.lane >= o) incl += y;
      }
      const uint32_t excl = incl - mine;
      if ((uint32_t)kk > excl && (uint32_t)kk <= incl) {   // exactly one lane owns the kk-th key
        uint32_t below = excl, bin = 4 * b.lane;
        if ((uint32_t)kk > below + c0) { below += c0; bin++;
          if ((uint32_t)kk > below + c1) { below += c1; bin++;
            if ((uint32_t)kk > below + c2) { below += c2; bin++; } } }
        hist[256] = bin;
        hist[257] = below;
      }
    }
    __syncthreads();
    prefix |= hist[256] << shift;
    kk -= (int)hist[257];
    __syncthreads();
  }
  return prefix;
}

// key[j] <- 1 for the k smallest (key, index) pairs, 0 otherwise
__device__ void blk_mark_k_smallest(Blk& b, uint32_t* key, int n, int k, uint32_t* hist) {
  uint32_t tau;
  int bound = n;
  if (k == 1) {
    unsigned long long best = ~0ull;
    for (int j = b.tid; j < n; j += kNT) {
      const unsigned long long x = ((unsigned long long)key[j] << 32) | (uint32_t)j;
      best = x < best ? x : best;
    }
    best = blk_min_u64(b, best);
    tau = (uint32_t)(best >> 32);
    bound = (int)(best & 0xFFFFFFFFu) + 1;
  } else {
    tau = blk_kth(b, key, n, k, hist);
    const int n_less = blk_count(b, key, n, [tau](uint32_t x, int) { return x < tau; });
    const int n_eq = blk_count(b, key, n, [tau](uint32_t x, int) { return x == tau; });
    const int need = k - n_less;
    if (n_eq != need) {  // ties at the threshold: lowest indices first
      int lo = 0, hi = n;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int c = blk_count(b, key, n, [tau, mid](uint32_t x, int j) { return x == tau && j < mid; });
        if (c >= need) hi = mid; else lo = mid + 1;
      }
      bound = lo;
    }
  }
  __syncthreads();
  for (int j = b.tid; j < n; j += kNT) {
    const uint32_t x = key[j];
    key[j] = (x < tau || (x == tau && j < bound)) ? 1u : 0u;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(kNT) ekv_score_select_kernel(const EkvScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int h = blockIdx.x, ll = blockIdx.y, gl = a.layer_begin + ll;
  const int T = a.n_slots, n = a.q_len, D = a.head_dim;
  const int rep = a.n_q_heads / a.n_kv_heads;
  const bool scored = a.policy == EKV_POLICY_H2O_HEAD || a.policy == EKV_POLICY_ROCO || a.policy == EKV_POLICY_TOVA;
  const int off = scored ? a.score_off : 0;
  const int W = T - off;
  const int rows = rep * n;

  float* sS = reinterpret_cast<float*>(smem);
  float* sQ = sS + W;
  float* sC = sQ + W;
  uint32_t* sKey = reinterpret_cast<uint32_t*>(sC + W);
  uint32_t* sKey2 = sKey + W;
  float* sRowM = reinterpret_cast<float*>(sKey2 + W);
  float* sRowL = sRowM + rows;
  Blk b;
  b.tid = threadIdx.x;
  b.lane = b.tid & 63;
  b.wave = b.tid >> 6;
  b.red = reinterpret_cast<unsigned long long*>(smem + ekv_align((size_t)(5 * W + 2 * rows) * 4, 16));
  b.phase = 0;
  uint32_t* sHist = reinterpret_cast<uint32_t*>(b.red) + 2 * kNWV * 8;   // 258 words

  const size_t head_row = ((size_t)gl * a.n_kv_heads + h) * a.cap;
  const size_t hq0 = (size_t)ll * a.n_q_heads + (size_t)h * rep;

  // ---- 0. fold the key-range splits into the attention output -------------------------------
  const int PS = D + 2;
  for (int idx = b.tid; idx < rows * D; idx += kNT) {
    const int row = idx / D, d = idx % D;  // row = r*n + i
    a.out[(hq0 * n + row) * D + d] = __float2half(ekv_fold_partials(a.partials + ((hq0 * n + row) * a.n_split) * PS, a.n_split, PS, d));
  }

  if (scored) {
    // ---- 1. exact softmax statistics of every query row (max, sum exp) over all T positions ----
    // all threads sweep the columns; 8 rows share one pair of block reductions
    if (a.accumulate) {
      for (int r0 = 0; r0 < rows; r0 += 8) {
        const int nr = min(8, rows - r0);
        const float* lg = a.logits + (hq0 * n + r0) * a.t_pad;
        float mx[8], sm[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) mx[i] = EKV_NEG_INF, sm[i] = 0.f;
        for (int j = b.tid; j < T; j += kNT) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (i < nr) mx[i] = fmaxf(mx[i], lg[(size_t)i * a.t_pad + j]);
        }
        blk_max8(b, mx);
        for (int j = b.tid; j < T; j += kNT) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (i < nr) sm[i] += expf(lg[(size_t)i * a.t_pad + j] - mx[i]);
        }
        blk_sum8(b, sm);
        if (b.tid == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (i < nr) sRowM[r0 + i] = mx[i], sRowL[r0 + i] = sm[i];
        }
      }
    }
    __syncthreads();
    // ---- 2. load the score rows, add this forward's folded probabilities ---------------------
    const float inv_rep_div = (float)rep;
    for (int j = b.tid; j < W; j += kNT) {
      float s = a.score_sum[head_row + j];
      float q = 0.f, c = 0.f;
      if (a.policy == EKV_POLICY_ROCO) {
        q = a.score_sq[head_row + j];
        c = a.score_cnt[head_row + j];
      }
      if (a.accumulate) {
        float colsum = 0.f, colsq = 0.f, last = 0.f;
        const float* col = a.logits + hq0 * n * a.t_pad + off + j;
        if (rep == 1) {
          // 8 independent loads in flight per thread (the rows of a column are t_pad floats apart)
          for (int i0 = 0; i0 < n; i0 += 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = (i0 + u < n) ? col[(size_t)(i0 + u) * a.t_pad] : EKV_NEG_INF;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if (i0 + u < n) {
                const float pb = expf(x[u] - sRowM[i0 + u]) / sRowL[i0 + u];
                colsum += pb;
                colsq += pb * pb;
                last = pb;
              }
            }
          }
        } else {
          for (int i = 0; i < n; ++i) {
            float pb = 0.f;
            float x[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = (r < rep) ? col[(size_t)(r * n + i) * a.t_pad] : EKV_NEG_INF;
#pragma unroll
            for (int r = 0; r < 8; ++r)
              if (r < rep) pb += expf(x[r] - sRowM[r * n + i]) / sRowL[r * n + i];
            for (int r = 8; r < rep; ++r) pb += expf(col[(size_t)(r * n + i) * a.t_pad] - sRowM[r * n + i]) / sRowL[r * n + i];
            pb = pb / inv_rep_div;
            colsum += pb;
            colsq += pb * pb;
            last = pb;
          }
        }
        if (a.policy == EKV_POLICY_TOVA) {
          s = a.tova_head_mean ? a.tova_row[(size_t)ll * a.t_pad + off + j] : last;
        } else {
          s += colsum;
          q += colsq;
        }
      }
      sS[j] = s;
      sQ[j] = q;
      sC[j] = c;
    }
    __syncthreads();
  }

  const int k = a.n_evict;
  if (k <= 0) {
    if (scored && a.accumulate) {
      for (int j = b.tid; j < W; j += kNT) {
        a.score_sum[head_row + j] = sS[j];
        if (a.policy == EKV_POLICY_ROCO) a.score_sq[head_row + j] = sQ[j];
      }
    }
    return;
  }

  // ---- 3. selection: k == 1 yields `victim` directly; k > 1 leaves flags in sKey (1 = evict) -------------
  int victim = -1;
  if (a.policy == EKV_POLICY_RANGE) {
    if (k == 1) {
      victim = a.range_start;
    } else {
      for (int j = b.tid; j < W; j += kNT) sKey[j] = (j >= a.range_start && j < a.range_start + k) ? 1u : 0u;
      __syncthreads();
    }
  } else if (a.policy == EKV_POLICY_ROCO) {
    for (int j = b.tid; j < W; j += kNT) {
      const float c = sC[j] + a.count_add;
      sC[j] = c;
      const float mean = sS[j] / c;
      float sd = sqrtf(sQ[j] / c - mean * mean);
      if (j >= W - a.roco_tail || j < a.win_lo) sd = 1e9f;
      sKey[j] = ekv_fkey(sd);
      sKey2[j] = ekv_fkey(mean);
    }
    __syncthreads();
    if (k == 1) {
      // victim = argmin mean over F = {k1 smallest std}.  Walk the candidates in increasing (mean, index)
      // order and take the first whose std rank is < k1: two block reductions per try instead of a k-select.
      for (int attempt = 0; attempt < 8 && victim < 0; ++attempt) {
        unsigned long long best = ~0ull;
        for (int j = b.tid; j < W; j += kNT) {
          const unsigned long long x = ((unsigned long long)sKey2[j] << 32) | (uint32_t)j;
          best = x < best ? x : best;
        }
        best = blk_min_u64(b, best);
        const int cand = (int)(best & 0xFFFFFFFFu);
        const uint32_t sk = sKey[cand];
        const int rank = blk_count(b, sKey, W, [sk, cand](uint32_t x, int j) { return x < sk || (x == sk && j < cand); });
        if (rank < a.roco_k1) {
          victim = cand;
        } else {
          if (b.tid == 0) sKey2[cand] = 0xFFFFFFFFu;  // not feasible: drop it from the walk
          __syncthreads();
        }
      }
    }
    if (victim < 0) {
      blk_mark_k_smallest(b, sKey, W, a.roco_k1, sHist);  // feasible set
      for (int j = b.tid; j < W; j += kNT) sKey[j] = sKey[j] ? sKey2[j] : 0xFFFFFFFFu;
      __syncthreads();
      blk_mark_k_smallest(b, sKey, W, k, sHist);
    }
  } else {  // h2o_head / tova: k smallest accumulated scores inside the candidate window
    if (k == 1) {
      unsigned long long best = ~0ull;
      for (int j = a.win_lo + b.tid; j < W - a.win_tail; j += kNT) {
        const unsigned long long x = ((unsigned long long)ekv_fkey(sS[j]) << 32) | (uint32_t)j;
        best = x < best ? x : best;
      }
      victim = (int)(blk_min_u64(b, best) & 0xFFFFFFFFu);
    } else {
      for (int j = b.tid; j < W; j += kNT)
        sKey[j] = (j >= a.win_lo && j < W - a.win_tail) ? ekv_fkey(sS[j]) : 0xFFFFFFFFu;
      __syncthreads();
      blk_mark_k_smallest(b, sKey, W, k, sHist);
    }
  }
  const bool roco = a.policy == EKV_POLICY_ROCO;

  if (victim >= 0) {
    // ---- 4a. single victim: everything behind it moves up by one, no scan needed ------------------------
    if (scored) {
      for (int d = b.tid; d < W; d += kNT) {
        const int j = d + (d >= victim ? 1 : 0);
        const bool tail = d == W - 1;
        a.score_sum[head_row + d] = tail ? 0.f : sS[j];
        if (roco) {
          a.score_sq[head_row + d] = tail ? 0.f : sQ[j];
          a.score_cnt[head_row + d] = tail ? 0.f : sC[j];
        }
      }
    }
    if (a.evict_ids != nullptr && b.tid == 0) a.evict_ids[(size_t)ll * a.n_kv_heads + h] = off + victim;
    // slot map: positions >= victim shift; the victim's row becomes the free tail (recycled by the next append)
    const int n_move = W - victim;          // entries victim .. W-1
    int32_t* sSlot = reinterpret_cast<int32_t*>(sKey);
    __syncthreads();
    for (int i = b.tid; i < n_move; i += kNT) sSlot[i] = a.slot_of_pos[head_row + off + victim + i];
    __syncthreads();
    for (int i = b.tid; i < n_move; i += kNT)
      a.slot_of_pos[head_row + off + victim + i] = (i == n_move - 1) ? sSlot[0] : sSlot[i + 1];
    return;
  }

  // ---- 4b. destinations: kept j -> #kept before j ; evicted j -> -(1 + #evicted before j) ----------
  {
    const int items = (W + kNT - 1) / kNT;
    const int c0 = min(W, b.tid * items), c1 = min(W, c0 + items);
    int kept = 0;
    for (int j = c0; j < c1; ++j) kept += sKey[j] ? 0 : 1;
    int incl = kept;
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(incl, o, 64);
      if (b.lane >= o) incl += y;
    }
    int* r = reinterpret_cast<int*>(b.red) + (b.phase & 1) * kNWV * 2;
    if (b.lane == 63) r[b.wave] = incl;
    __syncthreads();
    int base = incl - kept;
    for (int i = 0; i < b.wave; ++i) base += r[i];
    b.phase++;
    for (int j = c0; j < c1; ++j) {
      if (sKey[j]) {
        sKey[j] = (uint32_t)(-(1 + (j - base)));
      } else {
        sKey[j] = (uint32_t)base;
        base++;
      }
    }
    __syncthreads();
  }

  // ---- 5. write back compacted score rows, evict ids -------------------------------------------------
  if (scored) {
    for (int j = b.tid; j < W; j += kNT) {
      const int d = (int)sKey[j];
      if (d >= 0) {
        a.score_sum[head_row + d] = sS[j];
        if (roco) {
          a.score_sq[head_row + d] = sQ[j];
          a.score_cnt[head_row + d] = sC[j];
        }
      }
    }
    for (int i = b.tid; i < k; i += kNT) {
      a.score_sum[head_row + W - k + i] = 0.f;
      if (roco) {
        a.score_sq[head_row + W - k + i] = 0.f;
        a.score_cnt[head_row + W - k + i] = (float)i * a.count_tail_step;
      }
    }
  }
  if (a.evict_ids != nullptr) {
    for (int j = b.tid; j < W; j += kNT) {
      const int d = (int)sKey[j];
      if (d < 0) a.evict_ids[((size_t)ll * a.n_kv_heads + h) * k + (-1 - d)] = off + j;
    }
  }

  // ---- 6. slot-map compaction: kept rows keep their order, victims' rows become the free tail -----
  __syncthreads();
  int32_t* sSlot = reinterpret_cast<int32_t*>(sS);
  for (int j = b.tid; j < W; j += kNT) sSlot[j] = a.slot_of_pos[head_row + off + j];
  __syncthreads();
  for (int j = b.tid; j < W; j += kNT) {
    const int d = (int)sKey[j];
    const int dst = d >= 0 ? d : (W - k) + (-1 - d);
    a.slot_of_pos[head_row + off + dst] = sSlot[j];
  }
}

// 'tova' in encoding/ppl mode: one last-query row averaged over ALL kv heads (easykv/easykv.py:456, :847)
__global__ void __launch_bounds__(kNT) ekv_tova_headmean_kernel(const EkvScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ll = blockIdx.x;
  const int T = a.n_slots, n = a.q_len, H = a.n_kv_heads;
  const int rep = a.n_q_heads / H;
  float* sM = reinterpret_cast<float*>(smem);
  float* sL = sM + H * rep;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int hr = wave; hr < H * rep; hr += kNWV) {
    const float* lg = a.logits + (((size_t)ll * a.n_q_heads + hr) * n + (n - 1)) * a.t_pad;
    float mx = EKV_NEG_INF;
    for (int j = lane; j < T; j += 64) mx = fmaxf(mx, lg[j]);
    mx = ekv_wave_max(mx);
    float sm = 0.f;
    for (int j = lane; j < T; j += 64) sm += expf(lg[j] - mx);
    sm = ekv_wave_sum(sm);
    if (lane == 0) {
      sM[hr] = mx;
      sL[hr] = sm;
    }
  }
  __syncthreads();
  for (int j = tid; j < T; j += kNT) {
    float acc = 0.f;
    for (int h = 0; h < H; ++h) {
      float pb = 0.f;
      for (int r = 0; r < rep; ++r) {
        const int hr = h * rep + r;
        const float x = a.logits[(((size_t)ll * a.n_q_heads + hr) * n + (n - 1)) * a.t_pad + j];
        pb += expf(x - sM[hr]) / sL[hr];
      }
      if (rep > 1) pb = pb / (float)rep;
      acc += pb;
    }
    a.tova_row[(size_t)ll * a.t_pad + j] = acc / (float)H;
  }
}

}  // namespace

size_t ekv_score_lds_bytes(const EkvScoreArgs& a) {
  const bool scored = a.policy == EKV_POLICY_H2O_HEAD || a.policy == EKV_POLICY_ROCO || a.policy == EKV_POLICY_TOVA;
  const int W = a.n_slots - (scored ? a.score_off : 0);
  const int rows = (a.n_q_heads / a.n_kv_heads) * a.q_len;
  return ekv_align((size_t)(5 * W + 2 * rows) * 4, 16) + 2 * kNWV * 8 * 4 + 260 * 4;
}

hipError_t ekv_launch_score_select(const EkvScoreArgs& a, int layer_count, hipStream_t s) {
  const size_t lds = ekv_score_lds_bytes(a);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ekv_score_select_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ekv_score_select_kernel, dim3(a.n_kv_heads, layer_count), dim3(kNT), lds, s, a);
  return hipGetLastError();
}

hipError_t ekv_launch_tova_headmean(const EkvScoreArgs& a, int layer_count, hipStream_t s) {
  const size_t lds = (size_t)a.n_q_heads * 2 * 4;
  hipLaunchKernelGGL(ekv_tova_headmean_kernel, dim3(layer_count), dim3(kNT), lds, s, a);
  return hipGetLastError();
}
