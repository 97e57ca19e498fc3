// Scorer of a two-pass wide chunk step as the TAIL of the column-sum pass (ekv_attn_wide.inc, EKV_WIDE_MODE 2; round 5).
//
// Replaces, for those steps, the stand-alone ekv_score_select_kernel launch (same reference sites: accumulate
// easykv/easykv.py:443-457, count advance :460, select :462-490, score-row + K/V compaction :465-490 / :56-82 as a slot-map compaction).
// That launch was latency-bound and on the critical path of every wide step: W = 5098 columns are 87 KB of LDS with S / Q / C and the
// keys resident, i.e. ONE 1024-thread workgroup per CU and four dispatch rounds of ~27 us (112 us of a 0.93 ms configs[3] step, 20 %
// of a configs[2] step).  Here the workgroup that finishes a head's column sums scores and evicts the head itself:
//   * the score rows live in REGISTERS (thread-owned columns j = tid + 256 * it; S, Q, C: 3 x ITEMS), only the selection keys in LDS
//     (4 B per column, in the tile buffers the stream has left): the pass keeps its four workgroups per CU, all 1024 heads of a
//     32-layer launch are scored side by side instead of in four rounds, and the launch boundary in front of the scorer is gone;
//   * the arithmetic, the order of the sums and the exact selects (blk_mark_k_smallest: range histogram, refinement, radix / tie
//     fallbacks) are those of ekv_score_select_body, so the decisions are the stand-alone scorer's bit for bit.
// Measured (MI355X, cycle stamps per head at the configs[3] shape, -DEKV_TAIL_PROFILE; all 1024 heads in their tails at once): loads +
// accumulate 28.4 k cycles (105 MB of score rows and column sums: bandwidth), keys 9.1 k, select k1 18.6 k, mean keys + select k 23.1 k,
// scan 7.4 k, write-back 19.2 k = 105.6 k cycles ~ 50 us per head against 4 x 27 us of the stand-alone launch: configs[3] step 879 vs 937
// us same box, stride 64 387 vs 402, configs[2] 91.4 vs 92.8.  NOT for the RoPE-on-read builds: at their two workgroups per CU a 50 us
// tail leaves the CU's other workgroup streaming alone (configs[4] step 2434 vs 2385 us) — those keep the stand-alone scorer.
// Built and rejected: the selection keys in registers too (warm-started window -> range histogram with refinement -> bisection, the
// scheme of the slot-indexed decode tail; destinations by ballots + mbcnt instead of a scan over LDS): bit-identical, but 160 live
// registers against the pass's 128-register bound (four workgroups per CU) — 51-85 spilled registers, every phase of the tail 1.5-3x
// slower (177 k cycles per head) — and in the RoPE builds, which have the registers, the tail took the same 147 us as this version:
// the chain of ~40 workgroup barriers and its memory round trips, not the LDS sweeps of the selects, is what a head's tail costs.
// Also built and withdrawn: warm-started selects on the LDS keys (window around the head's previous threshold key: one counting / listing
// sweep, a reduction, one ranking, one marking sweep — 2 sweeps + 4 barriers instead of 5 + 8).  The std select hit its window in 99.5 % of
// the heads at the configs[3] shape (90 % stride 64, 78 % configs[2]) and took the SAME 18 k cycles; the mean select (its threshold jumps
// when the 96 smallest leave) missed every time.  With 16 waves per CU in the same phase at once the tail is bound by VALU issue — ~12
// instructions per key and sweep — not by the barrier chain: fewer instructions per key would help, fewer barriers do not.
// Only for heads whose column sums come from ONE workgroup (unsplit heads, one query-block group: every 32-layer launch of the BASELINE
// shapes).  Built, measured and removed: a last-arriver election (ekv_bank.arrive: workgroup barrier, agent-scope release by one lane,
// one atomic per workgroup, acquire in the elected one) for heads split over key ranges — bit-identical (tests/test_hip_wide_tail.py ran
// it), and SLOWER than the stand-alone scorer wherever it applied: the 8-layer configs[3] stage (256 heads x 2 splits) 320.6 vs 296.9 us,
// one layer per call with the column-sum pass deferred to the flush (1024 heads x 8 splits) 65.3 vs 56.6 us per layer, stride 64 39.3 vs
// 30.0 — a release fence in every workgroup of the launch, and the elected workgroup holds its slot for the ~50 us of a tail while the
// head's other workgroups have long left.
#pragma once

// Included after ekv_score_select.inc (EKV_SS_NT 256, EKV_SS_DEVICE_ONLY): Blk, blk_mark_k_smallest, kNT, kNWV.

constexpr int kTailItems = 24;                       // owned columns per thread: score rows of up to 24 * 256 = 6144 positions

__host__ __device__ inline size_t ekw_tail_lds_bytes(int W) {      // keys | reduction scratch | histogram | candidate list
  return ekv_align((size_t)W * 4, 16) + 2 * kNWV * 8 * 4 + 264 * 4 + kNT * 8;
}

// The score rows of the thread's columns, requested ahead of the tail (ekv_attn_resident.inc issues them before its column-sum phase: the
// round trip is over when the tail starts): the loads of step 1 below, same clamping.
template <int ITEMS>
__device__ __forceinline__ void ekw_tail_preload(const EkvScoreArgs& a, const int h, const int ll, float (&pS)[ITEMS], float (&pQ)[ITEMS], float (&pC)[ITEMS]) {
  const int W = a.n_slots - a.score_off;
  const bool roco = a.policy == EKV_POLICY_ROCO;
  const size_t head_row = ((size_t)(a.layer_begin + ll) * a.n_kv_heads + h) * a.cap;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    pS[it] = pQ[it] = pC[it] = 0.f;
    if (it * kNT < W) {
      const int j = min((int)threadIdx.x + it * kNT, W - 1);
      pS[it] = a.score_sum[head_row + j];
      if (roco) {
        pQ[it] = a.score_sq[head_row + j];
        pC[it] = a.score_cnt[head_row + j];
      }
    }
  }
}

// lds_cs / lds_cq (ekv_attn_resident.inc): this forward's column sums sit in LDS ([n_slots] each, complete) instead of the pass's
// partial rows in global memory; pS / pQ / pC: the score rows were requested by ekw_tail_preload
template <int ITEMS>
__device__ __forceinline__ void ekw_score_tail(const EkvScoreArgs& a, const int h, const int ll, char* smem, const float* lds_cs = nullptr,
                                               const float* lds_cq = nullptr, const float* pS = nullptr, const float* pQ = nullptr,
                                               const float* pC = nullptr) {
  static_assert(kNT == 256 || kNT == 512, "the tail runs on the 256-thread workgroups of the wide-block kernel and the 512-thread ones of the logits-resident kernel");
  const int tid = threadIdx.x;
  const int T = a.n_slots, off = a.score_off, W = T - off, k = a.n_evict;
  const bool roco = a.policy == EKV_POLICY_ROCO;
  const size_t head_row = ((size_t)(a.layer_begin + ll) * a.n_kv_heads + h) * a.cap;
  const float* const cp0 = a.colsum + ((size_t)ll * a.n_kv_heads + h) * a.n_col_parts * 2 * a.t_pad + off;
  uint32_t* const sKey = reinterpret_cast<uint32_t*>(smem);
  Blk b;
  b.tid = tid;
  b.lane = tid & 63;
  b.wave = tid >> 6;
  b.red = reinterpret_cast<unsigned long long*>(smem + ekv_align((size_t)W * 4, 16));
  b.phase = 0;
  uint32_t* const sHist = reinterpret_cast<uint32_t*>(b.red) + 2 * kNWV * 8;      // 264 words, then the select's candidate list (kNT uint64)
#ifdef EKV_TAIL_PROFILE      // cycle stamps per head in the (unused) tova_row scratch: tools/experiments/exp_widetail_prof.py
  unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(a.tova_row) + ((size_t)ll * a.n_kv_heads + h) * 8;
#define EKW_TSTAMP(i) do { __syncthreads(); if (threadIdx.x == 0) stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define EKW_TSTAMP(i) do { } while (0)
#endif
  EKW_TSTAMP(0);

  // ---- 1. score rows + this forward's column sums (parts in order 0, 1, ...: ekv_score_select_body's summation order) ----------
  float rS[ITEMS], rQ[ITEMS], rC[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    rS[it] = rQ[it] = rC[it] = 0.f;
    if (it * kNT < W) {                              // (workgroup-uniform)
      const int j = min(tid + it * kNT, W - 1);      // unconditional (clamped) loads
      float s, q = 0.f, c = 0.f;
      if (pS != nullptr) {                            // (compile-time after inlining)
        s = pS[it];
        q = pQ[it];
        c = pC[it];
      } else {
        s = a.score_sum[head_row + j];
        if (roco) {
          q = a.score_sq[head_row + j];
          c = a.score_cnt[head_row + j];
        }
      }
      float cs = 0.f, cq = 0.f;
      if (lds_cs != nullptr) {                        // (workgroup-uniform)
        cs += lds_cs[off + j];
        cq += lds_cq[off + j];
      } else {
        for (int part = 0; part < a.n_col_parts; ++part) {
          cs += cp0[(size_t)(2 * part) * a.t_pad + j];
          cq += cp0[(size_t)(2 * part + 1) * a.t_pad + j];
        }
      }
      rS[it] = s + cs;
      rQ[it] = q + cq;
      rC[it] = c;
    }
  }
  EKW_TSTAMP(1);
  if (k <= 0) {                                      // an accumulating step that evicts nothing (the scored dense prefix)
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int j = tid + it * kNT;
      if (j < W) {
        a.score_sum[head_row + j] = rS[it];
        if (roco) a.score_sq[head_row + j] = rQ[it];
      }
    }
    return;
  }

  // ---- 2. selection: flags in sKey (1 = evict) -------------------------------------------------------------------------------
  if (roco) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int j = tid + it * kNT;
      if (j < W) {
        const float c = rC[it] + a.count_add;
        rC[it] = c;
        const float mean = rS[it] / c;
        float sd = sqrtf(rQ[it] / c - mean * mean);
        if (j >= W - a.roco_tail || j < a.win_lo) sd = 1e9f;
        sKey[j] = ekv_fkey(sd);
      }
    }
    __syncthreads();
    EKW_TSTAMP(2);
    [[clang::always_inline]] blk_mark_k_smallest(b, sKey, W, a.roco_k1, sHist, ekv_fkey(1e9f));      // feasible set
    EKW_TSTAMP(3);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int j = tid + it * kNT;
      if (j < W) sKey[j] = sKey[j] ? ekv_fkey(rS[it] / rC[it]) : 0xFFFFFFFFu;
    }
    __syncthreads();
    [[clang::always_inline]] blk_mark_k_smallest(b, sKey, W, k, sHist);
  } else {                                           // h2o_head: k smallest accumulated scores inside the candidate window
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int j = tid + it * kNT;
      if (j < W) sKey[j] = (j >= a.win_lo && j < W - a.win_tail) ? ekv_fkey(rS[it]) : 0xFFFFFFFFu;
    }
    __syncthreads();
    [[clang::always_inline]] blk_mark_k_smallest(b, sKey, W, k, sHist);
  }

  EKW_TSTAMP(4);
  // slot-map cells of the owned columns: requested now, consumed behind the scan's barriers
  int32_t cell[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    cell[it] = 0;
    if (it * kNT < W) cell[it] = a.slot_of_pos[head_row + off + min(tid + it * kNT, W - 1)];
  }

  // ---- 3. destinations: kept j -> #kept before j ; evicted j -> -(1 + #evicted before j)  (ekv_score_select_body step 4b) --------
  {
    const int items = (W + kNT - 1) / kNT;
    const int c0 = min(W, tid * items), c1 = min(W, c0 + items);
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
    __syncthreads();       // (also: every thread's cell loads have completed before any thread stores to the slot map)
  }

  EKW_TSTAMP(5);
  // ---- 4. write back from the registers: compacted score rows, evict ids (ascending), compacted slot map -----------------------
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int j = tid + it * kNT;
    if (j < W) {
      const int d = (int)sKey[j];
      if (d >= 0) {
        a.score_sum[head_row + d] = rS[it];
        if (roco) {
          a.score_sq[head_row + d] = rQ[it];
          a.score_cnt[head_row + d] = rC[it];
        }
        a.slot_of_pos[head_row + off + d] = cell[it];
      } else {
        const int e = -1 - d;
        if (a.evict_ids != nullptr) a.evict_ids[((size_t)ll * a.n_kv_heads + h) * k + e] = off + j;
        a.slot_of_pos[head_row + off + (W - k) + e] = cell[it];      // the victims' rows become the free tail
      }
    }
  }
  for (int i = tid; i < k; i += kNT) {
    a.score_sum[head_row + W - k + i] = 0.f;
    if (roco) {
      a.score_sq[head_row + W - k + i] = 0.f;
      a.score_cnt[head_row + W - k + i] = (float)i * a.count_tail_step;
    }
  }
  EKW_TSTAMP(6);
}
