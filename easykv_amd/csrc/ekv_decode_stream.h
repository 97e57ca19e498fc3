// The streaming loop of the decode attention (shared by the split and the fused kernels).
//
// Mapping (D = 128): a K/V row is 256 B; 16 lanes x 16 B read one row fully coalesced along the
// head-dim axis, a wave64 load instruction covers 4 rows, each lane keeps kU = 8 K and 8 V loads in
// flight.  QK^T is v_dot2c_f32_f16 + a fused v_add_f32_dpp butterfly over the lanes of a row; softmax is
// lane-local online (exp2), so there is no cross-row traffic inside the loop.  One workgroup serves the
// REP query heads of one KV head (GQA: the K/V rows are read once).
#pragma once
#include "ekv_common.h"
#include "ekv_kernels.h"

constexpr int kU = 8;   // rows in flight per lane group (K and V each)

// head_dim 96 (any head_dim that is a multiple of 16 but not a power of two): a row keeps a power-of-two lane group (LPR = 16) of
// which only the first LIVE = D / 8 lanes hold a 16-byte piece; the idle lanes carry zeros through the dot products and reductions.
template <int D, int NW = 4, int KU = kU>
struct EkvDecodeGeom {
  static constexpr int LIVE = D / 8;  // lanes of a row's lane group that hold a 16-byte piece
  static constexpr int LPR = LIVE <= 4 ? 4 : (LIVE <= 8 ? 8 : 16);   // lanes per row (power of two)
  static constexpr int G = 64 / LPR;  // rows per wave-load
  static constexpr int RW = G * KU;   // rows per wave per iteration
  static constexpr int NP = NW;       // partials per workgroup = waves (lane groups are combined in-wave)
  static constexpr int PS = D + 2;    // (m, l, o[D])
};

// Streams positions [t0, t1) of KV head h.  SLOT_LDS: s_slot holds slot_of_pos[t0..t1) in LDS; otherwise s_slot is
// the head's row of the global slot map (t0 must be a multiple of 8) and the 8 indices of a lane group are fetched
// one iteration ahead.  Logits (q.k / sm_div) go to `logit_out` (+ `logit_stride` per query head): workspace or LDS.
//
// PHYS (fused kernel, t0 = 0, t1 = T): the rows are streamed in PHYSICAL order, 0 .. a.phys_extent-1, not through the slot
// map.  Eviction recycles rows in place, so after a few thousand steps of a score-driven policy the birth order of the
// live rows is a random permutation of their addresses, while attention does not care about the order: streaming by
// address keeps the HBM access pattern sequential whatever the eviction history was and takes the slot-map loads out of the
// loop.  (Measured on MI355X at T = 2049, D = 128: gathering whole 256-byte rows in random order costs nothing measurable,
// +-3 % run-to-run noise either way; the 188 -> 223 us drift of the first version over 1000+ steps was the roco select
// falling through to its bisection fallback, see ekv_decode_tail.h.)
// `s_dead` (LDS, one bit per row) marks free rows, the row the new token is being written to and the padding past the
// extent; logits are stored at the PHYSICAL row index (the scorer tail reads them back through the slot map).
// `mask_ready` (PHYS): finishes the dead-row bits (LDS barriers inside, so every wave calls it exactly once).  It runs after
// the loads of the wave's FIRST iteration have been issued — the K/V addresses do not depend on it — so the round trip of the
// free list hides behind the first K/V round trip instead of preceding it.
struct EkvNoop {
  __device__ __forceinline__ void operator()() const {}
};
// KU = rows in flight per lane group (K and V each): 8 (122 VGPRs in the fused kernel = four workgroups per CU).  16 is
// supported for plain keys in logical order and was tried in the split kernel (ekv_attn_decode.inc): no gain.
#ifndef EKV_ROPE_MOCK
#define EKV_ROPE_MOCK 0      // experiment builds of the RoPE-on-read decode stream: 1 = no table loads, 2 = no partner exchange
#endif
template <int D, int REP, bool ROPE, bool SLOT_LDS, int NW = 4, bool PHYS = false, int KU = kU, typename MaskReady = EkvNoop>
__device__ __forceinline__ void ekv_decode_stream(const EkvAttnArgs& a, const int32_t* s_slot, float* logit_out,
                                                  int logit_stride, int t0, int t1_in, int ll, int h, size_t head_row,
                                                  float (&m)[REP], float (&l)[REP], float (&o)[REP][8],
                                                  const uint8_t* s_dead = nullptr, MaskReady mask_ready = MaskReady(),
                                                  int n_rep_real = 0, int q_head0 = -1) {
  using Gm = EkvDecodeGeom<D, NW, KU>;
  constexpr int LPR = Gm::LPR, RW = Gm::RW, LIVE = Gm::LIVE;
  constexpr bool PADDED = LIVE != LPR;      // head_dim 96: lanes sub >= LIVE of a lane group are idle (zero pieces, no loads / stores)
  constexpr int kNW = NW;
  static_assert(!(PHYS && (ROPE || SLOT_LDS)), "physical-order streaming: plain keys, global slot map");
  static_assert((KU == 4 || KU == 8 || KU == 16) && (!PHYS || KU <= 8), "4, 8 or 16 rows per lane group; the dead-row mask is one byte per 8 rows");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane % LPR, grp = lane / LPR;
  const int t_new = a.n_slots - 1;  // the appended position
  int t1 = t1_in;
  const bool live_lane = !PADDED || sub < LIVE;
  // GQA factors that are not a power of two (or, in the split kernel, groups of <= 8 query heads of a wider factor): REP is the
  // padded count, `nrep` the query heads this workgroup really serves, from head `q_head0`; the padding heads r >= nrep repeat the
  // last real one (their logits / partials / outputs are never written out)
  const int nrep = n_rep_real > 0 ? n_rep_real : REP;
  const int hq0 = q_head0 >= 0 ? q_head0 : h * REP;

  uint4 qv[REP];
  float qf[REP][8];  // ROPE: rotated query q' (fp32)
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    const __half* qp = a.q + ((size_t)ll * a.n_q_heads + hq0 + min(r, nrep - 1)) * D;
    qv[r] = live_lane ? reinterpret_cast<const uint4*>(qp)[sub] : uint4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) qf[r][i] = 0.f;
    if (ROPE && live_lane) {
      // q' = q*cos[T-1] + rotate_half(q)*sin[T-1]   (llama_patch.py:311, :326)
      const int half_d = D / 2;
      const float* c = a.rope_cos + (size_t)t_new * D;
      const float* s = a.rope_sin + (size_t)t_new * D;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int d = sub * 8 + i;
        const int dp = d < half_d ? d + half_d : d - half_d;
        const float x = __half2float(qp[d]), y = __half2float(qp[dp]);
        qf[r][i] = x * c[d] + (d < half_d ? -y : y) * s[d];
      }
    }
  }

  // RoPE-on-read: keys are cached un-rotated and rotated by their SLOT index j at read time (llama_patch.py:312, :327):
  // k'[d] = k[d]*cos[j][d] + rotate_half(k)[d]*sin[j][d]; the partner k[d +- D/2] sits in lane sub ^ LPR/2 of the row's lane
  // group.  The row is rotated ONCE (fp32) and then dotted with each of the REP rotated queries.
  //
  // Where cos / sin come from (round 6).  A lane group walks KU CONSECUTIVE positions per iteration.  The tables are a rotation that is
  // linear in the position — row j = (cos(j*theta_f), sin(j*theta_f)), include/easykv_hip.h — so only the FIRST row of an iteration
  // is read from the table (the seed: exact table values) and the following KU - 1 rows advance the lane's 8 (cos, sin) pairs by the
  // angle-addition recurrence with the step constants (cos theta_f, sin theta_f) = table row 1, held in registers: 4 flops per pair
  // and step, a drift of a few fp32 roundings (< 1e-6 after 7 steps) before the next seed.  Before, every row loaded 64 B of fp32
  // table per lane — 512 B of table per 256-B key row through the vector-memory path, and 16 table registers per row IN FLIGHT next
  // to the K / V registers, which is what held the Llama-shape build at 4 rows in flight (25 of the 33 us over the plain stream:
  // profiles/r05 mocks).  The sine is kept pre-multiplied by the half's sign (-1 for d < D/2): (c, s~) obeys the same recurrence with
  // the step constant s~1, and the rotation is one fma + one multiply per element.  -DEKV_ROPE_EXACT=1 reads every row from the table;
  // so does the GQA x 8 build, whose eight fp32 queries leave no room for the 16 step registers (8 spilled VGPRs: 143 vs 135 us at
  // Hq = 64 / H = 8, measured).  Measured, Llama2-7B shape, budget 2048, same box: 219.3 us (round 5) / 225.5 (this code, every row from
  // the table) -> 198.6 us with the recurrence; plain keys 185.6; Mistral shape 95.0 -> 88.3.
#ifndef EKV_ROPE_EXACT
#define EKV_ROPE_EXACT 0
#endif
  constexpr bool kRopeExact = EKV_ROPE_EXACT || REP == 8;
  const float sgn = (sub < LIVE / 2) ? -1.f : 1.f;
  const int tab_off = (sub % (LIVE / 2)) * 8;      // cat(freqs, freqs): both halves of the lane group read the FIRST half of a row
  auto rope_seed = [&](int j, ekv_f2 (&cc)[4], ekv_f2 (&ss)[4]) {
#if EKV_ROPE_MOCK == 1      // (mock: no table loads — opaque constants, same arithmetic)
    float one = 1.f, zero = 0.25f;
    asm volatile("" : "+v"(one), "+v"(zero));
#pragma unroll
    for (int i = 0; i < 4; ++i) cc[i] = ekv_f2{one, one}, ss[i] = ekv_f2{zero, zero};
    (void)j;
#else
    const float4* c4 = reinterpret_cast<const float4*>(a.rope_cos + (size_t)j * D + tab_off);
    const float4* s4 = reinterpret_cast<const float4*>(a.rope_sin + (size_t)j * D + tab_off);
    const float4 c0 = c4[0], c1 = c4[1], s0 = s4[0], s1 = s4[1];
    cc[0] = ekv_f2{c0.x, c0.y}, cc[1] = ekv_f2{c0.z, c0.w}, cc[2] = ekv_f2{c1.x, c1.y}, cc[3] = ekv_f2{c1.z, c1.w};
    ss[0] = ekv_f2{s0.x, s0.y} * sgn, ss[1] = ekv_f2{s0.z, s0.w} * sgn, ss[2] = ekv_f2{s1.x, s1.y} * sgn, ss[3] = ekv_f2{s1.z, s1.w} * sgn;
#endif
  };
  ekv_f2 stp_c[4], stp_s[4];      // (cos theta_f, sgn * sin theta_f) of this lane's 8 frequencies
  if (ROPE && !kRopeExact) rope_seed(min(1, a.n_slots - 1), stp_c, stp_s);
  auto rope_advance = [&](ekv_f2 (&cc)[4], ekv_f2 (&ss)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const ekv_f2 cn = __builtin_elementwise_fma(cc[i], stp_c[i], -(ss[i] * stp_s[i]));
      ss[i] = __builtin_elementwise_fma(ss[i], stp_c[i], cc[i] * stp_s[i]);
      cc[i] = cn;
    }
  };
  auto rope_rotate = [&](const uint4& kv, const ekv_f2 (&cc)[4], const ekv_f2 (&ss)[4], float (&kp)[8]) {
    uint4 other;
#if EKV_ROPE_MOCK == 2      // (mock: no partner exchange)
    other = kv;
#else
    if (PADDED) {      // the partner piece d +- D/2 is LIVE/2 lanes away, not an xor pattern: ds_bpermute
      const int src = (lane & ~(LPR - 1)) | (sub < LIVE / 2 ? sub + LIVE / 2 : (sub < LIVE ? sub - LIVE / 2 : sub));
      other.x = __shfl(kv.x, src, 64);
      other.y = __shfl(kv.y, src, 64);
      other.z = __shfl(kv.z, src, 64);
      other.w = __shfl(kv.w, src, 64);
    } else {
      other.x = __shfl_xor(kv.x, LPR / 2, 64);
      other.y = __shfl_xor(kv.y, LPR / 2, 64);
      other.z = __shfl_xor(kv.z, LPR / 2, 64);
      other.w = __shfl_xor(kv.w, LPR / 2, 64);
    }
#endif
    const ekv_h8 kh = __builtin_bit_cast(ekv_h8, kv), oh = __builtin_bit_cast(ekv_h8, other);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const ekv_f2 k2 = {(float)kh[2 * i], (float)kh[2 * i + 1]}, o2 = {(float)oh[2 * i], (float)oh[2 * i + 1]};
      const ekv_f2 r2 = __builtin_elementwise_fma(k2, cc[i], o2 * ss[i]);
      kp[2 * i] = r2[0];
      kp[2 * i + 1] = r2[1];
    }
  };

  const __half* k_new_row = a.k_new + ((size_t)ll * a.n_kv_heads + h) * D;
  const __half* v_new_row = a.v_new + ((size_t)ll * a.n_kv_heads + h) * D;
  const int slot_base = SLOT_LDS ? t0 : 0;   // index origin of s_slot
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    m[r] = EKV_NEG_INF;
    l[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[r][i] = 0.f;
  }

  // The appended row is peeled off the loop: its K/V come from k_new / v_new and one lane group appends and scores it AFTER
  // the loop (below) — doing that first cost wave 0 three dependent round trips (slot index, k_new, v_new -> store) before
  // its first K/V load, while the other waves were already streaming.  The loop covers [t0, t_new): with T = budget + 1 =
  // 2049 that is exactly 16 full iterations instead of 17.
  const bool has_new = t_new >= t0 && t_new < t1;
  if (has_new) t1 = t_new;
  auto appended_row = [&]() {
    if (!(has_new && wave == 0 && grp == 0)) return;
    const int slot_new = s_slot[t_new - slot_base];
    uint4 kn = uint4{0, 0, 0, 0}, vn = uint4{0, 0, 0, 0};
    const size_t off = (head_row + slot_new) * D;          // append: the new row goes into the recycled slot
    if (live_lane) {
      kn = reinterpret_cast<const uint4*>(k_new_row)[sub];
      vn = reinterpret_cast<const uint4*>(v_new_row)[sub];
      reinterpret_cast<uint4*>(a.k_w + off)[sub] = kn;
      reinterpret_cast<uint4*>(a.v_w + off)[sub] = vn;
    }
    float kpn[8];
    if (ROPE) {
      ekv_f2 cn[4], sn[4];
      rope_seed(t_new, cn, sn);
      rope_rotate(kn, cn, sn, kpn);
    }
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      float acc;
      if (ROPE) {
        acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = fmaf(kpn[i], qf[r][i], acc);
      } else {
        acc = ekv_dot8(qv[r], kn, 0.f);
      }
      acc = ekv_group_sum<LPR>(acc) / a.sm_div;
      if (logit_out != nullptr && sub == 0 && r < nrep) logit_out[(size_t)r * logit_stride + (PHYS ? slot_new : t_new)] = acc;
      // one more row for this lane group's online softmax
      const float mn = fmaxf(m[r], acc);
      const float alpha = m[r] == EKV_NEG_INF ? 0.f : exp2f((m[r] - mn) * EKV_LOG2E);
      const float p = exp2f((acc - mn) * EKV_LOG2E);
      l[r] = l[r] * alpha + p;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[r][i] *= alpha;
      ekv_axpy8(p, vn, o[r]);
      m[r] = mn;
    }
  };
  if (!PHYS && t1 <= t0) {   // the split held only the appended row
    appended_row();
    return;
  }

  if (PHYS) t1 = a.phys_extent;   // loop bound: physical rows [0, E); which of them count is s_dead's business
  const int last_slot = PHYS ? 0 : s_slot[t1 - 1 - slot_base];
  const int idx_cap = (a.cap - KU) & ~(KU - 1);   // prefetches past t1 stay inside the head's map row (values unused); a multiple of KU like j0
  uint4 iv[KU / 4];                        // !SLOT_LDS: slot indices of rows j0..j0+KU-1 of the next iteration
#pragma unroll
  for (int i = 0; i < KU / 4; ++i) iv[i] = uint4{0, 0, 0, 0};
  if (!PHYS && !SLOT_LDS && t0 + wave * RW < t1) {
    const uint4* ip = reinterpret_cast<const uint4*>(s_slot + min(t0 + wave * RW + grp * KU, idx_cap));
#pragma unroll
    for (int i = 0; i < KU / 4; ++i) iv[i] = ip[i];
  }
  auto iteration = [&](const int base, auto&& after_issue) {
    uint4 kr[KU], vr[KU];
    const int j0 = base + grp * KU;
    int cur[KU];
#pragma unroll
    for (int i = 0; i < KU / 4; ++i) {
      cur[4 * i] = (int)iv[i].x;
      cur[4 * i + 1] = (int)iv[i].y;
      cur[4 * i + 2] = (int)iv[i].z;
      cur[4 * i + 3] = (int)iv[i].w;
    }
    if (!PHYS && !SLOT_LDS && base + kNW * RW < t1) {     // next iteration's indices (the map row has >= t_pad entries)
      const uint4* ip = reinterpret_cast<const uint4*>(s_slot + min(j0 + kNW * RW, idx_cap));
#pragma unroll
      for (int i = 0; i < KU / 4; ++i) iv[i] = ip[i];
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int j = j0 + u;
      const int jj = j < t1 ? j : t1 - 1;
      const int row = PHYS ? jj : (SLOT_LDS ? s_slot[jj - t0] : (j < t1 ? cur[u] : last_slot));
      const __half* kp = a.k + (head_row + row) * D;
      const __half* vp = a.v + (head_row + row) * D;
      // K/V rows are read exactly once per step and the cache (>1 GB) never fits L2/MALL: non-temporal loads
      // (measured on MI355X: 5.5 -> 6.1 TB/s on the pure stream)
      if (live_lane) {
        kr[u] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const ekv_u4*>(kp) + sub));
        vr[u] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const ekv_u4*>(vp) + sub));
      } else {
        kr[u] = vr[u] = uint4{0, 0, 0, 0};
      }
    }
    ekv_f2 rc[4], rs[4];      // ROPE: (cos, signed sin) of row j0 for this lane's 8 frequencies — the seed, read with the K / V rows
    if (ROPE) rope_seed(min(j0, t1 - 1), rc, rs);
    after_issue();
    // PHYS: bit u of dead8 = row j0+u is free / being appended / past the extent (j0 is a multiple of 8: one mask byte)
    const unsigned dead8 = PHYS ? (KU == 4 ? ((unsigned)s_dead[j0 >> 3] >> (j0 & 4)) & 0xFu : (unsigned)s_dead[j0 >> 3]) : 0u;
    if (PHYS && dead8 != 0u) {   // rare: a dead row may hold anything (0 * inf = NaN in the PV accumulation)
#pragma unroll
      for (int u = 0; u < KU; ++u)
        if ((dead8 >> u) & 1u) vr[u] = uint4{0, 0, 0, 0};
    }
    float sall[ROPE ? REP : 1][KU];   // ROPE: every row is rotated once, then dotted with all REP queries
    if (ROPE) {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        float kp[8];
        rope_rotate(kr[u], rc, rs, kp);
        if (u + 1 < KU) {      // (cos, sin) of the next position
          if (kRopeExact) rope_seed(min(j0 + u + 1, t1 - 1), rc, rs);
          else rope_advance(rc, rs);
        }
#pragma unroll
        for (int r = 0; r < REP; ++r) {
          float acc = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) acc = fmaf(kp[i], qf[r][i], acc);
          acc = ekv_group_sum<LPR>(acc);
          sall[r][u] = (j0 + u < t1) ? acc / a.sm_div : EKV_NEG_INF;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      float s[KU];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        if (ROPE) {
          s[u] = sall[r][u];
        } else {
          float acc = ekv_dot8(qv[r], kr[u], 0.f);
          acc = ekv_group_sum<LPR>(acc);
          s[u] = (PHYS ? !((dead8 >> u) & 1u) : (j0 + u < t1)) ? acc / a.sm_div : EKV_NEG_INF;
        }
      }
      // export the raw logits: lane `sub` of the group owns row j0+sub -> 8 consecutive floats per group
      // (D = 32 has only 4 lanes per row: each lane then owns rows sub and sub + 4)
      constexpr int NST = (KU + LPR - 1) / LPR;
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const int mu = sub + st * LPR;
        float mine = s[0];
#pragma unroll
        for (int u = 1; u < KU; ++u) mine = (mu == u) ? s[u] : mine;
        if (logit_out != nullptr && mu < KU && r < nrep && (PHYS ? !((dead8 >> mu) & 1u) : (j0 + mu < t1)))
          logit_out[(size_t)r * logit_stride + j0 + mu] = mine;
      }
      float mx = s[0];
#pragma unroll
      for (int u = 1; u < KU; ++u) mx = fmaxf(mx, s[u]);
      const float mn = fmaxf(m[r], mx);
      if (mn == EKV_NEG_INF) continue;  // whole group out of range
      const float alpha = exp2f((m[r] - mn) * EKV_LOG2E);
      l[r] *= alpha;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[r][i] *= alpha;
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const float p = exp2f((s[u] - mn) * EKV_LOG2E);
        l[r] += p;
        ekv_axpy8(p, vr[u], o[r]);
      }
      m[r] = mn;
    }
  };
  int base = t0 + wave * RW;
  if (PHYS) {
    if (t1 > (kNW - 1) * RW) {   // every wave has a first iteration: peel it around the mask completion
      iteration(base, mask_ready);
      base += kNW * RW;
    } else {
      mask_ready();
    }
  }
  for (; base < t1; base += kNW * RW) iteration(base, EkvNoop());
  appended_row();
}

// Combine the G lane groups of a wave (flash-decoding merge over lanes LPR, 2*LPR, ... apart); afterwards every lane
// holds the wave's (m, l, o) for its 8-wide slice of head_dim.
template <int D, int REP>
__device__ __forceinline__ void ekv_decode_wave_combine(float (&m)[REP], float (&l)[REP], float (&o)[REP][8]) {
  using Gm = EkvDecodeGeom<D>;
#pragma unroll
  for (int off = Gm::LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      const float mo = __shfl_xor(m[r], off, 64), lo = __shfl_xor(l[r], off, 64);
      const float mn = fmaxf(m[r], mo);
      const float wa = (m[r] == EKV_NEG_INF) ? 0.f : exp2f((m[r] - mn) * EKV_LOG2E);
      const float wb = (mo == EKV_NEG_INF) ? 0.f : exp2f((mo - mn) * EKV_LOG2E);
      l[r] = l[r] * wa + lo * wb;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[r][i] = o[r][i] * wa + __shfl_xor(o[r][i], off, 64) * wb;
      m[r] = mn;
    }
  }
}

// Wave partials -> LDS (call after ekv_decode_wave_combine, then __syncthreads(), then ekv_decode_reduce).
template <int D, int REP, int NW = 4>
__device__ __forceinline__ void ekv_decode_stash(float* s_part, const float (&m)[REP], const float (&l)[REP],
                                                 const float (&o)[REP][8]) {
  using Gm = EkvDecodeGeom<D, NW>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % Gm::LPR, grp = lane / Gm::LPR;
  if (grp != 0 || sub >= Gm::LIVE) return;
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    float* p = s_part + ((size_t)wave * REP + r) * Gm::PS;
    if (sub == 0) {
      p[0] = m[r];
      p[1] = l[r];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) p[2 + sub * 8 + i] = o[r][i];
  }
}

// Combined (max, sum, o[d]) of query head r over the workgroup's wave partials.
template <int D, int REP, int NW = 4>
__device__ __forceinline__ void ekv_decode_reduce(const float* s_part, int r, int d, float& mm, float& ls, float& os) {
  using Gm = EkvDecodeGeom<D, NW>;
  mm = EKV_NEG_INF;
#pragma unroll
  for (int i = 0; i < Gm::NP; ++i) mm = fmaxf(mm, s_part[((size_t)i * REP + r) * Gm::PS]);
  ls = 0.f;
  os = 0.f;
#pragma unroll
  for (int i = 0; i < Gm::NP; ++i) {
    const float* p = s_part + ((size_t)i * REP + r) * Gm::PS;
    const float w = (p[0] == EKV_NEG_INF) ? 0.f : exp2f((p[0] - mm) * EKV_LOG2E);
    ls += p[1] * w;
    os += p[2 + d] * w;
  }
}
