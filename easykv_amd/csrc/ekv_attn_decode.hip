// K1: single-token decode attention over the retained slots + logit export for the scorer.
//
// Replaces easykv/llama_patch.py:198-222 (mistral_patch.py:144-169) for q_len == 1, the K/V append of
// HF DynamicCache.update (call site llama_patch.py:193-196) and, with rope_on_read, the streaming
// rotation of llama_patch.py:310-327.
//
// Mapping (D = 128): a K/V row is 256 B; 16 lanes x 16 B read one row fully coalesced along the
// head-dim axis, a wave64 load instruction covers 4 rows, each lane keeps U = 8 K and 8 V loads in
// flight.  QK^T is v_dot2c_f32_f16 + a 4-step fused v_add_f32_dpp butterfly over the 16 lanes of a
// row; softmax is lane-local online (exp2), so there is no cross-row traffic inside the loop.  One
// workgroup serves the REP query heads of one KV head (GQA: the K/V rows are read once).
// HBM-bound: 2*H*T*D*2 bytes per layer-step; nothing is re-read.
#include "ekv_common.h"
#include "ekv_kernels.h"

namespace {

constexpr int kNW = 4;  // waves per workgroup
constexpr int kU = 8;   // rows in flight per lane group (K and V each)

template <int D, int REP, bool ROPE>
__global__ void __launch_bounds__(256) ekv_attn_decode_kernel(const EkvAttnArgs a) {
  constexpr int LPR = D / 8;    // lanes per row
  constexpr int G = 64 / LPR;   // rows per wave-load
  constexpr int RW = G * kU;    // rows per wave per iteration
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* s_slot = reinterpret_cast<int32_t*>(smem);
  float* s_part = reinterpret_cast<float*>(smem + ekv_align((size_t)a.rows_per_split * 4, 16));

  const int split = blockIdx.x, h = blockIdx.y, ll = blockIdx.z;
  const int gl = a.layer_begin + ll;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane % LPR, grp = lane / LPR;
  const int t0 = split * a.rows_per_split;
  const int t1 = min(a.n_slots, t0 + a.rows_per_split);
  const size_t head_row = ((size_t)gl * a.n_kv_heads + h) * a.cap;
  const int t_new = a.n_slots - 1;  // the appended position

  for (int i = tid; i < t1 - t0; i += 256) s_slot[i] = a.slot_of_pos[head_row + t0 + i];

  uint4 qv[REP];
  float qf[REP][8], qr[REP][8];  // ROPE: rotated query and its rotate_half partner, fp32
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    const __half* qp = a.q + ((size_t)ll * a.n_q_heads + h * REP + r) * D;
    qv[r] = reinterpret_cast<const uint4*>(qp)[sub];
    if (ROPE) {
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
      // partner values q'[d +- D/2] live in lane sub +- LPR/2 of the same row group
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float other = __shfl_xor(qf[r][i], LPR / 2, 64);
        qr[r][i] = (sub < LPR / 2) ? other : -other;
      }
    }
  }
  __syncthreads();

  const __half* k_new_row = a.k_new + ((size_t)ll * a.n_kv_heads + h) * D;
  const __half* v_new_row = a.v_new + ((size_t)ll * a.n_kv_heads + h) * D;
  if (t_new >= t0 && t_new < t1 && wave == 0 && grp == 0) {  // append: the new row goes into the recycled slot
    const size_t off = (head_row + s_slot[t_new - t0]) * D;
    reinterpret_cast<uint4*>(a.k_w + off)[sub] = reinterpret_cast<const uint4*>(k_new_row)[sub];
    reinterpret_cast<uint4*>(a.v_w + off)[sub] = reinterpret_cast<const uint4*>(v_new_row)[sub];
  }

  float m[REP], l[REP], o[REP][8];
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    m[r] = EKV_NEG_INF;
    l[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[r][i] = 0.f;
  }

  for (int base = t0 + wave * RW; base < t1; base += kNW * RW) {
    uint4 kr[kU], vr[kU];
    const int j0 = base + grp * kU;
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int j = j0 + u;
      const bool valid = j < t1;
      const int jj = valid ? j : t1 - 1;
      const int row = s_slot[jj - t0];
      // the appended position is read from k_new/v_new (pointer select, no branch in the hot loop)
      const bool is_new = jj == t_new;
      const __half* kp = is_new ? k_new_row : a.k + (head_row + row) * D;
      const __half* vp = is_new ? v_new_row : a.v + (head_row + row) * D;
      kr[u] = reinterpret_cast<const uint4*>(kp)[sub];
      vr[u] = reinterpret_cast<const uint4*>(vp)[sub];
    }
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      float s[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        float acc;
        if (ROPE) {
          // q'.(k*cos_j + rotate_half(k)*sin_j) == k.(q'*cos_j + qr*sin_j): rotate the query side per key
          const int j = min(j0 + u, t1 - 1);
          const float4* c4 = reinterpret_cast<const float4*>(a.rope_cos + (size_t)j * D + sub * 8);
          const float4* s4 = reinterpret_cast<const float4*>(a.rope_sin + (size_t)j * D + sub * 8);
          const float4 c0 = c4[0], c1 = c4[1], s0 = s4[0], s1 = s4[1];
          const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
          const float ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
          const ekv_h8 kh = __builtin_bit_cast(ekv_h8, kr[u]);
          acc = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) acc = fmaf((float)kh[i], fmaf(qr[r][i], ss[i], qf[r][i] * cc[i]), acc);
        } else {
          acc = ekv_dot8(qv[r], kr[u], 0.f);
        }
        acc = ekv_group_sum<LPR>(acc);
        s[u] = (j0 + u < t1) ? acc / a.sm_div : EKV_NEG_INF;
      }
      // export the raw logits: lane `sub` of the group owns row j0+sub -> 8 consecutive floats per group
      // (D = 32 has only 4 lanes per row: each lane then owns rows sub and sub + 4)
      constexpr int NST = (kU + LPR - 1) / LPR;
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const int mu = sub + st * LPR;
        float mine = s[0];
#pragma unroll
        for (int u = 1; u < kU; ++u) mine = (mu == u) ? s[u] : mine;
        if (mu < kU && j0 + mu < t1)
          a.logits[((size_t)ll * a.n_q_heads + h * REP + r) * a.t_pad + j0 + mu] = mine;
      }
      float mx = s[0];
#pragma unroll
      for (int u = 1; u < kU; ++u) mx = fmaxf(mx, s[u]);
      const float mn = fmaxf(m[r], mx);
      if (mn == EKV_NEG_INF) continue;  // whole group out of range
      const float alpha = exp2f((m[r] - mn) * EKV_LOG2E);
      l[r] *= alpha;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[r][i] *= alpha;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const float p = exp2f((s[u] - mn) * EKV_LOG2E);
        l[r] += p;
        ekv_axpy8(p, vr[u], o[r]);
      }
      m[r] = mn;
    }
  }

  // combine the kNW*G lane-group partials of this workgroup
  constexpr int NP = kNW * G;
  constexpr int PS = D + 2;
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    float* p = s_part + ((size_t)(wave * G + grp) * REP + r) * PS;
    if (sub == 0) {
      p[0] = m[r];
      p[1] = l[r];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) p[2 + sub * 8 + i] = o[r][i];
  }
  __syncthreads();
  for (int idx = tid; idx < REP * D; idx += 256) {
    const int r = idx / D, d = idx % D;
    float mm = EKV_NEG_INF;
    for (int i = 0; i < NP; ++i) mm = fmaxf(mm, s_part[((size_t)i * REP + r) * PS]);
    float ls = 0.f, os = 0.f;
    for (int i = 0; i < NP; ++i) {
      const float* p = s_part + ((size_t)i * REP + r) * PS;
      const float w = (p[0] == EKV_NEG_INF) ? 0.f : exp2f((p[0] - mm) * EKV_LOG2E);
      ls += p[1] * w;
      os += p[2 + d] * w;
    }
    float* dst = a.partials + (((size_t)ll * a.n_q_heads + h * REP + r) * a.n_split + split) * PS;
    if (d == 0) {
      dst[0] = mm;
      dst[1] = ls;
    }
    dst[2 + d] = os;
  }
}

template <int D, int REP>
hipError_t launch(const EkvAttnArgs& a, int layer_count, hipStream_t s) {
  constexpr int G = 64 / (D / 8);
  const size_t lds = ekv_align((size_t)a.rows_per_split * 4, 16) + (size_t)kNW * G * REP * (D + 2) * 4;
  const dim3 grid(a.n_split, a.n_kv_heads, layer_count);
  if (a.rope_cos != nullptr) {
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ekv_attn_decode_kernel<D, REP, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((ekv_attn_decode_kernel<D, REP, true>), grid, dim3(256), lds, s, a);
  } else {
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ekv_attn_decode_kernel<D, REP, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((ekv_attn_decode_kernel<D, REP, false>), grid, dim3(256), lds, s, a);
  }
  return hipGetLastError();
}

template <int D>
hipError_t launch_rep(const EkvAttnArgs& a, int rep, int layer_count, hipStream_t s) {
  switch (rep) {
    case 1: return launch<D, 1>(a, layer_count, s);
    case 2: return launch<D, 2>(a, layer_count, s);
    case 4: return launch<D, 4>(a, layer_count, s);
    case 8: return launch<D, 8>(a, layer_count, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

bool ekv_attn_decode_supported(int head_dim, int rep) {
  return (head_dim == 32 || head_dim == 64 || head_dim == 128) && (rep == 1 || rep == 2 || rep == 4 || rep == 8);
}

hipError_t ekv_launch_attn_decode(const EkvAttnArgs& a, int head_dim, int layer_count, hipStream_t s) {
  const int rep = a.n_q_heads / a.n_kv_heads;
  switch (head_dim) {
    case 32: return launch_rep<32>(a, rep, layer_count, s);
    case 64: return launch_rep<64>(a, rep, layer_count, s);
    case 128: return launch_rep<128>(a, rep, layer_count, s);
    default: return hipErrorInvalidValue;
  }
}
