// Shared device helpers for the easykv_amd HIP kernels (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define EKV_WAVE 64
#define EKV_LOG2E 1.4426950408889634f
#define EKV_NEG_INF (-__builtin_inff())

typedef _Float16 ekv_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 ekv_h8 __attribute__((ext_vector_type(8)));
typedef unsigned int ekv_u4 __attribute__((ext_vector_type(4)));
typedef float ekv_f2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float ekv_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

// All-reduce (sum) over aligned groups of LPR consecutive lanes, LPR in {4,8,16}; every lane gets the total.
// quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror, row_mirror: one fused v_add_f32_dpp each.
template <int LPR>
__device__ __forceinline__ float ekv_group_sum(float x) {
  x += ekv_dpp<0xB1>(x);
  x += ekv_dpp<0x4E>(x);
  if (LPR >= 8) x += ekv_dpp<0x141>(x);
  if (LPR >= 16) x += ekv_dpp<0x140>(x);
  return x;
}

__device__ __forceinline__ float ekv_dot8(const uint4& a, const uint4& b, float acc) {
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(ekv_h2, a.x), __builtin_bit_cast(ekv_h2, b.x), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(ekv_h2, a.y), __builtin_bit_cast(ekv_h2, b.y), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(ekv_h2, a.z), __builtin_bit_cast(ekv_h2, b.z), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(ekv_h2, a.w), __builtin_bit_cast(ekv_h2, b.w), acc, false);
  return acc;
}

__device__ __forceinline__ void ekv_axpy8(float p, const uint4& v, float (&o)[8]) {
  const ekv_h8 h = __builtin_bit_cast(ekv_h8, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = fmaf(p, (float)h[i], o[i]);
}

__device__ __forceinline__ float ekv_wave_max(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
  return x;
}
__device__ __forceinline__ float ekv_wave_sum(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}
__device__ __forceinline__ unsigned long long ekv_wave_min_u64(unsigned long long x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long y = __shfl_xor(x, off, 64);
    x = y < x ? y : x;
  }
  return x;
}

// Order-preserving float -> uint32 key (ascending); every NaN maps to the largest key, which is how
// torch.topk(largest=False) ranks NaN.
__device__ __forceinline__ uint32_t ekv_fkey(float x) {
  if (x != x) return 0xFFFFFFFFu;
  const uint32_t b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// Fold the key-range-split partials (m, l, o[D]) of one query row into o[d] / l.  All loads of a pass are issued
// together (BATCH splits per round trip): a naive loop serialises one L2 round trip per split (~10 us for 17 splits).
// `mm` / `ls` return the row's softmax statistics (max logit, sum of exp(logit - max)) over all splits.
template <int BATCH = 32>
__device__ __forceinline__ float ekv_fold_partials(const float* p0, int n_split, int PS, int d, float& mm, float& ls) {
  mm = EKV_NEG_INF;
  ls = 0.f;
  float os = 0.f;
  for (int s0 = 0; s0 < n_split; s0 += BATCH) {
    float mv[BATCH], lv[BATCH], ov[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const bool ok = s0 + i < n_split;
      const float* p = p0 + (size_t)(ok ? s0 + i : 0) * PS;
      const float pm = p[0], pl = p[1], po = p[2 + d];   // unconditional loads (clamped pointer), masked afterwards
      mv[i] = ok ? pm : EKV_NEG_INF;
      lv[i] = ok ? pl : 0.f;
      ov[i] = ok ? po : 0.f;
    }
    float mb = mm;
#pragma unroll
    for (int i = 0; i < BATCH; ++i) mb = fmaxf(mb, mv[i]);
    const float rescale = (mm == EKV_NEG_INF) ? 0.f : exp2f((mm - mb) * EKV_LOG2E);
    ls *= rescale;
    os *= rescale;
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const float w = (mv[i] == EKV_NEG_INF) ? 0.f : exp2f((mv[i] - mb) * EKV_LOG2E);
      ls += lv[i] * w;
      os += ov[i] * w;
    }
    mm = mb;
  }
  return os / ls;
}

template <int BATCH = 32>
__device__ __forceinline__ float ekv_fold_partials(const float* p0, int n_split, int PS, int d) {
  float mm, ls;
  return ekv_fold_partials<BATCH>(p0, n_split, PS, d, mm, ls);
}

// Batch = loads issued together per round trip.  Masked entries of a batch add exact zeros, so any batch that covers all
// partials gives the same result; a batch much wider than n_split only wastes clamped loads (C3: 32-wide batches over 8
// partials were 65 % of its scorer).  Batches are capped at 16 (8 for 1024-thread blocks): a 32-wide batch keeps 96 values live,
// and under a 512-thread launch bound (128 VGPRs) hipcc then serialises it into one exposed round trip PER SPLIT — 17 of them,
// 8.5 of the 19 us of the per-layer decode scorer.
template <int MAXB = 16>
__device__ __forceinline__ float ekv_fold_partials_auto(const float* p0, int n_split, int PS, int d, float& mm, float& ls) {
  if (MAXB <= 8 || n_split <= 8) return ekv_fold_partials<8>(p0, n_split, PS, d, mm, ls);
  if (n_split <= 16 || n_split > 24) return ekv_fold_partials<16>(p0, n_split, PS, d, mm, ls);
  return ekv_fold_partials<24>(p0, n_split, PS, d, mm, ls);   // 17..24 partials (decode, T ~ 2k: 17 splits) in ONE round trip
}
template <int MAXB = 16>
__device__ __forceinline__ float ekv_fold_partials_auto(const float* p0, int n_split, int PS, int d) {
  float mm, ls;
  return ekv_fold_partials_auto<MAXB>(p0, n_split, PS, d, mm, ls);
}

// Agent-scope store of a partial that ANOTHER workgroup of the same launch will read (global_store ... sc1: written through to the
// memory side, where the reader's sc1 loads find it).
__device__ __forceinline__ void ekv_store_sc1(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same fold over partials written by OTHER workgroups of the same launch (in-kernel fold by the last-arriving split): the
// loads are buffer loads with the sc1 cache policy (agent scope: served by memory-side caches, never by this CU's L1 or a stale
// line) from a wave-UNIFORM descriptor + per-lane offsets, so a batch still goes out as one round trip.  (__hip_atomic_load is
// issued one load at a time with a wait behind each: 51 serialised round trips made the fold cost 25 us; a descriptor built
// from a per-lane pointer makes hipcc wrap every load in a waterfall loop.)  Arithmetic identical to ekv_fold_partials.
template <int BATCH>
__device__ __forceinline__ float ekv_fold_partials_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned row_off, int n_split, int PS, int d) {
  float mm = EKV_NEG_INF, ls = 0.f, os = 0.f;
  for (int s0 = 0; s0 < n_split; s0 += BATCH) {
    float mv[BATCH], lv[BATCH], ov[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const bool ok = s0 + i < n_split;
      const unsigned off = (row_off + (unsigned)(ok ? s0 + i : 0) * (unsigned)PS) * 4u;
      const float pm = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 16));
      const float pl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 4u, 0, 16));
      const float po = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 8u + 4u * (unsigned)d, 0, 16));
      mv[i] = ok ? pm : EKV_NEG_INF;
      lv[i] = ok ? pl : 0.f;
      ov[i] = ok ? po : 0.f;
    }
    float mb = mm;
#pragma unroll
    for (int i = 0; i < BATCH; ++i) mb = fmaxf(mb, mv[i]);
    const float rescale = (mm == EKV_NEG_INF) ? 0.f : exp2f((mm - mb) * EKV_LOG2E);
    ls *= rescale;
    os *= rescale;
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const float w = (mv[i] == EKV_NEG_INF) ? 0.f : exp2f((mv[i] - mb) * EKV_LOG2E);
      ls += lv[i] * w;
      os += ov[i] * w;
    }
    mm = mb;
  }
  return os / ls;
}
__device__ __forceinline__ float ekv_fold_partials_buf_auto(__amdgpu_buffer_rsrc_t rsrc, unsigned row_off, int n_split, int PS, int d) {
  if (n_split <= 8) return ekv_fold_partials_buf<8>(rsrc, row_off, n_split, PS, d);       // (same batch choice as
  if (n_split <= 16 || n_split > 24) return ekv_fold_partials_buf<16>(rsrc, row_off, n_split, PS, d);   //  ekv_fold_partials_auto<16>)
  return ekv_fold_partials_buf<24>(rsrc, row_off, n_split, PS, d);
}

// Barrier that only orders LDS traffic: global loads of the next super-tile stay in flight across it
// (__syncthreads() would drain vmcnt(0) whenever a global store may be pending).
__device__ __forceinline__ void ekv_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

static inline __host__ __device__ size_t ekv_align(size_t x, size_t a) { return (x + a - 1) / a * a; }
