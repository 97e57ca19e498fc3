// head_dim dispatch and query-block planning of the chunk (q_len > 1) attention kernels.
#include <cstdlib>

#include "ekv_common.h"
#include "ekv_kernels.h"

#define EKV_DECL(d, m) hipError_t ekv_launch_attn_chunk_d##d##_m##m(const EkvAttnArgs&, int, int, hipStream_t, const EkvScoreArgs*);
EKV_DECL(32, 0) EKV_DECL(32, 1) EKV_DECL(32, 2) EKV_DECL(64, 0) EKV_DECL(64, 1) EKV_DECL(64, 2)
EKV_DECL(96, 0) EKV_DECL(96, 1) EKV_DECL(96, 2) EKV_DECL(128, 0) EKV_DECL(128, 1) EKV_DECL(128, 2)
#undef EKV_DECL

#define EKW_DECL(d, m) hipError_t ekv_launch_attn_wide_d##d##_m##m(const EkvAttnArgs&, int, int, hipStream_t, const EkvScoreArgs*);
EKW_DECL(64, 0) EKW_DECL(64, 2) EKW_DECL(128, 0) EKW_DECL(128, 2)
#undef EKW_DECL
#define EKW_DECL(d, m) hipError_t ekv_launch_attn_wide_rope_d##d##_m##m(const EkvAttnArgs&, int, int, hipStream_t, const EkvScoreArgs*);
EKW_DECL(64, 0) EKW_DECL(64, 2) EKW_DECL(128, 0) EKW_DECL(128, 2)
#undef EKW_DECL

// Two-pass scheme (16x16 kernel: statistics pass + exact pass with in-kernel column sums, ekv_attn_chunk.inc; wide-block kernel: one
// pass for output + row statistics, then a K-only column-sum pass, ekv_attn_wide.inc) for scored chunk
// steps.  It trades one extra read of K (and a third MFMA product) for the rep x n x T logits never touching HBM: at rep*n = 96
// the logits are 384 B per key against 512 B of K + V, written once and read once.  Measured on MI355X (docs/TUNING.md §8): in
// round 1 the one-pass path won at every BASELINE shape (C4 1.89 vs 2.31 ms); with the round-2 instruction diet of the MFMA
// kernel the two passes win from ~40 query rows up (C4: 1.17 vs 1.48 ms per step, stride 64: 0.43 vs 0.49), so `auto` picks
// them there.  Not with rope-on-read (every product is three MFMAs on the hi/lo pairs: C5 1.98 vs 2.59 ms).
// ekv_step.two_pass = 1 / -1 selects a scheme explicitly (every golden case runs under both).  tova needs the last query row
// itself, not column sums, and always uses the one-pass kernel.
// Round 4: with rope-on-read the two passes run on the wide-block kernel's RoPE variants (head_dim 64 / 128: K rotated in LDS, no
// logits in HBM) from the same 40 rows; head_dim 32 keeps the one-pass 16x16 kernel.
bool ekv_chunk_two_pass(int head_dim, int rep, int q_len, int policy, bool scored, bool accumulate, bool rope, int mode) {
  const bool rep_ok = rep == 1 || rep == 2 || rep == 4 || rep == 8 || rep == 16;   // rep query heads share a 16-lane row
  const bool can = q_len > 1 && scored && accumulate && policy != EKV_POLICY_TOVA && rep_ok;
  if (!can || mode < 0) return false;
  // measured crossover: 32 rows 0.34 (one pass) vs 0.36 ms, 48 rows 0.42 vs 0.41 ms.  RoPE-on-read steps keep the two passes too (round 5,
  // end): with the logits exported by the wide kernel's one pass and swept by the scorer, a configs[4] step is 2190 vs 2205 us
  return mode > 0 || ((!rope || head_dim == 64 || head_dim == 128) && rep * q_len >= 40);
}

bool ekv_attn_chunk_supported(int head_dim, int rep, int q_len) {
  return (head_dim == 32 || head_dim == 64 || head_dim == 96 || head_dim == 128) && rep >= 1 && rep <= 128 && q_len >= 1;
}

// A query block is <= 128 GQA-folded rows (rep x qb_rows).  qpw = 1 or 2: 16-row query tiles per wave of a 4-wave
// workgroup (<= 32 / <= 64 rows); qpw = 4 selects the 8-wave workgroup (2 tiles per wave x 4 query-tile waves, <= 128 rows).
void ekv_chunk_blocks(int rep, int q_len, int* qb_rows, int* n_qblocks, int* qpw) {
  int rows = q_len;
  if (rep * q_len > 128) rows = 128 / rep > 0 ? 128 / rep : 1;   // (64-row blocks on 4-wave workgroups: 324 vs 378 TFLOP/s on the dense prefix)
  // (65..128 rows stay ONE block on the 8-wave workgroup: two 4-wave blocks of <= 64 rows, even XCD-local so that the second K/V
  // read is an L2 hit, measured 1.55 vs 1.19 ms per C4 step)
  *qb_rows = rows;
  *n_qblocks = (q_len + rows - 1) / rows;
  const int r = rep * rows;
  *qpw = r <= 32 ? 1 : (r <= 64 ? 2 : 4);
}

// Wide query blocks (33..128 GQA-folded rows) run on the 32x32x16 kernel of ekv_attn_wide.inc: the dense prefix and every wide
// strided chunk step are bound by the MFMA kernel itself, not by HBM (docs/TUNING.md §3.5).  EKV_NO_WIDE=1 in the environment keeps
// the 16x16x32 kernel (A/B measurements on one box).
bool ekv_chunk_wide(int head_dim, int rep, int q_len, bool rope, bool two_pass, bool wants_logits) {
  static const bool off = [] { const char* e = std::getenv("EKV_NO_WIDE"); return e != nullptr && e[0] == '1'; }();
  if (off || q_len < 2 || (head_dim != 64 && head_dim != 128)) return false;
  int qb_rows, n_qblocks, qpw;
  ekv_chunk_blocks(rep, q_len, &qb_rows, &n_qblocks, &qpw);
  if (qpw < 2) return false;                                  // <= 32 rows: HBM-bound shapes, the small-tile kernels
  if (two_pass) return rep == 1 || rep == 2 || rep == 4 || rep == 8 || rep == 16;   // the column-sum pass folds the rep query heads in registers
  // only the RoPE builds export logits, and only for single-block steps (every key of the range is visited) of a power-of-two GQA factor
  return !wants_logits || (rope && n_qblocks == 1 && (rep == 1 || rep == 2 || rep == 4 || rep == 8 || rep == 16));
}

// rope_on_read: q' = q*cos[pos] + rotate_half(q)*sin[pos] with pos = T - n + i (llama_patch.py:311, :326), once per step,
// stored as an fp16 pair hi + lo (q' is an fp32 product; hi alone would cost ~5e-4 relative on the logits).
__global__ void __launch_bounds__(256) ekv_rope_q_kernel(const EkvAttnArgs a, int D, int n_rows) {
  // 256 / (D / 4) rows per workgroup, four consecutive d per thread (round 4: one 128-thread workgroup per row was 56 us of launch
  // overhead per configs[4] step — 153 600 workgroups)
  const int tpr = D / 4, rpb = 256 / tpr;
  const int r_in = blockIdx.x * rpb + threadIdx.x / tpr;       // (q head, query) of this layer
  if (r_in >= n_rows || (int)threadIdx.x >= rpb * tpr) return;      // (head_dim 96: 10 rows of 24 threads)
  const size_t row = (size_t)blockIdx.y * n_rows + r_in;       // (layer, q head, query)
  const int i = r_in % a.q_len;
  const int pos = a.n_slots - a.q_len + i;
  const __half* q = a.q + (size_t)blockIdx.y * n_rows * D + (size_t)(r_in / a.q_len) * a.q_hs + (size_t)i * a.q_ts;      // (ekv_step.q_*_stride)
  const int d0 = (threadIdx.x % tpr) * 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int d = d0 + e;
    const int dp = d < D / 2 ? d + D / 2 : d - D / 2;
    const float x = __half2float(q[d]), y = __half2float(q[dp]);
    const float qr = x * a.rope_cos[(size_t)pos * D + d] + (d < D / 2 ? -y : y) * a.rope_sin[(size_t)pos * D + d];
    const _Float16 hi = (_Float16)qr;
    reinterpret_cast<_Float16*>(a.q_rot_hi)[row * D + d] = hi;
    reinterpret_cast<_Float16*>(a.q_rot_lo)[row * D + d] = (_Float16)(qr - (float)hi);
  }
}

// 65..128-row blocks (qpw code 4): 8 waves x 2 query tiles — except the ONE-PASS kernel without rope-on-read, which runs SIXTEEN
// waves x 1 query tile (kernel code 8): <= 128 VGPRs per wave, so four waves per SIMD instead of two; the tile loop is a chain of
// LDS / MFMA / VALU latencies between two barriers and two waves per SIMD do not hide it (dense prefix 4906 tokens 411 -> 445
// TFLOP/s, 2048 tokens 319 -> 392).  The exact pass of the two-pass scheme stays on 8 waves: with 16 it writes twice the
// column-sum partial rows, which costs the scorer more than the attention kernel gains (C4 step 1.16 -> 1.25 ms, measured).
static int kernel_code(int qpw, bool rope, int mode) { return (qpw == 4 && !rope && mode == 0) ? 8 : qpw; }

// partial column-sum rows the exact pass writes per (head, query block) = its query-tile waves
int ekv_chunk_col_parts(int qpw, bool rope) { (void)rope; return qpw == 4 ? 4 : 2; }

// fuse_sc != nullptr: one-pass step with unsplit heads whose scorer runs as the tail of the attention kernel (no second launch)
// The scorer as the tail of the wide column-sum pass holds the score rows of a head in registers (24 columns per thread of a
// 256-thread workgroup) and the selection keys in the pass's tile buffers; only heads whose column sums come from ONE workgroup
// (a last-arriver election for split heads was built and measured slower than the stand-alone scorer: ekv_wide_tail.h)
bool ekv_wide_tail_supported(int W, int n_wg) {
  static const bool off = [] { const char* e = std::getenv("EKV_NO_WIDE_TAIL"); return e != nullptr && e[0] == '1'; }();     // (A/B switch)
  return !off && W >= 1 && W <= 24 * 256 && n_wg == 1;
}

// Kernel launches of ekv_launch_attn_chunk below for the same arguments: the query rotation of the 16x16 RoPE path, one launch per
// pass of the wide kernel (`passes` bits), two for the 16x16 two-pass scheme.  Keep next to the launch code.
int ekv_attn_chunk_launches(const EkvAttnArgs& a, int head_dim, bool two_pass, int passes) {
  const bool rope = a.rope_cos != nullptr;
  const bool wide = ekv_chunk_wide(head_dim, a.n_q_heads / a.n_kv_heads, a.q_len, rope, two_pass, a.logits != nullptr);
  if (wide) return ((passes & 1) ? 1 : 0) + ((two_pass && (passes & 2)) ? 1 : 0);
  return (rope ? 1 : 0) + (two_pass ? 2 : 1);
}

hipError_t ekv_launch_attn_chunk(const EkvAttnArgs& a, int head_dim, int layer_count, bool two_pass, hipStream_t s,
                                 const EkvScoreArgs* fuse_sc, int passes, const EkvScoreArgs* tail_sc) {
  if (two_pass && fuse_sc != nullptr) return hipErrorInvalidValue;
  if (tail_sc != nullptr && !(two_pass && (passes & 2))) return hipErrorInvalidValue;
  int qb_rows, n_qblocks, qpw;
  ekv_chunk_blocks(a.n_q_heads / a.n_kv_heads, a.q_len, &qb_rows, &n_qblocks, &qpw);
  if (qb_rows != a.qb_rows || n_qblocks != a.n_qblocks) return hipErrorInvalidValue;
  const bool rope = a.rope_cos != nullptr;
  const bool wide = ekv_chunk_wide(head_dim, a.n_q_heads / a.n_kv_heads, a.q_len, rope, two_pass, a.logits != nullptr);
  if (rope && !wide) {      // (the wide-block kernel rotates its query rows itself, in the lane that holds them)
    if (a.q_rot_hi == nullptr || a.q_rot_lo == nullptr) return hipErrorInvalidValue;
    const int n_rows = a.n_q_heads * a.q_len, rpb = 256 / (head_dim / 4);
    hipLaunchKernelGGL(ekv_rope_q_kernel, dim3((n_rows + rpb - 1) / rpb, layer_count), dim3(256), 0, s, a, head_dim, n_rows);
  }
  if (wide) {
    if (fuse_sc != nullptr || (two_pass && (a.stats == nullptr || a.colsum == nullptr || a.n_col_parts < 1 || a.n_col_parts > n_qblocks))) return hipErrorInvalidValue;
    if (!two_pass && a.stats != nullptr) return hipErrorInvalidValue;      // (mode 0 writes row statistics whenever the array is there)
    const int nwq = qpw == 4 ? 4 : 2;
    // one pass over K and V (output, and for a scored step every row's softmax statistics), then — scored steps — the column-sum
    // pass over K
#define EKW_GO(d, m, aa, shape, t) (rope ? ekv_launch_attn_wide_rope_d##d##_m##m(aa, shape, layer_count, s, t) : ekv_launch_attn_wide_d##d##_m##m(aa, shape, layer_count, s, t))
    hipError_t e = hipSuccess;
    if (passes & 1) {
      // a launch of at most one workgroup per CU (a layer-per-call model) runs 65..128-row blocks on 128-key tiles, 8 waves
      static const bool no_big = [] { const char* ev = std::getenv("EKV_NO_BIG_TILE"); return ev != nullptr && ev[0] == '1'; }();     // (A/B switch)
      // (... and key ranges long enough to hold several 128-key tiles: measured per one-layer call, 64-key / 128-key tiles — 96 rows x 640 keys
      //  per split 43.2 / 38.2 us; 64 rows x 272 keys 24.5 / 26.1; 32 layers x 8 KV heads x 1248 keys unsplit (configs[2]) 43.1 / 41.5)
      const bool small = (size_t)layer_count * a.n_kv_heads * a.n_split * a.n_qblocks <= 256 && a.rows_per_split >= 512;
      const int shape0 = (!rope && small && !no_big) ? (nwq == 4 ? 8 : 9) : nwq;      // workgroup-shape code of ekv_attn_wide.inc's entry (8 / 9: 128-key tiles)
      e = head_dim == 128 ? EKW_GO(128, 0, a, shape0, nullptr) : EKW_GO(64, 0, a, shape0, nullptr);
    }
    if (two_pass && (passes & 2) && e == hipSuccess) {
      EkvAttnArgs a2 = a;
      a2.score_tail = tail_sc != nullptr ? 1 : 0;
      e = head_dim == 128 ? EKW_GO(128, 2, a2, nwq, tail_sc) : EKW_GO(64, 2, a2, nwq, tail_sc);
    }
#undef EKW_GO
    return e;
  }
  if (tail_sc != nullptr) return hipErrorInvalidValue;
  if (two_pass && (a.stats == nullptr || a.colsum == nullptr || a.n_col_parts != ekv_chunk_col_parts(qpw, rope) * n_qblocks)) return hipErrorInvalidValue;
#define EKV_GO(d, m) ekv_launch_attn_chunk_d##d##_m##m(a, kernel_code(qpw, rope, m), layer_count, s, fuse_sc)
  hipError_t e = hipSuccess;
  switch (head_dim) {
    case 32: e = two_pass ? EKV_GO(32, 1) : EKV_GO(32, 0); if (two_pass && e == hipSuccess) e = EKV_GO(32, 2); break;
    case 64: e = two_pass ? EKV_GO(64, 1) : EKV_GO(64, 0); if (two_pass && e == hipSuccess) e = EKV_GO(64, 2); break;
    case 96: e = two_pass ? EKV_GO(96, 1) : EKV_GO(96, 0); if (two_pass && e == hipSuccess) e = EKV_GO(96, 2); break;
    case 128: e = two_pass ? EKV_GO(128, 1) : EKV_GO(128, 0); if (two_pass && e == hipSuccess) e = EKV_GO(128, 2); break;
    default: e = hipErrorInvalidValue;
  }
#undef EKV_GO
  return e;
}

// ---- small-row chunk step with the logits in LDS (ekv_chunk_lds.inc) -------------------------------------------------------
size_t ekv_chunk_lds_bytes_d32(int, int, int);
size_t ekv_chunk_lds_bytes_d64(int, int, int);
size_t ekv_chunk_lds_bytes_d128(int, int, int);
hipError_t ekv_launch_chunk_lds_d32(const EkvAttnArgs&, const EkvScoreArgs&, int, hipStream_t);
hipError_t ekv_launch_chunk_lds_d64(const EkvAttnArgs&, const EkvScoreArgs&, int, hipStream_t);
hipError_t ekv_launch_chunk_lds_d128(const EkvAttnArgs&, const EkvScoreArgs&, int, hipStream_t);

// Eligible: a scored, accumulating chunk step (plain keys, score rows over the whole cache) with at most 8 GQA-folded query
// rows whose logits fit LDS next to a second workgroup of the CU (<= 80 KB), one victim set per head.
bool ekv_chunk_lds_supported(const ekv_bank* bank, const ekv_step* st, int phys_extent, bool scored) {
  const int rep = bank->n_q_heads / bank->n_kv_heads;
  if (st->q_len < 2 || rep * st->q_len > 8 || (rep & (rep - 1)) != 0 || !scored || !st->accumulate || st->rope_on_read || st->score_off != 0) return false;
  if (st->phases != 0 || st->n_split == -1 || st->two_pass != 0 || !st->causal) return false;   // (two_pass = -1: "exported logits")
  if (st->policy == EKV_POLICY_TOVA && st->tova_head_mean) return false;   // needs every head of the layer first
  if (st->n_slots > 10 * 256 || st->n_evict >= st->n_slots || st->n_evict > 16) return false;
  size_t lds = 1u << 30;
  switch (bank->head_dim) {
    case 32: lds = ekv_chunk_lds_bytes_d32(rep * st->q_len, phys_extent, st->n_slots); break;
    case 64: lds = ekv_chunk_lds_bytes_d64(rep * st->q_len, phys_extent, st->n_slots); break;
    case 128: lds = ekv_chunk_lds_bytes_d128(rep * st->q_len, phys_extent, st->n_slots); break;
  }
  return lds <= 80 * 1024;
}

hipError_t ekv_launch_chunk_lds(const EkvAttnArgs& a, const EkvScoreArgs& sc, int head_dim, int layer_count, hipStream_t s) {
  switch (head_dim) {
    case 32: return ekv_launch_chunk_lds_d32(a, sc, layer_count, s);
    case 64: return ekv_launch_chunk_lds_d64(a, sc, layer_count, s);
    case 128: return ekv_launch_chunk_lds_d128(a, sc, layer_count, s);
  }
  return hipErrorInvalidValue;
}
