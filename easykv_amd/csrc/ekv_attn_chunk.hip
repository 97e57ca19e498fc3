// head_dim dispatch and query-block planning of the chunk (q_len > 1) attention kernels.
#include "ekv_common.h"
#include "ekv_kernels.h"

hipError_t ekv_launch_attn_chunk_d32(const EkvAttnArgs&, int, int, hipStream_t);
hipError_t ekv_launch_attn_chunk_d64(const EkvAttnArgs&, int, int, hipStream_t);
hipError_t ekv_launch_attn_chunk_d128(const EkvAttnArgs&, int, int, hipStream_t);

bool ekv_attn_chunk_supported(int head_dim, int rep, int q_len) {
  return (head_dim == 32 || head_dim == 64 || head_dim == 128) && rep >= 1 && rep <= 128 && q_len >= 1;
}

// A query block is <= 128 GQA-folded rows (rep x qb_rows); qpw = query tiles per wave.
void ekv_chunk_blocks(int rep, int q_len, int* qb_rows, int* n_qblocks, int* qpw) {
  int rows = q_len;
  if (rep * q_len > 128) rows = 128 / rep > 0 ? 128 / rep : 1;
  *qb_rows = rows;
  *n_qblocks = (q_len + rows - 1) / rows;
  const int r = rep * rows;
  *qpw = r <= 32 ? 1 : (r <= 64 ? 2 : 4);
}

// rope_on_read: q' = q*cos[pos] + rotate_half(q)*sin[pos] with pos = T - n + i (llama_patch.py:311, :326), once per step,
// stored as an fp16 pair hi + lo (q' is an fp32 product; hi alone would cost ~5e-4 relative on the logits).
__global__ void __launch_bounds__(128) ekv_rope_q_kernel(const EkvAttnArgs a, int D) {
  const size_t row = (size_t)blockIdx.y * a.n_q_heads * a.q_len + blockIdx.x;   // (layer, q head, query)
  const int i = blockIdx.x % a.q_len;
  const int pos = a.n_slots - a.q_len + i;
  const __half* q = a.q + row * D;
  for (int d = threadIdx.x; d < D; d += 128) {
    const int dp = d < D / 2 ? d + D / 2 : d - D / 2;
    const float x = __half2float(q[d]), y = __half2float(q[dp]);
    const float qr = x * a.rope_cos[(size_t)pos * D + d] + (d < D / 2 ? -y : y) * a.rope_sin[(size_t)pos * D + d];
    const _Float16 hi = (_Float16)qr;
    reinterpret_cast<_Float16*>(a.q_rot_hi)[row * D + d] = hi;
    reinterpret_cast<_Float16*>(a.q_rot_lo)[row * D + d] = (_Float16)(qr - (float)hi);
  }
}

hipError_t ekv_launch_attn_chunk(const EkvAttnArgs& a, int head_dim, int layer_count, hipStream_t s) {
  if (a.rope_cos != nullptr) {
    if (a.q_rot_hi == nullptr || a.q_rot_lo == nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ekv_rope_q_kernel, dim3(a.n_q_heads * a.q_len, layer_count), dim3(128), 0, s, a, head_dim);
  }
  int qb_rows, n_qblocks, qpw;
  ekv_chunk_blocks(a.n_q_heads / a.n_kv_heads, a.q_len, &qb_rows, &n_qblocks, &qpw);
  if (qb_rows != a.qb_rows || n_qblocks != a.n_qblocks) return hipErrorInvalidValue;
  switch (head_dim) {
    case 32: return ekv_launch_attn_chunk_d32(a, qpw, layer_count, s);
    case 64: return ekv_launch_attn_chunk_d64(a, qpw, layer_count, s);
    case 128: return ekv_launch_attn_chunk_d128(a, qpw, layer_count, s);
    default: return hipErrorInvalidValue;
  }
}
