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

hipError_t ekv_launch_attn_chunk(const EkvAttnArgs& a, int head_dim, int layer_count, hipStream_t s) {
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
