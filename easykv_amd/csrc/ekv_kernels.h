// Internal launch interface between the C ABI (ekv_abi.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/easykv_hip.h"

// Workspace carve-up for one ekv_step_attend call.
struct EkvWs {
  float* logits;    // [layer_count][Hq][q_len][t_pad]   raw q.k/sm_div of every live position
  float* partials;  // [layer_count][Hq][q_len][n_split][D+2]   (m, l, o[D]) per key-range split
  float* tova_row;  // [layer_count][t_pad]   head-averaged last-query row (tova_head_mean)
  float* big_rows;  // [layer_count][H][3][t_pad] working copies of the score rows when they exceed one CU's LDS, else null
  float* stats;     // two-pass chunk steps (see EkvAttnArgs)
  float* colsum;
  int32_t two_pass, n_col_parts;
  float* row_stats;
  int32_t fold_in_kernel;   // chunk step whose attention kernel writes the final output itself (no partials, no fold)
  int32_t wide;             // chunk step on the wide-query-block kernel (ekv_attn_wide.inc): ONE partial per split, ONE column-sum row
  int32_t resident;         // whole scored chunk step on the logits-resident kernel (ekv_attn_resident.inc): one launch, unsplit
  int32_t fused_nw;     // waves per workgroup the fused decode kernel would use for this launch (4 or 8)
  __half* q_keep;   // deferred wide two-pass chunk steps: [layer_count][Hq][q_len][D] raw queries kept for the flush's column-sum pass
  __half* q_rot;    // rope_on_read chunk steps: [2][layer_count][Hq][q_len][D] rotated queries, fp16 hi then lo
  int32_t t_pad, n_split, rows_per_split;
  int32_t n_partials;   // partials per query row the scorer folds (chunk kernels emit 2 per split)
  int32_t qb_rows, n_qblocks;
  size_t bytes;
};

struct EkvAttnArgs {
  const __half* k;
  const __half* v;
  __half* k_w;  // same buffers, writable (append of the new rows)
  __half* v_w;
  const int32_t* slot_of_pos;
  const __half* q;
  const __half* k_new;
  const __half* v_new;
  float* logits;
  float* partials;
  const float* rope_cos;
  const float* rope_sin;
  __half* q_rot_hi;  // chunk kernels with rope_on_read: queries rotated by ekv_rope_q_kernel (hi + lo fp16 pair)
  __half* q_rot_lo;
  float* row_stats;    // with out_direct, one-pass scored steps: [layer_count][Hq][q_len][2] final (max, sum exp) per query row
  __half* out_direct;  // chunk kernels, unsplit heads: fold the two key halves in the kernel and write the fp16 output here
  float* stats;      // two-pass chunk steps: [layer_count][Hq][q_len][2*n_split][2] (max, sum exp) per key-range half split
  float* colsum;     // two-pass chunk steps: [layer_count][H][n_col_parts][2][t_pad] column sums of pbar and pbar^2
  int32_t n_col_parts;   // = query-tile waves per workgroup (2 or 4) * n_qblocks
  int32_t n_q_heads, n_kv_heads, cap, n_slots, q_len, n_split, rows_per_split, t_pad, layer_begin, causal;
  int32_t qb_rows, n_qblocks;  // chunk kernels: queries per query block, number of query blocks
  int32_t phys_extent;         // fused decode step, physical-order stream: live rows have physical index < phys_extent
  int32_t l_pad;               // fused decode step: pitch of a logits row in LDS (t_pad, or align(phys_extent, 64))
  uint32_t* arrive;            // split decode kernel: non-null = fold the key-range partials in the kernel (last-arriving split of a
                               //   head) and write the fp16 output to out_direct; the bank's counters [n_layers][H], 0 when idle
  float sm_div;
  // deferred column-sum pass of the wide-block kernel (ekv_step.defer_layers, chunk steps): the one pass of a layer keeps its raw
  // queries in q_keep ([layers][Hq][q_len][D], this call's slice); the column-sum pass at the flush reads them as `q` and takes the
  // chunk's own K rows from the cache slots (new_in_cache) instead of k_new
  __half* q_keep;
  int32_t new_in_cache;
  // column-sum pass of the wide-block kernel: 1 = the scorer of the step runs as the tail of this launch (ekv_wide_tail.h; the
  // EkvScoreArgs are the launch's second argument); set only for heads whose column sums ONE workgroup writes
  int32_t score_tail;
  // row strides in elements (ekv_step.*_stride, ABI 8; always filled in: the dense layout is q_ts = D, q_hs = q_len * D, ...): row
  // (layer ll, head hd, token i) of q sits at ((size_t)ll * n_q_heads * q_len) * D + hd * q_hs + i * q_ts, k_new / v_new and out alike
  int32_t q_ts, q_hs, kv_ts, kv_hs, o_ts, o_hs;
  int32_t n_stat_parts;     // column-sum pass: (max, sum) partials per query row in `stats` when that differs from this launch's n_split (0 = n_split)
};

struct EkvScoreArgs {
  int32_t* slot_of_pos;
  float* score_sum;
  float* score_sq;
  float* score_cnt;
  const float* logits;
  const float* partials;
  const float* colsum;   // non-null: column sums from the two-pass chunk kernel replace the logits
  float* big_rows;       // non-null: score rows wider than one CU's LDS — the scorer keeps its working copies of S / Q / C in this
  int big_stride;        //   scratch ([layer_count][H][3][big_stride] floats) and only the selection keys in LDS
  const float* row_stats;   // non-null: final row statistics written by the chunk kernel (its in-kernel fold)
  int32_t n_col_parts;
  float* tova_row;
  __half* out;
  int32_t* evict_ids;
  int32_t n_q_heads, n_kv_heads, head_dim, cap, n_slots, q_len, n_split, t_pad, layer_begin;
  int32_t score_off, policy, accumulate, n_evict, win_lo, win_tail, roco_k1, roco_tail, range_start, tova_head_mean,
      causal;
  float count_add, count_tail_step;
  int32_t skip_fold;   // 1: the attention output was already folded by ekv_fold_kernel (scorer off the critical path)
  int32_t o_ts, o_hs;  // row strides of `out` in elements (see EkvAttnArgs)
  // slot-indexed score rows (ekv_step.phases & EKV_PHASE_SLOT_ROWS, fused decode step; ekv_decode_tail.h): score_sum / score_sq /
  // score_cnt are indexed by physical row, score_cnt holds the count base, birth[row] the order key, slot_state[head] = (g, next birth, threshold hint of the decode step, threshold hint of the chunk_lds kernel — the hints in any layout)
  int32_t* birth;
  float* slot_state;
  float* cnt_tail;        // parked tail of the ordered count row (second half of ekv_bank.birth)
  int32_t slot_tail_ok;   // EKV_PHASE_SLOT_TAIL_OK: the newest `tail` entries are known to have consecutive births
};

EkvWs ekv_plan_workspace(const ekv_bank* bank, const ekv_step* step, void* base);

hipError_t ekv_launch_attn_decode(const EkvAttnArgs& a, int head_dim, int layer_count, hipStream_t s);
// passes (wide-block kernel, two-pass scheme): bit 0 = the one pass (output + row statistics), bit 1 = the column-sum pass
// tail_sc (wide-block kernel, two passes, passes & 2): the step's scorer runs as the tail of the column-sum pass (ekv_wide_tail.h)
hipError_t ekv_launch_attn_chunk(const EkvAttnArgs& a, int head_dim, int layer_count, bool two_pass, hipStream_t s,
                                 const EkvScoreArgs* fuse_sc = nullptr, int passes = 3, const EkvScoreArgs* tail_sc = nullptr);
// kernel launches ekv_launch_attn_chunk issues for these arguments (the dry run's count; lives next to the launch code)
int ekv_attn_chunk_launches(const EkvAttnArgs& a, int head_dim, bool two_pass, int passes);
// can the scorer of a two-pass wide step run as the tail of its column-sum pass: W score columns, n_wg workgroups per head
bool ekv_wide_tail_supported(int W, int n_wg);
// logits-resident scored chunk step (ekv_attn_resident.inc): the whole step of an unsplit head in ONE launch, K and V read once
bool ekv_attn_resident_supported(int head_dim, int rep, int q_len, int n_slots, int W);
hipError_t ekv_launch_attn_resident(const EkvAttnArgs& a, const EkvScoreArgs& sc, int layer_count, hipStream_t s);
size_t ekv_score_lds_bytes_nt256(const EkvScoreArgs& a);
bool ekv_score_rows_exceed_lds(int W, int rows);   // generic scorer: S / Q / C + keys of W columns do not fit 160 KB of LDS
bool ekv_chunk_two_pass(int head_dim, int rep, int q_len, int policy, bool scored, bool accumulate, bool rope, int mode);
// the launch runs on the wide-query-block kernel (32x32x16 MFMA, ekv_attn_wide.inc): 33..128 GQA-folded rows per query block,
// plain or RoPE-on-read keys, head_dim 64 / 128, and either the two-pass scheme (rep in {1, 2, 4, 8, 16}) or a step that exports no logits
bool ekv_chunk_wide(int head_dim, int rep, int q_len, bool rope, bool two_pass, bool wants_logits);
hipError_t ekv_launch_tova_headmean(const EkvScoreArgs& a, int layer_count, hipStream_t s);
hipError_t ekv_launch_score_select(const EkvScoreArgs& a, int layer_count, hipStream_t s);
bool ekv_attn_decode_supported(int head_dim, int rep);
int ekv_decode_fused_nw(int n_heads_in_launch);
bool ekv_decode_fused_supported(int head_dim, int rep, int n_slots, int t_pad, int l_pad, int n_evict, int cap, int nw);
int ekv_fused_logit_pad(const ekv_bank* bank, const ekv_step* st, int t_pad);
hipError_t ekv_launch_decode_fused(const EkvAttnArgs& a, const EkvScoreArgs& sc, int head_dim, int layer_count, int nw, hipStream_t s);
bool ekv_attn_chunk_supported(int head_dim, int rep, int q_len);
void ekv_chunk_blocks(int rep, int q_len, int* qb_rows, int* n_qblocks, int* qpw);
int ekv_chunk_col_parts(int qpw, bool rope);
size_t ekv_score_lds_bytes(const EkvScoreArgs& a);
bool ekv_decode_score_supported(const EkvScoreArgs& sc);
hipError_t ekv_launch_decode_score(const EkvScoreArgs& sc, int layer_count, hipStream_t s);
hipError_t ekv_launch_fold(const EkvScoreArgs& sc, int layer_count, hipStream_t s);

// Small-row chunk step with the logits in LDS (ekv_chunk_lds.inc): whole step in one launch, K and V read once.
bool ekv_chunk_lds_supported(const ekv_bank* bank, const ekv_step* st, int phys_extent, bool scored);
hipError_t ekv_launch_chunk_lds(const EkvAttnArgs& a, const EkvScoreArgs& sc, int head_dim, int layer_count, hipStream_t s);
