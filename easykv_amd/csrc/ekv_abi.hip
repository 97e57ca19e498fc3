// C ABI of the MI355X-native budgeted-KV attention path (see include/easykv_hip.h) + utility kernels.
#include <algorithm>
#include <cstdlib>

#include "ekv_common.h"
#include "ekv_kernels.h"

namespace {

// ---------------------------------------------------------------------------------------------
// utility kernels
// ---------------------------------------------------------------------------------------------
__global__ void ekv_iota_rows_kernel(int32_t* slot, int cap, size_t n_rows) {
  const size_t row = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < cap; j += gridDim.x * blockDim.x) slot[row * cap + j] = j;
  (void)n_rows;
}

// easykv/easykv.py:242-245 (mode 0), :412-416 (modes 1, 2)
__global__ void ekv_state_init_kernel(float* s, float* q, float* c, int cap, int width, int mode, int stride,
                                      size_t row0) {
  const size_t row = row0 + blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < cap; j += gridDim.x * blockDim.x) {
    float cv = 0.f;
    if (j < width) {
      if (mode == 0) cv = (float)(width - 1 - j);
      else if (mode == 1) cv = (float)(width - j) - (float)stride;
      else cv = j < width - stride ? 0.f : -(float)(j - (width - stride));
    }
    s[row * cap + j] = 0.f;
    q[row * cap + j] = 0.f;
    c[row * cap + j] = cv;
  }
}

// one 16-byte lane per 8 halfs; rows of D halfs
template <bool GATHER>
__global__ void ekv_rows_copy_kernel(__half* bank_k, __half* bank_v, const int32_t* slot, __half* lin_k, __half* lin_v,
                                     int n_kv_heads, int cap, int D, int layer_begin, int pos_begin, int n) {
  const int h = blockIdx.y, ll = blockIdx.z;
  const size_t head_row = ((size_t)(layer_begin + ll) * n_kv_heads + h) * cap;
  const int lpr = D / 8;
  const int rows_per_block = blockDim.x / lpr;
  if ((int)threadIdx.x >= rows_per_block * lpr) return;      // (head_dim 96: 21 rows of 12 pieces per 256 threads)
  const int sub = threadIdx.x % lpr;
  for (int i = blockIdx.x * rows_per_block + threadIdx.x / lpr; i < n; i += gridDim.x * rows_per_block) {
    const int row = slot[head_row + pos_begin + i];
    uint4* bk = reinterpret_cast<uint4*>(bank_k + (head_row + row) * D) + sub;
    uint4* bv = reinterpret_cast<uint4*>(bank_v + (head_row + row) * D) + sub;
    uint4* lk = reinterpret_cast<uint4*>(lin_k + (((size_t)ll * n_kv_heads + h) * n + i) * D) + sub;
    uint4* lv = reinterpret_cast<uint4*>(lin_v + (((size_t)ll * n_kv_heads + h) * n + i) * D) + sub;
    if (GATHER) {
      *lk = *bk;
      *lv = *bv;
    } else {
      *bk = *lk;
      *bv = *lv;
    }
  }
}

// Reference-shaped physical compaction (easykv/easykv.py:56-82) in place, identity layout.  One workgroup per
// (tensor, head, layer).  Destination d >= first victim takes source d + #victims <= source: a forward memmove by 1 .. n_evict rows.
// Chunks of 256 / (D/8) * CH rows ascend; inside a chunk every thread has its source rows in registers before any thread
// stores (one barrier).  Nothing else needs ordering: chunk c+1 reads rows above everything chunk c writes, and chunk c+1's writes
// only reach rows chunk c had read before ITS barrier — so the loads of chunk c+1 are issued BEFORE the stores of chunk c
// (two register sets), no thread ever waits for a store to complete, and there is one barrier per chunk instead of two.
template <int CH, bool SINGLE>
__global__ void __launch_bounds__(256) ekv_compact_inplace_kernel(__half* k, __half* v, const int32_t* evict, int n_kv_heads,
                                                                  int cap, int D, int layer_begin, int n_slots, int n_evict) {
  extern __shared__ int32_t s_ev[];
  const int which = blockIdx.x, h = blockIdx.y, ll = blockIdx.z;
  char* base = reinterpret_cast<char*>((which == 0 ? k : v) + ((size_t)(layer_begin + ll) * n_kv_heads + h) * cap * D);
  for (int i = threadIdx.x; i < n_evict; i += 256) s_ev[i] = evict[((size_t)ll * n_kv_heads + h) * n_evict + i];
  __syncthreads();
  const int lpr = D / 8, rpb = 256 / lpr;
  const int tix = min((int)threadIdx.x, rpb * lpr - 1);      // (head_dim 96: the 4 threads past 21 rows x 12 pieces repeat the last piece)
  const int sub = tix % lpr, rg = tix / lpr;
  const int first = s_ev[0], n_keep = n_slots - n_evict;
  const int row_bytes = D * 2;
  // source row of destination d = d + #{e : ev[e] - e <= d} (ev ascending, so ev[e] - e is non-decreasing: a branch-free binary
  // search with a launch-uniform number of steps; the single-victim decode step needs none).  Rows past the end are clamped:
  // the loads are unconditional.
  int n_bits = 0;
  while ((1 << n_bits) < n_evict + 1) ++n_bits;
  auto src_of = [&](int d) __attribute__((always_inline)) {
    d = min(d, n_keep - 1);
    if (SINGLE) return d + 1;              // (d >= first; template parameter: no victim walk between the loads of a chunk)
    int cnt = 0;                           // largest cnt with ev[cnt - 1] - (cnt - 1) <= d
    for (int b = n_bits - 1; b >= 0; --b) {
      const int c = cnt + (1 << b);
      const int e = min(c, n_evict) - 1;
      cnt = (c <= n_evict && s_ev[e] - e <= d) ? c : cnt;
    }
    return d + cnt;
  };
  if (first >= n_keep) return;
  ekv_u4 ra[CH], rb[CH];      // two register sets, roles alternate (a copy nxt -> cur would wait for the look-ahead loads)
  const int step = rpb * CH;
  auto load = [&](ekv_u4 (&r)[CH], int d0) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CH; ++c) r[c] = __builtin_nontemporal_load(reinterpret_cast<const ekv_u4*>(base + (size_t)src_of(d0 + c * rpb + rg) * row_bytes + sub * 16));
  };
  auto store = [&](const ekv_u4 (&r)[CH], int d0) __attribute__((always_inline)) {
    // every thread's rows of THIS chunk have landed (the CH newer loads stay in flight), then the stores
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int d = d0 + c * rpb + rg;
      if (d < n_keep) __builtin_nontemporal_store(r[c], reinterpret_cast<ekv_u4*>(base + (size_t)d * row_bytes + sub * 16));
    }
  };
  load(ra, first);
  for (int d0 = first; d0 < n_keep; d0 += 2 * step) {
    load(rb, d0 + step);
    store(ra, d0);
    load(ra, d0 + 2 * step);
    store(rb, d0 + step);
  }
}

// EKV_POLICY_RANGE ('recency' / 'random', easykv/easykv.py:343-362, :491-499, :105-112): every head of every layer drops the
// same contiguous positions [start, start + k).  Nothing is scored, so nothing needs LDS-resident rows: only the slot map is
// compacted — entries behind the range move down by k, the victims' rows become the free tail [T - k, T) — whatever the cache
// length.  One workgroup per (head, layer); chunks ascend and every chunk is read completely before it is written, and a
// chunk's sources lie at or beyond the next chunk's destinations, so no entry is overwritten before it has moved.
__global__ void __launch_bounds__(256) ekv_range_evict_kernel(int32_t* slot_of_pos, int32_t* evict_ids, int n_kv_heads, int cap,
                                                              int layer_begin, int T, int start, int k) {
  extern __shared__ int32_t s_vict[];
  const int h = blockIdx.x, ll = blockIdx.y, tid = threadIdx.x;
  int32_t* map = slot_of_pos + ((size_t)(layer_begin + ll) * n_kv_heads + h) * cap;
  for (int i = tid; i < k; i += 256) {
    s_vict[i] = map[start + i];
    if (evict_ids != nullptr) evict_ids[((size_t)ll * n_kv_heads + h) * k + i] = start + i;
  }
  __syncthreads();
  constexpr int CH = 8;
  for (int d0 = start; d0 < T - k; d0 += 256 * CH) {
    int32_t buf[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) buf[c] = map[min(d0 + c * 256 + tid + k, T - 1)];   // unconditional (clamped) loads
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int d = d0 + c * 256 + tid;
      if (d < T - k) map[d] = buf[c];
    }
    __syncthreads();
  }
  for (int i = tid; i < k; i += 256) map[T - k + i] = s_vict[i];
}

// ---- ordered <-> slot-indexed score rows (ekv_decode_tail.h, "slot-indexed score rows") ------------------------------------------
// One workgroup per (head, layer); everything is read into LDS before anything is written (the conversions are in place).
// to_slots: entry j of the ordered rows (row = slot_of_pos[j]) becomes S[row], Q[row], C0[row] = C[j] (g = 0), birth[row] = j; the
// next birth is n_slots.  Entries of the slot map below n_slots are dead afterwards; the free list [n_slots, cap) stays.
__global__ void __launch_bounds__(256) ekv_rows_to_slots_kernel(const int32_t* slot_of_pos, float* S, float* Q, float* Cn, int32_t* birth,
                                                                float* cnt_tail, float* slot_state, int n_kv_heads, int cap, int layer_begin, int T) {
  extern __shared__ float s_rows[];      // [4][T]
  const int h = blockIdx.x, ll = blockIdx.y, tid = threadIdx.x;
  const size_t head = (size_t)(layer_begin + ll) * n_kv_heads + h, head_row = head * cap;
  // the ordered count row's tail [T, cap) — what the next appended entries start from (zeros after decode steps, 0, -1, -2 ... after
  // a strided chunk step) — has no place in a row-indexed array: parked, and put back by ekv_rows_to_order
  for (int j = T + tid; j < cap; j += 256) cnt_tail[head_row + j] = Cn ? Cn[head_row + j] : 0.f;
  for (int j = tid; j < T; j += 256) {
    s_rows[j] = S[head_row + j];
    s_rows[T + j] = Q ? Q[head_row + j] : 0.f;
    s_rows[2 * T + j] = Cn ? Cn[head_row + j] : 0.f;
    reinterpret_cast<int32_t*>(s_rows)[3 * T + j] = slot_of_pos[head_row + j];
  }
  __syncthreads();
  for (int j = tid; j < T; j += 256) {
    const int row = reinterpret_cast<const int32_t*>(s_rows)[3 * T + j];
    S[head_row + row] = s_rows[j];
    if (Q) Q[head_row + row] = s_rows[T + j];
    if (Cn) Cn[head_row + row] = s_rows[2 * T + j];
    birth[head_row + row] = j;
  }
  if (tid == 0) {
    slot_state[4 * head] = 0.f;
    reinterpret_cast<int32_t*>(slot_state)[4 * head + 1] = T;
    reinterpret_cast<uint32_t*>(slot_state)[4 * head + 2] = 0u;      // no threshold hint yet
    reinterpret_cast<uint32_t*>(slot_state)[4 * head + 3] = 0u;
  }
}

// to_order: the live rows are the rows that are not on the free list [T, cap); the order index of a row is the rank of its birth
// among them (counted: births are unique).  Rebuilds slot_of_pos[0, T), S / Q / C (C = C0 + g) in order, zero tails.
__global__ void __launch_bounds__(256) ekv_rows_to_order_kernel(int32_t* slot_of_pos, float* S, float* Q, float* Cn, const int32_t* birth,
                                                                const float* cnt_tail, const float* slot_state, int n_kv_heads, int cap, int layer_begin, int T) {
  extern __shared__ float s_rows[];      // [4][cap]: S, Q, C0, birth (-1 = not live)
  const int h = blockIdx.x, ll = blockIdx.y, tid = threadIdx.x;
  const size_t head = (size_t)(layer_begin + ll) * n_kv_heads + h, head_row = head * cap;
  int32_t* s_b = reinterpret_cast<int32_t*>(s_rows) + 3 * (size_t)cap;
  const float g = slot_state[4 * head];
  for (int r = tid; r < cap; r += 256) {
    s_rows[r] = S[head_row + r];
    s_rows[cap + r] = Q ? Q[head_row + r] : 0.f;
    s_rows[2 * cap + r] = Cn ? Cn[head_row + r] : 0.f;
    s_b[r] = birth[head_row + r];
  }
  __syncthreads();
  for (int i = T + tid; i < cap; i += 256) s_b[slot_of_pos[head_row + i]] = -1;      // the free list: distinct rows
  __syncthreads();
  for (int r = tid; r < cap; r += 256) {
    const int b = s_b[r];
    if (b >= 0) {
      int rank = 0;
      for (int x = 0; x < cap; ++x) {
        const int bx = s_b[x];
        rank += (bx >= 0 && bx < b) ? 1 : 0;
      }
      slot_of_pos[head_row + rank] = r;
      S[head_row + rank] = s_rows[r];
      if (Q) Q[head_row + rank] = s_rows[cap + r];
      if (Cn) Cn[head_row + rank] = s_rows[2 * cap + r] + g;
    }
  }
  // tails: S / Q are zero behind the live entries in every flow; the count tail is the parked one (an evicting slot-layout step
  // has zeroed its front entry, like the ordered step does)
  for (int j = T + tid; j < cap; j += 256) {
    S[head_row + j] = 0.f;
    if (Q) Q[head_row + j] = 0.f;
    if (Cn) Cn[head_row + j] = cnt_tail[head_row + j];
  }
}

// A launch failure must be reported as THIS call's, not as whatever sticky-free error an earlier, unrelated runtime call of the
// thread left behind: every entry point drops the stale last-error state first, then reads it back after its own launches.
inline void drop_stale_error() { (void)hipGetLastError(); }
inline int launch_status() { return hipGetLastError() == hipSuccess ? EKV_OK : EKV_E_LAUNCH; }

int check_bank(const ekv_bank* b) {
  if (!b || !b->k || !b->v || !b->slot_of_pos) return EKV_E_ARG;
  if (b->n_layers <= 0 || b->n_kv_heads <= 0 || b->n_q_heads % b->n_kv_heads || b->cap <= 0) return EKV_E_ARG;
  if (b->head_dim != 32 && b->head_dim != 64 && b->head_dim != 96 && b->head_dim != 128) return EKV_E_UNSUPPORTED;
  return EKV_OK;
}

int check_layers(const ekv_bank* b, int begin, int count) {
  return (begin < 0 || count <= 0 || begin + count > b->n_layers) ? EKV_E_ARG : EKV_OK;
}

}  // namespace

// Physical extent E of a step (every live row has a physical index < E): the caller's value when it is consistent, else cap.
static int step_extent(const ekv_bank* bank, const ekv_step* st) {
  return (st->phys_extent >= st->n_slots && st->phys_extent <= bank->cap) ? st->phys_extent : bank->cap;
}

// Pitch of a logits row in the fused decode kernel's LDS: logical positions (RoPE-on-read streams in position order, the
// rotation needs the position index) or physical rows [0, E).
int ekv_fused_logit_pad(const ekv_bank* bank, const ekv_step* st, int t_pad) {
  return st->rope_on_read ? t_pad : (int)ekv_align((size_t)step_extent(bank, st), 64);
}

EkvWs ekv_plan_workspace(const ekv_bank* bank, const ekv_step* st, void* base) {
  EkvWs w{};
  const int T = st->n_slots;
  const int rep = bank->n_q_heads / bank->n_kv_heads;
  const bool scored =
      st->policy == EKV_POLICY_H2O_HEAD || st->policy == EKV_POLICY_ROCO || st->policy == EKV_POLICY_TOVA;
  w.t_pad = (int)ekv_align((size_t)T, 64);
  w.qb_rows = 1;
  w.n_qblocks = 1;
  int qpw = 1;
  if (st->q_len > 1) ekv_chunk_blocks(rep, st->q_len, &w.qb_rows, &w.n_qblocks, &qpw);
  // key-range splits
  const int wg_unit = st->q_len == 1 ? 128 : 64;
  int n_split = st->n_split;
  w.fused_nw = ekv_decode_fused_nw(st->layer_count * bank->n_kv_heads);
  if (n_split <= 0 && st->q_len == 1 && st->layer_count * bank->n_kv_heads >= 256 && w.fused_nw == 8 &&
      ekv_decode_fused_supported(bank->head_dim, rep, T, w.t_pad, ekv_fused_logit_pad(bank, st, w.t_pad), st->n_evict, bank->cap, 8)) {
    n_split = 1;   // >= 1 head per CU: one 8-wave workgroup per head beats key-range splits + a second kernel (GQA shapes)
  }
  // A whole scored step small enough for the logits-resident kernel (ekv_attn_resident.inc: one launch, one workgroup per head, K and V
  // read once) runs there, unsplit, whatever the number of heads in the launch.  Measured per step (us, resident / two passes): 256
  // (head, layer) pairs of configs[2] 49.8 / 81.9-82.4; 1024 pairs of 64 rows x 1152 keys 193.7-197.0 / 225.3-226.2; blocks of 9..32
  // rows against the 16x16 kernel + scorer tail: 256 pairs, 32 / 16 rows 44.0 / 43.3 against 71.5-73.9 / 61.5-62.2; <= 32 rows x 2064 keys
  // 63.7-64.0 against 110.7-111.8.  Not for steps that
  // force a scheme or a split, run in phases or defer their scorer (a layer-per-call model: those launches hold 8..32 heads, and their
  // column-sum pass + scorer run once over all layers).  It is planned as an unsplit two-pass step of the wide-block kernel (the
  // workspace of one is never touched).
  w.resident = (st->q_len > 1 && st->phases == 0 && st->defer_layers == 0 && st->two_pass == 0 && n_split <= 1 &&
                (st->policy == EKV_POLICY_H2O_HEAD || st->policy == EKV_POLICY_ROCO) && st->accumulate && !st->rope_on_read && w.n_qblocks == 1 &&
                st->score_off >= 0 && st->score_off < T && ekv_attn_resident_supported(bank->head_dim, rep, st->q_len, T, T - st->score_off)) ? 1 : 0;
  if (w.resident) n_split = 1;
  const bool two_pass_plan = w.resident || (st->q_len > 1 && ekv_chunk_two_pass(bank->head_dim, rep, st->q_len, st->policy, scored, st->accumulate != 0, st->rope_on_read != 0, st->two_pass));
  const bool wide_plan = w.resident || (st->q_len > 1 && ekv_chunk_wide(bank->head_dim, rep, st->q_len, st->rope_on_read != 0, two_pass_plan,
                                                                        !two_pass_plan && scored && st->accumulate != 0));
  if (n_split <= 0 && st->q_len > 1) {
    // Chunk steps (two or three workgroups per CU): a split costs a partial per query row and split, a fold in the scorer and a
    // shorter stream per workgroup, so the grid is filled to the 256..512 workgroups that are resident at a time — not to 1024:
    // the fewest splits (powers of two, <= 8) that give every CU a workgroup, doubled once more if a split then still streams
    // >= 1024 rows.  Measured (us per step, MI355X, round 3 sweep; this rule / what the 1024- or 3072-target picked):
    //   wide kernel, 8 KV heads x 1 layer, 64 rows, T = 1248 / 2176 / 5098:  46.7 / 54.7 / 76.2   (50.0 / 61.4 / 84.8)
    //   wide kernel, 32 heads x 1 layer, 96 rows:                            63.9 / 76.6 / 110.5  (69.6 / 91.6 / 155.9)
    //   wide kernel, 128 (head, layer) pairs, 96 rows:                       97.8 / 110.8 / 182.0 (108.6 / 132.5 / 208.6)
    //   wide kernel, 256 pairs (configs[2]: 8 KV heads x 32 layers):         87.2 / 131.7 / 239.7 (116.3 / 154.1 / 265.4)
    //   RoPE-on-read (16x16 kernel), 32 heads x 1 layer, 96 rows, T = 2176 / 4205:  127.4 / 166.4  (152.3 / 228.7)
    //   16x16 kernel, GQA x4 stride 8 (32 rows), 64 pairs, T = 2176 / 4205:  57.9 / 87.9   (72.9 / 110.0);  256 pairs: 109.4 / 190.4 (126.8 / 207.5)
    //   16x16 kernel, stride 16 MHA, 256 pairs:                              93.3 / 155.6  (99.4 / 173.0)
    const int wgs = st->layer_count * bank->n_kv_heads * w.n_qblocks;
    n_split = 1;
    while (wgs * n_split < 256 && n_split < 8) n_split *= 2;
    if (wgs * n_split < 512 && n_split < 8 && T / (2 * n_split) >= 1024) n_split *= 2;
    n_split = std::max(1, std::min(n_split, (T + 255) / 256));
  }
  if (n_split <= 0) {      // decode: >= 1024 workgroups (4 per CU, one round), never fewer than 128 positions per split
    const int wgs = st->layer_count * bank->n_kv_heads;
    n_split = std::max(1, std::min((1024 + wgs - 1) / wgs, (T + 127) / 128));
    // decode launches of a few heads (one layer per call): the launch is latency-bound and ends with the fold of the key-range
    // partials, whose loads go out in batches of 8 — up to 8 splits are ONE round trip.  Measured at 32 heads, T = 2049
    // (us per layer, attention + in-kernel fold): 5 splits 15.4, 6..8 13.7, 10..17 14.6..14.7.
    if (n_split > 8 && wgs * 8 >= 256) n_split = 8;
  }
  if (st->q_len > 1) {
    // the chunk kernel caches the slot indices of its key range in LDS next to its tiles and query block: bound the range
    // (an unsplit 32 k-slot head would need 128 KB of indices alone; split heads fold through the partials instead)
    const int max_rows = st->rope_on_read ? 6144 : 16384;
    n_split = std::max(n_split, (T + max_rows - 1) / max_rows);
  }
  int rows = (int)ekv_align((size_t)(T + n_split - 1) / n_split, wg_unit);
  w.rows_per_split = rows;
  w.n_split = (T + rows - 1) / rows;
  w.two_pass = two_pass_plan ? 1 : 0;
  w.wide = wide_plan ? 1 : 0;
  // partials per query row: one per split (decode, wide chunk kernel) or one per split and key half (16x16 chunk kernel)
  w.n_partials = (st->q_len == 1 || w.wide) ? w.n_split : 2 * w.n_split;
  const size_t rowsq = (size_t)st->layer_count * bank->n_q_heads * st->q_len;
  size_t off = 0;
  char* p = static_cast<char*>(base);
  w.logits = nullptr;
  w.stats = w.colsum = nullptr;
  // column-sum partial rows per head: query-tile waves per workgroup x query blocks; the wide kernel combines them itself
  if (w.wide) {
    // the column-sum pass of the wide kernel walks the query blocks of a (head, key range) inside the workgroup and leaves ONE row of
    // column sums; a launch of few (head, layer) pairs (one layer of a decoder stack) spreads them over up to 16 workgroups
    const int wgs = st->layer_count * bank->n_kv_heads * w.n_split;
    w.n_col_parts = std::max(1, std::min(std::min(w.n_qblocks, 16), (512 + wgs - 1) / wgs));
  } else {
    w.n_col_parts = ekv_chunk_col_parts(qpw, st->rope_on_read != 0) * w.n_qblocks;
  }
  if (w.two_pass) {   // statistics partials + column sums instead of the logits
    w.stats = reinterpret_cast<float*>(p + off);
    off += ekv_align(rowsq * w.n_partials * 2 * 4, 256);
    w.colsum = reinterpret_cast<float*>(p + off);
    off += ekv_align((size_t)st->layer_count * bank->n_kv_heads * w.n_col_parts * 2 * w.t_pad * 4, 256);
  } else if (scored && st->accumulate) {   // the scorer only needs the logits when it accumulates
    w.logits = reinterpret_cast<float*>(p + off);
    off += ekv_align(rowsq * w.t_pad * 4, 256);
  }
  w.partials = reinterpret_cast<float*>(p + off);
  off += ekv_align(rowsq * w.n_partials * (bank->head_dim + 2) * 4, 256);
  w.tova_row = reinterpret_cast<float*>(p + off);
  off += ekv_align((size_t)st->layer_count * w.t_pad * 4, 256);
  w.big_rows = nullptr;
  if (scored && ekv_score_rows_exceed_lds(T - st->score_off, w.two_pass ? 0 : rep * st->q_len)) {   // W > ~10 000: rows in scratch, keys in LDS
    w.big_rows = reinterpret_cast<float*>(p + off);
    off += ekv_align((size_t)st->layer_count * bank->n_kv_heads * 3 * w.t_pad * 4, 256);
  }
  // unsplit chunk steps fold the two key halves inside the attention kernel (one-pass scored steps also get the final row
  // statistics from it)
  w.fold_in_kernel = (st->q_len > 1 && w.n_split == 1) ? 1 : 0;
  w.row_stats = nullptr;
  if (w.fold_in_kernel && !w.two_pass && scored && st->accumulate) {   // one-pass scored: the scorer still needs (M, L) per row
    w.row_stats = reinterpret_cast<float*>(p + off);
    off += ekv_align(rowsq * 2 * 4, 256);
  }
  // deferred scorer of a layer-per-call model, wide two-pass chunk steps: the column-sum pass is deferred with it (one launch over
  // all layers at the flush instead of a 256-workgroup launch per layer) — it needs every layer's raw queries
  w.q_keep = nullptr;
  if (st->defer_layers > 0 && st->q_len > 1 && w.wide && w.two_pass) {
    w.q_keep = reinterpret_cast<__half*>(p + off);
    off += ekv_align(rowsq * bank->head_dim * 2, 256);
  }
  w.q_rot = nullptr;
  if (st->rope_on_read && st->q_len > 1 && !w.wide) {      // (16x16 kernel only: the wide-block kernel rotates Q in its prologue)
    w.q_rot = reinterpret_cast<__half*>(p + off);
    off += ekv_align(2 * rowsq * bank->head_dim * 2, 256);
  }
  w.bytes = off;
  return w;
}

extern "C" {

int ekv_abi_version(void) { return EKV_ABI_VERSION; }

const char* ekv_strerror(int code) {
  switch (code) {
    case EKV_OK: return "ok";
    case EKV_E_ARG: return "invalid argument (null pointer or inconsistent sizes)";
    case EKV_E_UNSUPPORTED: return "unsupported shape (head_dim, group size, q_len or row width)";
    case EKV_E_WORKSPACE: return "workspace too small";
    case EKV_E_LAUNCH: return "kernel launch failed";
    default: return "unknown error";
  }
}

// Deferred scorer (ekv_step.defer_layers): the workspace is laid out for all deferred layers whatever this call launches
static ekv_step defer_layout_step(const ekv_step* st) {
  ekv_step full = *st;
  if (st->defer_layers > 0) full.layer_count = st->defer_layers;
  return full;
}

size_t ekv_workspace_bytes(const ekv_bank* bank, const ekv_step* step) {
  if (!bank || !step) return 0;
  const ekv_step full = defer_layout_step(step);
  return ekv_plan_workspace(bank, &full, nullptr).bytes;
}

// what a dry run reports about the dispatch: one_launch = the whole step is ONE launch; n_launches = kernel launches of the call
struct EkvPlanOut {
  int32_t one_launch, n_launches;
};
static int step_attend_impl(const ekv_bank*, const ekv_step*, const void*, const void*, const void*, void*, int32_t*, const float*,
                            const float*, void*, size_t, void*, bool, EkvPlanOut*);

int ekv_step_plan(const ekv_bank* bank, const ekv_step* st, int32_t* n_split, int32_t* fused) {
  if (int e = check_bank(bank)) return e;
  if (!st || !n_split || !fused) return EKV_E_ARG;
  const EkvWs ws = ekv_plan_workspace(bank, st, nullptr);
  *n_split = ws.n_split;
  // one launch for the whole step: the fused decode kernel, the logits-in-LDS chunk kernel, or a chunk step whose scorer runs
  // as the tail of the attention kernel — asked of the dispatch itself (dry run); a step the dispatch would refuse plans as 0
  EkvPlanOut po{};
  if (step_attend_impl(bank, st, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true, &po) != EKV_OK) po.one_launch = 0;
  *fused = (po.one_launch && (st->phases & ~(EKV_PHASE_SLOT_ROWS | EKV_PHASE_SLOT_TAIL_OK)) == 0) ? 1 : 0;
  return EKV_OK;
}

int ekv_step_info(const ekv_bank* bank, const ekv_step* st, int32_t* info, int32_t n_info) {
  if (int e = check_bank(bank)) return e;
  if (!st || !info || n_info < 1) return EKV_E_ARG;
  int32_t n_split = 0, fused = 0;
  if (int e = ekv_step_plan(bank, st, &n_split, &fused)) return e;
  const EkvWs ws = ekv_plan_workspace(bank, st, nullptr);
  EkvPlanOut po{};
  if (step_attend_impl(bank, st, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true, &po) != EKV_OK) po.n_launches = 0;
  const int32_t v[EKV_STEP_INFO_N] = {n_split, fused, ws.two_pass, ws.wide, ws.n_qblocks, ws.qb_rows, ws.n_col_parts, ws.fold_in_kernel, po.n_launches};
  for (int i = 0; i < n_info && i < EKV_STEP_INFO_N; ++i) info[i] = v[i];
  for (int i = EKV_STEP_INFO_N; i < n_info; ++i) info[i] = 0;
  return EKV_OK;
}

int ekv_bank_reset(const ekv_bank* bank, void* stream) {
  if (int e = check_bank(bank)) return e;
  drop_stale_error();
  const size_t rows = (size_t)bank->n_layers * bank->n_kv_heads;
  hipLaunchKernelGGL(ekv_iota_rows_kernel, dim3((bank->cap + 255) / 256, (unsigned)rows), dim3(256), 0,
                     static_cast<hipStream_t>(stream), bank->slot_of_pos, bank->cap, rows);
  if (bank->arrive != nullptr && hipMemsetAsync(bank->arrive, 0, rows * 4, static_cast<hipStream_t>(stream)) != hipSuccess) return EKV_E_LAUNCH;
  return launch_status();
}

int ekv_state_init(const ekv_bank* bank, int32_t layer_begin, int32_t layer_count, int32_t width, int32_t mode,
                   int32_t stride, void* stream) {
  if (int e = check_bank(bank)) return e;
  if (int e = check_layers(bank, layer_begin, layer_count)) return e;
  if (!bank->score_sum || !bank->score_sq || !bank->score_cnt || width < 0 || width > bank->cap || mode < 0 || mode > 2)
    return EKV_E_ARG;
  const size_t row0 = (size_t)layer_begin * bank->n_kv_heads;
  drop_stale_error();
  hipLaunchKernelGGL(ekv_state_init_kernel, dim3((bank->cap + 255) / 256, layer_count * bank->n_kv_heads), dim3(256), 0,
                     static_cast<hipStream_t>(stream), bank->score_sum, bank->score_sq, bank->score_cnt, bank->cap, width,
                     mode, stride, row0);
  return launch_status();
}

}  // extern "C"

// Which one-launch decode steps run on the slot-indexed layout: plain keys, a scored policy over the whole cache (score_off = 0), no
// protected sink window (win_lo = 0: the recent tail is a birth threshold, a sink window would need ranks), at most one victim,
// GQA factor <= 4 and an extent of at most 9 (4-wave workgroups) / 5 (8-wave) rows per thread — the builds whose thread-owned
// columns stay in registers without spills.
// second half of ekv_bank.birth: the parked tail of the ordered count row (float), same [layer][head][cap] indexing
static float* ekv_cnt_tail(const ekv_bank* bank) {
  return reinterpret_cast<float*>(bank->birth + (size_t)bank->n_layers * bank->n_kv_heads * bank->cap);
}

static bool ekv_slot_rows_supported_impl(const ekv_bank* bank, const ekv_step* st, int phys_extent, int fused_nw, int t_pad) {
  const bool scored = st->policy == EKV_POLICY_H2O_HEAD || st->policy == EKV_POLICY_ROCO || st->policy == EKV_POLICY_TOVA;
  const int rep = bank->n_q_heads / bank->n_kv_heads;
  if (!bank->birth || !bank->slot_state || !scored || st->q_len != 1 || st->rope_on_read || st->n_evict > 1) return false;
  if (!bank->score_sq || !bank->score_cnt) return false;      // (the tail keeps the count base of EVERY policy's appended row in score_cnt)
  if (st->score_off != 0 || st->win_lo != 0 || st->tova_head_mean || rep > 4 || !st->accumulate) return false;
  if (st->count_add != (float)(int)st->count_add) return false;      // counts stay exact integers (count = base + running sum)
  const int max_rows = 6144;      // (columns per thread in registers: 5 / 9 up to 2560 / 2304 rows, 12 / 24 beyond — 8-wave / 4-wave build)
  if (phys_extent > max_rows || bank->cap > 9600) return false;      // (cap: ekv_rows_to_order stages four rows of `cap` words in LDS)
  return 2 * ekv_align((size_t)phys_extent, 256) <= 3 * ekv_align((size_t)t_pad, 256);      // (LDS: two rows over [0, E) instead of three over [0, T))
}

// Body of ekv_step_attend.  `dry` (ekv_step_check): every argument / shape / capability test of the real call, in the same
// order, and a return right before the first launch — nothing is launched, no pointer is dereferenced, the workspace is not
// needed.
static int step_attend_impl(const ekv_bank* bank, const ekv_step* st, const void* q, const void* k_new, const void* v_new,
                            void* out, int32_t* evict_ids, const float* rope_cos, const float* rope_sin, void* workspace,
                            size_t workspace_bytes, void* stream, bool dry, EkvPlanOut* plan_out) {
  int32_t one_launch_dummy = 0;
  int32_t* const one_launch = plan_out ? &plan_out->one_launch : &one_launch_dummy;
  if (plan_out) plan_out->one_launch = plan_out->n_launches = 0;
  if (int e = check_bank(bank)) return e;
  if (!st) return EKV_E_ARG;
  // EKV_PHASE_SLOT_ROWS: the bank's score rows are in the slot-indexed layout (ekv_rows_to_slots) — only the one-launch decode step
  // runs on it; everything below sees the remaining phase bits
  const bool slot_rows = (st->phases & EKV_PHASE_SLOT_ROWS) != 0;
  const bool slot_tail_ok = slot_rows && (st->phases & EKV_PHASE_SLOT_TAIL_OK) != 0;
  ekv_step st_plain = *st;
  st_plain.phases &= ~(EKV_PHASE_SLOT_ROWS | EKV_PHASE_SLOT_TAIL_OK);
  st = &st_plain;
  if (!dry && (!q || !k_new || !v_new || !out || !workspace)) return EKV_E_ARG;
  if (int e = check_layers(bank, st->layer_begin, st->layer_count)) return e;
  const int T = st->n_slots, n = st->q_len;
  if (n < 1 || T < n || T > bank->cap || st->n_evict < 0 || st->n_evict >= T) return EKV_E_ARG;
  const bool scored =
      st->policy == EKV_POLICY_H2O_HEAD || st->policy == EKV_POLICY_ROCO || st->policy == EKV_POLICY_TOVA;
  if (scored && (!bank->score_sum || st->score_off < 0 || st->score_off >= T)) return EKV_E_ARG;
  if (st->policy == EKV_POLICY_ROCO && (!bank->score_sq || !bank->score_cnt)) return EKV_E_ARG;
  if (st->policy < EKV_POLICY_NONE || st->policy > EKV_POLICY_RANGE) return EKV_E_ARG;
  if (!dry && st->rope_on_read && (!rope_cos || !rope_sin)) return EKV_E_ARG;
  const int W = T - (scored ? st->score_off : 0);
  if (st->n_evict > 0) {
    if (st->policy == EKV_POLICY_NONE) return EKV_E_ARG;
    if (st->policy == EKV_POLICY_RANGE && (st->range_start < 0 || st->range_start + st->n_evict > T)) return EKV_E_ARG;
    if (st->policy == EKV_POLICY_ROCO && (st->roco_k1 < st->n_evict || st->roco_k1 > W)) return EKV_E_ARG;
    if ((st->policy == EKV_POLICY_H2O_HEAD || st->policy == EKV_POLICY_TOVA) &&
        (st->win_lo < 0 || st->win_tail < 0 || W - st->win_tail - st->win_lo < st->n_evict))
      return EKV_E_ARG;
  }
  const int rep = bank->n_q_heads / bank->n_kv_heads;
  // row strides (ABI 8): both zero = dense; q_len = 1 makes the token stride irrelevant and only takes head rows head_dim apart
  int32_t strides[6] = {st->q_token_stride, st->q_head_stride, st->kv_token_stride, st->kv_head_stride, st->out_token_stride, st->out_head_stride};
  for (int i = 0; i < 6; i += 2) {
    int32_t& ts = strides[i];
    int32_t& hs = strides[i + 1];
    if (ts < 0 || hs < 0) return EKV_E_ARG;
    if (n == 1) {
      if (hs != 0 && hs != bank->head_dim) return EKV_E_UNSUPPORTED;
      ts = hs = 0;
    }
    if (ts == 0 && hs == 0) {
      ts = bank->head_dim;
      hs = n * bank->head_dim;
    } else if (ts < bank->head_dim || hs < bank->head_dim || (ts & 7) || (hs & 7)) {
      return EKV_E_ARG;
    }
  }
  if (st->defer_layers != 0) {   // deferred scorer: decode AND chunk steps (ABI 5), explicit splits, attention + fold now / scorer later
    if (st->defer_layers < 0 || st->defer_index < 0 || st->defer_index + st->layer_count > st->defer_layers || st->n_split <= 0 ||
        (st->phases != (1 | 4) && st->phases != 8))
      return EKV_E_ARG;
  }
  const ekv_step layout = defer_layout_step(st);
  // (dry: a non-null dummy base — several tests below read "is this array present" off the carved pointers)
  EkvWs ws = ekv_plan_workspace(bank, &layout, dry ? reinterpret_cast<void*>(uintptr_t(256)) : workspace);
  if (!dry && ws.bytes > workspace_bytes) return EKV_E_WORKSPACE;
  if (st->defer_layers > 0) {    // this call's slice of the per-layer arrays
    const size_t rows0 = (size_t)st->defer_index * bank->n_q_heads * n;
    if (ws.logits) ws.logits += rows0 * ws.t_pad;
    ws.partials += rows0 * ws.n_partials * (bank->head_dim + 2);
    ws.tova_row += (size_t)st->defer_index * ws.t_pad;
    if (ws.big_rows) ws.big_rows += (size_t)st->defer_index * bank->n_kv_heads * 3 * ws.t_pad;
    // chunk steps: what the deferred scorer reads of the attention launches — column sums (two passes) or row statistics (one pass)
    if (ws.stats) ws.stats += rows0 * ws.n_partials * 2;
    if (ws.colsum) ws.colsum += (size_t)st->defer_index * bank->n_kv_heads * ws.n_col_parts * 2 * ws.t_pad;
    if (ws.row_stats) ws.row_stats += rows0 * 2;
    if (ws.q_keep) ws.q_keep += rows0 * bank->head_dim;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);

  EkvAttnArgs aa{};
  aa.k = static_cast<const __half*>(bank->k);
  aa.v = static_cast<const __half*>(bank->v);
  aa.k_w = static_cast<__half*>(bank->k);
  aa.v_w = static_cast<__half*>(bank->v);
  aa.slot_of_pos = bank->slot_of_pos;
  aa.q = static_cast<const __half*>(q);
  aa.k_new = static_cast<const __half*>(k_new);
  aa.v_new = static_cast<const __half*>(v_new);
  aa.logits = ws.logits;
  aa.partials = ws.partials;
  // (dry run: a non-null dummy — the dispatch reads "RoPE-on-read?" off this pointer, nothing is launched)
  aa.rope_cos = st->rope_on_read ? (dry ? reinterpret_cast<const float*>(uintptr_t(256)) : rope_cos) : nullptr;
  aa.rope_sin = st->rope_on_read ? (dry ? reinterpret_cast<const float*>(uintptr_t(256)) : rope_sin) : nullptr;
  aa.q_rot_hi = ws.q_rot;
  aa.q_rot_lo = ws.q_rot ? ws.q_rot + (size_t)st->layer_count * bank->n_q_heads * n * bank->head_dim : nullptr;
  aa.out_direct = ws.fold_in_kernel ? static_cast<__half*>(out) : nullptr;
  aa.row_stats = ws.row_stats;
  aa.stats = ws.stats;
  aa.colsum = ws.colsum;
  aa.n_col_parts = ws.n_col_parts;
  aa.n_q_heads = bank->n_q_heads;
  aa.n_kv_heads = bank->n_kv_heads;
  aa.cap = bank->cap;
  aa.n_slots = T;
  aa.q_len = n;
  aa.n_split = ws.n_split;
  aa.rows_per_split = ws.rows_per_split;
  aa.t_pad = ws.t_pad;
  aa.layer_begin = st->layer_begin;
  aa.causal = st->causal;
  aa.qb_rows = ws.qb_rows;
  aa.n_qblocks = ws.n_qblocks;
  aa.sm_div = st->sm_div;
  aa.q_keep = ws.q_keep;
  aa.phys_extent = step_extent(bank, st);
  aa.l_pad = ekv_fused_logit_pad(bank, st, ws.t_pad);
  aa.q_ts = strides[0], aa.q_hs = strides[1], aa.kv_ts = strides[2], aa.kv_hs = strides[3], aa.o_ts = strides[4], aa.o_hs = strides[5];

  EkvScoreArgs sa{};
  sa.slot_of_pos = bank->slot_of_pos;
  sa.score_sum = bank->score_sum;
  sa.score_sq = bank->score_sq;
  sa.score_cnt = bank->score_cnt;
  sa.logits = ws.logits;
  sa.partials = ws.partials;
  sa.tova_row = ws.tova_row;
  sa.colsum = ws.colsum;
  sa.row_stats = ws.row_stats;
  sa.n_col_parts = ws.n_col_parts;
  sa.out = static_cast<__half*>(out);
  sa.evict_ids = evict_ids;
  sa.n_q_heads = bank->n_q_heads;
  sa.n_kv_heads = bank->n_kv_heads;
  sa.head_dim = bank->head_dim;
  sa.cap = bank->cap;
  sa.n_slots = T;
  sa.q_len = n;
  sa.n_split = ws.n_partials;
  sa.t_pad = ws.t_pad;
  sa.layer_begin = st->layer_begin;
  sa.score_off = st->score_off;
  sa.policy = st->policy;
  sa.accumulate = st->accumulate;
  sa.n_evict = st->n_evict;
  sa.win_lo = st->win_lo;
  sa.win_tail = st->win_tail;
  sa.roco_k1 = st->roco_k1;
  sa.roco_tail = st->roco_tail;
  sa.range_start = st->range_start;
  sa.tova_head_mean = st->tova_head_mean;
  sa.causal = st->causal;
  sa.count_add = st->count_add;
  sa.count_tail_step = st->count_tail_step;
  sa.o_ts = strides[4], sa.o_hs = strides[5];

  sa.big_rows = ws.big_rows;
  sa.big_stride = ws.t_pad;
  sa.slot_state = bank->slot_state;      // (word [3] of a head: threshold hint of the logits-in-LDS chunk kernel, whatever the layout)

  drop_stale_error();
  // whole decode step in one launch when no head has to be split
  if (n == 1 && st->phases == 0 && ws.n_split == 1 &&
      ekv_decode_fused_supported(bank->head_dim, rep, T, ws.t_pad, aa.l_pad, st->n_evict, bank->cap, ws.fused_nw)) {
    if (slot_rows) {
      if (!ekv_slot_rows_supported_impl(bank, st, aa.phys_extent, ws.fused_nw, ws.t_pad)) return EKV_E_UNSUPPORTED;
      sa.birth = bank->birth;
      sa.slot_state = bank->slot_state;
      sa.cnt_tail = ekv_cnt_tail(bank);
      sa.slot_tail_ok = slot_tail_ok ? 1 : 0;
    }
    *one_launch = 1;
    if (plan_out) plan_out->n_launches = 1;
    if (dry) return EKV_OK;
    return ekv_launch_decode_fused(aa, sa, bank->head_dim, st->layer_count, ws.fused_nw, s) == hipSuccess ? EKV_OK : EKV_E_LAUNCH;
  }
  if (slot_rows) return EKV_E_UNSUPPORTED;      // every other kernel reads the ordered layout: ekv_rows_to_order first

  // small-row chunk step (configs[1]: stride 8): one launch, logits in LDS, K and V read once
  if (n > 1 && ekv_chunk_lds_supported(bank, st, aa.phys_extent, scored) &&
      (st->layer_count * bank->n_kv_heads >= 256 || T <= 1024)) {
    *one_launch = 1;
    if (plan_out) plan_out->n_launches = 1;
    if (dry) return EKV_OK;
    return ekv_launch_chunk_lds(aa, sa, bank->head_dim, st->layer_count, s) == hipSuccess ? EKV_OK : EKV_E_LAUNCH;
  }

  // phases: 0 = whole step; else a bit mask: 1 attention kernel, 2 scorer (fold + score), 4 fold only, 8 scorer
  // without the fold (4 and 8 let the caller run the scorer on a side stream, off the critical path)
  const int ph = st->phases;
  if (ph < 0 || ph > 15 || ((ph & 2) && (ph & (4 | 8)))) return EKV_E_ARG;
  // Whole chunk step in ONE launch: unsplit heads (the kernel folds its own output), one-pass logits, a scored policy, and the
  // scorer's LDS rows fit next to two workgroups per CU.  The scorer of a head then runs as the tail of the workgroup that
  // streamed it and overlaps the K/V stream of the workgroups still running (tova_head_mean needs all heads of a layer first).
  // (not on the wide-block kernel: it has no scorer tail — an unsplit scored step of 33..64 rows that exports no logits, i.e. the
  //  first strided chunk of an encoding-mode prefill, runs as wide attention + scorer launch)
  const bool fuse_chunk = n > 1 && ph == 0 && ws.fold_in_kernel && !ws.two_pass && !ws.wide && scored && ws.n_qblocks == 1 &&
                          rep * n <= 64 && !(st->policy == EKV_POLICY_TOVA && st->tova_head_mean && st->accumulate) &&
                          ekv_score_lds_bytes_nt256(sa) <= 64 * 1024 && st->n_split != -1;
  if (fuse_chunk) sa.skip_fold = 1;
  if (fuse_chunk) *one_launch = 1;
  // Two-pass step on the wide-block kernel (whole step, or the flush of a layer-per-call step whose column-sum pass was deferred):
  // the scorer runs as the TAIL of the column-sum pass (ekv_wide_tail.h, round 5) — score rows in registers, keys in the pass's
  // tile buffers, four workgroups per CU — instead of a 1024-thread-per-CU scorer launch behind it.  Only for heads whose column sums
  // come from one workgroup (no key-range splits, one query-block group): split heads keep the stand-alone scorer (measured faster).
  // Not with RoPE-on-read: those passes run two workgroups per CU, where a head's ~50 us tail costs more stream than the launch it saves.
  const bool flush_colsum = (ph & 8) && !(ph & 1) && n > 1 && ws.q_keep != nullptr;
  // The flush's column-sum launch covers ALL deferred layers: with >= 512 (head, layer, query-block group) units it runs UNSPLIT whatever
  // key-range split the one-layer calls of the one pass used (that split exists to fill the chip from 32 heads) — 1024 workgroups in one
  // resident round instead of 8192 short ones, and the scorer as its tail.
  // ADVICE r5 asked for this branch to be tied to the tail or measured on its own.  Measured (round 6, one layer per call, flush over 40
  // layers x 40 heads, configs[4] shape with RoPE-on-read, where the tail does NOT run): unsplit flush 92.4-93.0 us per layer against
  // 94.1-94.8 us with the one-layer calls' key-range split kept — so a flush runs unsplit WHENEVER the key range fits what an unsplit
  // workgroup may walk (the row bound ekv_plan_workspace applies to n_split: slot entries of the range live in LDS — 6144 rows with
  // RoPE-on-read, 16384 plain), tail or not; beyond that bound it keeps the split.
  const bool tail_shape = n > 1 && ws.wide && ws.two_pass && ws.big_rows == nullptr && scored && st->accumulate &&
                          st->policy != EKV_POLICY_TOVA && !st->rope_on_read;
  const bool flush_unsplit = flush_colsum && ws.wide && ws.two_pass && T <= (st->rope_on_read ? 6144 : 16384) &&
                             (size_t)st->layer_count * bank->n_kv_heads * ws.n_col_parts >= 512;
  const int tail_wgs = (flush_unsplit ? 1 : ws.n_split) * ws.n_col_parts;
  const bool tail_step = tail_shape && (ph == 0 || flush_colsum) && ekv_wide_tail_supported(W, tail_wgs);

  // How the step ends, decided BEFORE anything is launched: a shape no scorer can take must be refused while the bank is
  // still untouched (the attention kernel appends the new rows).
  //   fold_only  nothing to score and nothing to evict ('full', any unknown policy string), or phases = attention + fold
  //   range_only 'recency' / 'random': no score rows at all, only the slot map is compacted — any cache length
  const bool wants_scorer = ph == 0 || (ph & (2 | 8));
  const bool fold_only = (!scored && st->n_evict == 0) || !wants_scorer;
  const bool range_only = wants_scorer && st->policy == EKV_POLICY_RANGE;
  const bool fast_scorer = wants_scorer && !fold_only && !range_only && !fuse_chunk && ekv_decode_score_supported(sa);
  if (wants_scorer && !fold_only && !range_only && !fuse_chunk && !fast_scorer && ekv_score_lds_bytes(sa) > 160 * 1024)
    return EKV_E_UNSUPPORTED;   // even the selection keys alone exceed one CU's LDS (W > ~39 000): see DESIGN.md "size limits"
  if (n == 1 ? !ekv_attn_decode_supported(bank->head_dim, rep) : !ekv_attn_chunk_supported(bank->head_dim, rep, n)) return EKV_E_UNSUPPORTED;

  // decode split path whose partials are folded right behind the attention kernel (attention + fold phases, or a step that has
  // nothing to score): the last-arriving split of a head folds them inside the attention kernel — no fold launch
  // (GQA factors > 8 run several query-head groups per KV head, ekv_attn_decode.inc: the arrival counter counts one group's splits)
  const bool fold_in_decode = n == 1 && rep <= 8 && bank->arrive != nullptr && (ph == 0 || (ph & 1)) && !ws.fold_in_kernel &&
                              ((ph & 4) || (ph == 0 && (fold_only || range_only)));
  // ---- the launch sequence.  ONE walk serves the real call and the dry run (ekv_step_check / ekv_step_plan / ekv_step_info): `go`
  // counts a launch and, unless dry, issues it — so ekv_step_info's n_launches IS the sequence below, not a mirror of it.
  int nl = 0;
  auto done = [&](int rc) {
    if (plan_out) plan_out->n_launches = rc == EKV_OK ? nl : 0;
    return rc;
  };
  auto go = [&](int n_kernels, auto&& launch) -> bool {
    nl += n_kernels;
    return dry || launch() == hipSuccess;
  };
  auto chunk = [&](const EkvAttnArgs& A, bool two_pass, const EkvScoreArgs* fuse, int passes, const EkvScoreArgs* tail) -> bool {
    return go(ekv_attn_chunk_launches(A, bank->head_dim, two_pass, passes),
              [&] { return ekv_launch_attn_chunk(A, bank->head_dim, st->layer_count, two_pass, s, fuse, passes, tail); });
  };
  if (fold_in_decode) {
    aa.arrive = bank->arrive + (size_t)st->layer_begin * bank->n_kv_heads;
    aa.out_direct = static_cast<__half*>(out);
  }
  if (ph != 0 && !(ph & 1)) {
  } else if (n == 1) {
    if (!go(1, [&] { return ekv_launch_attn_decode(aa, bank->head_dim, st->layer_count, s); })) return done(EKV_E_LAUNCH);
  } else {
    if (ws.resident && tail_step && ph == 0 && ws.n_split == 1) {
      // small enough for the logits to stay in the register file: one launch, K and V read once (ekv_attn_resident.inc; the planner
      // decides: a step that FORCES the two-pass kernels — ekv_step.two_pass = 1 — or a split keeps them)
      EkvAttnArgs ar = aa;
      ar.out_direct = static_cast<__half*>(out);
      *one_launch = 1;
      return done(go(1, [&] { return ekv_launch_attn_resident(ar, sa, st->layer_count, s); }) ? EKV_OK : EKV_E_LAUNCH);
    }
    if (tail_step && ph == 0) {
      // one pass (output / partials + row statistics) -> fold of the key-range partials, if any -> column-sum pass with the scorer as
      // its tail: two launches for an unsplit head
      sa.skip_fold = 1;
      if (!chunk(aa, true, nullptr, 1, nullptr)) return done(EKV_E_LAUNCH);
      if (!ws.fold_in_kernel && !go(1, [&] { return ekv_launch_fold(sa, st->layer_count, s); })) return done(EKV_E_LAUNCH);
      return done(chunk(aa, true, nullptr, 2, &sa) ? EKV_OK : EKV_E_LAUNCH);
    }
    // (deferred wide two-pass step: only the one pass now — the column-sum pass runs with the scorer at the flush)
    if (!chunk(aa, ws.two_pass != 0, fuse_chunk ? &sa : nullptr, ws.q_keep != nullptr ? 1 : 3, nullptr)) return done(EKV_E_LAUNCH);
  }
  if (ph == 1 || fuse_chunk) return done(EKV_OK);

  if ((ph & 4) || fold_only || range_only) {
    // (not even the fold when the attention kernel has already written the output)
    if (!(ph & 8) || (ph & 4)) {
      if (!ws.fold_in_kernel && !fold_in_decode && !go(1, [&] { return ekv_launch_fold(sa, st->layer_count, s); })) return done(EKV_E_LAUNCH);
    }
    if (fold_only) return done(EKV_OK);
  }
  if (range_only) {
    if (st->n_evict > 0) {
      const bool ok = go(1, [&] {
        hipLaunchKernelGGL(ekv_range_evict_kernel, dim3(bank->n_kv_heads, st->layer_count), dim3(256), (size_t)st->n_evict * 4, s,
                           bank->slot_of_pos, evict_ids, bank->n_kv_heads, bank->cap, st->layer_begin, T, st->range_start, st->n_evict);
        return hipGetLastError();
      });
      return done(ok ? EKV_OK : EKV_E_LAUNCH);
    }
    return done(EKV_OK);
  }
  if (flush_colsum) {
    // the flush of a deferred chunk step: the column-sum pass of ALL layers (queries from the kept copies, the chunk's own rows from
    // the cache slots the one pass of each layer wrote them to), then the scorer — as the tail of that pass where it can be
    EkvAttnArgs a2 = aa;
    a2.q = ws.q_keep;
    a2.q_keep = nullptr;
    a2.new_in_cache = 1;
    a2.q_ts = bank->head_dim, a2.q_hs = n * bank->head_dim;      // (the kept copies are dense)
    if (flush_unsplit) {
      a2.n_stat_parts = ws.n_split;
      a2.n_split = 1;
      a2.rows_per_split = ws.t_pad;
    }
    if (!chunk(a2, true, nullptr, 2, tail_step ? &sa : nullptr)) return done(EKV_E_LAUNCH);
    if (tail_step) return done(EKV_OK);
  }
  sa.skip_fold = ((ph & 8) || ws.fold_in_kernel) ? 1 : 0;
  if (fast_scorer)   // decode steps: the fast scorer (same tail as the fused kernel)
    return done(go(1, [&] { return ekv_launch_decode_score(sa, st->layer_count, s); }) ? EKV_OK : EKV_E_LAUNCH);
  if (st->policy == EKV_POLICY_TOVA && st->tova_head_mean && st->accumulate) {
    if (!go(1, [&] { return ekv_launch_tova_headmean(sa, st->layer_count, s); })) return done(EKV_E_LAUNCH);
  }
  return done(go(1, [&] { return ekv_launch_score_select(sa, st->layer_count, s); }) ? EKV_OK : EKV_E_LAUNCH);
}

extern "C" {

int ekv_step_attend(const ekv_bank* bank, const ekv_step* st, const void* q, const void* k_new, const void* v_new,
                    void* out, int32_t* evict_ids, const float* rope_cos, const float* rope_sin, void* workspace,
                    size_t workspace_bytes, void* stream) {
  return step_attend_impl(bank, st, q, k_new, v_new, out, evict_ids, rope_cos, rope_sin, workspace, workspace_bytes, stream, false, nullptr);
}

int ekv_step_check(const ekv_bank* bank, const ekv_step* st) {
  return step_attend_impl(bank, st, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, true, nullptr);
}

int ekv_gather_ordered(const ekv_bank* bank, int32_t layer_begin, int32_t layer_count, int32_t n_slots, void* k_out,
                       void* v_out, void* stream) {
  if (int e = check_bank(bank)) return e;
  if (int e = check_layers(bank, layer_begin, layer_count)) return e;
  if (!k_out || !v_out || n_slots < 0 || n_slots > bank->cap) return EKV_E_ARG;
  if (n_slots == 0) return EKV_OK;
  drop_stale_error();
  const int rpb = 256 / (bank->head_dim / 8);
  hipLaunchKernelGGL((ekv_rows_copy_kernel<true>), dim3((n_slots + rpb - 1) / rpb, bank->n_kv_heads, layer_count),
                     dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<__half*>(bank->k),
                     static_cast<__half*>(bank->v), bank->slot_of_pos, static_cast<__half*>(k_out),
                     static_cast<__half*>(v_out), bank->n_kv_heads, bank->cap, bank->head_dim, layer_begin, 0, n_slots);
  return launch_status();
}

int ekv_scatter_rows(const ekv_bank* bank, int32_t layer_begin, int32_t layer_count, int32_t pos_begin, int32_t n,
                     const void* k_in, const void* v_in, void* stream) {
  if (int e = check_bank(bank)) return e;
  if (int e = check_layers(bank, layer_begin, layer_count)) return e;
  if (!k_in || !v_in || pos_begin < 0 || n < 0 || pos_begin + n > bank->cap) return EKV_E_ARG;
  if (n == 0) return EKV_OK;
  drop_stale_error();
  const int rpb = 256 / (bank->head_dim / 8);
  hipLaunchKernelGGL((ekv_rows_copy_kernel<false>), dim3((n + rpb - 1) / rpb, bank->n_kv_heads, layer_count),
                     dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<__half*>(bank->k),
                     static_cast<__half*>(bank->v), bank->slot_of_pos,
                     const_cast<__half*>(static_cast<const __half*>(k_in)),
                     const_cast<__half*>(static_cast<const __half*>(v_in)), bank->n_kv_heads, bank->cap, bank->head_dim,
                     layer_begin, pos_begin, n);
  return launch_status();
}

int ekv_rows_to_slots(const ekv_bank* bank, int32_t layer_begin, int32_t layer_count, int32_t n_slots, void* stream) {
  if (int e = check_bank(bank)) return e;
  if (int e = check_layers(bank, layer_begin, layer_count)) return e;
  if (!bank->score_sum || !bank->birth || !bank->slot_state || n_slots < 0 || n_slots > bank->cap) return EKV_E_ARG;
  const size_t lds = (size_t)4 * n_slots * 4;
  if (lds > 150 * 1024) return EKV_E_UNSUPPORTED;
  drop_stale_error();
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ekv_rows_to_slots_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ekv_rows_to_slots_kernel, dim3(bank->n_kv_heads, layer_count), dim3(256), lds, static_cast<hipStream_t>(stream),
                     bank->slot_of_pos, bank->score_sum, bank->score_sq, bank->score_cnt, bank->birth, ekv_cnt_tail(bank), bank->slot_state,
                     bank->n_kv_heads, bank->cap, layer_begin, n_slots);
  return launch_status();
}

int ekv_rows_to_order(const ekv_bank* bank, int32_t layer_begin, int32_t layer_count, int32_t n_slots, void* stream) {
  if (int e = check_bank(bank)) return e;
  if (int e = check_layers(bank, layer_begin, layer_count)) return e;
  if (!bank->score_sum || !bank->birth || !bank->slot_state || n_slots < 0 || n_slots > bank->cap) return EKV_E_ARG;
  const size_t lds = (size_t)4 * bank->cap * 4;
  if (lds > 150 * 1024) return EKV_E_UNSUPPORTED;
  drop_stale_error();
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ekv_rows_to_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ekv_rows_to_order_kernel, dim3(bank->n_kv_heads, layer_count), dim3(256), lds, static_cast<hipStream_t>(stream),
                     bank->slot_of_pos, bank->score_sum, bank->score_sq, bank->score_cnt, bank->birth, ekv_cnt_tail(bank), bank->slot_state,
                     bank->n_kv_heads, bank->cap, layer_begin, n_slots);
  return launch_status();
}

int ekv_compact_inplace(const ekv_bank* bank, int32_t layer_begin, int32_t layer_count, int32_t n_slots, int32_t n_evict,
                        const int32_t* evict_ids, void* stream) {
  if (int e = check_bank(bank)) return e;
  if (int e = check_layers(bank, layer_begin, layer_count)) return e;
  if (!evict_ids || n_evict <= 0 || n_evict >= n_slots || n_slots > bank->cap) return EKV_E_ARG;
  drop_stale_error();
  static const int ch = [] { const char* e = std::getenv("EKV_COMPACT_CH"); return e ? std::atoi(e) : 16; }();   // (tuning knob; 4 / 8 / 16 rows per thread in flight: 4.3 / 4.5 / 4.65 TB/s)
#define EKV_CI(CHV) hipLaunchKernelGGL((n_evict == 1 ? ekv_compact_inplace_kernel<CHV, true> : ekv_compact_inplace_kernel<CHV, false>), dim3(2, bank->n_kv_heads, layer_count), dim3(256), (size_t)n_evict * 4, \
                     static_cast<hipStream_t>(stream), static_cast<__half*>(bank->k), static_cast<__half*>(bank->v),                      \
                     evict_ids, bank->n_kv_heads, bank->cap, bank->head_dim, layer_begin, n_slots, n_evict)
  if (ch <= 2) EKV_CI(2); else if (ch <= 4) EKV_CI(4); else if (ch <= 8) EKV_CI(8); else EKV_CI(16);
#undef EKV_CI
  return launch_status();
}

}  // extern "C"
