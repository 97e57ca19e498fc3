/*
 * easykv_hip.h — C ABI of the MI355X-native budgeted-KV-cache attention path.
 *
 * Drop-in boundary for the ONE hot path of DRSY/EasyKV (reference paths relative to
 * /root/reference):  per layer and per model forward the reference does, in Python,
 *
 *   patched attention forward      easykv/llama_patch.py:198-222, :310-327   (mistral_patch.py:144-169, :225-241)
 *   GQA fold of the probabilities  easykv/easykv.py:188-196
 *   score accumulation             easykv/easykv.py:287-300, :443-457, :693-707
 *   victim selection               easykv/easykv.py:310-347, :462-493, :711-742
 *   K/V compaction                 easykv/easykv.py:56-82, :105-112
 *   score-state compaction         easykv/easykv.py:315-333, :465-490
 *
 * The reference has no FFI of its own (it is 100 % Python); the entry points below are what a
 * ctypes binding for this path binds (see INTEGRATION.md for the reference-side stub).  Plain
 * pointers and sizes only, no torch types.  Every launch is asynchronous on the `stream`
 * argument (a hipStream_t passed as void*), re-entrant per stream, and keeps no global state.
 * Return value: 0 on success, a negative EKV_E_* code otherwise (ekv_strerror() describes it).
 *
 * Memory model ("bank" = the layers resident on this GPU, all device memory, caller-owned):
 *
 *   k, v          fp16  [n_layers][n_kv_heads][cap][head_dim]   physical rows ("slots")
 *   slot_of_pos   int32 [n_layers][n_kv_heads][cap]             permutation: logical position -> row.
 *                       positions [0, n_slots) are live, in the reference's birth order;
 *                       positions [n_slots, cap) hold the free rows.
 *   score_sum     fp32  [n_layers][n_kv_heads][cap]             S  (sum p),      index j <-> position score_off + j
 *   score_sq      fp32  [n_layers][n_kv_heads][cap]             Q  (sum p^2)     (roco only)
 *   score_cnt     fp32  [n_layers][n_kv_heads][cap]             C  (#queries)    (roco only)
 *   arrive        u32   [n_layers][n_kv_heads]                  optional arrival counters (see ekv_bank)
 *
 * Eviction never moves K/V rows: the victim's row is recycled for the next token and only the
 * 4-byte slot map is compacted (order-preserving, exactly the reference's list semantics).
 * ekv_gather_ordered() materialises the ordered [T][D] view the HF legacy-tuple boundary needs;
 * ekv_compact_inplace() is the reference-shaped physical compaction for banks kept in identity
 * layout.
 */
#ifndef EASYKV_HIP_H
#define EASYKV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EKV_ABI_VERSION 8

/* kv_policy strings of the reference -> codes (easykv/easykv.py:288-300, :310-362) */
enum {
  EKV_POLICY_NONE = 0,     /* 'full' or any unknown string: attention only                    */
  EKV_POLICY_H2O_HEAD = 1, /* 'h2o_head'                                                      */
  EKV_POLICY_ROCO = 2,     /* 'roco'                                                          */
  EKV_POLICY_TOVA = 3,     /* 'tova'                                                          */
  EKV_POLICY_RANGE = 4     /* 'recency' / 'random': host-chosen contiguous range, all heads   */
};

enum {
  EKV_OK = 0,
  EKV_E_ARG = -1,          /* inconsistent sizes / null pointer          */
  EKV_E_UNSUPPORTED = -2,  /* head_dim / group size / width not built    */
  EKV_E_WORKSPACE = -3,    /* workspace too small                        */
  EKV_E_LAUNCH = -4        /* hipLaunch failure (hipGetLastError)        */
};

typedef struct ekv_bank {
  void *k, *v;
  int32_t *slot_of_pos;
  float *score_sum, *score_sq, *score_cnt;
  int32_t n_layers, n_q_heads, n_kv_heads, head_dim, cap;
  uint32_t *arrive; /* optional (ABI 3), uint32 [n_layers][n_kv_heads], zeroed by ekv_bank_reset and left zero by every call:
                       arrival counters of the split decode path.  With it the key-range partials of a head are folded inside the
                       attention kernel by the last split to arrive (no fold launch); NULL = separate fold kernel.            */
  int32_t *birth;   /* optional (ABI 6), int32 [2][n_layers][n_kv_heads][cap] (order keys; the parked tail of the count rows) + */
  float *slot_state;/* float [n_layers][n_kv_heads][4]: the slot-indexed layout of the score rows (EKV_PHASE_SLOT_ROWS); NULL = the
                       bank only ever holds the ordered layout                                                                */
} ekv_bank;

/* One model forward over `layer_count` layers starting at `layer_begin` (all layers share the
 * geometry, as in the reference where every layer's cache has the same length). */
typedef struct ekv_step {
  int32_t layer_begin, layer_count;
  int32_t q_len;        /* 1 = decode, stride = prefill chunk                                            */
  int32_t n_slots;      /* T: live positions INCLUDING the q_len new ones                                */
  int32_t score_off;    /* P: first position covered by the score rows (decoding mode: prompt length)    */
  int32_t policy;       /* EKV_POLICY_*                                                                   */
  int32_t accumulate;   /* 1: add this forward's probabilities to the score rows                         */
  int32_t n_evict;      /* victims per (layer, head) after the attention; 0 = none                       */
  int32_t win_lo;       /* h2o/tova: first candidate index; roco: leading indices forced to 1e9 (sink)   */
  int32_t win_tail;     /* h2o/tova: trailing indices excluded (recent window)                           */
  int32_t roco_k1;      /* roco stage 1: size of the smallest-std feasible set                           */
  int32_t roco_tail;    /* roco: trailing indices forced to 1e9 (10 in the reference)                    */
  int32_t range_start;  /* EKV_POLICY_RANGE: evict positions [range_start, range_start + n_evict)        */
  int32_t tova_head_mean; /* encoding/ppl 'tova': one head-averaged last-query row for all heads (:456)  */
  int32_t causal;       /* 1: the last q_len positions are the chunk itself, causal inside it            */
  int32_t rope_on_read; /* 1: streaming variant, rotate keys by position index at read time             */
  int32_t n_split;      /* key-range splits per head (0 = choose)                                        */
  int32_t phases;       /* 0 = whole step; else bit mask: 1 attention kernel, 2 scorer (fold+score), 4 fold only, 8 scorer
                           without the fold (1|4 then 8 lets the caller run the scorer on a side stream)          */
  float count_add;      /* added to C before selection (1 decode, stride prefill); 0 = leave             */
  float count_tail_step;/* C tail after compaction: tail[i] = i * count_tail_step (0 decode, -1 prefill) */
  float sm_div;         /* logits are divided by this (sqrt(head_dim))                                   */
  int32_t two_pass;     /* scored chunk steps: 0 = library decides (two passes from 40 GQA-folded query rows, one pass below,
                           for tova and head_dim 32 with rope_on_read; 9..64 rows against <= 1280 plain keys — <= 32 rows: 2560 — at head_dim 128: the
                           one-launch logits-resident kernel), 1 = two passes — on the wide-block kernel one pass over K and V (output +
                           row statistics) and a K-only column-sum pass; on the 16x16 kernel a statistics pass + an exact pass —
                           whenever the shape allows it, -1 = always one pass with exported logits                      */
  int32_t phys_extent;  /* E: every live row of these layers has a physical index < E (n_slots <= E <= cap), and rows
                           [0, E) hold initialised memory.  0 = unknown (cap is used).  The one-launch decode step streams
                           the rows [0, E) in PHYSICAL order (sequential HBM reads however fragmented the slot map is) and
                           masks the dead ones, so a tight E saves reading free rows.  With the free list kept as this
                           library keeps it (freed rows first, then never-used rows ascending) E is simply the high-water
                           mark of n_slots.                                                                          */
  int32_t defer_layers; /* > 0: DEFERRED SCORER for a model that issues one layer per call (a real decoder stack: layer l + 1's
                           query depends on layer l's output, easykv/easykv.py:264-269).  The attention output of a layer is
                           needed at once, its scoring / eviction only before the NEXT forward: the per-layer call (decode steps,
                           and since ABI 5 chunk steps of any q_len; phases = 1|4: attention + fold — for two-pass steps of the
                           wide-block kernel the column-sum pass is deferred as well and runs in the phases = 8 call) leaves its logits and partials in slice `defer_index` of a workspace laid
                           out for `defer_layers` layers, and ONE call with phases = 8 over all those layers
                           (layer_count = defer_layers, defer_index = 0) scores and evicts for the whole token.  Both calls must
                           pass the same explicit n_split and the same workspace.  0 = off.                               */
  int32_t defer_index;  /* slice of this call's first layer in the deferred workspace                                      */
  /* Row strides of the call's tensors, in ELEMENTS (ABI 8; SURVEY.md §8b "raw device pointers + explicit strides").  A row is the
   * head_dim contiguous halfs of one (head, token); inside a layer's block it sits at  head * head_stride + token * token_stride,
   * and the layers of a multi-layer call follow each other n_heads * q_len * head_dim elements apart.  Both zero = the dense layout
   * [heads][q_len][head_dim] (token_stride = head_dim, head_stride = q_len * head_dim).  The patched attention modules of HF
   * transformers hand over [1, heads, q_len, head_dim] VIEWS of the projections' [1, q_len, heads * head_dim] output
   * (llama_patch.py:176-182 does the same transpose): token_stride = heads * head_dim, head_stride = head_dim — read in place, no
   * copy kernels in front of the step, and `out` written straight in the [q_len][heads * head_dim] layout o_proj wants
   * (llama_patch.py:230-232 transposes back).  Strides must be multiples of 8 (16-byte row loads); for q_len = 1 only
   * head_stride = head_dim is accepted (token_stride is irrelevant).                                                      */
  int32_t q_token_stride, q_head_stride;      /* q                                                                          */
  int32_t kv_token_stride, kv_head_stride;    /* k_new and v_new                                                            */
  int32_t out_token_stride, out_head_stride;  /* out                                                                        */
} ekv_step;

int ekv_abi_version(void);
const char *ekv_strerror(int code);

/* bytes of scratch ekv_step_attend needs for (bank, step) */
size_t ekv_workspace_bytes(const ekv_bank *bank, const ekv_step *step);

/* How ekv_step_attend will run (bank, step): *n_split = key-range splits per head, *fused = 1 when the whole
 * step is ONE launch (the fused decode kernel, the logits-in-LDS chunk kernel, or a chunk step whose scorer runs as the tail
 * of its attention kernel), 0 when it is an attention launch (two for the two-pass chunk scheme) + fold / scorer launches. */
int ekv_step_plan(const ekv_bank *bank, const ekv_step *step, int32_t *n_split, int32_t *fused);

/* The dispatch decisions behind ekv_step_plan, for host code that must shape its calls after them instead of mirroring the rules
 * (ABI 5): info[0 .. n_info) <- { n_split, fused, two_pass (1 = statistics + column-sum scheme, no logits in HBM), wide (1 = the
 * 32x32x16 wide-block kernel, which walks all query blocks of a head inside one launch), n_qblocks, qb_rows, n_col_parts,
 * fold_in_kernel, n_launches (ABI 7: kernel launches this call issues — a two-pass wide step is 2 since the scorer became the tail
 * of its column-sum pass, 3 before; 0 = the dispatch refuses the step) }; entries beyond EKV_STEP_INFO_N are zeroed.  The reference has no counterpart (its keep_attention prefix
 * materialises the r x r map, easykv/easykv.py:396-405); easykv_amd.api uses it to decide whether a scored prefix goes down as
 * one step or in query blocks. */
#define EKV_STEP_INFO_N 9
int ekv_step_info(const ekv_bank *bank, const ekv_step *step, int32_t *info, int32_t n_info);

/* slot_of_pos <- identity for the whole bank */
int ekv_bank_reset(const ekv_bank *bank, void *stream);

/* Score-row initialisation for `layer_count` layers from `layer_begin` over width W:
 *   S = Q = 0;  C[j] = c0 - j for j < ramp_from ... see easykv/easykv.py:242-245, :412-416.
 * mode 0 (decoding, :245):            C[j] = (W-1) - j
 * mode 1 (prefill, keep_attention):   C[j] = (W - j) - stride
 * mode 2 (prefill, no keep):          C[j] = 0 for j < W-stride, else -(j-(W-stride))            */
int ekv_state_init(const ekv_bank *bank, int32_t layer_begin, int32_t layer_count, int32_t width, int32_t mode,
                   int32_t stride, void *stream);

/* The fused step: append q_len new K/V rows, attention of the q_len queries over the n_slots live
 * positions, GQA fold, score accumulation, victim selection, slot-map + score-row compaction.
 *   q      fp16 [layer_count][n_q_heads][q_len][head_dim]    (dense, or rows at ekv_step.q_*_stride)
 *   k_new  fp16 [layer_count][n_kv_heads][q_len][head_dim]   (already rotated unless rope_on_read; ekv_step.kv_*_stride)
 *   v_new  fp16 [layer_count][n_kv_heads][q_len][head_dim]   (same strides as k_new)
 *   out    fp16 [layer_count][n_q_heads][q_len][head_dim]    (ekv_step.out_*_stride)
 *   evict_ids int32 [layer_count][n_kv_heads][n_evict] or NULL: evicted logical positions, ascending
 *   rope_cos/rope_sin fp32 [>= n_slots][head_dim] or NULL; layout cat(freqs, freqs) as in the reference (llama_patch.py:74-98):
 *                    the kernels read the first half of a row for both halves of the head.  The tables must be what RoPE tables are —
 *                    a rotation LINEAR in the position, row j = (cos(j * theta_f), sin(j * theta_f)) for per-frequency angles
 *                    theta_f of any scaling (linear / NTK / llama3 / yarn frequencies) — : the decode stream reads only the first
 *                    row of every run of consecutive positions from the table and advances (cos, sin) to the following rows by the
 *                    angle-addition recurrence with row 1 as the step (easykv_amd.KVBank.set_rope verifies the property)
 * After the call the bank holds n_slots - n_evict live positions (the caller tracks that number). */
int ekv_step_attend(const ekv_bank *bank, const ekv_step *step, const void *q, const void *k_new, const void *v_new,
                    void *out, int32_t *evict_ids, const float *rope_cos, const float *rope_sin, void *workspace,
                    size_t workspace_bytes, void *stream);

/* Dry run of ekv_step_attend (ABI 4): every argument / shape / capability test of the real call for (bank, step), in the same
 * order, and nothing launched — no data pointer, workspace or stream is needed.  Returns what ekv_step_attend would return
 * before its first launch (EKV_OK, EKV_E_ARG, EKV_E_UNSUPPORTED).  For callers that split a step over several calls (the
 * deferred scorer: per-layer phases = 1|4 calls append rows long before the phases = 8 call that scores them), so that a shape
 * the LAST call would refuse is refused before the FIRST one touches the bank. */
int ekv_step_check(const ekv_bank *bank, const ekv_step *step);

/* Slot-indexed score rows (ABI 6).  In the ordered layout an eviction shifts every entry of S / Q / C and of the slot map behind
 * the victim: a decode step rewrites all of them (28 KB per head at 2 k slots), and written bytes cost about twice what read
 * bytes do in the middle of the K/V stream.  ekv_rows_to_slots re-indexes the score rows of `layer_count` layers (n_slots live
 * entries each) by PHYSICAL row — S[row], Q[row], a count base C0[row] in score_cnt (count = C0 + the head's running sum of
 * count_add, slot_state[head][0]) and an order key birth[row] (the rank of `birth` among the live rows is the entry's order index;
 * the next birth is slot_state[head][1] as int32; [2] = the last decode step's roco threshold key, a hint for the next select; [3] = the same for the logits-in-LDS chunk kernel,
 * which uses it on the ordered layout too) — after which a step with EKV_PHASE_SLOT_ROWS set in `phases` moves nothing on an
 * eviction: the victim's row goes to the front of the free list (slot_of_pos[n_slots - 1]), S / Q of the live rows are rewritten,
 * C0 / birth only for the appended row.  Decisions (victims, ties to the older entry) and reported ids (order indices) are those
 * of the ordered layout.  While a layer is in this layout slot_of_pos[0, n_slots) is undefined and only steps with the flag may
 * run on it; ekv_rows_to_order converts back (slot map, score rows and zero tails exactly as the ordered steps would have left
 * them, counts included while all count_add were integers).  ekv_step_check / ekv_step_attend return EKV_E_UNSUPPORTED for a
 * flagged step the layout does not cover: anything but a one-launch decode step (q_len = 1, phases otherwise 0) with plain keys, a
 * scored policy with accumulate, score_off = 0, win_lo = 0, n_evict <= 1, GQA factor <= 4, phys_extent <= 6144, cap <= 9600, and a bank
 * that carries score_sq and score_cnt (the count base of every policy's appended row lives in score_cnt). */
#define EKV_PHASE_SLOT_ROWS 16
/* with EKV_PHASE_SLOT_ROWS: the caller guarantees that this step's protected tail (roco: 10 entries; h2o_head: win_tail) is not
 * longer than that of any earlier evicting step since ekv_rows_to_slots — the newest entries then have consecutive births and the
 * kernel skips the counting check (one block reduction) that otherwise proves it, with an exact bisection when it fails */
#define EKV_PHASE_SLOT_TAIL_OK 32
int ekv_rows_to_slots(const ekv_bank *bank, int32_t layer_begin, int32_t layer_count, int32_t n_slots, void *stream);
int ekv_rows_to_order(const ekv_bank *bank, int32_t layer_begin, int32_t layer_count, int32_t n_slots, void *stream);

/* Ordered view: k_out/v_out fp16 [layer_count][n_kv_heads][n_slots][head_dim] <- rows in position order */
int ekv_gather_ordered(const ekv_bank *bank, int32_t layer_begin, int32_t layer_count, int32_t n_slots, void *k_out,
                       void *v_out, void *stream);

/* Load ordered rows into the bank at positions [pos_begin, pos_begin+n) (prefix prefill / cache import):
 *   k_in/v_in fp16 [layer_count][n_kv_heads][n][head_dim] */
int ekv_scatter_rows(const ekv_bank *bank, int32_t layer_begin, int32_t layer_count, int32_t pos_begin, int32_t n,
                     const void *k_in, const void *v_in, void *stream);

/* Reference-shaped physical compaction (easykv/easykv.py:56-82) for a bank in identity layout:
 * removes `n_evict` ascending positions per (layer, head) from the first n_slots rows in place,
 * order-preserving.  evict_ids int32 [layer_count][n_kv_heads][n_evict]. */
int ekv_compact_inplace(const ekv_bank *bank, int32_t layer_begin, int32_t layer_count, int32_t n_slots,
                        int32_t n_evict, const int32_t *evict_ids, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EASYKV_HIP_H */
