/*
 * buglab_hip.h -- C ABI of libbuglab_hip.so, the MI355X (gfx950) kernels behind BugLab's
 * `gnn-mlp` message-passing + scoring-head hot path.
 *
 * Conventions:
 *   - every function is extern "C", returns 0 on success, a positive hipError_t or a negative
 *     BL_E* argument error otherwise; bl_last_error() gives the text for the calling thread;
 *   - the CALLER owns every buffer (PyTorch-ROCm tensors -> data_ptr()); nothing here allocates;
 *   - all index arrays are int32, all values float32, matrices row-major with an explicit
 *     leading dimension where one is given; pointers must be 16-byte aligned and every leading
 *     dimension / width a multiple of 4 floats;
 *   - the last argument is the hipStream_t to launch on
 *     (torch.cuda.current_stream().cuda_stream); entry points never synchronise.
 *
 * The reference has no native code at all: each entry point below names the reference Python
 * (or the third-party op called from it) that it replaces.  Paths are relative to
 * /root/reference.
 */
#ifndef BUGLAB_HIP_H
#define BUGLAB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BL_OK 0
#define BL_EINVAL (-1) /* bad argument (null pointer, misaligned, unsupported size) */
#define BL_ERANGE (-2) /* dimension outside what the kernels were built for */

/* activation codes shared by every entry point.  BL_ACT_GELU_AGG is accepted only where a segmented max is involved
 * (bl_segment_max_fwd / _bwd, bl_mp_layer_t.msg_act): GELU applied to the AGGREGATE, out = gelu(max_i x_i), instead of to every
 * item before the max, out = max_i gelu(x_i) (BL_ACT_GELU) -- the two placements of ptgnn's `message_activation`. */
enum { BL_ACT_NONE = 0, BL_ACT_RELU = 1, BL_ACT_SIGMOID = 2, BL_ACT_TANH = 3, BL_ACT_GELU = 4, BL_ACT_GELU_AGG = 5 };

const char* bl_last_error(void);
int bl_version(void);
/* Deterministic-gradient mode (also switched on by BL_DETERMINISTIC=1 in the environment at first use).  The kernels that
 * add into one address from several workgroups -- the split-K weight gradients (bl_gemm_wgrad*, bl_gemm_wgrad_routed_x6),
 * the column sums of bl_layernorm_bwd / bl_act_bwd / bl_rowdot_bwd -- then flush in a fixed workgroup order (a turn
 * counter per output tile), bl_scatter_add_rows and bl_embed_subtoken_max_bwd walk their rows serially per column: every
 * gradient is bit-identical from run to run, at a cost (the flushes serialise).  bl_embed_subtoken_max_bwd_sorted is
 * reproducible when every token is ONE chunk (the collator does that in this mode).  Not covered: the relational-attention
 * bias gradients (bl_rel_attn_bias_bwd, bl_rel_value_bias_bwd).  The reference has no counterpart (torch's index_add /
 * scatter backward are atomic too); this exists so that a parity failure can be reproduced. */
void bl_set_deterministic(int32_t on);
int32_t bl_get_deterministic(void);

/* Rows of a (virtually) concatenated, gathered operand:
 *   row r = [ x[0][idx[0][r], 0:width[0]] ; x[1][idx[1][r], 0:width[1]] ; ... ]   (nsrc <= 3)
 * idx[j] == NULL means the identity (row r of x[j]).  This is how `torch.cat((h[src], h[tgt]), -1)`
 * and the heads' `torch.cat((a[i], b[j]), -1)` are consumed without ever being materialised. */
typedef struct {
  const float* x[3];
  const int32_t* idx[3];
  int32_t ld[3];
  int32_t width[3];
  int32_t nsrc;
} bl_rows_t;

/* counter-based dropout (oracle: oracle/buglab_oracle.py::dropout_keep_mask).  p == 0 disables. */
typedef struct {
  float p;
  uint32_t seed;
  uint32_t stream;
} bl_dropout_t;

/* ---------------------------------------------------------------------------------------------
 * M0  node embedder: out[n,:] = Dropout(max_{s < lens[n]} table[ids[n,s], :]), argsub = winning s.
 * Replaces ptgnn StrElementRepresentationModel (configured at buglab/models/modelregistry.py:59-82:
 * subtoken splitting, <= 6 subtokens, "max" combination).
 * bwd: g_table[ids[n, argsub[n,h]], h] += g_out[n,h] (masked by the same dropout), fp32 atomics.
 * drop_before_pool: 0 (default spec) = dropout on the pooled rows, out = drop(max_s emb); 1 = dropout on the embedded subtokens
 * before the pooling, out = max_s drop(emb)[n, s, :] with mask index (n S + s) H + h (the other placement ptgnn's
 * SubtokenUnitEmbedder may have: DESIGN.md section 2); backward scales by the winner's mask bit. */
int bl_embed_subtoken_max_fwd(const float* table, int32_t V, int32_t H, const int32_t* ids, const int32_t* lens,
                              int32_t N, int32_t S, bl_dropout_t drop, int32_t drop_before_pool, float* out, int32_t ld_out,
                              int8_t* argsub, void* stream);
int bl_embed_subtoken_max_bwd(const float* g_out, int32_t ld_g, const int32_t* ids, const int8_t* argsub, int32_t N,
                              int32_t S, int32_t H, int32_t V, bl_dropout_t drop, int32_t drop_before_pool, float* g_table,
                              void* stream);
/* The same gradient from a token-sorted occurrence list: occ[i] = n * S + s (every valid subtoken slot),
 * sorted by ids[n, s] and cut in chunks of one token each (chunk_ptr [nchunks + 1], chunk_tok [nchunks];
 * the collator uses <= 256 occurrences per chunk).  One atomic per (chunk, channel) instead of one per
 * (node, channel): subtoken frequencies are Zipfian and same-address atomics serialise. */
int bl_embed_subtoken_max_bwd_sorted(const float* g_out, int32_t ld_g, const int32_t* occ, const int32_t* chunk_ptr,
                                     const int32_t* chunk_tok, int32_t nchunks, const int8_t* argsub, int32_t S,
                                     int32_t H, bl_dropout_t drop, int32_t drop_before_pool, float* g_table, void* stream);

/* The embedder with the node model's other `subtoken_combination` values (reference buglab/models/modelregistry.py:65-66 sets "max"
 * unless the caller's node_representations says otherwise; ptgnn's StrElementRepresentationModel also takes "sum" and "mean"):
 * combination BL_POOL_MAX = the three entry points above; BL_POOL_SUM: out[n] = drop(sum_{s < lens[n]} emb[n, s]);
 * BL_POOL_MEAN: the sum / lens[n].  argsub is used by BL_POOL_MAX only (NULL otherwise); bwd needs lens for sum / mean (every
 * slot s < lens[n] receives g_out[n] (/ lens[n]), under its own mask bit with drop_before_pool); the sorted form needs lens for
 * BL_POOL_MEAN. */
#define BL_POOL_MAX 0
#define BL_POOL_SUM 1
#define BL_POOL_MEAN 2
int bl_embed_subtoken_pool_fwd(const float* table, int32_t V, int32_t H, const int32_t* ids, const int32_t* lens, int32_t N, int32_t S,
                               int32_t combination, bl_dropout_t drop, int32_t drop_before_pool, float* out, int32_t ld_out,
                               int8_t* argsub, void* stream);
int bl_embed_subtoken_pool_bwd(const float* g_out, int32_t ld_g, const int32_t* ids, const int32_t* lens, const int8_t* argsub, int32_t N,
                               int32_t S, int32_t H, int32_t V, int32_t combination, bl_dropout_t drop, int32_t drop_before_pool,
                               float* g_table, void* stream);
int bl_embed_subtoken_pool_bwd_sorted(const float* g_out, int32_t ld_g, const int32_t* occ, const int32_t* chunk_ptr,
                                      const int32_t* chunk_tok, int32_t nchunks, const int32_t* lens, const int8_t* argsub, int32_t S,
                                      int32_t H, int32_t combination, bl_dropout_t drop, int32_t drop_before_pool, float* g_table,
                                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Grouped, gathered fp32 GEMM on MFMA (v_mfma_f32_32x32x2_f32, exact fp32).
 *   for group g (rows group_ptr[g] .. group_ptr[g+1]):  C[r, 0:N] = drop(act(rows(a)[r, 0:K] . B_g + bias))
 *   B_g = b + group_w[g] * b_group_stride ; if b_is_nk == 0, B_g is [K, N] row-major (ldb),
 *   else B_g is [N, K] row-major and is used transposed (C = A . B_g^T).
 * group_ptr == NULL: one group covering rows 0..M.  group_w == NULL: identity.
 * Replaces, per edge type, ptgnn MlpMessagePassingLayer's gather + cat + Linear
 * (call site buglab/models/gnnlayerdefs.py:6-23), ptgnn's dense node update, and the heads'
 * nn.Linear / MLP layers (buglab/models/layers/mlp.py:6-20, localizationmodule.py:22-24, 56-60,
 * fixermodules.py:36-39, 71-73, 116-124).  With b_is_nk = 1 it is the input-gradient GEMM. */
int bl_gemm_rows(const bl_rows_t* a, const float* b, int64_t b_group_stride, int32_t ldb, int32_t b_is_nk,
                 const float* bias, const int32_t* group_ptr, const int32_t* group_w, int32_t G, int32_t M, int32_t N,
                 int32_t K, int32_t act, bl_dropout_t drop, float* c, int32_t ldc, void* stream);

/* Input-gradient GEMM with the max-aggregation ROUTING folded into the operand load:
 *   row r of the left operand is  G[r, k] = (winner[idx[r], k] == r) ? a.x[0][idx[r], k] : 0
 * (a has exactly one gathered source: a.x[0] = d loss / d aggregate per node [N, K], idx = target node
 * of message r; `winner` = the arg-max table of bl_segment_max_fwd), then C = G . B_g^T as in
 * bl_gemm_rows with b_is_nk = 1.  Replaces torch_scatter's scatter_max backward (a gather of the
 * gradient at arg) + the autograd of the per-type Linear, without materialising the [E, Dm] gradient. */
int bl_gemm_rows_routed(const bl_rows_t* a, const int32_t* winner, int32_t ld_winner, const float* b,
                        int64_t b_group_stride, int32_t ldb, const int32_t* group_ptr, const int32_t* group_w, int32_t G,
                        int32_t M, int32_t N, int32_t K, float* c, int32_t ldc, void* stream);

/* ---------------------------------------------------------------------------------------------
 * fp32-accurate GEMM on the bf16 matrix cores ("bf16x6"): every fp32 operand is split once into
 * three bf16 terms hi + mid + lo (bl_pack_bf16x3*: every row is three bf16 planes back to
 * back, [hi x D | mid x D | lo x D]) and a product is evaluated as the six MFMA terms hh + hm + mh + hl + lh + mm
 * with fp32 accumulation; dropped terms are < 2^-26 of the product, below fp32's own rounding.
 * Same contract as bl_gemm_rows (C[rows of g] = A . B_g), B given by bl_pack_weights_x6, no
 * bias/activation epilogue.  win_bits != NULL selects the routed left operand of bl_gemm_rows_routed,
 * with the routing given as bl_segment_max_fwd's per-message bitmask (row r keeps channel k iff bit k
 * of win_bits[r * ld_bits ...] is set; ld_bits in 32-bit words).  Source widths: multiples of 32. */
typedef struct {
  const uint16_t* xp[3];  /* packed matrices (bl_pack_bf16x3) */
  const int32_t* idx[3];  /* row gather index or NULL */
  int32_t width[3];       /* k's taken from each source (the packed matrix's D) */
  int32_t nsrc;
} bl_rows_packed_t;
int bl_pack_bf16x3(const float* x, int32_t ld, int64_t R, int32_t D, uint16_t* out, void* stream);
/* the same into the columns col_off .. col_off + D of a packed matrix whose rows are D_total wide: how the
 * [stash ; current] input of a ConcatResidual layer (gnnlayerdefs.py:24-38) is packed without a concatenated copy */
int bl_pack_bf16x3_cols(const float* x, int32_t ld, int64_t R, int32_t D, int32_t D_total, int32_t col_off, uint16_t* out,
                        void* stream);
/* Weights of the bf16x6 row GEMM, packed and TILED: per group ceil(N/128) * (K/32) blocks of 24 KB
 * (12288 uint16), block (tile, stage) = [i (2)][plane (3)][row_lo (64)][k-group (4)][8] for column
 * n = 128 tile + 64 i + row_lo and k = 32 stage + 8 k-group + 0..7; columns past N are zero.
 * w is [G][K][N] if w_is_kn (forward weights W[t]: C = A . W) or [G][N][K] (C = A . w^T). */
int bl_pack_weights_x6(const float* w, int32_t G, int32_t K, int32_t N, int32_t w_is_kn, uint16_t* out, void* stream);
/* All the operand copies of the weights a training step needs, in ONE launch (they are re-made after every optimiser
 * step): a table of jobs in DEVICE memory, built once per model.  kind 0 / 1 = bl_pack_weights_x6 with w_is_kn = kind
 * (w [G][N][K] / [G][K][N]); kind 2 = the fp32 transpose out[g][n][k] = w[g][k][n] (the W^T of bl_routed_dgrad_nodes);
 * kind 3 / 4 = bl_pack_weights_x6w (the wide row GEMM's image) with w_is_kn = kind - 3.
 * first_block = sum of bl_pack_job_blocks(...) of the jobs before; total_blocks = that sum over all jobs. */
typedef struct {
  const float* w;
  void* out;
  int32_t kind, G, K, N;
  int32_t first_block, pad_;
} bl_pack_job_t;
int64_t bl_pack_job_blocks(int32_t kind, int32_t G, int32_t K, int32_t N);
int bl_pack_weights_multi(const bl_pack_job_t* jobs_device, int32_t njobs, int32_t total_blocks, void* stream);
int bl_gemm_rows_x6(const bl_rows_packed_t* a, const uint32_t* win_bits, int32_t ld_bits, const uint16_t* bp,
                    int64_t b_group_stride, const int32_t* group_ptr, const int32_t* group_w, int32_t G, int32_t M,
                    int32_t N, int32_t K, float* c, int32_t ldc, void* stream);
/* Wide form of bl_gemm_rows_x6 (csrc/bl_gemm_x6w.hip): 128 x 256 output tile, 8 waves, operands DMA'd into a double-buffered
 * LDS image (global_load_lds), ping-pong schedule.  Same contract, same summation order per element -- results are bit-identical
 * to bl_gemm_rows_x6 -- for the message GEMMs with >= 256 output columns and long K (ptgnn MlpMessagePassingLayer's per-type
 * Linear and its input gradient, call site buglab/models/gnnlayerdefs.py:6-23: the ConcatResidual layers of the hidden-128
 * model, every layer at hidden 256).  Shapes the layer calls send here: bl_gemm_rows_x6w_ok(N, K) -- N a multiple of 256, K a
 * multiple of 64 and >= 256 (the entry point itself takes any K >= 64 that is a multiple of 64).  The
 * weights come in their own image, bl_pack_weights_x6w (a 48 KB block per group, 256-column tile and 32-k stage in the exact
 * LDS layout; bl_packed_weight_elems_x6w uint16 elements; kinds 3 / 4 of bl_pack_weights_multi).  bl_set_rows_tile(128) makes bl_gemm_rows_x6w_ok return 0 (measurement switch; returns the previous tile). */
int32_t bl_gemm_rows_x6w_ok(int32_t N, int32_t K);
int32_t bl_set_rows_tile(int32_t cols);
int64_t bl_packed_weight_elems_x6w(int32_t G, int32_t K, int32_t N);
int bl_pack_weights_x6w(const float* w, int32_t G, int32_t K, int32_t N, int32_t w_is_kn, uint16_t* out, void* stream);
int bl_gemm_rows_x6w(const bl_rows_packed_t* a, const uint32_t* win_bits, int32_t ld_bits, const uint16_t* bp,
                     int64_t b_group_stride, const int32_t* group_ptr, const int32_t* group_w, int32_t G, int32_t M,
                     int32_t N, int32_t K, float* c, int32_t ldc, void* stream);
/* the same with bl_gemm_rows' epilogue, C = drop(act(A . B_g + bias)): ptgnn MlpMessagePassingLayer's dense node update
 * Linear -> tanh -> Dropout (call site buglab/models/gnnlayerdefs.py:6-23) on the bf16 matrix cores */
int bl_gemm_rows_x6_epi(const bl_rows_packed_t* a, const uint16_t* bp, int64_t b_group_stride, const int32_t* group_ptr,
                        const int32_t* group_w, int32_t G, int32_t M, int32_t N, int32_t K, const float* bias, int32_t act,
                        bl_dropout_t drop, float* c, int32_t ldc, void* stream);
/* Extended epilogues of the plain (one group) bf16x6 row GEMM: what the Linear layers of the relational transformer block
 * (reference buglab/models/layers/relational_transformer.py:104-124, multihead_attention.py:27-35) hand to the next kernel
 * without an elementwise pass in between (csrc/bl_great_layer.hip).
 *   BL_X6_EPI_ACT        c = drop(act(A . B + bias)), fp32                                     (= bl_gemm_rows_x6_epi)
 *   BL_X6_EPI_ACT_PACK   the same, written ONLY in bl_pack_bf16x3's form to c_packed [M, 3 N]  (act = relu: linear1 -> linear2)
 *   BL_X6_EPI_RES        c = A . B + res[row, 0:N], fp32                    (input gradient + the residual branch's gradient)
 *   BL_X6_EPI_MASK_PACK  c = (y != 0) ? A . B x mask_scale : 0 with y = drop(relu(z)) given by its packed form y_packed [M, 3 N]
 *                        (only the hi plane is read: y != 0 <=> kept and z > 0), written only packed to c_packed;
 *                        colsum[0:N] (optional) += the column sums of c (unordered atomics: not in deterministic mode)
 *                        -- the gradient through dropout(relu(.)) of linear1, its bias gradient and the operand of its
 *                        weight / input gradient GEMMs in the epilogue of linear2's input gradient. */
#define BL_X6_EPI_ACT 0
#define BL_X6_EPI_ACT_PACK 1
#define BL_X6_EPI_RES 2
#define BL_X6_EPI_MASK_PACK 3
typedef struct {
  int32_t form;
  const float* bias; /* ACT forms: [N] or NULL */
  int32_t act;
  bl_dropout_t drop;
  const float* res; /* RES */
  int32_t ld_res;
  const uint16_t* y_packed; /* MASK_PACK */
  float mask_scale;
  float* colsum;
  uint16_t* c_packed; /* ACT_PACK, MASK_PACK */
} bl_x6_epi_t;
int bl_gemm_rows_x6_epi2(const bl_rows_packed_t* a, const uint16_t* bp, int32_t M, int32_t N, int32_t K, const bl_x6_epi_t* epi,
                         float* c, int32_t ldc, void* stream);
/* bf16x6 form of bl_gemm_wgrad_routed (below): `a` packed rows, g_node_packed = bl_pack_bf16x3 of the
 * node gradient [*, N]; the message-major operands are transposed on the fly by gfx950's transposing
 * LDS read (ds_read_b64_tr_b16).  N and the source widths must be multiples of 32. */
int bl_gemm_wgrad_routed_x6(const bl_rows_packed_t* a, const uint16_t* g_node_packed, const int32_t* g_idx,
                            const uint32_t* win_bits, int32_t ld_bits, const int32_t* group_ptr, const int32_t* group_w,
                            int32_t G, int32_t M, int32_t N, int32_t K, float* gw, int64_t gw_group_stride, int32_t ld_gw,
                            void* stream);

/* Tile of the two bf16x6 weight-gradient GEMMs above.  256 (default): a 256 x 128 output tile per 8-wave workgroup with
 * double-buffered stages wherever K is a multiple of 256 and every source width a multiple of 128 -- the routed operand is
 * staged once per message instead of once per 128-feature half; 128: the 128 x 128 tile everywhere.  Same results either
 * way (same products, same per-tile accumulation order); a measurement switch (tools/gemm_bench.py).  Returns the
 * previous setting. */
int32_t bl_set_wgrad_tile(int32_t rows);

/* Most rows one workgroup of the two bf16x6 weight-gradient GEMMs above reduces before it adds its output tile into gw
 * (fp32 atomics: one flush = one tile of them).  The chunk is the smallest one that fills an integer number of rounds of
 * resident workgroups and stays <= the cap; rows < 256 are ignored.  Returns the previous cap. */
int32_t bl_set_wgrad_kchunk_cap(int32_t rows);

/* bf16x6 form of bl_gemm_wgrad (below), no routing: gw[g] += rows(a)^T . g_packed[g_idx[r] or r, 0:N] -- the weight
 * gradient of a plain Linear (the dense node update) from packed operands */
int bl_gemm_wgrad_x6(const bl_rows_packed_t* a, const uint16_t* g_packed, const int32_t* g_idx, const int32_t* group_ptr,
                     const int32_t* group_w, int32_t G, int32_t M, int32_t N, int32_t K, float* gw, int64_t gw_group_stride,
                     int32_t ld_gw, void* stream);

/* Weight-gradient GEMM (reduction over rows, split across row chunks, fp32 atomics):
 *   gw[group_w[g]][0:K, 0:N] += rows(a)[rows of g, 0:K]^T . g_c[rows of g, 0:N]
 * gw must be zeroed (or hold the running gradient) by the caller.  autograd equivalent:
 * the weight gradient of every Linear named above. */
int bl_gemm_wgrad(const bl_rows_t* a, const float* g_c, int32_t ld_g, const int32_t* group_ptr,
                  const int32_t* group_w, int32_t G, int32_t M, int32_t N, int32_t K, float* gw,
                  int64_t gw_group_stride, int32_t ld_gw, void* stream);
/* the same with the routed right operand  g_c[r, n] = (winner[g_idx[r], n] == r) ? g_node[g_idx[r], n] : 0 */
int bl_gemm_wgrad_routed(const bl_rows_t* a, const float* g_node, int32_t ld_g, const int32_t* g_idx,
                         const int32_t* winner, int32_t ld_winner, const int32_t* group_ptr, const int32_t* group_w,
                         int32_t G, int32_t M, int32_t N, int32_t K, float* gw, int64_t gw_group_stride, int32_t ld_gw,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * M2(+M3a)  segmented max with argmax, optional fused LayerNorm.
 *   out[v, d] = max_{i in seg_ptr[v]..seg_ptr[v+1]} act(x[item(i), d]),  item(i) = seg_items ? seg_items[i] : i
 *   (act = BL_ACT_GELU_AGG: out[v, d] = gelu(max_i x[item(i), d]) -- the winner is the largest RAW item)
 *   arg[v, d] = that item (ties: first in segment order), -1 and out = 0 for an empty segment;
 *   if ln_g != NULL also  ln_out[v,:] = LayerNorm(out[v,:]; ln_g, ln_b, eps), mean[v], rstd[v];
 *   if dact != NULL also  dact[v, d] = act'(x[arg[v, d], d]) (0 for an empty segment): with it the
 *   backward pass needs only [nseg, D] arrays, never the [items, D] pre-activations again;
 *   if winbits != NULL also  winbits[item, w] bit b = (arg[seg(item), 32 w + b] == item), ceil(D/32)
 *   words per item: the routing table in the form the bf16x6 routed GEMMs read;
 *   seg_order (optional): a permutation of 0..nseg-1 = the order segments are processed in (the collator puts
 *   high-degree nodes first: one wave works through a 512-item hub for about as long as the whole launch takes).
 * Replaces torch_scatter.scatter_max at ptgnn's "max" aggregation (gnnlayerdefs.py:11,21) and at
 * buglab/models/layers/localizationmodule.py:56-58, plus ptgnn's nn.LayerNorm. */
int bl_segment_max_fwd(const float* x, int32_t ldx, const int32_t* seg_ptr, const int32_t* seg_items, int32_t nseg,
                       int32_t D, int32_t act, float* out, int32_t* arg, const float* ln_g, const float* ln_b,
                       float eps, float* ln_out, float* mean, float* rstd, float* dact, uint32_t* winbits,
                       const int32_t* seg_order, void* stream);

/* backward of the segmented max, gather form (deterministic, no atomics):
 *   g_x[i, d] = (arg[seg_of[i], d] == i) ? g_out[seg_of[i], d] * act'(x[i, d]) : 0      (g_x may alias x) */
int bl_segment_max_bwd(const float* g_out, const int32_t* arg, const float* x, int32_t ldx, const int32_t* seg_of,
                       int32_t nitems, int32_t D, int32_t act, float* g_x, void* stream);

/* LayerNorm backward (rows of even D <= 512):  g_x (times post_scale elementwise if not NULL -- used to
 * fold the message activation's derivative `dact` in), and g_gamma/g_beta ACCUMULATED with fp32 atomics.
 * g_x_packed (if not NULL; D % 8 == 0) also receives the result in bl_pack_bf16x3's packed form, the
 * operand of the bf16x6 routed GEMMs; g_x may then be NULL. */
int bl_layernorm_bwd(const float* g_y, const float* x, const float* mean, const float* rstd, const float* gamma,
                     int32_t nrows, int32_t D, float* g_x, float* g_gamma, float* g_beta, const float* post_scale,
                     uint16_t* g_x_packed, void* stream);

/* LayerNorm backward where the normalised tensor was  x + dropout(branch)  (the two sublayers of the relational transformer
 * block, reference relational_transformer.py:104-124): g_x (fp32, unmasked) is the residual's gradient, and the BRANCH's
 * gradient -- g_x through the dropout mask of the branch, element index row * D + d -- leaves in bl_pack_bf16x3's form
 * (g_branch_packed [nrows, 3 D], the operand of the branch Linear's two gradient GEMMs) with its column sums added to
 * g_bias (the Linear's bias gradient; may be NULL).  D a multiple of 8 up to 512.  branch_drop.p == 0: no mask. */
int bl_layernorm_bwd_branch(const float* g_y, const float* x, const float* mean, const float* rstd, const float* gamma,
                            int32_t nrows, int32_t D, float* g_x, float* g_gamma, float* g_beta, bl_dropout_t branch_drop,
                            float* g_bias, uint16_t* g_branch_packed, void* stream);

/* activation(+dropout) backward from the OUTPUT y of y = drop(act(z + bias)); g_bias (if not NULL)
 * accumulates column sums of g_z with fp32 atomics.  g_z may alias g_y.  (GELU is not supported
 * here: it needs the pre-activation, see bl_segment_max_bwd.) */
int bl_act_bwd(const float* g_y, const float* y, int32_t nrows, int32_t N, int32_t ld, int32_t act, bl_dropout_t drop,
               float* g_z, float* g_bias, void* stream);
/* the same with the result also (or only: g_z may then be NULL) in bl_pack_bf16x3's packed form [nrows, 3 N] (N % 8 == 0):
 * the operand of the bf16x6 input- and weight-gradient GEMMs of a Linear */
int bl_act_bwd_packed(const float* g_y, const float* y, int32_t nrows, int32_t N, int32_t ld, int32_t act, bl_dropout_t drop,
                      float* g_z, float* g_bias, uint16_t* g_z_packed, void* stream);

/* M1 backward, last step: gradient w.r.t. node states from the per-message input gradients
 *   g_h[n, 0:Din] (+)= sum_{e in src CSR of n} g_a[e, 0:Din] + sum_{e in tgt CSR of n} g_a[e, Din:2Din]
 * (deterministic segmented sums; replaces autograd's index_add of the two h[...] gathers).
 * tgt_ptr == tgt_msgs == NULL: source half only (messages built from h[src] alone, `ggnn`). */
int bl_mp_scatter_grad(const float* g_a, int32_t ld_ga, const int32_t* src_ptr, const int32_t* src_msgs,
                       const int32_t* tgt_ptr, const int32_t* tgt_msgs, int32_t N, int32_t Din, int32_t accumulate,
                       float* g_h, int32_t ld_gh, const int32_t* node_order,
                       void* stream);

/* the same with the columns 0..split-1 written to g_h_lo and split..Din-1 to g_h_hi: the gradients of the two
 * inputs of a folded ConcatResidual ([stash ; current]), each in its own contiguous matrix */
int bl_mp_scatter_grad_split(const float* g_a, int32_t ld_ga, const int32_t* src_ptr, const int32_t* src_msgs,
                             const int32_t* tgt_ptr, const int32_t* tgt_msgs, int32_t N, int32_t Din, int32_t split,
                             float* g_h_lo, int32_t ld_lo, float* g_h_hi, int32_t ld_hi, const int32_t* node_order,
                             void* stream);

/* Routed input gradient of the message layer from the NON-ZEROS of the message gradient alone (vector units, exact
 * fp32): the max aggregation gives every (node, channel) gradient to ONE message, so only N * Dm of the E * Dm entries
 * the routed GEMM multiplies are non-zero.  gq [*, Dm] fp32 = d loss / d (winning pre-activation) per node, wt =
 * W transposed [T, Dm, 2 Din], win_bits / type_ptr as in bl_gemm_rows_x6 (messages type-major, target-sorted inside a type).
 *   bl_routed_dgrad_vec    writes the per-message rows g_a[e, :] = sum_{d won by e} gq[tgt(e), d] * W[type(e)][:, d]
 *                          (the result of the routed bl_gemm_rows_x6 up to fp32 summation order);
 *   bl_routed_dgrad_nodes  adds them straight into the node gradient: g_h[src(e), 0:Din] += g_a[e, 0:Din] per message,
 *                          g_h[tgt, 0:Din] += sum over a run of messages sharing the target of g_a[e, Din:2Din] (fp32 atomics;
 *                          the caller zeroes g_h).  Columns < split go to g_h_lo, the rest to g_h_hi (split == Din: one output).
 *                          = bl_routed_dgrad_vec + bl_mp_scatter_grad without the [E, 2 Din] round trip through memory.
 * Both replace the autograd of ptgnn's gather + per-type Linear + torch_scatter.scatter_max (call site
 * buglab/models/gnnlayerdefs.py:6-23).  bl_routed_dgrad_vec_ok: whether (Dm, K2 = 2 Din) is supported (W[t]^T must fit
 * one 128 KB LDS block: Dm in {64, 128}, K2 in {128, 256}, Dm * K2 <= 32768). */
int32_t bl_routed_dgrad_vec_ok(int32_t Dm, int32_t K2);
int bl_routed_dgrad_vec(const float* gq, int32_t ld_gq, const int32_t* msg_tgt, const uint32_t* win_bits, int32_t ld_bits,
                        const int32_t* type_ptr, int32_t T, const float* wt, int32_t E, int32_t Dm, int32_t K2, float* g_a,
                        int32_t ld_ga, void* stream);
int bl_routed_dgrad_nodes(const float* gq, int32_t ld_gq, const int32_t* msg_src, const int32_t* msg_tgt,
                          const uint32_t* win_bits, int32_t ld_bits, const int32_t* type_ptr, int32_t T, const float* wt,
                          int32_t E, int32_t Dm, int32_t Din, int32_t split, float* g_h_lo, int32_t ld_lo, float* g_h_hi,
                          int32_t ld_hi, void* stream);
/* bl_routed_dgrad_nodes with the SOURCE half kept per message: g_src[e, 0:Din] = g_a[e, 0:Din] as plain rows (g_src != NULL;
 * summed per node afterwards by bl_mp_scatter_grad over the source CSR with accumulate = 1), the target half by atomics as
 * above.  Halves the fp32 atomics, which bound the all-atomic form (one 4-byte atomic per L2 channel per clock). */
int bl_routed_dgrad_nodes_rows(const float* gq, int32_t ld_gq, const int32_t* msg_src, const int32_t* msg_tgt,
                               const uint32_t* win_bits, int32_t ld_bits, const int32_t* type_ptr, int32_t T, const float* wt,
                               int32_t E, int32_t Dm, int32_t Din, int32_t split, float* g_h_lo, int32_t ld_lo, float* g_h_hi,
                               int32_t ld_hi, float* g_src, int32_t ld_src, void* stream);

/* GRU cell of the gated (`ggnn`) node update -- the elementwise part of torch.nn.GRUCell (gate order
 * r | z | n) after gi = x W_i + b_i and gh = h W_h + b_h [N, 3D] were produced by bl_gemm_rows:
 *   h' = drop((1 - z) * tanh(gi_n + r * gh_n) + z * h).  Replaces ptgnn GatedMessagePassingLayer's
 * nn.GRUCell state update (reference call site buglab/models/gnnlayerdefs.py:42-68).
 * bwd writes g_gi, g_gh [N, 3D] and the direct part of g_h (z * g_out) [N, D]. */
int bl_gru_cell_fwd(const float* gi, const float* gh, const float* h, int32_t ld_h, int32_t N, int32_t D,
                    bl_dropout_t drop, float* out, void* stream);
int bl_gru_cell_bwd(const float* g_out, const float* gi, const float* gh, const float* h, int32_t ld_h, int32_t N,
                    int32_t D, bl_dropout_t drop, float* g_gi, float* g_gh, float* g_h, void* stream);

/* Time recurrence of one BIDIRECTIONAL GRU layer over a padded [B, L] minibatch (`seq-gru`): replaces torch.nn.GRU(bidirectional,
 * batch_first) applied to a PackedSequence (reference buglab/models/seqmodel.py:119-126, called at :385-392 between
 * pack_padded_sequence(lengths, enforce_sorted=False) and pad_packed_sequence): every sequence runs over its own length, the
 * reverse direction starts at its last real token, positions >= lens[b] come back as zeros.  Row b * L + t everywhere.
 *   gi   [B L, >= 6 Hh]  x W_ih + b_ih of both directions, columns [direction][r | z | n][Hh]  (one row GEMM outside)
 *   w_hh [2][Hh][3 Hh]   gh = h . w_hh[direction] + b_hh[direction]   (torch's weight_hh_l{k}{_reverse}, transposed); b_hh [2][3 Hh]
 *   out  [B L, >= 2 Hh]  forward direction in columns 0 .. Hh - 1, reverse in Hh .. 2 Hh - 1
 *   saved  bl_gru_scan_saved_elems floats kept for bl_gru_scan_bwd (NULL: forward only), gates [2][B L][r | z | n | gh_n] then h_prev [2][B L][Hh]
 * Hh in {32, 64, 128}.  One workgroup per (sequence, direction); exact fp32 arithmetic.
 * bwd: g_out [B L, >= 2 Hh] -> g_gi [B L, >= 6 Hh] (gradient of gi; zeros at padded rows) and g_gh [2][B L][3 Hh] (gradient of
 * the recurrent pre-activations: the caller forms g_w_hh[d] = h_prev[d]^T g_gh[d] (bl_gemm_wgrad on saved's h_prev block) and
 * g_b_hh[d] = column sums of g_gh[d]). */
int64_t bl_gru_scan_saved_elems(int32_t B, int32_t L, int32_t Hh);
int bl_gru_scan_fwd(const float* gi, int32_t ld_gi, const float* w_hh, const float* b_hh, const int32_t* lens, int32_t B, int32_t L,
                    int32_t Hh, float* out, int32_t ld_out, float* saved, void* stream);
int bl_gru_scan_bwd(const float* g_out, int32_t ld_g, const float* w_hh, const float* saved, const int32_t* lens, int32_t B, int32_t L,
                    int32_t Hh, float* g_gi, int32_t ld_ggi, float* g_gh, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ONE MlpMessagePassingLayer per call (SURVEY.md section 8b's minimum set).  Replaces ptgnn's
 * MlpMessagePassingLayer.forward and its autograd; kwargs of the reference call site
 * buglab/models/gnnlayerdefs.py:6-23 map to Din = input_state_dimension, Dm = message_dimension,
 * Dout = output_state_dimension, T = num_edge_types, aggregation "max", drop.p = dropout_rate.
 *   msg_act BL_ACT_GELU_AGG (ptgnn's order as recollected, the product's default):
 *     m_e = [h_src ; h_tgt] . W[type(e)];  a_v = gelu(max_{e -> v} m_e) (0 if none);
 *   msg_act BL_ACT_GELU: m_e = gelu([h_src ; h_tgt] . W[type(e)]);  a_v = max_{e -> v} m_e (0 if none);  BL_ACT_NONE: no gelu;
 *   h'_v = Dropout(tanh(LayerNorm(a_v; ln_g, ln_b, ln_eps) . Wd + bd))
 * Widths: Din, Dm multiples of 32 (the message GEMMs run as bf16x6), Dm <= 512.  Messages are type-major
 * (type_ptr [T+1]) and target-sorted inside a type; tgt_ptr/tgt_msgs and src_ptr/src_msgs are the CSRs node ->
 * incoming / outgoing message ids; node_order (optional) = processing order of the per-node kernels. */
typedef struct {
  int32_t N, E, T, Din, Dm, Dout;
  const int32_t *msg_src, *msg_tgt, *type_ptr, *tgt_ptr, *tgt_msgs, *src_ptr, *src_msgs, *node_order;
  const float* W;                 /* [T, 2 Din, Dm] */
  const float *ln_g, *ln_b;       /* [Dm] */
  const float* Wd;                /* [Dm, Dout] */
  const float* bd;                /* [Dout] */
  int32_t msg_act;                /* BL_ACT_GELU_AGG, BL_ACT_GELU or BL_ACT_NONE */
  float ln_eps;
  bl_dropout_t drop;
  const float* Wt;                /* optional (backward only): W transposed, [T, Dm, 2 Din].  When given and
                                   * bl_routed_dgrad_vec_ok(Dm, 2 Din), the input gradient is computed from the non-zeros of
                                   * the routed message gradient (bl_routed_dgrad_nodes; bl_routed_dgrad_vec + bl_mp_scatter_grad
                                   * in the deterministic mode) instead of the routed matrix-core GEMM; NULL: matrix cores */
  const uint16_t* Wd_packed;      /* optional: bl_pack_weights_x6(Wd, 1, Dm, Dout, w_is_kn = 1).  When given (and Dm, Dout are
                                   * multiples of 32) the dense node update runs as bf16x6 GEMMs: forward, input gradient and
                                   * weight gradient; the same layer's backward call must then also carry ... */
  const uint16_t* Wd_packed_bwd;  /* ... bl_pack_weights_x6(Wd, 1, Dout, Dm, w_is_kn = 0), the form of g_ln = g_z . Wd^T.
                                   * NULL / NULL: exact-fp32 MFMA GEMMs (bl_gemm_rows / bl_gemm_wgrad) */
  int32_t num_hub_slots;          /* how many leading entries of node_order may be hubs (nodes with very long target segments get
                                   * a whole workgroup in the segmented max); < 0: unknown -- the first 4096 are looked at;
                                   * 0: no hubs -- node_order is then not read at all (the collators list the hubs first and every
                                   * other node in natural order, i.e. the identity; any order gives the same results) */
  int32_t aggregation;            /* BL_AGG_MAX (0: the reference's recipe, gnnlayerdefs.py:11,21), BL_AGG_SUM or BL_AGG_MEAN -- ptgnn's other
                                   * message_aggregation_function values: a_v = act(sum_{e -> v} m_e (/ in-degree)), 0 for a node without
                                   * messages; msg_act must then be BL_ACT_GELU_AGG or BL_ACT_NONE; no winner table (winner_out untouched), every
                                   * message receives its target's gradient in backward (unrouted GEMMs on the matrix cores) */
} bl_mp_layer_t;
#define BL_AGG_MAX 0
#define BL_AGG_SUM 1
#define BL_AGG_MEAN 2

/* buffer sizes (bytes): `saved` is written by forward and read by backward; the workspace is scratch of one call.
 * backward: 0 = forward call, 1 = backward call, 2 = backward call that will take the bl_routed_dgrad_nodes_rows path (Wt
 * given, shape supported, deterministic mode off): [E, Din] source-half rows instead of the [E, 2 Din] per-message gradient,
 * 3 = forward-only call (bl_mp_layer_fwd with saved == NULL: the workspace then also holds the packed input and the
 * LayerNorm output) */
int64_t bl_mp_layer_saved_bytes(int32_t N, int32_t E, int32_t Din, int32_t Dm, int32_t msg_act);
int64_t bl_mp_layer_workspace_bytes(int32_t N, int32_t E, int32_t Din, int32_t Dm, int32_t Dout, int32_t backward);
/* uint16 elements of the packed weights a layer call takes: bl_pack_weights_x6(W, T, 2 Din, Dm, w_is_kn = 1) for
 * forward, bl_pack_weights_x6(W, T, Dm, 2 Din, w_is_kn = 0) for backward (pack once per optimiser step) -- or, where
 * bl_mp_layer_weight_image(Din, Dm, for_backward) returns 1 (the layer's GEMM of that direction has a multiple of 256
 * output columns: the wide row GEMM runs it), bl_pack_weights_x6w with the same arguments.  The answer depends on the shape
 * and on bl_set_rows_tile only: choose the tile before packing. */
int32_t bl_mp_layer_weight_image(int32_t Din, int32_t Dm, int32_t for_backward);
int64_t bl_mp_layer_packed_weight_elems(int32_t T, int32_t Din, int32_t Dm, int32_t for_backward);

/* forward.  The layer input is h_lo [N, width_lo] alone (h_hi NULL, width_lo == Din) or the virtual concatenation
 * [h_lo ; h_hi] of a ConcatResidual layer (gnnlayerdefs.py:24-38), never materialised.  winner_out (optional,
 * int32 [N, Dm]): the arg-max message per (node, channel), -1 = none.  saved == NULL: forward-only call (the reference's
 * predict / evaluate path, buglab/models/gnn.py:606-645) -- nothing is stored for a backward pass (no routing bitmask,
 * activation derivative, aggregate, LayerNorm statistics), ws is sized by bl_mp_layer_workspace_bytes(..., 3). */
int bl_mp_layer_fwd(const bl_mp_layer_t* L, const float* h_lo, int32_t ld_lo, int32_t width_lo, const float* h_hi,
                    int32_t ld_hi, const uint16_t* w_packed, float* h_out, int32_t* winner_out, void* saved, void* ws,
                    void* stream);
/* backward.  g_h_lo / g_h_hi are written; g_W, g_ln_g, g_ln_b, g_Wd, g_bd are ACCUMULATED into (fp32 atomics):
 * hand in zeroed buffers or the running gradients.  side_stream (optional): the two weight-gradient GEMMs run
 * there next to the input-gradient chain; with join_side == 0 they are left running (the caller joins the
 * streams before it reads the weight gradients, and keeps `saved` / `ws` alive until then). */
int bl_mp_layer_bwd(const bl_mp_layer_t* L, const float* h_out, const float* g_out, const uint16_t* w_packed_bwd,
                    const void* saved, void* ws, float* g_h_lo, int32_t ld_lo, int32_t width_lo, float* g_h_hi,
                    int32_t ld_hi, float* g_W, float* g_ln_g, float* g_ln_b, float* g_Wd, float* g_bd, void* stream,
                    void* side_stream, int32_t join_side);

/* Backward of the layer's node update  h_out = drop(tanh(LayerNorm(agg) . Wd + bd))  from g_out, in one kernel per 64-node
 * tile (ptgnn MlpMessagePassingLayer's LayerNorm -> Linear -> tanh -> Dropout tail; call site
 * buglab/models/gnnlayerdefs.py:6-23):
 *   g_z = g_out . dropout mask . (1 - tanh^2)      -> g_z_packed [nrows, 3 Dout] (bl_pack_bf16x3 form; the operand of the dense
 *                                                     weight-gradient GEMM), g_bias [Dout] += its column sums (may be NULL)
 *   g_ln = g_z . Wd^T                              (bf16x6; wd_packed_bwd = bl_pack_weights_x6(Wd, 1, Dm, Dout, w_is_kn = 0))
 *   gq = LayerNorm backward(g_ln; agg, mean, rstd, ln_g) x dact   -> gq [nrows, Dm] fp32 and / or gq_packed [nrows, 3 Dm]
 *   g_ln_g, g_ln_b [Dm] += the gamma / beta gradients
 * = bl_act_bwd_packed + bl_gemm_rows_x6_epi + bl_layernorm_bwd without the g_z / g_ln round trips.  dact (the message
 * activation's derivative at the winners) may be NULL (= 1).  Shapes: bl_node_update_bwd_ok(Dm, Dout) -- Dm 128 or 256,
 * Dout a multiple of 32 up to 256, deterministic mode off (the column sums are flushed by unordered atomics). */
int32_t bl_node_update_bwd_ok(int32_t Dm, int32_t Dout);
int bl_node_update_bwd(const float* g_out, const float* h_out, int32_t nrows, int32_t Dout, bl_dropout_t drop,
                       const uint16_t* wd_packed_bwd, const float* agg, const float* mean, const float* rstd, const float* ln_g,
                       const float* dact, int32_t Dm, uint16_t* g_z_packed, float* g_bias, float* gq, uint16_t* gq_packed,
                       float* g_ln_g, float* g_ln_b, void* stream);
/* A/B switch: 1 (default) = bl_mp_layer_bwd uses bl_node_update_bwd where it applies, 0 = the three kernels it replaces.
 * Returns the previous value. */
int32_t bl_set_fused_node_bwd(int32_t on);

/* optional per-kernel timing of the launches made inside bl_mp_layer_fwd / _bwd (HIP events on the stream each
 * kernel is launched on); read after a device synchronisation.  bench.py's roofline numbers come from here. */
int bl_prof_enable(int32_t on);
int bl_prof_reset(void);
int bl_prof_num_kinds(void);
const char* bl_prof_kind_name(int32_t kind);
int bl_prof_read(int32_t kind, double* ms, double* flop, int64_t* launches, int32_t* overlapped);
double bl_prof_read_bytes(int32_t kind); /* algorithmic bytes recorded with a memory-bound kind's launches (0: none) */

/* ---------------------------------------------------------------------------------------------
 * fp32-accurate GEMMs on the fp16 matrix cores ("f16x3", csrc/bl_gemm_h3.hip): the second operand split of the message-passing
 * GEMMs.  An fp32 operand x, multiplied by a power-of-two scale s of its tensor, is split into TWO fp16 planes hi + lo
 * (22+ significant bits where |x s| >= 2^-3; an absolute error of 2^-25 / s below; finite values saturate at 65504 / s); a
 * product is three fp16 MFMA terms hh + hl + lh with fp32 accumulation, times 1 / (s_a s_b): half the matrix-pipe work and
 * two thirds of the operand bytes of bf16x6 at fp32-class accuracy (error vs fp64 below a plain fp32 matmul's) -- for tensors
 * whose range is bounded or measured.  Same contracts as the bf16x6 entry points they mirror (ptgnn MlpMessagePassingLayer's
 * per-type Linear forward / backward, buglab/models/gnnlayerdefs.py:6-23).
 *   packed rows: two planes back to back per row, [hi x D | lo x D] halves (4 D bytes);
 *   bl_pack_f16x2: rows of x[:, 0:D] into columns col_off .. col_off + D of D_total-wide packed rows (a ConcatResidual pair is
 *     packed in two calls); effective scale = scale x (amax_dev ? 2^(14 - ceil(log2 *amax_dev)) : 1) -- amax_dev: device float
 *     holding max |x| of the tensor (gradient tensors, whose magnitude is not known in advance; bl_amax or a producing kernel);
 *   bl_amax: *amax_dev = max(*amax_dev, max |x[0:n]|) (zero it first);
 *   bl_pack_weights_h3 / bl_packed_weight_elems_h3: tiled weight image (16 KB block per group, 128-column tile, 32-k stage);
 *   bl_gemm_rows_h3: C[rows of g] = out_scale x rows(a) . B_g (+ routed left operand with win_bits); out_scale = 1 / (s_a s_b),
 *     divided by 2^(14 - ceil(log2 *a_amax_dev)) when a_amax_dev is given (left operand packed with a device amax);
 *   bl_gemm_wgrad_h3: gW_g += out_scale x rows(a)^T . G rows (g_idx gather, win_bits routing as bl_gemm_wgrad_routed_x6);
 *     g_amax_dev: the device amax G was packed with. */
#define BL_H3_ROW_SCALE 256.0f /* layer inputs: |h| <= 1.25 after tanh x dropout, embedding rows O(1); saturation at 255.9 */
#define BL_H3_W_SCALE 64.0f    /* weights: saturation at 1023 */
/* which split the message GEMMs of bl_mp_layer_fwd / _bwd use: 1 = f16x3 (default), 0 = bf16x6, 2 = f16x1 -- the f16x3 images
 * and kernels with the high-plane term only (fp16 operands, fp32 accumulation and results, gradient operand scaled by its
 * device-side amax): the reduced-precision mode behind the reference's `train.py --amp` (/root/reference/buglab/models/train.py:8,106),
 * never the default and never the benchmarked headline.  BL_MSG_GEMM=x6 / h3 / amp in the environment sets the initial value.
 * Returns the previous mode.  bl_mp_layer_weight_image returns 2 (f16x3 image) in modes 1 and 2. */
int32_t bl_set_msg_gemm_mode(int32_t mode);
int32_t bl_get_msg_gemm_mode(void);
/* packing threads that had to saturate a finite value at +-65504 since the last reset, on the current device (synchronises; -1 if
 * the counter is unavailable): 0 in a healthy run -- the trainer checks it once per epoch (runtime/trainer.py). */
int64_t bl_h3_saturation_events(int32_t reset);
int bl_pack_f16x2(const float* x, int32_t ld, int64_t R, int32_t D, int32_t D_total, int32_t col_off, float scale,
                  const float* amax_dev, uint16_t* out, void* stream);
int bl_amax(const float* x, int64_t n, float* amax_dev, void* stream);
int64_t bl_packed_weight_elems_h3(int32_t G, int32_t K, int32_t N);
int bl_pack_weights_h3(const float* w, int32_t G, int32_t K, int32_t N, int32_t w_is_kn, float scale, uint16_t* out, void* stream);
int bl_gemm_rows_h3(const bl_rows_packed_t* a, const uint32_t* win_bits, int32_t ld_bits, const uint16_t* bp, int64_t b_group_stride,
                    const int32_t* group_ptr, const int32_t* group_w, int32_t G, int32_t M, int32_t N, int32_t K, float out_scale,
                    const float* a_amax_dev, float* c, int32_t ldc, void* stream);
int bl_gemm_wgrad_h3(const bl_rows_packed_t* a, const uint16_t* g_packed, const int32_t* g_idx, const uint32_t* win_bits,
                     int32_t ld_bits, const int32_t* group_ptr, const int32_t* group_w, int32_t G, int32_t M, int32_t N, int32_t K,
                     float out_scale, const float* g_amax_dev, float* gw, int64_t gw_group_stride, int32_t ld_gw, void* stream);

/* Box calibration (measurement support for bench.py; nothing on the training path calls these): boxes
 * of the pool differ by several per cent in the shader clock they sustain at the package power limit, so the bench line
 * carries what THIS chip delivers on two fixed kernels next to the paper peaks.  The caller times the launch with HIP events.
 *   bl_calib_mfma_bf16: `workgroups` x 4 waves, each `iters` x 4 independent chains of v_mfma_f32_32x32x16_bf16 from registers;
 *     *flop = the launch's floating-point operations (dense bf16 MFMA: paper peak 2 500 TF/s).
 *   bl_calib_stream_copy: 16 B / lane grid-stride copy of nbytes (multiple of 16, aligned): 2 x nbytes of HBM traffic. */
int bl_calib_mfma_bf16(int32_t iters, int32_t workgroups, float* sink, double* flop, void* stream);
int bl_calib_stream_copy(const void* src, void* dst, int64_t nbytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Heads.
 * H7  scatter_log_softmax (buglab/models/utils.py:15-28) over CSR segments:
 *     y[i] = x[i] - max_seg - log(sum_seg exp(x - max_seg) + eps)
 * bwd: g_x[i] = g_y[i] - exp(y[i]) * sum_seg g_y */
int bl_segment_log_softmax_fwd(const float* x, const int32_t* seg_ptr, const int32_t* seg_items, int32_t nseg,
                               float eps, float* y, void* stream);
int bl_segment_log_softmax_bwd(const float* g_y, const float* y, const int32_t* seg_ptr, const int32_t* seg_items,
                               int32_t nseg, float* g_x, void* stream);

/* final score layer Linear(H -> 1): y[r] = x[r, :] . w + b   (mlp.py:17, localizationmodule.py:24, 60) */
int bl_rowdot_fwd(const float* x, int32_t ldx, const float* w, const float* b, int32_t R, int32_t H, float* y,
                  void* stream);
/* g_x[r,:] = g_y[r] * w ;  g_w += sum_r g_y[r] x[r,:] ;  g_b += sum_r g_y[r]   (atomics) */
int bl_rowdot_bwd(const float* g_y, const float* x, int32_t ldx, const float* w, int32_t R, int32_t H, float* g_x,
                  int32_t ld_gx, float* g_w, float* g_b, void* stream);

/* out[idx[r], 0:width] += src[r, col_off : col_off + width]  (fp32 atomics).  Backward of every
 * `reprs[node_idx]` gather in buglab/models/gnn.py:170-172, 261-289 and of the rewrite-embedding
 * lookup in fixermodules.py:36. */
int bl_scatter_add_rows(const float* src, int32_t ld_src, int32_t col_off, int32_t width, const int32_t* idx,
                        int32_t R, float* out, int32_t ld_out, void* stream);
/* out[r, 0:width] = x[idx[r], 0:width]: ONE gather of every node row the heads reference
 * (buglab/models/gnn.py:170-172, 261-289 gather `reprs[node_idx]` once per scorer); the scorers then work on
 * the compact [R, H] copy, so backward has one [N, H] scatter instead of one per gather. */
int bl_gather_rows(const float* x, int32_t ld_x, const int32_t* idx, int32_t R, int32_t width, float* out,
                   int32_t ld_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whole scoring heads per call (the kernels are the ones above; these entry points sequence them).
 * H3/H4/H5  score[r] = w2 . relu(concat_j(x_j[idx_j[r]]) . W1 + b1) + b2 -- the MLP(k H -> H -> 1) of
 * buglab/models/layers/mlp.py:6-20 behind TextRepairModule / SingleCandidateNodeSelectorModule /
 * CandidatePairSelectorModule (fixermodules.py:36-39, 71-73, 116-124).  `hidden` [R, H] is kept for backward.
 * bwd ACCUMULATES into g_W1 [K, H], g_b1, g_w2, g_b2 and into every non-NULL g_x[j] (the gradient matrix of source j,
 * rows addressed through a->idx[j]; several sources may name the same matrix). */
int bl_gather_concat_mlp_score_fwd(const bl_rows_t* a, const float* W1, const float* b1, const float* w2, const float* b2,
                                   int32_t R, int32_t H, float* hidden, float* score, void* stream);
int64_t bl_gather_concat_mlp_score_workspace_bytes(int32_t R, int32_t H, int32_t K);
int bl_gather_concat_mlp_score_bwd(const bl_rows_t* a, const float* W1, const float* w2, const float* hidden,
                                   const float* g_score, int32_t R, int32_t H, void* ws, float* g_W1, float* g_b1,
                                   float* g_w2, float* g_b2, float* const* g_x, const int32_t* ld_gx, void* stream);
/* H1  candidate localization scores (localizationmodule.py:54-60), before the NO_BUG logit and the log-softmax:
 *   s = x[cand] . Ws + bs;  pool[g] = max over graph g's candidates (contiguous rows cand_ptr[g]..cand_ptr[g+1]);
 *   score[c] = w . sigmoid([x[cand[c]] ; pool[cand_graph[c]]] . W1 + b1)
 * bwd ACCUMULATES into g_x (rows addressed through cand) and the parameter gradients. */
int64_t bl_localization_scores_saved_bytes(int32_t C, int32_t B, int32_t H);
int64_t bl_localization_scores_workspace_bytes(int32_t C, int32_t B, int32_t H, int32_t backward);
int bl_localization_scores_fwd(const float* x, int32_t ld_x, const int32_t* cand, const int32_t* cand_graph,
                               const int32_t* cand_ptr, int32_t C, int32_t B, int32_t H, const float* Ws, const float* bs,
                               const float* W1, const float* b1, const float* w, void* saved, void* ws, float* score,
                               void* stream);
int bl_localization_scores_bwd(const float* x, int32_t ld_x, const int32_t* cand, const int32_t* cand_graph,
                               const int32_t* cand_ptr, int32_t C, int32_t B, int32_t H, const float* Ws, const float* W1,
                               const float* w, const void* saved, void* ws, const float* g_score, float* g_x,
                               int32_t ld_gx, float* g_Ws, float* g_bs, float* g_W1, float* g_b1, float* g_w, void* stream);

/* ---------------------------------------------------------------------------------------------
 * H2 + H6 + H7 + H8 on the training path: everything between the scorers' logits and the scalar loss in one kernel per
 * direction.  Replaces LocalizationModule.forward's NO_BUG logit / log-softmax / torch.where pick / clamp at log 0.995 /
 * abstain term / (weighted) mean and accuracy counters (buglab/models/layers/localizationmodule.py:63-124), the joint
 * log-softmax of the three repair scorers' logits over the location groups and the arg-max flags
 * (buglab/models/gnn.py:295-311, utils.py:15-28), the fixers' -logprob[target] and counters (fixermodules.py:41-53, 86-98,
 * 134-147) and the loss assembly loc + w_buggy * (text + var + swap) / B (gnn.py:221-251).
 * Items of the localization log-softmax are the C candidate rows followed by one NO_BUG slot per graph (value 1.0). */
typedef struct {
  int32_t B, C, Rt, Rv, Rs, G;                 /* graphs, candidate rows, text / var / swap logits, repair location groups */
  const float* loc_scores;                     /* [C] candidate scores (bl_localization_scores_fwd) */
  const float* repair_logits;                  /* [Rt + Rv + Rs]: text | var | swap */
  const int32_t *loc_group_ptr, *loc_group_items;        /* CSR graph -> its items among 0 .. C + B - 1 */
  const int32_t* candidate_ptr;                /* [B + 1]: the candidate rows of graph b are candidate_ptr[b] .. candidate_ptr[b+1] */
  const uint8_t* has_bug;                      /* [B] */
  const int32_t* correct_candidate_idxs;       /* [B] row of the buggy location (read where has_bug) */
  const int32_t *repair_group_ptr, *repair_group_items;  /* CSR location group -> its logits */
  const int32_t* logit_group[3];               /* location group of every text / var / swap logit */
  const int32_t* target[3];                    /* indices (into their slice) of the correct text / var / swap rewrites */
  int32_t ntarget[3];
  float w_buggy, abstain_weight;
} bl_bug_loss_t;
/* stats[16]: B, graphs localized correctly, NO_BUG graphs, NO_BUG graphs predicted so, -sum of the location logprobs, then
 * (arg-max hits, count) for text / var / swap rewrites, loss, repair loss (times w_buggy), buggy graphs, 1, 0. */
int bl_bug_loss_fwd(const bl_bug_loss_t* d, float* loc_logprobs, float* repair_logprobs, float* group_max, float* loss,
                    float* stats, void* stream);
/* g_loss: device scalar; scratch: C + B + Rt + Rv + Rs floats; writes every entry of g_loc_scores [C] and g_repair_logits */
int bl_bug_loss_bwd(const bl_bug_loss_t* d, const float* loc_logprobs, const float* repair_logprobs, const float* g_loss,
                    float* scratch, float* g_loc_scores, float* g_repair_logits, void* stream);

/* ---------------------------------------------------------------------------------------------
 * `seq-great` / `seq-rat` relational-transformer block (reference buglab/models/layers/relational_transformer.py,
 * relational_multihead_attention.py, multihead_attention.py): the row-wise kernels around the MFMA GEMMs.
 * q (pre-scaled by dk^-0.5), k, v, the attention context and their gradients are [B, H, L, dk] (one [L, dk] matrix per
 * (sample, head) = one GEMM group); scores / probabilities are [B * H * L, L].  The minibatch's edges come as a CSR over
 * query rows (b * L + i): ekey = key position, ecode = 2 * edge_type + direction (0: the query is the edge's source). */
/* y = LayerNorm(x + r) (r may be NULL); z_out (optional) receives x + r; the backward is bl_layernorm_bwd.
 * nn.LayerNorm of relational_transformer.py:84-85 + the residual adds of :113,122 */
int bl_add_layernorm_fwd(const float* x, const float* r, const float* gamma, const float* beta, float eps, int32_t nrows,
                         int32_t D, float* z_out, float* y, float* mean, float* rstd, void* stream);
/* S[(b, h, i), key] += edge term, relational_multihead_attention.py:90-152 (index_put_ with accumulate=True):
 * mode 0: <bias[code][h, :], q[b, h, i, :]> (what seq-great runs, :135-152); mode 1: bias[code][h] * sum_d k[b, h, key, d] */
int bl_rel_attn_bias_fwd(const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L, int32_t H,
                         int32_t dk, int32_t mode, const float* qk, const float* bias_f, const float* bias_r, float* S,
                         void* stream);
/* gradients of the edge terms from dS: g_q += (mode 0, row-owned), g_k += (mode 1, atomics), g_bias_* += (atomics) */
int bl_rel_attn_bias_bwd(const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L, int32_t H,
                         int32_t dk, int32_t mode, int32_t T, const float* qk, const float* bias_f, const float* bias_r,
                         const float* dS, float* g_q, float* g_k, float* g_bias_f, float* g_bias_r, void* stream);
/* softmax over keys with the padding keys (key >= lens[row / rows_per_sample]) masked, in place (multihead_attention.py:65-71) */
int bl_masked_softmax_fwd(float* S, int32_t R, int32_t L, int32_t rows_per_sample, const int32_t* lens, void* stream);
/* ... and nn.Dropout on the probabilities in the same pass (multihead_attention.py:72): S <- P, Pd <- dropout(P) with mask
 * element row * L + k (= bl_dropout_inplace on a copy of P); drop.p == 0: Pd is not written */
int bl_masked_softmax_dropout_fwd(float* S, int32_t R, int32_t L, int32_t rows_per_sample, const int32_t* lens, bl_dropout_t drop,
                                  float* Pd, void* stream);
/* dP <- P * (dP - sum_k P dP) */
int bl_softmax_bwd(const float* P, float* dP, int32_t R, int32_t L, void* stream);
/* the same with dP arriving as the gradient of dropout(P): it passes the mask first (bl_dropout_inplace + bl_softmax_bwd) */
int bl_softmax_dropout_bwd(const float* P, float* dP, int32_t R, int32_t L, bl_dropout_t drop, void* stream);
/* The attention probabilities of `seq-great` in one kernel (multihead_attention.py:54-72 with the edge terms of
 * relational_multihead_attention.py:135-152, mode 0): P[(b, h, i), :] = softmax_keys(q[b, h, i, :] . k[b, h, :, :]^T + edge terms,
 * keys >= lens[b] masked), Pd = dropout(P) (mask element row * L + key; drop.p == 0: Pd not written).  q (pre-scaled), k:
 * [B, H, L, dk]; the edge CSR may be NULL (no entries); T edge types, bias_f / bias_r [T, H * dk].  Replaces the grouped
 * Q.K^T GEMM + bl_rel_attn_bias_fwd + bl_masked_softmax_dropout_fwd (three passes over the [B H L, L] scores).
 * bl_rel_attn_probs_ok: whether the shape is handled (L % 4 == 0, L <= 1024, dk in {16, 32, 64}, K^T of one head in 64 KB,
 * 2 T dk <= 1024). */
int32_t bl_rel_attn_probs_ok(int32_t L, int32_t dk, int32_t T);
int bl_rel_attn_probs_fwd(const float* q, const float* k, const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode,
                          int32_t B, int32_t L, int32_t H, int32_t dk, int32_t T, const float* bias_f, const float* bias_r,
                          const int32_t* lens, bl_dropout_t drop, float* P, float* Pd, void* stream);
/* ... and their backward: g_ctx [B, H, L, dk] = gradient of the context, v, q [B, H, L, dk], P from the forward call ->
 * dS [B H L, L] = d loss / d scores (the grouped GEMMs dQ = dS.K and dK = dS^T.Q read it);  with an edge CSR also
 * gq_edge [B, H, L, dk] = the edge terms' part of d loss / d q (every row is written: zeros where a row has no entries) and
 * g_bias_f / g_bias_r [T, H dk] += (atomics).  Replaces dO.V^T GEMM + bl_softmax_dropout_bwd + bl_rel_attn_bias_bwd. */
int bl_rel_attn_probs_bwd(const float* g_ctx, const float* v, const float* P, const float* q, const int32_t* row_ptr,
                          const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L, int32_t H, int32_t dk, int32_t T,
                          const float* bias_f, const float* bias_r, bl_dropout_t drop, float* dS, float* gq_edge, float* g_bias_f,
                          float* g_bias_r, void* stream);
/* The attention's four tall-and-skinny products for head dimension 32 (bl_attn_mm32_ok: dk == 32, L % 4 == 0, L <= 1164),
 * exact-fp32 matrix cores, the head's [L, 32] matrix whole in LDS and the [G L, L] operand streamed once:
 *   bl_attn_rows_times        out[(g, i), :] = (sum_k A[(g, i), k] M[g, k, :] (+ add[(g, i), :])) * scale   -- P.V, dS.K
 *   bl_attn_transposed_times  out[g, k, :]   = sum_i A[(g, i), k] Bm[g, i, :]                              -- P^T.dO, dS^T.Q
 * (the grouped bl_gemm_rows / bl_gemm_wgrad calls of multihead_attention.py:73-77 and their autograd). */
int32_t bl_attn_mm32_ok(int32_t L, int32_t dk);
int bl_attn_rows_times(const float* A, const float* M, int32_t G, int32_t L, int32_t dk, const float* add, float scale, float* out,
                       void* stream);
int bl_attn_transposed_times(const float* A, const float* Bm, int32_t G, int32_t L, int32_t dk, float* out, void* stream);
/* The same four kernels on HEAD VIEWS: a [B, H, L, dk] operand given by a base pointer and three strides (element (b, h, l, d) at
 * p[b sb + h sh + l sl + d]; 16-byte aligned base, strides multiples of 4), so that q / k / v are read straight from the QKV
 * projection's [B L, H 3 dk] output (per head [q | k | v], multihead_attention.py:46-50: p = qkv + which dk, sb = L 3 H dk,
 * sh = 3 dk, sl = 3 H dk), the context is written as [B L, H dk] and the gradients land in the layout the projection's
 * gradient GEMMs read -- no permuted copies.  q_scale / bm_scale multiply q where it is loaded (the reference's pre-scaled
 * queries, multihead_attention.py:54: same rounding as a scaled copy).  P, Pd, dS, gq_edge, add stay [B H L, *] contiguous. */
typedef struct {
  float* p;
  int64_t sb;
  int32_t sh, sl;
} bl_head_view_t;
int bl_rel_attn_probs_fwd_v(const bl_head_view_t* q, float q_scale, const bl_head_view_t* k, const int32_t* row_ptr, const int32_t* ekey,
                            const int32_t* ecode, int32_t B, int32_t L, int32_t H, int32_t dk, int32_t T, const float* bias_f,
                            const float* bias_r, const int32_t* lens, bl_dropout_t drop, float* P, float* Pd, void* stream);
int bl_rel_attn_probs_bwd_v(const bl_head_view_t* g_ctx, const bl_head_view_t* v, const float* P, const bl_head_view_t* q, float q_scale,
                            const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L, int32_t H, int32_t dk,
                            int32_t T, const float* bias_f, const float* bias_r, bl_dropout_t drop, float* dS, float* gq_edge,
                            float* g_bias_f, float* g_bias_r, void* stream);
/* a_drop (p > 0): A is read through the counter-hash dropout mask, element index (g L + i) L + k -- A = the probabilities P of
 * bl_rel_attn_probs_fwd_v called with Pd == NULL: nn.Dropout on the attention probabilities (multihead_attention.py:72) without
 * a stored copy of the dropped matrix */
/* out_packed (optional; out may then be NULL): the result in bl_pack_bf16x3's form as columns of a packed [B L, 3 W] matrix --
 * element (b, h, l, d), plane pl at p[((b L + l) 3 + pl) W + col0 + h hs + d] (16-byte aligned base, W a multiple of 8, col0 and hs
 * multiples of 4): the context as the output projection's operand (W = H dk, col0 = 0, hs = dk), the gradients of q / k / v as the
 * operand of the QKV projection's gradient GEMMs (W = 3 H dk, col0 = which dk, hs = 3 dk) */
typedef struct {
  uint16_t* p;
  int32_t W, col0, hs;
} bl_packed_head_view_t;
int bl_attn_rows_times_v(const float* A, const bl_head_view_t* M, int32_t B, int32_t H, int32_t L, int32_t dk, const float* add, float scale,
                         const bl_head_view_t* out, bl_dropout_t a_drop, const bl_packed_head_view_t* out_packed, void* stream);
int bl_attn_transposed_times_v(const float* A, const bl_head_view_t* Bm, float bm_scale, int32_t B, int32_t H, int32_t L, int32_t dk,
                               const bl_head_view_t* out, bl_dropout_t a_drop, const bl_packed_head_view_t* out_packed, void* stream);
/* `rat` edge value biases (relational_multihead_attention.py:155-178): ctx[b, h, i, :] += P[(b, h, i), key] * vb[code][h, :] */
int bl_rel_value_bias_fwd(const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L, int32_t H,
                          int32_t dk, const float* P, const float* vb_f, const float* vb_r, float* ctx, void* stream);
int bl_rel_value_bias_bwd(const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L, int32_t H,
                          int32_t dk, int32_t T, const float* P, const float* g_ctx, const float* vb_f, const float* vb_r,
                          float* dP, float* g_vb_f, float* g_vb_r, void* stream);
/* in-place counter-hash dropout (forward and backward are the same call) */
int bl_dropout_inplace(float* x, int64_t n, bl_dropout_t drop, void* stream);

/* bl_add_layernorm_fwd with a lane owning four consecutive channels (D a multiple of 4, 16-byte aligned pointers) and,
 * if y_packed != NULL, the result also in bl_pack_bf16x3's form [nrows, 3 D]: the next Linear's operand without a packing pass */
int bl_add_layernorm_fwd_packed(const float* x, const float* r, const float* gamma, const float* beta, float eps, int32_t nrows,
                                int32_t D, float* z_out, float* y, float* mean, float* rstd, uint16_t* y_packed, void* stream);

/* One relational transformer encoder layer per call (csrc/bl_great_layer.hip): reference
 * buglab/models/layers/relational_transformer.py:104-124 in the configuration `seq-great` runs -- normalisation "postnorm"
 * (both sublayers normalised by norm1, :123-124), rezero off, relu feed-forward, vector query edge bias
 * (relational_multihead_attention.py:135-152), no edge value biases; head dimension 32.
 *   x1  = norm1(x + dropout(out_proj(attention(qkv_proj(x)))))       out = norm1(x1 + dropout(linear2(dropout(relu(linear1(x1))))))
 * Weights come packed (bl_pack_weights_x6 of the [in, out] matrices: w_is_kn = 1 for forward, the *_bwd images w_is_kn = 0 for the
 * input-gradient GEMMs; NULL in a forward-only description).  Activations are [B L, *] row-major fp32; x_packed (optional) is
 * bl_pack_bf16x3 of x -- the previous layer's out_packed -- and must stay alive until the backward call; out_packed (optional)
 * receives the packed output.  Dropout counters as the op-by-op path (bl_gemm_rows_x6_epi / bl_rel_attn_probs_fwd): element
 * index row * width + column of the tensor the mask applies to.
 * bl_great_layer_ok: dk == 32, bl_rel_attn_probs_ok(L, dk, T), H dk and FF multiples of 32, H dk <= 512, fewer than 2^32 elements
 * in any dropout index space, deterministic mode off. */
typedef struct {
  int32_t B, L, H, dk, T, FF;
  const int32_t* row_ptr; /* edge CSR over query rows (bl_rel_attn_probs_fwd); all three NULL: no entries */
  const int32_t* ekey;
  const int32_t* ecode;
  const int32_t* lens;  /* [B] */
  const float* bias_f;  /* [T, H dk] */
  const float* bias_r;
  const float* norm_g;  /* [H dk] norm1 */
  const float* norm_b;
  const float* lin1_b;  /* [FF] */
  const float* lin2_b;  /* [H dk] */
  const uint16_t *qkv_w, *out_w, *lin1_w, *lin2_w;                 /* forward images */
  const uint16_t *qkv_w_bwd, *out_w_bwd, *lin1_w_bwd, *lin2_w_bwd; /* input-gradient images */
  float ln_eps;
  bl_dropout_t drop_attn;      /* on the attention probabilities */
  bl_dropout_t drop_att_out;   /* dropout1: on the attention branch */
  bl_dropout_t drop_ff_hidden; /* inside the feed-forward block */
  bl_dropout_t drop_ff_out;    /* dropout2: on the feed-forward branch */
} bl_great_layer_t;
/* gradient buffers, all ACCUMULATED into (fp32 atomics): zeroed buffers or the running gradients */
typedef struct {
  float *qkv_w, *out_w, *lin1_w, *lin1_b, *lin2_w, *lin2_b; /* [D, 3 D], [D, D], [D, FF], [FF], [FF, D], [D] */
  float *norm_g, *norm_b;                                   /* [D] */
  float *bias_f, *bias_r;                                   /* [T, D] (may be NULL without edge entries) */
} bl_great_layer_grads_t;
int32_t bl_great_layer_ok(int32_t B, int32_t L, int32_t H, int32_t dk, int32_t T, int32_t FF);
/* `saved` (forward -> backward; own_xp: the call packs x itself because no x_packed is handed in) and workspace bytes
 * (backward: 0 = forward, 1 = backward, 3 = forward-only call with saved == NULL) */
int64_t bl_great_layer_saved_bytes(int32_t B, int32_t L, int32_t H, int32_t dk, int32_t FF, int32_t own_xp);
int64_t bl_great_layer_workspace_bytes(int32_t B, int32_t L, int32_t H, int32_t dk, int32_t FF, int32_t backward);
int bl_great_layer_fwd(const bl_great_layer_t* d, const float* x, const uint16_t* x_packed, float* out, uint16_t* out_packed, void* saved,
                       void* ws, void* stream);
/* side_stream (optional): the four weight-gradient GEMMs run there next to the input-gradient chain; joined before the call returns */
int bl_great_layer_bwd(const bl_great_layer_t* d, const uint16_t* x_packed, const float* g_out, const void* saved, void* ws, float* g_x,
                       const bl_great_layer_grads_t* g, void* stream, void* side_stream);

/* ---------------------------------------------------------------------------------------------
 * T1  optimiser on flat fp32 buffers: global-norm clip (buglab/models/train.py:104, clip 0.5) fused
 * with Adam (buglab/models/utils.py:51-52).  bl_sqnorm writes sum(g^2) to *out (device scalar);
 * bl_adam_clip_step reads it on the device -- no host sync.  The sum is taken in one fixed order (per-block partials in
 * `scratch`, bl_sqnorm_scratch_bytes() bytes, then one block): equal gradients give a bit-equal norm on every replica. */
int64_t bl_sqnorm_scratch_bytes(void);
int bl_sqnorm(const float* g, int64_t n, float* out, float* scratch, void* stream);
int bl_adam_clip_step(float* param, const float* grad, float* m, float* v, int64_t n, const float* grad_sqnorm,
                      float grad_prescale, float clip_norm, float lr, float beta1, float beta2, float eps,
                      int32_t step, void* stream);
/* data-parallel form: `grad` is the all-reduced SUM over ranks of (graphs on the rank) x (the rank's gradient) and
 * *batch_total (device) the all-reduced number of graphs: the update uses grad / *batch_total; *batch_total <= 0 (no
 * rank had a minibatch) leaves parameters and moments untouched.  No value returns to the host. */
int bl_adam_clip_step_dp(float* param, const float* grad, float* m, float* v, int64_t n, const float* grad_sqnorm,
                         const float* batch_total, float clip_norm, float lr, float beta1, float beta2, float eps,
                         int32_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BUGLAB_HIP_H */
