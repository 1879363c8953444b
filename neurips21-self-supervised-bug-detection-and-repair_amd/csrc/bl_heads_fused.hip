// Scoring heads, one C call per head and direction (bl_gather_concat_mlp_score_*).
// The kernels are the library's own (gathered MFMA GEMM with bias+activation epilogue, row dot, segmented max,
// activation backward, weight-gradient GEMM, scatter-add); these entry points only sequence them, so that a
// training step is a few dozen calls into the library instead of a few hundred.
#include "bl_common.h"

namespace {
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
#define BLH_TRY(call_)            \
  do {                            \
    int rc_ = (call_);            \
    if (rc_ != BL_OK) return rc_; \
  } while (0)

bl_rows_t one_source(const float* x, int ld, int width, const int32_t* idx) {
  bl_rows_t r;
  r.x[0] = x; r.idx[0] = idx; r.ld[0] = ld; r.width[0] = width;
  r.x[1] = r.x[2] = nullptr; r.idx[1] = r.idx[2] = nullptr; r.ld[1] = r.ld[2] = 0; r.width[1] = r.width[2] = 0;
  r.nsrc = 1;
  return r;
}
const bl_dropout_t kNoDrop = {0.f, 0u, 0u};
}  // namespace

// ---- H3 / H4 / H5: score[r] = w2 . relu(concat_j(x_j[idx_j[r]]) . W1 + b1) + b2 --------------------------------
// reference: MLP(k H -> H -> 1) of buglab/models/layers/mlp.py:6-20 as used by fixermodules.py:36-39, 71-73, 116-124
extern "C" int bl_gather_concat_mlp_score_fwd(const bl_rows_t* a, const float* W1, const float* b1, const float* w2,
                                              const float* b2, int32_t R, int32_t H, float* hidden, float* score,
                                              void* stream) {
  if (R == 0) return BL_OK;
  BL_CHECK_ARG(a && W1 && b1 && w2 && hidden && score && H > 0 && H % 4 == 0, "bl_gather_concat_mlp_score_fwd: null pointer or bad H");
  int K = 0;
  for (int j = 0; j < a->nsrc; ++j) K += a->width[j];
  BLH_TRY(bl_gemm_rows(a, W1, 0, H, 0, b1, nullptr, nullptr, 1, R, H, K, BL_ACT_RELU, kNoDrop, hidden, H, stream));
  BLH_TRY(bl_rowdot_fwd(hidden, H, w2, b2, R, H, score, stream));
  return BL_OK;
}

extern "C" int64_t bl_gather_concat_mlp_score_workspace_bytes(int32_t R, int32_t H, int32_t K) {
  return (int64_t)(al256((size_t)R * H * 4) + al256((size_t)R * K * 4));
}

// g_W1 / g_b1 / g_w2 / g_b2 and every non-NULL g_x[j] (the gradient matrix of source j, rows addressed through
// idx_j like the source itself) are ACCUMULATED into; several sources may share one g_x matrix.
extern "C" int bl_gather_concat_mlp_score_bwd(const bl_rows_t* a, const float* W1, const float* w2, const float* hidden,
                                              const float* g_score, int32_t R, int32_t H, void* ws, float* g_W1,
                                              float* g_b1, float* g_w2, float* g_b2, float* const* g_x,
                                              const int32_t* ld_gx, void* stream) {
  if (R == 0) return BL_OK;
  BL_CHECK_ARG(a && W1 && w2 && hidden && g_score && ws && g_W1 && g_b1 && g_w2 && g_x && ld_gx,
               "bl_gather_concat_mlp_score_bwd: null pointer");
  int K = 0;
  for (int j = 0; j < a->nsrc; ++j) K += a->width[j];
  float* g_hid = (float*)ws;
  float* g_a = (float*)((char*)ws + al256((size_t)R * H * 4));
  BLH_TRY(bl_rowdot_bwd(g_score, hidden, H, w2, R, H, g_hid, H, g_w2, g_b2, stream));
  BLH_TRY(bl_act_bwd(g_hid, hidden, R, H, H, BL_ACT_RELU, kNoDrop, g_hid, g_b1, stream));  // in place: g_z
  BLH_TRY(bl_gemm_wgrad(a, g_hid, H, nullptr, nullptr, 1, R, H, K, g_W1, 0, H, stream));
  bool any = false;
  for (int j = 0; j < a->nsrc; ++j) any = any || g_x[j] != nullptr;
  if (!any) return BL_OK;
  bl_rows_t gz = one_source(g_hid, H, H, nullptr);
  BLH_TRY(bl_gemm_rows(&gz, W1, 0, H, 1, nullptr, nullptr, nullptr, 1, R, K, H, BL_ACT_NONE, kNoDrop, g_a, K, stream));
  int off = 0;
  for (int j = 0; j < a->nsrc; ++j) {
    if (g_x[j]) {
      BL_CHECK_ARG(a->idx[j] != nullptr, "bl_gather_concat_mlp_score_bwd: source %d has a gradient target but no row index", j);
      BLH_TRY(bl_scatter_add_rows(g_a, K, off, a->width[j], a->idx[j], R, g_x[j], ld_gx[j], stream));
    }
    off += a->width[j];
  }
  return BL_OK;
}

// ---- H1: candidate localization scores (before the NO_BUG logit and the per-graph log-softmax) ------------------
//   s = x[cand] . Ws + bs ;  pool[g] = max over the candidates of graph g ;  l1 = sigmoid([x[cand] ; pool[g(c)]] . W1 + b1)
//   score[c] = l1[c] . w          reference buglab/models/layers/localizationmodule.py:54-60
// saved (forward -> backward): summary-max argmax [B, H] int32, pooled [B, H], l1 [C, H]
extern "C" int64_t bl_localization_scores_saved_bytes(int32_t C, int32_t B, int32_t H) {
  return (int64_t)(al256((size_t)B * H * 4) * 2 + al256((size_t)C * H * 4));
}
extern "C" int64_t bl_localization_scores_workspace_bytes(int32_t C, int32_t B, int32_t H, int32_t backward) {
  if (!backward) return (int64_t)al256((size_t)C * H * 4);                         // summary
  return (int64_t)(al256((size_t)C * H * 4) + al256((size_t)C * 2 * H * 4) + al256((size_t)B * H * 4) + al256((size_t)C * H * 4));
}

extern "C" int bl_localization_scores_fwd(const float* x, int32_t ld_x, const int32_t* cand, const int32_t* cand_graph,
                                          const int32_t* cand_ptr, int32_t C, int32_t B, int32_t H, const float* Ws,
                                          const float* bs, const float* W1, const float* b1, const float* w, void* saved,
                                          void* ws, float* score, void* stream) {
  if (C == 0) return BL_OK;
  BL_CHECK_ARG(x && cand && cand_graph && cand_ptr && Ws && bs && W1 && b1 && w && saved && ws && score && H % 4 == 0 && H <= 512,
               "bl_localization_scores_fwd: null pointer or unsupported H");
  int32_t* arg = (int32_t*)saved;
  float* pooled = (float*)((char*)saved + al256((size_t)B * H * 4));
  float* l1 = (float*)((char*)saved + 2 * al256((size_t)B * H * 4));
  float* summary = (float*)ws;
  bl_rows_t a = one_source(x, ld_x, H, cand);
  BLH_TRY(bl_gemm_rows(&a, Ws, 0, H, 0, bs, nullptr, nullptr, 1, C, H, H, BL_ACT_NONE, kNoDrop, summary, H, stream));
  BLH_TRY(bl_segment_max_fwd(summary, H, cand_ptr, nullptr, B, H, BL_ACT_NONE, pooled, arg, nullptr, nullptr, 0.f, nullptr, nullptr,
                             nullptr, nullptr, nullptr, nullptr, stream));
  bl_rows_t a2 = one_source(x, ld_x, H, cand);
  a2.x[1] = pooled; a2.idx[1] = cand_graph; a2.ld[1] = H; a2.width[1] = H; a2.nsrc = 2;
  BLH_TRY(bl_gemm_rows(&a2, W1, 0, H, 0, b1, nullptr, nullptr, 1, C, H, 2 * H, BL_ACT_SIGMOID, kNoDrop, l1, H, stream));
  BLH_TRY(bl_rowdot_fwd(l1, H, w, nullptr, C, H, score, stream));
  return BL_OK;
}

// g_x [*, H] (rows addressed through `cand`) and the parameter gradients are ACCUMULATED into.
extern "C" int bl_localization_scores_bwd(const float* x, int32_t ld_x, const int32_t* cand, const int32_t* cand_graph,
                                          const int32_t* cand_ptr, int32_t C, int32_t B, int32_t H, const float* Ws,
                                          const float* W1, const float* w, const void* saved, void* ws,
                                          const float* g_score, float* g_x, int32_t ld_gx, float* g_Ws, float* g_bs,
                                          float* g_W1, float* g_b1, float* g_w, void* stream) {
  if (C == 0) return BL_OK;
  BL_CHECK_ARG(x && cand && cand_graph && cand_ptr && Ws && W1 && w && saved && ws && g_score && g_x && g_Ws && g_bs && g_W1 && g_b1 && g_w,
               "bl_localization_scores_bwd: null pointer");
  const int32_t* arg = (const int32_t*)saved;
  const float* pooled = (const float*)((const char*)saved + al256((size_t)B * H * 4));
  const float* l1 = (const float*)((const char*)saved + 2 * al256((size_t)B * H * 4));
  char* p = (char*)ws;
  float* g_l1 = (float*)p;               p += al256((size_t)C * H * 4);   // then g_z1 in place
  float* g_cat = (float*)p;              p += al256((size_t)C * 2 * H * 4);
  float* g_pool = (float*)p;             p += al256((size_t)B * H * 4);
  float* g_sum = (float*)p;
  hipStream_t st = (hipStream_t)stream;
  BLH_TRY(bl_rowdot_bwd(g_score, l1, H, w, C, H, g_l1, H, g_w, nullptr, stream));
  BLH_TRY(bl_act_bwd(g_l1, l1, C, H, H, BL_ACT_SIGMOID, kNoDrop, g_l1, g_b1, stream));
  bl_rows_t a2 = one_source(x, ld_x, H, cand);
  a2.x[1] = pooled; a2.idx[1] = cand_graph; a2.ld[1] = H; a2.width[1] = H; a2.nsrc = 2;
  BLH_TRY(bl_gemm_wgrad(&a2, g_l1, H, nullptr, nullptr, 1, C, H, 2 * H, g_W1, 0, H, stream));
  bl_rows_t gz = one_source(g_l1, H, H, nullptr);
  BLH_TRY(bl_gemm_rows(&gz, W1, 0, H, 1, nullptr, nullptr, nullptr, 1, C, 2 * H, H, BL_ACT_NONE, kNoDrop, g_cat, 2 * H, stream));
  BLH_TRY(bl_scatter_add_rows(g_cat, 2 * H, 0, H, cand, C, g_x, ld_gx, stream));           // d / d x[cand], first half
  if (hipMemsetAsync(g_pool, 0, (size_t)B * H * 4, st) != hipSuccess) { bl_set_error("bl_localization_scores_bwd: memset failed"); return BL_EINVAL; }
  BLH_TRY(bl_scatter_add_rows(g_cat, 2 * H, H, H, cand_graph, C, g_pool, H, stream));      // d / d pooled
  // through the per-graph max: candidates are contiguous per graph (seg_of = cand_graph)
  BLH_TRY(bl_segment_max_bwd(g_pool, arg, g_sum, H, cand_graph, C, H, BL_ACT_NONE, g_sum, stream));
  // summary = x[cand] . Ws + bs
  BLH_TRY(bl_act_bwd(g_sum, g_sum, C, H, H, BL_ACT_NONE, kNoDrop, g_sum, g_bs, stream));   // column sums -> g_bs
  bl_rows_t a = one_source(x, ld_x, H, cand);
  BLH_TRY(bl_gemm_wgrad(&a, g_sum, H, nullptr, nullptr, 1, C, H, H, g_Ws, 0, H, stream));
  bl_rows_t gs = one_source(g_sum, H, H, nullptr);
  BLH_TRY(bl_gemm_rows(&gs, Ws, 0, H, 1, nullptr, nullptr, nullptr, 1, C, H, H, BL_ACT_NONE, kNoDrop, g_cat, H, stream));
  BLH_TRY(bl_scatter_add_rows(g_cat, H, 0, H, cand, C, g_x, ld_gx, stream));
  return BL_OK;
}
