// One relational transformer encoder layer of `seq-great` per C call: bl_great_layer_fwd / bl_great_layer_bwd
// (reference buglab/models/layers/relational_transformer.py:104-124 "postnorm", rezero off; attention =
// relational_multihead_attention.py:72-152 with the vector query bias, multihead_attention.py:46-80).
//
// What the call removes against the op-by-op path (hip_ops.gather_linear / rel_attention / add_layernorm, one C call and one or
// more launches per arrow of the reference's graph):
//   * no permuted copies of q / k / v / context or their gradients: the attention kernels address the QKV projection's own
//     [B L, H 3 dk] output (and the [B L, H dk] context) through head views (bl_head_view_t); the query scale dk^-0.5 is applied
//     where q is loaded;
//   * the dropped attention probabilities are never stored: P.V and P^T.dO read P through the counter-hash mask;
//   * no packing pass in front of a Linear whose input a kernel of the layer just produced: LayerNorm writes its result also in
//     bl_pack_bf16x3's form, linear1's epilogue writes relu + dropout ONLY packed (the fp32 hidden activations never exist);
//   * no elementwise backward kernels: the gradient through dropout(Linear(.)) of a residual branch leaves the LayerNorm
//     backward already masked, packed and column-summed (bl_layernorm_bwd_branch); the gradient through dropout(relu(.)) of
//     linear1 is the epilogue of linear2's input-gradient GEMM (BL_X6_EPI_MASK_PACK); the two residual sums are the epilogues
//     of linear1's and the QKV projection's input-gradient GEMMs (BL_X6_EPI_RES).
//   * the attention products write what a Linear reads next in packed form themselves: the context (P.V) and the gradients of
//     q / k / v (dS.K, dS^T.Q, P^T.dO) land as columns of the packed operands (bl_packed_head_view_t);
//   forward   8 launches: QKV GEMM, probabilities, P.V (-> packed context), output GEMM(+dropout), add+LayerNorm, linear1
//             (+bias, relu, dropout -> packed), linear2 (+bias, dropout), add+LayerNorm
//   backward  14: LN backward, 2 GEMMs each for linear2 / linear1, LN backward, 2 GEMMs for the output projection,
//             P^T.dO, probabilities' backward, dS.K, dS^T.Q (-> packed), 2 GEMMs for the QKV projection; the four weight-gradient
//             GEMMs on the side stream when one is given (each is one round of 128 x 128 tiles at these shapes -- a third of
//             what a CU can hold -- and depends only on its Linear's packed output gradient), joined before the call returns
// The caller owns every buffer (`saved` lives from forward to backward, `ws` during the call); nothing is allocated here.
#include "bl_common.h"

namespace {
inline size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

struct Shape {
  int B, L, H, dk, FF, T;
  size_t R, D, G;
};
inline Shape shape_of(int B, int L, int H, int dk, int FF, int T) {
  Shape s = {B, L, H, dk, FF, T, (size_t)B * L, (size_t)H * dk, (size_t)B * H};
  return s;
}

struct Saved {  // forward -> backward
  uint16_t* xp;      // [R, 3 D]   the layer input, packed (only when the caller did not hand one in)
  float* qkv;        // [R, 3 D]   per head [q | k | v]
  float* P;          // [G L, L]   attention probabilities BEFORE dropout (the mask is applied where P is read)
  uint16_t* ctx_p;   // [R, 3 D]   packed attention context
  float* z1;         // [R, D]     x + attention branch
  float *mean1, *rstd1;
  uint16_t* x1p;     // [R, 3 D]   packed LayerNorm output = linear1's input
  uint16_t* hid_p;   // [R, 3 FF]  packed dropout(relu(linear1))
  float* z2;         // [R, D]     x1 + feed-forward branch
  float *mean2, *rstd2;
  size_t bytes;
};
Saved carve_saved(void* base, const Shape& s, bool own_xp) {
  Saved v;
  char* p = static_cast<char*>(base);
  size_t o = 0;
#define TAKE(field_, type_, n_) v.field_ = reinterpret_cast<type_*>(p + o); o += al((size_t)(n_) * sizeof(type_));
  TAKE(xp, uint16_t, own_xp ? s.R * 3 * s.D : 0)
  TAKE(qkv, float, s.R * 3 * s.D)
  TAKE(P, float, s.G * s.L * s.L)
  TAKE(ctx_p, uint16_t, s.R * 3 * s.D)
  TAKE(z1, float, s.R * s.D)
  TAKE(mean1, float, s.R)
  TAKE(rstd1, float, s.R)
  TAKE(x1p, uint16_t, s.R * 3 * s.D)
  TAKE(hid_p, uint16_t, s.R * 3 * s.FF)
  TAKE(z2, float, s.R * s.D)
  TAKE(mean2, float, s.R)
  TAKE(rstd2, float, s.R)
  v.bytes = o;
  return v;
}

struct WsFwd {
  float *att, *x1, *ff;  // [R, D] each
  size_t bytes;
};
WsFwd carve_fwd(void* base, const Shape& s) {
  WsFwd v;
  char* p = static_cast<char*>(base);
  size_t o = 0;
  TAKE(att, float, s.R * s.D)
  TAKE(x1, float, s.R * s.D)
  TAKE(ff, float, s.R * s.D)
  v.bytes = o;
  return v;
}

struct WsBwd {
  float* g_z2;        // [R, D]
  uint16_t* g_ff_p;   // [R, 3 D]
  uint16_t* g_h_p;    // [R, 3 FF]
  float* g_x1;        // [R, D]
  float* g_z1;        // [R, D]
  uint16_t* g_att_p;  // [R, 3 D]
  float* g_ctx;       // [R, D]
  float* dS;          // [G L, L]
  float* gq_edge;     // [G L, dk]
  uint16_t* g_qkv_p;  // [R, 9 D]
  size_t bytes;
};
WsBwd carve_bwd(void* base, const Shape& s) {
  WsBwd v;
  char* p = static_cast<char*>(base);
  size_t o = 0;
  TAKE(g_z2, float, s.R * s.D)
  TAKE(g_ff_p, uint16_t, s.R * 3 * s.D)
  TAKE(g_h_p, uint16_t, s.R * 3 * s.FF)
  TAKE(g_x1, float, s.R * s.D)
  TAKE(g_z1, float, s.R * s.D)
  TAKE(g_att_p, uint16_t, s.R * 3 * s.D)
  TAKE(g_ctx, float, s.R * s.D)
  TAKE(dS, float, s.G * s.L * s.L)
  TAKE(gq_edge, float, s.R * s.D)
  TAKE(g_qkv_p, uint16_t, s.R * 9 * s.D)
  v.bytes = o;
  return v;
}
#undef TAKE

inline bl_rows_packed_t one_source(const uint16_t* xp, int width) {
  bl_rows_packed_t r = {};
  r.xp[0] = xp;
  r.idx[0] = nullptr;
  r.width[0] = width;
  r.nsrc = 1;
  return r;
}
inline bl_head_view_t view_of(float* base, int col0, const Shape& s, int row_width, int head_width) {
  bl_head_view_t v;
  v.p = base + col0;
  v.sb = (int64_t)s.L * row_width;
  v.sh = head_width;
  v.sl = row_width;
  return v;
}
inline bool has_drop(const bl_dropout_t& d) { return d.p > 0.f; }

#define GL_TRY(call_)            \
  do {                           \
    const int rc_ = (call_);     \
    if (rc_ != BL_OK) return rc_; \
  } while (0)

// fork / join events of the side stream (the four weight-gradient GEMMs of the backward call run there, next to the
// input-gradient chain), one set per device
constexpr int kMaxDevices = 16;
struct SideEvents { hipEvent_t fork[4] = {nullptr, nullptr, nullptr, nullptr}; hipEvent_t join = nullptr; };
SideEvents g_side_events[kMaxDevices];
SideEvents* ensure_events() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  SideEvents& e = g_side_events[dev];
  if (e.join) return &e;
  for (int i = 0; i < 4; ++i)
    if (hipEventCreateWithFlags(&e.fork[i], hipEventDisableTiming) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&e.join, hipEventDisableTiming) != hipSuccess) return nullptr;
  return &e;
}

int check_desc(const char* who, const bl_great_layer_t* d) {
  BL_CHECK_ARG(d, "%s: null layer description", who);
  BL_CHECK_ARG(bl_great_layer_ok(d->B, d->L, d->H, d->dk, d->T, d->FF), "%s: shape B=%d L=%d H=%d dk=%d T=%d FF=%d is outside bl_great_layer_ok "
               "(or the deterministic mode is on)", who, d->B, d->L, d->H, d->dk, d->T, d->FF);
  BL_CHECK_ARG((d->row_ptr == nullptr) == (d->ekey == nullptr) && (d->ekey == nullptr) == (d->ecode == nullptr), "%s: partial edge CSR", who);
  BL_CHECK_ARG(d->lens && d->bias_f && d->bias_r && d->norm_g && d->norm_b && d->lin1_b && d->lin2_b, "%s: null parameter", who);
  BL_CHECK_ARG(d->qkv_w && d->out_w && d->lin1_w && d->lin2_w, "%s: null packed weights", who);
  return BL_OK;
}
}  // namespace

extern "C" int32_t bl_great_layer_ok(int32_t B, int32_t L, int32_t H, int32_t dk, int32_t T, int32_t FF) {
  if (B <= 0 || L <= 0 || H <= 0 || FF <= 0) return 0;
  const long long D = (long long)H * dk, R = (long long)B * L;
  if (!bl_attn_mm32_ok(L, dk) || !bl_rel_attn_probs_ok(L, dk, T)) return 0;
  if (D % 32 != 0 || FF % 32 != 0 || D > 512) return 0;                       // bf16x6 source widths; LayerNorm backward rows
  if (R * FF >= (1ll << 32) || R * 3 * D >= (1ll << 32) || R * H * L >= (1ll << 32)) return 0;  // 32-bit dropout counters
  if (bl_get_deterministic()) return 0;  // the bias gradient of linear1 leaves a GEMM epilogue by unordered atomics
  return 1;
}

extern "C" int64_t bl_great_layer_saved_bytes(int32_t B, int32_t L, int32_t H, int32_t dk, int32_t FF, int32_t own_xp) {
  return (int64_t)carve_saved(nullptr, shape_of(B, L, H, dk, FF, 1), own_xp != 0).bytes;
}

extern "C" int64_t bl_great_layer_workspace_bytes(int32_t B, int32_t L, int32_t H, int32_t dk, int32_t FF, int32_t backward) {
  const Shape s = shape_of(B, L, H, dk, FF, 1);
  if (backward == 1) return (int64_t)carve_bwd(nullptr, s).bytes;
  const size_t f = carve_fwd(nullptr, s).bytes;
  return (int64_t)(backward == 3 ? f + carve_saved(nullptr, s, true).bytes : f);
}

extern "C" int bl_great_layer_fwd(const bl_great_layer_t* d, const float* x, const uint16_t* x_packed, float* out, uint16_t* out_packed,
                                  void* saved, void* ws, void* stream) {
  GL_TRY(check_desc("bl_great_layer_fwd", d));
  BL_CHECK_ARG(x && out && ws, "bl_great_layer_fwd: null pointer");
  const Shape s = shape_of(d->B, d->L, d->H, d->dk, d->FF, d->T);
  const int R = (int)s.R, D = (int)s.D, FF = s.FF, dk = s.dk;
  const WsFwd w = carve_fwd(ws, s);
  // forward-only call (saved == NULL): what a backward pass would read lives in the workspace instead
  const Saved sv = saved ? carve_saved(saved, s, x_packed == nullptr) : carve_saved(static_cast<char*>(ws) + w.bytes, s, true);
  const float scale = 1.0f / sqrtf((float)dk);

  const uint16_t* xp = x_packed;
  if (xp == nullptr) {
    BlProfScope ps(BL_PROF_PACK_ROWS, 0.0, stream);
    GL_TRY(bl_pack_bf16x3(x, D, R, D, sv.xp, stream));
    xp = sv.xp;
  }
  {  // QKV projection (multihead_attention.py:27-31, bias=False)
    BlProfScope ps(BL_PROF_LINEAR_FWD, 2.0 * R * D * 3.0 * D, stream);
    const bl_rows_packed_t a = one_source(xp, D);
    GL_TRY(bl_gemm_rows_x6(&a, nullptr, 0, d->qkv_w, 0, nullptr, nullptr, 1, R, 3 * D, D, sv.qkv, 3 * D, stream));
  }
  const bl_head_view_t q = view_of(sv.qkv, 0, s, 3 * D, 3 * dk), k = view_of(sv.qkv, dk, s, 3 * D, 3 * dk),
                       v = view_of(sv.qkv, 2 * dk, s, 3 * D, 3 * dk);
  {  // softmax(q k^T / sqrt(dk) + edge terms), dropout (multihead_attention.py:54-72, relational_multihead_attention.py:135-152)
    BlProfScope ps(BL_PROF_ATTN_PROBS_FWD, 0.0, stream, 4.0 * s.G * s.L * (s.L + 2.0 * dk));
    GL_TRY(bl_rel_attn_probs_fwd_v(&q, scale, &k, d->row_ptr, d->ekey, d->ecode, s.B, s.L, s.H, dk, s.T, d->bias_f, d->bias_r, d->lens,
                                   d->drop_attn, sv.P, nullptr, stream));
  }
  {  // context = dropout(P) . V, written packed as [B L, 3 H dk]: the output projection's operand
    BlProfScope ps(BL_PROF_ATTN_ROWS_TIMES, 2.0 * s.G * s.L * s.L * dk, stream, 4.0 * s.G * s.L * (s.L + 2.0 * dk));
    const bl_packed_head_view_t cp = {sv.ctx_p, D, 0, dk};
    GL_TRY(bl_attn_rows_times_v(sv.P, &v, s.B, s.H, s.L, dk, nullptr, 1.0f, nullptr, d->drop_attn, &cp, stream));
  }
  {  // output projection + dropout1 (multihead_attention.py:35, relational_transformer.py:108-110)
    BlProfScope ps(BL_PROF_LINEAR_FWD, 2.0 * R * D * (double)D, stream);
    const bl_rows_packed_t a = one_source(sv.ctx_p, D);
    GL_TRY(bl_gemm_rows_x6_epi(&a, d->out_w, 0, nullptr, nullptr, 1, R, D, D, nullptr, BL_ACT_NONE, d->drop_att_out, w.att, D, stream));
  }
  {  // x1 = norm1(x + attention branch)
    BlProfScope ps(BL_PROF_ADD_LAYERNORM, 0.0, stream, 4.0 * R * D * 4.0 + 6.0 * R * D);
    GL_TRY(bl_add_layernorm_fwd_packed(x, w.att, d->norm_g, d->norm_b, d->ln_eps, R, D, sv.z1, w.x1, sv.mean1, sv.rstd1, sv.x1p, stream));
  }
  {  // dropout(relu(linear1(x1))) -- only its packed form is ever stored (relational_transformer.py:117-118)
    BlProfScope ps(BL_PROF_LINEAR_FWD, 2.0 * R * D * (double)FF, stream);
    const bl_rows_packed_t a = one_source(sv.x1p, D);
    bl_x6_epi_t e = {};
    e.form = BL_X6_EPI_ACT_PACK;
    e.bias = d->lin1_b;
    e.act = BL_ACT_RELU;
    e.drop = d->drop_ff_hidden;
    e.c_packed = sv.hid_p;
    GL_TRY(bl_gemm_rows_x6_epi2(&a, d->lin1_w, R, FF, D, &e, nullptr, 0, stream));
  }
  {  // dropout2(linear2(.))
    BlProfScope ps(BL_PROF_LINEAR_FWD, 2.0 * R * D * (double)FF, stream);
    const bl_rows_packed_t a = one_source(sv.hid_p, FF);
    GL_TRY(bl_gemm_rows_x6_epi(&a, d->lin2_w, 0, nullptr, nullptr, 1, R, D, FF, d->lin2_b, BL_ACT_NONE, d->drop_ff_out, w.ff, D, stream));
  }
  {  // out = norm1(x1 + feed-forward branch)  (sic: norm1 again, relational_transformer.py:123-124)
    BlProfScope ps(BL_PROF_ADD_LAYERNORM, 0.0, stream, 4.0 * R * D * 4.0 + (out_packed ? 6.0 * R * D : 0.0));
    GL_TRY(bl_add_layernorm_fwd_packed(w.x1, w.ff, d->norm_g, d->norm_b, d->ln_eps, R, D, sv.z2, out, sv.mean2, sv.rstd2, out_packed, stream));
  }
  return BL_OK;
}

extern "C" int bl_great_layer_bwd(const bl_great_layer_t* d, const uint16_t* x_packed, const float* g_out, const void* saved, void* ws,
                                  float* g_x, const bl_great_layer_grads_t* g, void* stream, void* side_stream) {
  GL_TRY(check_desc("bl_great_layer_bwd", d));
  BL_CHECK_ARG(g_out && saved && ws && g_x && g, "bl_great_layer_bwd: null pointer");
  BL_CHECK_ARG(d->qkv_w_bwd && d->out_w_bwd && d->lin1_w_bwd && d->lin2_w_bwd, "bl_great_layer_bwd: null packed weights (backward images)");
  BL_CHECK_ARG(g->qkv_w && g->out_w && g->lin1_w && g->lin1_b && g->lin2_w && g->lin2_b && g->norm_g && g->norm_b,
               "bl_great_layer_bwd: null gradient buffer");
  BL_CHECK_ARG(d->row_ptr == nullptr || (g->bias_f && g->bias_r), "bl_great_layer_bwd: edge entries need the edge-bias gradient buffers");
  const Shape s = shape_of(d->B, d->L, d->H, d->dk, d->FF, d->T);
  const int R = (int)s.R, D = (int)s.D, FF = s.FF, dk = s.dk;
  const Saved sv = carve_saved(const_cast<void*>(saved), s, x_packed == nullptr);
  const WsBwd w = carve_bwd(ws, s);
  const uint16_t* xp = x_packed ? x_packed : sv.xp;
  const float scale = 1.0f / sqrtf((float)dk);
  hipStream_t st = (hipStream_t)stream, side = side_stream ? (hipStream_t)side_stream : st;
  const bool two = side != st;
  SideEvents* ev = two ? ensure_events() : nullptr;
  BL_CHECK_ARG(!two || ev, "bl_great_layer_bwd: cannot create the side stream's events");
  void* wst = (void*)side;  // where the weight-gradient GEMMs go
  // the side stream may start a weight gradient once the main chain has produced its second operand
#define GL_FORK(i_)                                   \
  if (two) {                                          \
    (void)hipEventRecord(ev->fork[i_], st);           \
    (void)hipStreamWaitEvent(side, ev->fork[i_], 0);  \
  }

  {  // out = norm1(z2), z2 = x1 + dropout2(linear2(h)): g_z2 (the residual's gradient) and, masked + packed, the branch's
    BlProfScope ps(BL_PROF_LAYERNORM_BWD_BRANCH, 0.0, stream, 4.0 * R * D * 3.0 + 6.0 * R * D, two);
    GL_TRY(bl_layernorm_bwd_branch(g_out, sv.z2, sv.mean2, sv.rstd2, d->norm_g, R, D, w.g_z2, g->norm_g, g->norm_b, d->drop_ff_out, g->lin2_b,
                                   w.g_ff_p, stream));
  }
  GL_FORK(0)
  {
    BlProfScope ps(BL_PROF_LINEAR_WGRAD, 2.0 * R * D * (double)FF, wst, 0.0, two);
    const bl_rows_packed_t a = one_source(sv.hid_p, FF);
    GL_TRY(bl_gemm_wgrad_x6(&a, w.g_ff_p, nullptr, nullptr, nullptr, 1, R, D, FF, g->lin2_w, 0, D, wst));
  }
  {  // g_h = g_ff . W2^T through dropout(relu(.)): masked by the packed forward activations, column sums = linear1's bias gradient
    BlProfScope ps(BL_PROF_LINEAR_DGRAD, 2.0 * R * D * (double)FF, stream, 0.0, two);
    const bl_rows_packed_t a = one_source(w.g_ff_p, D);
    bl_x6_epi_t e = {};
    e.form = BL_X6_EPI_MASK_PACK;
    e.y_packed = sv.hid_p;
    e.mask_scale = has_drop(d->drop_ff_hidden) ? 1.0f / (1.0f - d->drop_ff_hidden.p) : 1.0f;
    e.colsum = g->lin1_b;
    e.c_packed = w.g_h_p;
    GL_TRY(bl_gemm_rows_x6_epi2(&a, d->lin2_w_bwd, R, FF, D, &e, nullptr, 0, stream));
  }
  GL_FORK(1)
  {
    BlProfScope ps(BL_PROF_LINEAR_WGRAD, 2.0 * R * D * (double)FF, wst, 0.0, two);
    const bl_rows_packed_t a = one_source(sv.x1p, D);
    GL_TRY(bl_gemm_wgrad_x6(&a, w.g_h_p, nullptr, nullptr, nullptr, 1, R, FF, D, g->lin1_w, 0, FF, wst));
  }
  {  // g_x1 = g_h . W1^T + g_z2
    BlProfScope ps(BL_PROF_LINEAR_DGRAD, 2.0 * R * D * (double)FF, stream, 0.0, two);
    const bl_rows_packed_t a = one_source(w.g_h_p, FF);
    bl_x6_epi_t e = {};
    e.form = BL_X6_EPI_RES;
    e.res = w.g_z2;
    e.ld_res = D;
    GL_TRY(bl_gemm_rows_x6_epi2(&a, d->lin1_w_bwd, R, D, FF, &e, w.g_x1, D, stream));
  }
  {  // x1 = norm1(z1), z1 = x + dropout1(out_proj(context))
    BlProfScope ps(BL_PROF_LAYERNORM_BWD_BRANCH, 0.0, stream, 4.0 * R * D * 3.0 + 6.0 * R * D, two);
    GL_TRY(bl_layernorm_bwd_branch(w.g_x1, sv.z1, sv.mean1, sv.rstd1, d->norm_g, R, D, w.g_z1, g->norm_g, g->norm_b, d->drop_att_out, nullptr,
                                   w.g_att_p, stream));
  }
  GL_FORK(2)
  {
    BlProfScope ps(BL_PROF_LINEAR_WGRAD, 2.0 * R * D * (double)D, wst, 0.0, two);
    const bl_rows_packed_t a = one_source(sv.ctx_p, D);
    GL_TRY(bl_gemm_wgrad_x6(&a, w.g_att_p, nullptr, nullptr, nullptr, 1, R, D, D, g->out_w, 0, D, wst));
  }
  {
    BlProfScope ps(BL_PROF_LINEAR_DGRAD, 2.0 * R * D * (double)D, stream, 0.0, two);
    const bl_rows_packed_t a = one_source(w.g_att_p, D);
    GL_TRY(bl_gemm_rows_x6(&a, nullptr, 0, d->out_w_bwd, 0, nullptr, nullptr, 1, R, D, D, w.g_ctx, D, stream));
  }
  const bl_head_view_t q = view_of(sv.qkv, 0, s, 3 * D, 3 * dk), k = view_of(sv.qkv, dk, s, 3 * D, 3 * dk),
                       v = view_of(sv.qkv, 2 * dk, s, 3 * D, 3 * dk);
  // the gradients of q / k / v leave the attention kernels as columns of the packed [B L, 9 D] operand of the QKV projection's GEMMs
  const bl_packed_head_view_t gq = {w.g_qkv_p, 3 * D, 0, 3 * dk}, gk = {w.g_qkv_p, 3 * D, dk, 3 * dk}, gv = {w.g_qkv_p, 3 * D, 2 * dk, 3 * dk};
  const bl_head_view_t gc = view_of(w.g_ctx, 0, s, D, dk);
  const bl_dropout_t nodrop = {0.f, 0u, 0u};
  const double mm_flop = 2.0 * s.G * s.L * s.L * dk, mm_bytes = 4.0 * s.G * s.L * (s.L + 2.0 * dk);
  {  // g_v = dropout(P)^T . dO  (on the side stream next to the probabilities' backward it only took bandwidth from that kernel:
     // 217 -> 293 us, the step 1.8 % slower -- tools/experiments/README.md)
    BlProfScope ps(BL_PROF_ATTN_TRANSPOSED_TIMES, mm_flop, stream, mm_bytes, two);
    GL_TRY(bl_attn_transposed_times_v(sv.P, &gc, 1.0f, s.B, s.H, s.L, dk, nullptr, d->drop_attn, &gv, stream));
  }
  const bool edges = d->row_ptr != nullptr;
  {  // dS from dO . V^T, the dropout mask, the softmax and the edge terms
    BlProfScope ps(BL_PROF_ATTN_PROBS_BWD, 0.0, stream, 4.0 * s.G * s.L * (2.0 * s.L + 3.0 * dk), two);
    GL_TRY(bl_rel_attn_probs_bwd_v(&gc, &v, sv.P, &q, scale, d->row_ptr, d->ekey, d->ecode, s.B, s.L, s.H, dk, s.T, d->bias_f, d->bias_r,
                                   d->drop_attn, w.dS, edges ? w.gq_edge : nullptr, g->bias_f, g->bias_r, stream));
  }
  {  // g_q = (dS . K + edge part) / sqrt(dk)
    BlProfScope ps(BL_PROF_ATTN_ROWS_TIMES, mm_flop, stream, mm_bytes, two);
    GL_TRY(bl_attn_rows_times_v(w.dS, &k, s.B, s.H, s.L, dk, edges ? w.gq_edge : nullptr, scale, nullptr, nodrop, &gq, stream));
  }
  {  // g_k = dS^T . (q / sqrt(dk))
    BlProfScope ps(BL_PROF_ATTN_TRANSPOSED_TIMES, mm_flop, stream, mm_bytes, two);
    GL_TRY(bl_attn_transposed_times_v(w.dS, &q, scale, s.B, s.H, s.L, dk, nullptr, nodrop, &gk, stream));
  }
  GL_FORK(3)
  {
    BlProfScope ps(BL_PROF_LINEAR_WGRAD, 2.0 * R * D * 3.0 * D, wst, 0.0, two);
    const bl_rows_packed_t a = one_source(xp, D);
    GL_TRY(bl_gemm_wgrad_x6(&a, w.g_qkv_p, nullptr, nullptr, nullptr, 1, R, 3 * D, D, g->qkv_w, 0, 3 * D, wst));
  }
  {  // g_x = g_qkv . Wqkv^T + g_z1
    BlProfScope ps(BL_PROF_LINEAR_DGRAD, 2.0 * R * D * 3.0 * D, stream, 0.0, two);
    const bl_rows_packed_t a = one_source(w.g_qkv_p, 3 * D);
    bl_x6_epi_t e = {};
    e.form = BL_X6_EPI_RES;
    e.res = w.g_z1;
    e.ld_res = D;
    GL_TRY(bl_gemm_rows_x6_epi2(&a, d->qkv_w_bwd, R, D, 3 * D, &e, g_x, D, stream));
  }
  if (two) {  // the workspace (every weight gradient's second operand) is the caller's again when the call's stream gets here
    (void)hipEventRecord(ev->join, side);
    (void)hipStreamWaitEvent(st, ev->join, 0);
  }
#undef GL_FORK
  return BL_OK;
}
