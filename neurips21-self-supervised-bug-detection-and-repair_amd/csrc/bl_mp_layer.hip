// One MlpMessagePassingLayer per C call: bl_mp_layer_fwd / bl_mp_layer_bwd (SURVEY.md section 8b's minimum
// set; kwargs of the reference call site buglab/models/gnnlayerdefs.py:6-23) + the optional per-kernel
// HIP-event timing bench.py reads (bl_prof_*).
//
//   forward   pack h (one or two sources: a ConcatResidual input [stash ; current] is packed in place, no
//             concatenated copy) -> bf16x6 message GEMM -> segmented max + GELU + LayerNorm (+ routing bitmask,
//             activation derivative at the winners) -> dense + tanh + dropout
//   backward  node update's backward in one kernel (bl_node_update_bwd: act/dropout backward + bias gradient, dense input
//             gradient, LayerNorm backward, packed results; three kernels where its shapes do not apply) -> dense weight
//             gradient (side stream) || routed bf16x6 weight gradient (side stream) || routed input gradient (vector
//             units from the non-zeros, or bf16x6 GEMM) -> segmented sums over the src / tgt CSRs (two outputs for a
//             folded concat)
//   forward-only (saved == NULL): the same forward without any store that only a backward pass reads
// The caller owns every buffer: `saved` lives from forward to backward, `ws` only during the call
// (sizes from bl_mp_layer_saved_bytes / bl_mp_layer_workspace_bytes); nothing is allocated here except three
// HIP events per device (created once) used to fork / join the side stream.
#include <vector>

#include <string.h>

#include "bl_common.h"

// ---- per-kernel timing ----------------------------------------------------------------------------
namespace {
struct ProfRec {
  int kind;
  double flop;
  hipEvent_t e0, e1;
  bool overlapped;
  double bytes;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_pool;
const char* kProfNames[] = {"pack_rows",      "msg_gemm_x6",  "segment_max_ln", "dense_fwd",    "act_bwd",       "dense_wgrad",
                            "dense_dgrad",    "layernorm_bwd", "msg_wgrad_x6",   "msg_dgrad_x6", "node_grad_sums", "msg_dgrad_nodes",
                            "node_update_bwd", "msg_gemm_h3", "msg_wgrad_h3", "msg_dgrad_h3", "pack_gq_h3",
                            // the relational transformer block (csrc/bl_great_layer.hip; BL_PROF_GREAT_* in bl_common.h)
                            "linear_x6_epi", "linear_dgrad_x6", "gemm_wgrad_x6", "attn_probs_fwd", "attn_probs_bwd", "attn_rows_times",
                            "attn_transposed_times", "add_layernorm", "layernorm_bwd_branch"};
constexpr int kProfKinds = sizeof(kProfNames) / sizeof(kProfNames[0]);

hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

struct ProfScope {  // brackets the launches made while it is alive with two events on `st`
  hipStream_t st;
  bool on;
  ProfScope(int kind, double flop, hipStream_t s, bool overlapped) : st(s), on(g_prof_on) {
    if (!on) return;
    ProfRec r{kind, flop, prof_event(), prof_event(), overlapped, 0.0};
    (void)hipEventRecord(r.e0, st);
    g_prof.push_back(r);
  }
  ~ProfScope() {
    if (on) (void)hipEventRecord(g_prof.back().e1, st);
  }
};
}  // namespace

// the same for the other per-layer entry points of the library (declared in bl_common.h)
BlProfScope::BlProfScope(int kind, double flop, void* stream, double bytes, bool overlapped) : st(stream), on(g_prof_on) {
  if (!on) return;
  ProfRec r{kind, flop, prof_event(), prof_event(), overlapped, bytes};
  (void)hipEventRecord(r.e0, (hipStream_t)st);
  g_prof.push_back(r);
}
BlProfScope::~BlProfScope() {
  if (on) (void)hipEventRecord(g_prof.back().e1, (hipStream_t)st);
}

extern "C" int bl_prof_enable(int32_t on) {
  g_prof_on = on != 0;
  return BL_OK;
}
extern "C" int bl_prof_reset(void) {
  for (auto& r : g_prof) {
    g_prof_pool.push_back(r.e0);
    g_prof_pool.push_back(r.e1);
  }
  g_prof.clear();
  return BL_OK;
}
extern "C" int bl_prof_num_kinds(void) { return kProfKinds; }
extern "C" const char* bl_prof_kind_name(int32_t kind) { return kind >= 0 && kind < kProfKinds ? kProfNames[kind] : ""; }
// totals of one kind since the last reset; the caller must have synchronised the device
extern "C" int bl_prof_read(int32_t kind, double* ms, double* flop, int64_t* launches, int32_t* overlapped) {
  double t = 0, f = 0;
  int64_t n = 0;
  int ov = 0;
  for (auto& r : g_prof) {
    if (r.kind != kind) continue;
    float dt = 0.f;
    hipError_t e = hipEventElapsedTime(&dt, r.e0, r.e1);
    if (e != hipSuccess) {
      bl_set_error("bl_prof_read: %s (synchronise the device first)", hipGetErrorString(e));
      return (int)e;
    }
    t += dt;
    f += r.flop;
    ov |= r.overlapped ? 1 : 0;
    ++n;
  }
  if (ms) *ms = t;
  if (flop) *flop = f;
  if (launches) *launches = n;
  if (overlapped) *overlapped = ov;
  return BL_OK;
}
// algorithmic bytes the launches of a (memory-bound) kind were recorded with since the last reset (0 for the GEMM kinds)
extern "C" double bl_prof_read_bytes(int32_t kind) {
  double b = 0;
  for (auto& r : g_prof)
    if (r.kind == kind) b += r.bytes;
  return b;
}

// ---- buffer carving ---------------------------------------------------------------------------------
extern bool g_h3_one_term;  // csrc/bl_gemm_h3.hip: the f16x3 GEMMs evaluate the high-plane term only (mode 2)
namespace {
inline size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
inline size_t packed_w_elems(int G, int K, int N) { return (size_t)G * ((N + 127) / 128) * (K / 32) * 12288; }

struct Saved {  // forward -> backward
  uint16_t* hp;      // [N, 3 Din]  packed layer input
  float* dact;       // [N, Dm]     message activation derivative at each winner (GELU only)
  uint32_t* bits;    // [E, Dm/32]  routing bitmask
  float* agg;        // [N, Dm]
  float* mean;       // [N]
  float* rstd;       // [N]
  float* ln_out;     // [N, Dm] fp32, or (dense node update as bf16x6) the bf16x3-packed form [N, 3 Dm] in the same region
  size_t bytes;
};
Saved carve_saved(void* base, int N, int E, int Din, int Dm, int msg_act) {
  Saved s;
  char* p = (char*)base;
  size_t o = 0;
  auto take = [&](size_t b) { char* q = p ? p + o : nullptr; o += al(b); return q; };
  s.hp = (uint16_t*)take((size_t)N * 3 * Din * 2);
  s.dact = (float*)(msg_act == BL_ACT_NONE ? nullptr : take((size_t)N * Dm * 4));
  s.bits = (uint32_t*)take((size_t)E * (Dm / 32) * 4);
  s.agg = (float*)take((size_t)N * Dm * 4);
  s.mean = (float*)take((size_t)N * 4);
  s.rstd = (float*)take((size_t)N * 4);
  s.ln_out = (float*)take((size_t)N * Dm * 6);
  s.bytes = o;
  return s;
}
struct WsInfer {  // forward-only call (saved == NULL): the three buffers a forward pass cannot do without
  float* pre;      // [E, Dm]     pre-activations of the messages
  uint16_t* hp;    // [N, 3 Din]  packed layer input
  float* ln_out;   // [N, Dm] fp32 or its packed form [N, 3 Dm]
  size_t bytes;
};
WsInfer carve_infer(void* base, int N, int E, int Din, int Dm) {
  WsInfer w;
  char* p = (char*)base;
  size_t o = 0;
  auto take = [&](size_t b) { char* q = p ? p + o : nullptr; o += al(b); return q; };
  w.pre = (float*)take((size_t)E * Dm * 4);
  w.hp = (uint16_t*)take((size_t)N * 3 * Din * 2);
  w.ln_out = (float*)take((size_t)N * Dm * 6);
  w.bytes = o;
  return w;
}
struct WsBwd {
  float* g_z;      // [N, Dout] fp32, or its bf16x3-packed form [N, 3 Dout] (same region)
  float* g_ln;     // [N, Dm]
  uint16_t* gqp;   // [N, 3 Dm] (bf16x3) or [N, 2 Dm] (f16x2)
  float* g_a;      // [E, 2 Din]; [E, Din] (source halves only) when the node sums are fused into the input gradient
  float* amax;     // f16x3 message GEMMs: max |gq| of this layer, on the device
  size_t bytes;
};
WsBwd carve_bwd(void* base, int N, int E, int Din, int Dm, int Dout, bool full_ga = true) {
  WsBwd w;
  char* p = (char*)base;
  size_t o = 0;
  auto take = [&](size_t b) { char* q = p ? p + o : nullptr; o += al(b); return q; };
  w.g_z = (float*)take((size_t)N * Dout * 6);
  w.g_ln = (float*)take((size_t)N * Dm * 4);
  w.gqp = (uint16_t*)take((size_t)N * 3 * Dm * 2);
  w.g_a = (float*)take((size_t)E * (full_ga ? 2 : 1) * Din * 4);
  w.amax = (float*)take(256);
  w.bytes = o;
  return w;
}

// Message GEMMs (forward, weight gradient, routed input gradient) as f16x3 (csrc/bl_gemm_h3.hip: two fp16 planes per operand,
// three MFMA terms) instead of bf16x6.  The operands of the forward GEMM and the weight gradient's left operand are layer inputs
// and weights -- bounded tensors, fixed power-of-two scales BL_H3_ROW_SCALE / BL_H3_W_SCALE; the routed gradient operand gq has
// no bound known in advance: its amax is taken on the device and the scale derived from it there (no host round trip).  The
// dense node update stays on bf16x6 (its operands g_z / LayerNorm outputs are produced packed by fused kernels).
// Mode 2 ("f16x1", `train.py --amp`): the same packed images and kernels, high-plane term only (g_h3_one_term in bl_gemm_h3.hip) --
// fp16 operands, fp32 accumulation and fp32 results, the gradient operand scaled by its device-side amax (what a GradScaler is for).
int g_msg_h3 = -1;
bool msg_h3() {
  if (g_msg_h3 < 0) {
    const char* e = getenv("BL_MSG_GEMM");
    g_msg_h3 = (e && (e[0] == 'x' || e[0] == 'b')) ? 0 : ((e && (strcmp(e, "amp") == 0 || strcmp(e, "f16x1") == 0)) ? 2 : 1);  // "x6" / "bf16x6" -> 0; "amp" / "f16x1" -> 2; default f16x3
    g_h3_one_term = g_msg_h3 == 2;
  }
  return g_msg_h3 >= 1;
}

bool g_fused_node_bwd = true;  // bl_set_fused_node_bwd: act backward -> dense input gradient -> LayerNorm backward in one kernel

// fork / join events of the side stream, one set per device (a process that drives several GPUs -- tests, tools -- must not
// record an event created on another device)
constexpr int kMaxDevices = 16;
struct SideEvents { hipEvent_t fork1 = nullptr, fork2 = nullptr, join = nullptr; };
SideEvents g_side_events[kMaxDevices];
SideEvents* ensure_events() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  SideEvents& e = g_side_events[dev];
  if (e.join) return &e;
  if (hipEventCreateWithFlags(&e.fork1, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e.fork2, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e.join, hipEventDisableTiming) != hipSuccess)
    return nullptr;
  return &e;
}

inline int saved_act_code(const bl_mp_layer_t* L) {
  return (L->aggregation == BL_AGG_MEAN && L->msg_act == BL_ACT_NONE) ? BL_ACT_GELU_AGG : L->msg_act;
}

int check_layer(const bl_mp_layer_t* L, const char* who) {
  BL_CHECK_ARG(L && L->N > 0 && L->E >= 0 && L->T > 0, "%s: bad sizes", who);
  BL_CHECK_ARG(L->Din % 32 == 0 && L->Dm % 32 == 0 && L->Dout % 4 == 0 && L->Din > 0 && L->Dm > 0 && L->Dm <= 512 && L->Dout > 0,
               "%s: the fused layer needs Din, Dm multiples of 32 (bf16x6 GEMMs), Dm <= 512", who);
  // (a minibatch without messages has empty per-message arrays: PyTorch hands out NULL for those)
  BL_CHECK_ARG(L->type_ptr && L->tgt_ptr && L->src_ptr && (L->E == 0 || (L->msg_src && L->msg_tgt && L->tgt_msgs && L->src_msgs)),
               "%s: null graph index array", who);
  BL_CHECK_ARG(L->W && L->ln_g && L->ln_b && L->Wd && L->bd, "%s: null parameter", who);
  BL_CHECK_ARG(L->msg_act == BL_ACT_NONE || L->msg_act == BL_ACT_GELU || L->msg_act == BL_ACT_GELU_AGG,
               "%s: message activation must be none, gelu (per message) or gelu_agg (on the aggregate)", who);
  BL_CHECK_ARG(L->aggregation >= BL_AGG_MAX && L->aggregation <= BL_AGG_MEAN, "%s: aggregation must be BL_AGG_MAX, BL_AGG_SUM or BL_AGG_MEAN", who);
  BL_CHECK_ARG(L->aggregation == BL_AGG_MAX || L->msg_act != BL_ACT_GELU,
               "%s: sum / mean aggregation takes the activation on the aggregate (BL_ACT_GELU_AGG) or none: a per-message activation would need "
               "the [E, Dm] messages again in backward", who);
  return BL_OK;
}
}  // namespace

// A/B switch (tests, bench): 1 (default) = the node update's backward chain runs as bl_node_update_bwd where the shapes allow
// (Dm 128 / 256, dense node update on the bf16x6 path, deterministic mode off); 0 = three kernels.  Returns the previous value.
extern "C" int32_t bl_set_fused_node_bwd(int32_t on) {
  const int32_t prev = g_fused_node_bwd ? 1 : 0;
  g_fused_node_bwd = on != 0;
  return prev;
}

extern "C" int64_t bl_mp_layer_saved_bytes(int32_t N, int32_t E, int32_t Din, int32_t Dm, int32_t msg_act) {
  return (int64_t)carve_saved(nullptr, N, E, Din, Dm, msg_act).bytes;
}
// forward scratch: the [E, Dm] pre-activations; backward scratch: g_z, g_ln, packed node gradient, [E, 2 Din] input gradients
extern "C" int64_t bl_mp_layer_workspace_bytes(int32_t N, int32_t E, int32_t Din, int32_t Dm, int32_t Dout, int32_t backward) {
  if (backward == 3) return (int64_t)carve_infer(nullptr, N, E, Din, Dm).bytes;
  if (backward) return (int64_t)carve_bwd(nullptr, N, E, Din, Dm, Dout, backward != 2).bytes;
  return (int64_t)al((size_t)E * Dm * 4);
}
// Which image of the per-type weights a layer call expects: 1 = bl_pack_weights_x6w (the wide row GEMM takes this shape:
// message GEMM N = Dm, K = 2 Din; input-gradient GEMM N = 2 Din, K = Dm), 0 = bl_pack_weights_x6.
extern "C" int32_t bl_mp_layer_weight_image(int32_t Din, int32_t Dm, int32_t for_backward) {
  if (msg_h3()) return 2;  // bl_pack_weights_h3 (kinds 5 / 6 of bl_pack_weights_multi)
  return for_backward ? bl_gemm_rows_x6w_ok(2 * Din, Dm) : bl_gemm_rows_x6w_ok(Dm, 2 * Din);
}
extern "C" int32_t bl_set_msg_gemm_mode(int32_t mode) {
  (void)msg_h3();
  const int32_t prev = g_msg_h3;
  g_msg_h3 = mode == 2 ? 2 : (mode ? 1 : 0);
  g_h3_one_term = g_msg_h3 == 2;
  return prev;
}
extern "C" int32_t bl_get_msg_gemm_mode(void) {
  (void)msg_h3();
  return g_msg_h3;
}

extern "C" int64_t bl_mp_layer_packed_weight_elems(int32_t T, int32_t Din, int32_t Dm, int32_t for_backward) {
  if (msg_h3()) return for_backward ? bl_packed_weight_elems_h3(T, Dm, 2 * Din) : bl_packed_weight_elems_h3(T, 2 * Din, Dm);
  if (bl_mp_layer_weight_image(Din, Dm, for_backward))
    return for_backward ? bl_packed_weight_elems_x6w(T, Dm, 2 * Din) : bl_packed_weight_elems_x6w(T, 2 * Din, Dm);
  return (int64_t)(for_backward ? packed_w_elems(T, Dm, 2 * Din) : packed_w_elems(T, 2 * Din, Dm));
}

#define BL_TRY(call_)        \
  do {                       \
    int rc_ = (call_);       \
    if (rc_ != BL_OK) return rc_; \
  } while (0)

extern "C" int bl_mp_layer_fwd(const bl_mp_layer_t* L, const float* h_lo, int32_t ld_lo, int32_t width_lo, const float* h_hi,
                               int32_t ld_hi, const uint16_t* w_packed, float* h_out, int32_t* winner_out, void* saved,
                               void* ws, void* stream) {
  BL_TRY(check_layer(L, "bl_mp_layer_fwd"));
  const int N = L->N, E = L->E, T = L->T, Din = L->Din, Dm = L->Dm, Dout = L->Dout;
  // saved == NULL: forward-only call (model.predict, evaluate.py) -- nothing is kept for a backward pass: no routing bitmask, no
  // activation derivative, no aggregate / mean / rstd stores; the packed input and the LayerNorm output live in `ws`
  const bool infer = saved == nullptr;
  BL_CHECK_ARG(h_lo && w_packed && h_out && (ws || (E == 0 && !infer)), "bl_mp_layer_fwd: null buffer");
  BL_CHECK_ARG((h_hi == nullptr && width_lo == Din) || (h_hi != nullptr && width_lo > 0 && width_lo < Din && width_lo % 32 == 0),
               "bl_mp_layer_fwd: width_lo must be Din (one source) or a multiple of 32 below Din (two sources)");
  hipStream_t st = (hipStream_t)stream;
  // node_order lists the hubs first and the other nodes in natural order: without hubs it is the identity, and the per-node
  // kernels skip the load (one dependent round trip less per wave)
  const int32_t* node_order = L->num_hub_slots == 0 ? nullptr : L->node_order;
  // ("mean" without an activation still needs the [N, Dm] derivative array: it carries the 1 / in-degree)
  Saved S = carve_saved(saved, N, E, Din, Dm, saved_act_code(L));
  float* pre = (float*)ws;
  if (infer) {
    const WsInfer I = carve_infer(ws, N, E, Din, Dm);
    pre = I.pre;
    S.hp = I.hp;
    S.ln_out = I.ln_out;
    S.dact = nullptr; S.bits = nullptr; S.agg = nullptr; S.mean = nullptr; S.rstd = nullptr;
  }
  const bool h3 = msg_h3();
  if (h3) {
    ProfScope ps(0, 0.0, st, false);
    BL_TRY(bl_pack_f16x2(h_lo, ld_lo, N, h_hi ? width_lo : Din, Din, 0, BL_H3_ROW_SCALE, nullptr, S.hp, st));
    if (h_hi) BL_TRY(bl_pack_f16x2(h_hi, ld_hi, N, Din - width_lo, Din, width_lo, BL_H3_ROW_SCALE, nullptr, S.hp, st));
  } else {
    ProfScope ps(0, 0.0, st, false);
    if (h_hi == nullptr) {
      BL_TRY(bl_pack_bf16x3(h_lo, ld_lo, N, Din, S.hp, st));
    } else {
      BL_TRY(bl_pack_bf16x3_cols(h_lo, ld_lo, N, width_lo, Din, 0, S.hp, st));
      BL_TRY(bl_pack_bf16x3_cols(h_hi, ld_hi, N, Din - width_lo, Din, width_lo, S.hp, st));
    }
  }
  bl_rows_packed_t a;
  a.xp[0] = S.hp; a.xp[1] = S.hp; a.xp[2] = nullptr;
  a.idx[0] = L->msg_src; a.idx[1] = L->msg_tgt; a.idx[2] = nullptr;
  a.width[0] = Din; a.width[1] = Din; a.width[2] = 0;
  a.nsrc = 2;
  if (h3) {
    ProfScope ps(13, 2.0 * E * (2.0 * Din) * Dm, st, false);
    BL_TRY(bl_gemm_rows_h3(&a, nullptr, 0, w_packed, bl_packed_weight_elems_h3(1, 2 * Din, Dm), L->type_ptr, nullptr, T, E, Dm, 2 * Din,
                           1.0f / (BL_H3_ROW_SCALE * BL_H3_W_SCALE), nullptr, pre, Dm, st));
  } else {
    ProfScope ps(1, 2.0 * E * (2.0 * Din) * Dm, st, false);
    if (bl_mp_layer_weight_image(Din, Dm, 0))  // >= 256 output columns: the wide (128 x 256 tile, LDS-DMA) form, bit-identical
      BL_TRY(bl_gemm_rows_x6w(&a, nullptr, 0, w_packed, bl_packed_weight_elems_x6w(1, 2 * Din, Dm), L->type_ptr, nullptr, T, E, Dm,
                              2 * Din, pre, Dm, st));
    else
      BL_TRY(bl_gemm_rows_x6(&a, nullptr, 0, w_packed, (int64_t)packed_w_elems(1, 2 * Din, Dm), L->type_ptr, nullptr, T, E, Dm,
                             2 * Din, pre, Dm, st));
  }
  // dense node update on the bf16x6 path when the caller packed Wd (bl_pack_weights_x6(Wd, 1, Dm, Dout, w_is_kn = 1)): the
  // LayerNorm epilogue of the segmented max then emits its result in packed form and no fp32 copy is kept
  const bool dense_x6 = L->Wd_packed != nullptr && Dm % 32 == 0 && Dout % 32 == 0;
  {
    ProfScope ps(2, 0.0, st, false);
    // (E == 0: every segment is empty and the kernel never dereferences `pre`)
    if (L->aggregation != BL_AGG_MAX)  // ptgnn's "sum" / "mean" (no winners: winner_out is left untouched)
      BL_TRY(bl_segment_sum_fwd_impl(pre, Dm, L->tgt_ptr, L->tgt_msgs, N, Dm, L->msg_act, L->aggregation == BL_AGG_MEAN, S.agg, L->ln_g, L->ln_b,
                                     L->ln_eps, dense_x6 ? nullptr : S.ln_out, S.mean, S.rstd, S.dact, node_order,
                                     dense_x6 ? (uint16_t*)S.ln_out : nullptr, st));
    else
    BL_TRY(bl_segment_max_fwd_impl(pre, Dm, L->tgt_ptr, L->tgt_msgs, N, Dm, L->msg_act, S.agg, winner_out, L->ln_g, L->ln_b,
                                   L->ln_eps, dense_x6 ? nullptr : S.ln_out, S.mean, S.rstd, S.dact, S.bits, node_order,
                                   dense_x6 ? (uint16_t*)S.ln_out : nullptr, L->num_hub_slots, st));
  }
  {
    ProfScope ps(3, 2.0 * N * (double)Dm * Dout, st, false);
    if (dense_x6) {
      bl_rows_packed_t d;
      d.xp[0] = (const uint16_t*)S.ln_out; d.xp[1] = d.xp[2] = nullptr; d.idx[0] = d.idx[1] = d.idx[2] = nullptr;
      d.width[0] = Dm; d.width[1] = d.width[2] = 0; d.nsrc = 1;
      BL_TRY(bl_gemm_rows_x6_epi(&d, L->Wd_packed, 0, nullptr, nullptr, 1, N, Dout, Dm, L->bd, BL_ACT_TANH, L->drop, h_out, Dout, st));
    } else {
      bl_rows_t d;
      d.x[0] = S.ln_out; d.idx[0] = nullptr; d.ld[0] = Dm; d.width[0] = Dm; d.nsrc = 1;
      d.x[1] = d.x[2] = nullptr; d.idx[1] = d.idx[2] = nullptr; d.ld[1] = d.ld[2] = 0; d.width[1] = d.width[2] = 0;
      BL_TRY(bl_gemm_rows(&d, L->Wd, 0, Dout, 0, L->bd, nullptr, nullptr, 1, N, Dout, Dm, BL_ACT_TANH, L->drop, h_out, Dout, st));
    }
  }
  return BL_OK;
}

extern "C" int bl_mp_layer_bwd(const bl_mp_layer_t* L, const float* h_out, const float* g_out, const uint16_t* w_packed_bwd,
                               const void* saved, void* ws, float* g_h_lo, int32_t ld_lo, int32_t width_lo, float* g_h_hi,
                               int32_t ld_hi, float* g_W, float* g_ln_g, float* g_ln_b, float* g_Wd, float* g_bd,
                               void* stream, void* side_stream, int32_t join_side) {
  BL_TRY(check_layer(L, "bl_mp_layer_bwd"));
  const int N = L->N, E = L->E, T = L->T, Din = L->Din, Dm = L->Dm, Dout = L->Dout;
  BL_CHECK_ARG(h_out && g_out && saved && ws && g_h_lo && g_W && g_ln_g && g_ln_b && g_Wd && g_bd, "bl_mp_layer_bwd: null buffer");
  BL_CHECK_ARG(w_packed_bwd || (L->Wt && L->E > 0 && bl_routed_dgrad_vec_ok(L->Dm, 2 * L->Din)) || L->E == 0,
               "bl_mp_layer_bwd: w_packed_bwd may only be NULL when the input gradient takes the Wt path");
  BL_CHECK_ARG((g_h_hi == nullptr && width_lo == Din) || (g_h_hi != nullptr && width_lo > 0 && width_lo < Din),
               "bl_mp_layer_bwd: width_lo must be Din (one output) or below Din (two outputs)");
  hipStream_t st = (hipStream_t)stream, side = side_stream ? (hipStream_t)side_stream : st;
  const int32_t* node_order = L->num_hub_slots == 0 ? nullptr : L->node_order;  // (identity without hubs: see bl_mp_layer_fwd)
  const bool two = side != st;
  SideEvents* ev = two ? ensure_events() : nullptr;
  if (two) BL_CHECK_ARG(ev != nullptr, "bl_mp_layer_bwd: cannot create HIP events");
  Saved S = carve_saved(const_cast<void*>(saved), N, E, Din, Dm, saved_act_code(L));
  const uint32_t* bits = L->aggregation == BL_AGG_MAX ? S.bits : nullptr;  // sum / mean: every message receives its target's gradient
  // the input gradient from the non-zeros of the routed message gradient (vector units) when the caller supplied W^T and
  // W[t]^T fits one LDS block; node sums fused in (atomics) unless the deterministic mode asks for a fixed summation order
  const bool vec_dgrad = L->Wt != nullptr && E > 0 && bl_routed_dgrad_vec_ok(Dm, 2 * Din) && L->aggregation == BL_AGG_MAX;
  const bool fused_sums = vec_dgrad && !bl_get_deterministic();
  WsBwd B = carve_bwd(ws, N, E, Din, Dm, Dout, !fused_sums);

  // dense node update on the bf16x6 path: forward must have run with Wd_packed too (it decides the form of `saved`)
  const bool dense_x6 = L->Wd_packed != nullptr && L->Wd_packed_bwd != nullptr && Dm % 32 == 0 && Dout % 32 == 0;
  BL_CHECK_ARG(dense_x6 || L->Wd_packed == nullptr, "bl_mp_layer_bwd: Wd_packed given without Wd_packed_bwd");
  // act backward -> dense input gradient -> LayerNorm backward x activation derivative in ONE kernel (csrc/bl_node_bwd.hip)
  const bool fused_node = dense_x6 && g_fused_node_bwd && bl_node_update_bwd_ok(Dm, Dout);
  // f16x3 message GEMMs: gq leaves its producer in fp32 (B.g_ln); its packed form is made below, once its amax is known
  const bool h3 = msg_h3();
  if (h3 && E > 0 && hipMemsetAsync(B.amax, 0, 4, st) != hipSuccess) { bl_set_error("bl_mp_layer_bwd: memset failed"); return BL_EINVAL; }
  if (fused_node) {
    ProfScope ps(12, 2.0 * N * (double)Dm * Dout, st, false);
    BL_TRY(bl_node_update_bwd_impl(g_out, h_out, N, Dout, L->drop, L->Wd_packed_bwd, S.agg, S.mean, S.rstd, L->ln_g, S.dact, Dm,
                                   (uint16_t*)B.g_z, g_bd, (vec_dgrad || h3) ? B.g_ln : nullptr, h3 ? nullptr : B.gqp, g_ln_g, g_ln_b,
                                   (h3 && E > 0) ? B.amax : nullptr, st));
  } else {  // y = drop(tanh(z)): g_z (packed for the bf16x6 GEMMs), bias gradient
    ProfScope ps(4, 0.0, st, false);
    BL_TRY(bl_act_bwd_impl(g_out, h_out, N, Dout, Dout, BL_ACT_TANH, L->drop, dense_x6 ? nullptr : B.g_z, g_bd,
                           dense_x6 ? (uint16_t*)B.g_z : nullptr, st));
  }
  bl_rows_t r1;
  r1.x[1] = r1.x[2] = nullptr; r1.idx[0] = r1.idx[1] = r1.idx[2] = nullptr; r1.ld[1] = r1.ld[2] = 0; r1.width[1] = r1.width[2] = 0; r1.nsrc = 1;
  bl_rows_packed_t p1;
  p1.xp[1] = p1.xp[2] = nullptr; p1.idx[0] = p1.idx[1] = p1.idx[2] = nullptr; p1.width[1] = p1.width[2] = 0; p1.nsrc = 1;
  if (two) {
    (void)hipEventRecord(ev->fork1, st);
    (void)hipStreamWaitEvent(side, ev->fork1, 0);
  }
  {  // dense weight gradient, next to the input-gradient chain
    ProfScope ps(5, 2.0 * N * (double)Dm * Dout, side, two);
    if (dense_x6) {
      p1.xp[0] = (const uint16_t*)S.ln_out; p1.width[0] = Dm;
      BL_TRY(bl_gemm_wgrad_x6(&p1, (const uint16_t*)B.g_z, nullptr, nullptr, nullptr, 1, N, Dout, Dm, g_Wd, 0, Dout, side));
    } else {
      r1.x[0] = S.ln_out; r1.ld[0] = Dm; r1.width[0] = Dm;
      BL_TRY(bl_gemm_wgrad(&r1, B.g_z, Dout, nullptr, nullptr, 1, N, Dout, Dm, g_Wd, 0, Dout, side));
    }
  }
  if (!fused_node) {
    ProfScope ps(6, 2.0 * N * (double)Dm * Dout, st, two);
    if (dense_x6) {
      p1.xp[0] = (const uint16_t*)B.g_z; p1.width[0] = Dout;
      // (the epilogue form with nothing in it: a kernel instantiation of its own, so that per-kernel profiles do not mix
      // this small GEMM with the message GEMM's launches)
      bl_dropout_t nodrop = {0.f, 0u, 0u};
      BL_TRY(bl_gemm_rows_x6_epi(&p1, L->Wd_packed_bwd, 0, nullptr, nullptr, 1, N, Dm, Dout, nullptr, BL_ACT_NONE, nodrop, B.g_ln, Dm, st));
    } else {
      r1.x[0] = B.g_z; r1.ld[0] = Dout; r1.width[0] = Dout;
      bl_dropout_t nodrop = {0.f, 0u, 0u};
      BL_TRY(bl_gemm_rows(&r1, L->Wd, 0, Dout, 1, nullptr, nullptr, nullptr, 1, N, Dm, Dout, BL_ACT_NONE, nodrop, B.g_ln, Dm, st));
    }
  }
  if (!fused_node) {  // LayerNorm backward x activation derivative at the winners -> packed d loss / d (winning pre-activation)
    ProfScope ps(7, 0.0, st, two);
    // (the fp32 form of the result, for the vector input gradient, overwrites g_ln in place: the kernel is row-local)
    BL_TRY(bl_layernorm_bwd(B.g_ln, S.agg, S.mean, S.rstd, L->ln_g, N, Dm, (vec_dgrad || h3) ? B.g_ln : nullptr, g_ln_g, g_ln_b, S.dact,
                            h3 ? nullptr : B.gqp, st));
  }
  if (h3 && E > 0) {  // amax of gq on the device -> its power-of-two scale -> two fp16 planes
    ProfScope ps(16, 0.0, st, two);
    if (!fused_node) BL_TRY(bl_amax(B.g_ln, (int64_t)N * Dm, B.amax, st));  // (the fused node-update backward took it on the way)
    BL_TRY(bl_pack_f16x2(B.g_ln, Dm, N, Dm, Dm, 0, 1.0f, B.amax, B.gqp, st));
  }
  // The routed weight gradient (side stream) reads what is ready at this point: the packed layer input, the packed gradient operand, the
  // routing bits.  WHEN it is forked decides what it runs next to.  Layers up to 128 wide: here, side by side with the routed input
  // gradient.  Wider layers (the hidden-256 configurations): behind the segmented sums, i.e. next to the bandwidth-bound tail of this layer
  // and the head of the next one -- measured on one box, two interleaved pairs (tools/experiments/fork_probe.sh,
  // profiles/r06zzn_fork_probe.log): c3 shard 1 664 / 1 660 -> 1 684 / 1 692 graphs/s with the late fork, c2 4 917 / 4 942 -> 4 904 / 4 911
  // (and 4 768 / 4 774 when forked between the input gradient and the sums).
  const bool late_fork = two && Dout >= 256;
  auto launch_wgrad = [&]() -> int {
    bl_rows_packed_t a;
    a.xp[0] = S.hp; a.xp[1] = S.hp; a.xp[2] = nullptr;
    a.idx[0] = L->msg_src; a.idx[1] = L->msg_tgt; a.idx[2] = nullptr;
    a.width[0] = Din; a.width[1] = Din; a.width[2] = 0;
    a.nsrc = 2;
    if (two) {
      (void)hipEventRecord(ev->fork2, st);
      (void)hipStreamWaitEvent(side, ev->fork2, 0);
    }
    if (h3) {
      ProfScope ps(14, 2.0 * E * (2.0 * Din) * Dm, side, two);
      BL_TRY(bl_gemm_wgrad_h3(&a, B.gqp, L->msg_tgt, bits, Dm / 32, L->type_ptr, nullptr, T, E, Dm, 2 * Din, 1.0f / BL_H3_ROW_SCALE, B.amax,
                              g_W, (int64_t)2 * Din * Dm, Dm, side));
    } else {
      ProfScope ps(8, 2.0 * E * (2.0 * Din) * Dm, side, two);
      if (bits)
        BL_TRY(bl_gemm_wgrad_routed_x6(&a, B.gqp, L->msg_tgt, bits, Dm / 32, L->type_ptr, nullptr, T, E, Dm, 2 * Din, g_W,
                                       (int64_t)2 * Din * Dm, Dm, side));
      else
        BL_TRY(bl_gemm_wgrad_x6(&a, B.gqp, L->msg_tgt, L->type_ptr, nullptr, T, E, Dm, 2 * Din, g_W, (int64_t)2 * Din * Dm, Dm, side));
    }
    return BL_OK;
  };
  if (E > 0) {
    if (!late_fork) BL_TRY(launch_wgrad());
    bl_rows_packed_t g;
    g.xp[0] = B.gqp; g.xp[1] = g.xp[2] = nullptr; g.idx[0] = L->msg_tgt; g.idx[1] = g.idx[2] = nullptr;
    g.width[0] = Dm; g.width[1] = g.width[2] = 0; g.nsrc = 1;
    if (fused_sums) {
      ProfScope ps(11, 2.0 * N * (2.0 * Din) * Dm, st, two);  // FLOPs of the non-zeros: one winner per (node, channel)
      const int split = g_h_hi ? width_lo : Din;
      if (ld_lo == split) {
        if (hipMemsetAsync(g_h_lo, 0, (size_t)N * split * 4, st) != hipSuccess) { bl_set_error("bl_mp_layer_bwd: memset failed"); return BL_EINVAL; }
      } else if (hipMemset2DAsync(g_h_lo, (size_t)ld_lo * 4, 0, (size_t)split * 4, N, st) != hipSuccess) { bl_set_error("bl_mp_layer_bwd: memset failed"); return BL_EINVAL; }
      if (g_h_hi) {
        if (ld_hi == Din - split) {
          if (hipMemsetAsync(g_h_hi, 0, (size_t)N * (Din - split) * 4, st) != hipSuccess) { bl_set_error("bl_mp_layer_bwd: memset failed"); return BL_EINVAL; }
        } else if (hipMemset2DAsync(g_h_hi, (size_t)ld_hi * 4, 0, (size_t)(Din - split) * 4, N, st) != hipSuccess) { bl_set_error("bl_mp_layer_bwd: memset failed"); return BL_EINVAL; }
      }
      // target halves by atomics (one per run of equal targets), source halves as rows + a segmented sum over the source CSR:
      // all-atomic, the kernel is bound by the L2's one fp32 atomic per channel per clock (0.475 vs 0.404 ms at c2's layer shape;
      // tools/dgrad_bench.py times both forms through bl_routed_dgrad_nodes / bl_routed_dgrad_nodes_rows)
      BL_TRY(bl_routed_dgrad_nodes_rows(B.g_ln, Dm, L->msg_src, L->msg_tgt, S.bits, Dm / 32, L->type_ptr, T, L->Wt, E, Dm, Din, split,
                                        g_h_lo, ld_lo, g_h_hi, ld_hi, B.g_a, Din, st));
      BL_TRY(bl_mp_scatter_src_accum_impl(B.g_a, Din, L->src_ptr, L->src_msgs, N, Din, split, g_h_lo, ld_lo, g_h_hi, ld_hi,
                                          node_order, L->num_hub_slots, st));
    } else if (vec_dgrad) {
      ProfScope ps(9, 2.0 * N * (2.0 * Din) * Dm, st, two);
      BL_TRY(bl_routed_dgrad_vec(B.g_ln, Dm, L->msg_tgt, S.bits, Dm / 32, L->type_ptr, T, L->Wt, E, Dm, 2 * Din, B.g_a, 2 * Din, st));
    } else if (h3) {
      ProfScope ps(15, 2.0 * E * (2.0 * Din) * Dm, st, two);
      BL_TRY(bl_gemm_rows_h3(&g, bits, Dm / 32, w_packed_bwd, bl_packed_weight_elems_h3(1, Dm, 2 * Din), L->type_ptr, nullptr, T, E, 2 * Din,
                             Dm, 1.0f / BL_H3_W_SCALE, B.amax, B.g_a, 2 * Din, st));
    } else {
      ProfScope ps(9, 2.0 * E * (2.0 * Din) * Dm, st, two);
      if (bl_mp_layer_weight_image(Din, Dm, 1))
        BL_TRY(bl_gemm_rows_x6w(&g, bits, Dm / 32, w_packed_bwd, bl_packed_weight_elems_x6w(1, Dm, 2 * Din), L->type_ptr, nullptr, T,
                                E, 2 * Din, Dm, B.g_a, 2 * Din, st));
      else
        BL_TRY(bl_gemm_rows_x6(&g, bits, Dm / 32, w_packed_bwd, (int64_t)packed_w_elems(1, Dm, 2 * Din), L->type_ptr, nullptr, T, E,
                               2 * Din, Dm, B.g_a, 2 * Din, st));
    }
  }
  if (!fused_sums) {
    ProfScope ps(10, 0.0, st, two);
    // E == 0: both CSRs are empty and g_a is never read
    BL_TRY(bl_mp_scatter_grad_hubs_impl(B.g_a, 2 * Din, L->src_ptr, L->src_msgs, L->tgt_ptr, L->tgt_msgs, N, Din, width_lo, g_h_lo, ld_lo,
                                        g_h_hi, ld_hi, node_order, L->num_hub_slots, st));
  }
  if (E > 0 && late_fork) BL_TRY(launch_wgrad());
  if (two && join_side) {
    (void)hipEventRecord(ev->join, side);
    (void)hipStreamWaitEvent(st, ev->join, 0);
  }
  return BL_OK;
}
