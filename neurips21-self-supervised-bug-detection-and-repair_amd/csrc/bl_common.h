// Shared device/host helpers for libbuglab_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/buglab_hip.h"

#define BL_WAVE 64

// ---- error plumbing ---------------------------------------------------------------------------
void bl_set_error(const char* fmt, ...);
extern "C" int32_t bl_get_deterministic(void);
int bl_num_cus();  // bl_core.hip
int bl_max_lds_per_block();  // bl_core.hip: LDS bytes a workgroup may declare on the current device
// n zeroed turn counters for one launch on `stream`, or nullptr when the deterministic mode is off (bl_core.hip)
unsigned* bl_order_counters(int n, void* stream);
// per-device counter of f16x2 packing threads that had to saturate a value (csrc/bl_gemm_h3.hip); NULL if it cannot be allocated
unsigned* bl_h3_sat_counter();

// internal forms of two exported entry points with an extra bf16x3-packed output (csrc/bl_graph_ops.hip; used by the
// fused layer calls, whose dense node-update GEMMs run as bf16x6)
int bl_segment_max_fwd_impl(const float* x, int32_t ldx, const int32_t* seg_ptr, const int32_t* seg_items, int32_t nseg, int32_t D,
                            int32_t act, float* out, int32_t* arg, const float* ln_g, const float* ln_b, float eps, float* ln_out,
                            float* mean, float* rstd, float* dact, uint32_t* winbits, const int32_t* seg_order,
                            uint16_t* ln_out_packed, int32_t num_hub_slots, void* stream);
int bl_segment_sum_fwd_impl(const float* x, int32_t ldx, const int32_t* seg_ptr, const int32_t* seg_items, int32_t nseg, int32_t D,
                            int32_t act, int32_t mean_agg, float* out, const float* ln_g, const float* ln_b, float eps, float* ln_out,
                            float* mean, float* rstd, float* dact, const int32_t* seg_order, uint16_t* ln_out_packed, void* stream);
int bl_act_bwd_impl(const float* g_y, const float* y, int32_t nrows, int32_t N, int32_t ld, int32_t act, bl_dropout_t drop,
                    float* g_z, float* g_bias, uint16_t* g_z_packed, void* stream);
int bl_mp_scatter_src_accum_impl(const float* g_src, int32_t ld_src, const int32_t* src_ptr, const int32_t* src_msgs, int32_t N,
                                 int32_t Din, int32_t split, float* g_h_lo, int32_t ld_lo, float* g_h_hi, int32_t ld_hi,
                                 const int32_t* node_order, int32_t hub_slots, void* stream);
// bl_mp_scatter_grad / _split with the number of hub entries at the front of node_order (g_h_hi == NULL: one output)
int bl_mp_scatter_grad_hubs_impl(const float* g_a, int32_t ld_ga, const int32_t* src_ptr, const int32_t* src_msgs, const int32_t* tgt_ptr,
                                 const int32_t* tgt_msgs, int32_t N, int32_t Din, int32_t split, float* g_h_lo, int32_t ld_lo, float* g_h_hi,
                                 int32_t ld_hi, const int32_t* node_order, int32_t hub_slots, void* stream);

int bl_node_update_bwd_impl(const float* g_out, const float* h_out, int32_t nrows, int32_t Dout, bl_dropout_t drop,
                            const uint16_t* wd_packed_bwd, const float* agg, const float* mean, const float* rstd, const float* ln_g,
                            const float* dact, int32_t Dm, uint16_t* g_z_packed, float* g_bias, float* gq, uint16_t* gq_packed,
                            float* g_ln_g, float* g_ln_b, float* gq_amax, void* stream);

// per-kernel timing inside a per-layer entry point (csrc/bl_mp_layer.hip: bl_prof_*): brackets the launches made while it is alive
// with two HIP events on `stream` when profiling is enabled.  Kinds: indices into bl_prof_kind_name's table.
struct BlProfScope {
  void* st;
  bool on;
  BlProfScope(int kind, double flop, void* stream, double bytes = 0.0, bool overlapped = false);
  ~BlProfScope();
};
enum {
  BL_PROF_PACK_ROWS = 0,
  BL_PROF_LINEAR_FWD = 17,
  BL_PROF_LINEAR_DGRAD,
  BL_PROF_LINEAR_WGRAD,
  BL_PROF_ATTN_PROBS_FWD,
  BL_PROF_ATTN_PROBS_BWD,
  BL_PROF_ATTN_ROWS_TIMES,
  BL_PROF_ATTN_TRANSPOSED_TIMES,
  BL_PROF_ADD_LAYERNORM,
  BL_PROF_LAYERNORM_BWD_BRANCH
};

#define BL_CHECK_ARG(cond, ...)   \
  do {                            \
    if (!(cond)) {                \
      bl_set_error(__VA_ARGS__);  \
      return BL_EINVAL;           \
    }                             \
  } while (0)
#define BL_LAUNCH_CHECK(name)                                            \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      bl_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return (int)e_;                                                    \
    }                                                                    \
  } while (0)

static inline bool bl_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a DEVICE's copy of a kernel: raised once per (call site, device) -- a
// process may drive several devices (done: the call site's own `static bool [BL_MAX_DEVICES]`)
#define BL_MAX_DEVICES 64
static inline int bl_raise_lds_limit_once(const void* fn, int bytes, bool (&done)[BL_MAX_DEVICES]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BL_MAX_DEVICES) return BL_EINVAL;
  if (done[dev]) return BL_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return BL_EINVAL;
  done[dev] = true;
  return BL_OK;
}

// ---- counter-based dropout: identical to oracle/buglab_oracle.py::dropout_keep_mask -----------
__host__ __device__ static inline uint32_t bl_lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
struct bl_drop_dev {  // what kernels receive
  uint32_t key;
  uint32_t thresh;  // keep iff (hash >> 8) >= thresh ; thresh == 0 -> dropout off
  float scale;
};
static inline bl_drop_dev bl_make_drop(bl_dropout_t d) {
  bl_drop_dev r;
  r.key = bl_lowbias32(d.seed ^ (d.stream * 0x9E3779B9u));
  r.thresh = (d.p > 0.f) ? (uint32_t)(d.p * 16777216.0f) : 0u;
  r.scale = (d.p > 0.f) ? 1.0f / (1.0f - d.p) : 1.0f;
  return r;
}
__device__ __forceinline__ bool bl_keep(const bl_drop_dev& d, uint32_t idx) {
  return (bl_lowbias32(idx + d.key) >> 8) >= d.thresh;
}

// ---- activations ------------------------------------------------------------------------------
__device__ __forceinline__ float bl_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float bl_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float bl_act(int act, float x) {
  switch (act) {
    case BL_ACT_RELU: return x > 0.f ? x : 0.f;
    case BL_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
    case BL_ACT_TANH: return tanhf(x);
    case BL_ACT_GELU: return bl_gelu(x);
    default: return x;
  }
}
// derivative expressed through the OUTPUT y = act(z) (relu / sigmoid / tanh / none)
__device__ __forceinline__ float bl_act_grad_from_out(int act, float y) {
  switch (act) {
    case BL_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case BL_ACT_SIGMOID: return y * (1.f - y);
    case BL_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// ---- streaming accesses -------------------------------------------------------------------------------
// Results that are written once and read by a LATER kernel, and are far larger than the 32 MB of L2 (the [E, Dm] / [E, 2 Din] fp32 rows of
// the message GEMMs: 328 / 655 MB per launch), leave with the `nt` hint: as plain stores they are allocated in the XCD's L2 like
// anything else and push out the operand rows the kernel gathers.  Measured on the routed input-gradient GEMM at the c2 layer
// shape: 0.263 -> 0.211 ms per launch, and the segmented sums that read the rows next 0.157 -> 0.148 ms; c2 step 13.40 -> 12.85 ms
// (tools/experiments/nt_probe.sh, profiles/r06zze_nt_probe.log).
typedef float bl_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bl_store_streaming(float* p, const float4& v) {
  __builtin_nontemporal_store(__builtin_bit_cast(bl_f32x4, v), reinterpret_cast<bl_f32x4*>(p));
}

// ---- wave-level reductions (64 lanes) -----------------------------------------------------------
__device__ __forceinline__ float bl_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float bl_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- bf16x3 split of an fp32 value (csrc/bl_gemm_x6.hip): x = hi + mid + lo up to 2^-27 |x| ------
__device__ __forceinline__ uint16_t f2bf_rne(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ void split3(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
  h = f2bf_rne(x);
  if ((h & 0x7F80u) == 0x7F80u || x != x) {  // +-inf / NaN (or a value that rounds to inf): keep it in `hi` alone
    if (x != x) h = 0x7FC0u;                   // (inf - inf would put NaN into the other planes)
    m = l = 0;
    return;
  }
  const float r1 = x - bf2f(h);  // exact
  m = f2bf_rne(r1);
  const float r2 = r1 - bf2f(m);  // exact
  l = f2bf_rne(r2);
}

// ---- f16x2 split of an fp32 value (csrc/bl_gemm_h3.hip): x = hi + lo up to 2^-24 |x| while both planes are normal fp16 numbers
// (|x| >= 2^-3); below that the error is absolute, <= 2^-25.  The caller has multiplied x by the tensor's power-of-two scale.
// A finite value beyond fp16's range saturates at +-65504; +-inf / NaN: hi = x's fp16 image, lo = 0 (inf - inf would be NaN
// anyway: the product is NaN or inf either way, as in split3).
// sat is SET (never cleared) when a finite value was clamped: the packers count those events (bl_h3_saturation_events).
__device__ __forceinline__ void split2h(float x, uint16_t& h, uint16_t& l, bool& sat) {
  if (fabsf(x) > 65504.f && fabsf(x) <= 3.402823466e38f) {
    x = copysignf(65504.f, x);
    sat = true;
  }
  const _Float16 hh = (_Float16)x;  // round to nearest even
  h = __builtin_bit_cast(uint16_t, hh);
  if ((h & 0x7C00u) == 0x7C00u) {  // inf / NaN
    l = 0;
    return;
  }
  const float r = x - (float)hh;  // exact
  l = __builtin_bit_cast(uint16_t, (_Float16)r);
}

// Ordered flush (deterministic mode): workgroup number `turn` of a counter's sequence may add only after workgroup
// turn - 1 has left.  Workgroups are dispatched in increasing linear id and the callers number them so that a lower turn
// never has a higher id within its XCD's range, so the one being waited for is always running or already done; the spin
// is bounded anyway -- a wrong assumption must cost reproducibility, not hang the device.
__device__ __forceinline__ void bl_ordered_enter(unsigned* ctr, unsigned turn) {
  if (ctr == nullptr) return;
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != turn && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(16);
  }
  __syncthreads();
}
__device__ __forceinline__ void bl_ordered_leave(unsigned* ctr, unsigned turn) {
  if (ctr == nullptr) return;
  __threadfence();  // this workgroup's atomics have reached L2 before the next one is let in
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(ctr, turn + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

