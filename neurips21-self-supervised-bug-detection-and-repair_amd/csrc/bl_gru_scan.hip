// `seq-gru`: the time recurrence of one bidirectional GRU layer over a padded [B, L] minibatch as ONE launch per direction pair
// and pass -- the role torch.nn.GRU(bidirectional=True, batch_first=True) over a PackedSequence plays in the reference
// (/root/reference/buglab/models/seqmodel.py:119-126 construction, :385-392 call: pack_padded_sequence(lengths, enforce_sorted=False)
// -> GRU -> pad_packed_sequence, i.e. every sequence runs over its OWN length, the reverse direction starts at its last real token,
// positions >= length come back as zeros).
//
// Split of the work: the input projections of all time steps and both directions, gi = x W_ih + b_ih [B L, 2 x 3 Hh], are one MFMA
// row GEMM outside (and so are their gradients); what is left is sequential in t and tiny per step -- gh = h W_hh + b_hh (Hh x 3 Hh
// MACs per sequence) and the gate arithmetic.  One workgroup per (sequence, direction), 3 Hh threads: thread j keeps column j of
// W_hh (Hh floats) in registers for the whole scan, h lives in LDS and is read as broadcast float4s, gi is loaded two steps
// ahead of the arithmetic.  Exact fp32 FMAs (no split products needed: the recurrence is latency-bound, ~1 us per step).
// torch gate order [r | z | n]:  r = sig(gi_r + gh_r), z = sig(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h.
//
// Backward walks the steps in reverse processing order with W_hh's ROWS split over the threads (thread (k, part) holds
// W_hh[k, part Hh : (part + 1) Hh]): dh_{t-1} = d_gh W_hh^T + dh z.  It writes d_gi (-> input / W_ih / b_ih gradients through the row
// GEMM's backward) and d_gh per step; the recurrent weight gradient h_prev^T d_gh and the bias column sums are GEMM-shaped and run
// afterwards on the library's weight-gradient kernel (hip_ops._GruScan.backward) from the `saved` h_prev rows.
//
// saved (forward -> backward): the gates [2 directions][B L rows][4 Hh] = r, z, n, gh_n, then h_prev [2][B L][Hh] (a matrix of its own:
// it is the left operand of the weight-gradient GEMM).  Padded rows: h_prev = 0 (they enter that GEMM with d_gh = 0 and must not
// hold NaNs), the gates unwritten and unread.
#include "bl_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int HH>
__global__ __launch_bounds__(3 * HH) void gru_scan_fwd_kernel(const float* __restrict__ gi, int ld_gi, const float* __restrict__ w_hh,
                                                               const float* __restrict__ b_hh, const int* __restrict__ lens, int L,
                                                               long long R, float* __restrict__ out, int ld_out, float* __restrict__ saved) {
  __shared__ __attribute__((aligned(16))) float hs[HH];
  __shared__ float pre[3 * HH];
  __shared__ float gin[HH];
  const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x, gate = j / HH, u = j % HH;
  float w[HH];
#pragma unroll
  for (int k = 0; k < HH; ++k) w[k] = w_hh[((size_t)dir * HH + k) * 3 * HH + j];
  const float bias = b_hh[dir * 3 * HH + j];
  int len = lens[b];
  len = len < 0 ? 0 : (len > L ? L : len);
  const size_t row0 = (size_t)b * L;
  // padded positions: zeros in this direction's half of the output (pad_packed_sequence) and in the saved h_prev rows
  for (int i = j; i < (L - len) * HH; i += 3 * HH) {
    const size_t row = row0 + len + i / HH;
    out[row * ld_out + dir * HH + i % HH] = 0.f;
    if (saved) saved[(size_t)2 * R * 4 * HH + ((size_t)dir * R + row) * HH + i % HH] = 0.f;
  }
  if (j < HH) hs[j] = 0.f;
  __syncthreads();
  const float* gcol = gi + (size_t)dir * 3 * HH + j;
  // gi two steps ahead (a step is ~0.6 us: less than a round trip to HBM)
  float g_next = len > 0 ? gcol[(row0 + (dir ? len - 1 : 0)) * ld_gi] : 0.f;
  float g_next2 = len > 1 ? gcol[(row0 + (dir ? len - 2 : 1)) * ld_gi] : 0.f;
  for (int s = 0; s < len; ++s) {
    const int t = dir ? len - 1 - s : s;
    const float g = g_next;
    g_next = g_next2;
    if (s + 2 < len) g_next2 = gcol[(row0 + (dir ? t - 2 : t + 2)) * ld_gi];
    // four independent chains (a step is latency: one chain of HH dependent FMAs would be most of it)
    // (pairs: v_pk_fma_f32 retires two fp32 FMAs per lane and issue slot -- the 128 FMAs per thread are most of a step)
    f32x2 p0 = {bias, 0.f}, p1 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < HH; k += 4) {
      const float4 h4 = *reinterpret_cast<const float4*>(&hs[k]);  // same address in every lane: an LDS broadcast
      p0 = __builtin_elementwise_fma(f32x2{h4.x, h4.y}, f32x2{w[k], w[k + 1]}, p0);
      p1 = __builtin_elementwise_fma(f32x2{h4.z, h4.w}, f32x2{w[k + 2], w[k + 3]}, p1);
    }
    const float acc = (p0.x + p0.y) + (p1.x + p1.y);
    if (gate < 2) {
      pre[j] = 1.f / (1.f + expf(-(g + acc)));
    } else {
      pre[j] = acc;
      gin[u] = g;
    }
    __syncthreads();
    if (j < HH) {
      const float r = pre[u], z = pre[HH + u], ghn = pre[2 * HH + u], hp = hs[u];
      const float n = tanhf(gin[u] + r * ghn);
      const float hn = (1.f - z) * n + z * hp;
      hs[u] = hn;
      out[(row0 + t) * ld_out + dir * HH + u] = hn;
      if (saved) {
        float* sv = saved + ((size_t)dir * R + row0 + t) * 4 * HH;
        sv[u] = r;
        sv[HH + u] = z;
        sv[2 * HH + u] = n;
        sv[3 * HH + u] = ghn;
        saved[(size_t)2 * R * 4 * HH + ((size_t)dir * R + row0 + t) * HH + u] = hp;
      }
    }
    __syncthreads();
  }
}

template <int HH>
__global__ __launch_bounds__(3 * HH) void gru_scan_bwd_kernel(const float* __restrict__ g_out, int ld_g, const float* __restrict__ w_hh,
                                                               const float* __restrict__ saved, const int* __restrict__ lens, int L,
                                                               long long R, float* __restrict__ g_gi, int ld_ggi, float* __restrict__ g_gh) {
  __shared__ __attribute__((aligned(16))) float dgh[3 * HH];
  __shared__ float ps[3][HH];
  __shared__ float dhr[HH];
  const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x, part = j / HH, u = j % HH;
  float w[HH];  // W_hh[u, part HH + i]
#pragma unroll
  for (int i = 0; i < HH; ++i) w[i] = w_hh[((size_t)dir * HH + u) * 3 * HH + part * HH + i];
  int len = lens[b];
  len = len < 0 ? 0 : (len > L ? L : len);
  const size_t row0 = (size_t)b * L;
  for (int i = j; i < (L - len) * 3 * HH; i += 3 * HH) {  // padded positions carry no gradient
    const size_t row = row0 + len + i / (3 * HH);
    g_gi[row * ld_ggi + dir * 3 * HH + i % (3 * HH)] = 0.f;
    g_gh[((size_t)dir * R + row) * 3 * HH + i % (3 * HH)] = 0.f;
  }
  if (j < HH) dhr[j] = 0.f;
  __syncthreads();
  // this step's saved gates / incoming gradient are loaded one step ahead (a global round trip per step would be the step)
  float nx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define GRU_BWD_LOAD(t_)                                                                         \
  if (j < HH) {                                                                                  \
    const float* sv_ = saved + ((size_t)dir * R + row0 + (t_)) * 4 * HH;                         \
    nx[0] = sv_[u]; nx[1] = sv_[HH + u]; nx[2] = sv_[2 * HH + u]; nx[3] = sv_[3 * HH + u];       \
    nx[4] = saved[(size_t)2 * R * 4 * HH + ((size_t)dir * R + row0 + (t_)) * HH + u];            \
    nx[5] = g_out[(row0 + (t_)) * ld_g + dir * HH + u];                                          \
  }
  if (len > 0) GRU_BWD_LOAD(dir ? 0 : len - 1)
  for (int s = len - 1; s >= 0; --s) {
    const int t = dir ? len - 1 - s : s;
    float carry = 0.f;
    const float r = nx[0], z = nx[1], n = nx[2], ghn = nx[3], hp = nx[4], go = nx[5];
    if (s > 0) GRU_BWD_LOAD(dir ? len - s : s - 1)
    if (j < HH) {
      const float dh = go + dhr[u];
      const float dnp = dh * (1.f - z) * (1.f - n * n);
      const float drp = dnp * ghn * r * (1.f - r);
      const float dzp = dh * (hp - n) * z * (1.f - z);
      float* gi_row = g_gi + (row0 + t) * ld_ggi + dir * 3 * HH;
      gi_row[u] = drp;
      gi_row[HH + u] = dzp;
      gi_row[2 * HH + u] = dnp;
      float* gh_row = g_gh + ((size_t)dir * R + row0 + t) * 3 * HH;
      gh_row[u] = dgh[u] = drp;
      gh_row[HH + u] = dgh[HH + u] = dzp;
      gh_row[2 * HH + u] = dgh[2 * HH + u] = dnp * r;
      carry = dh * z;
    }
    __syncthreads();
    f32x2 p0 = {0.f, 0.f}, p1 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < HH; i += 4) {
      const float4 d4 = *reinterpret_cast<const float4*>(&dgh[part * HH + i]);
      p0 = __builtin_elementwise_fma(f32x2{d4.x, d4.y}, f32x2{w[i], w[i + 1]}, p0);
      p1 = __builtin_elementwise_fma(f32x2{d4.z, d4.w}, f32x2{w[i + 2], w[i + 3]}, p1);
    }
    ps[part][u] = (p0.x + p0.y) + (p1.x + p1.y);
    __syncthreads();
    if (j < HH) dhr[u] = ps[0][u] + ps[1][u] + ps[2][u] + carry;
    __syncthreads();
  }
}

int check_scan(const char* who, const void* a, const void* b, const void* c, const int32_t* lens, int32_t B, int32_t L, int32_t Hh) {
  BL_CHECK_ARG(a && b && c && lens, "%s: null pointer", who);
  BL_CHECK_ARG(B > 0 && L > 0, "%s: B and L must be positive", who);
  BL_CHECK_ARG(Hh == 32 || Hh == 64 || Hh == 128, "%s: hidden size per direction must be 32, 64 or 128 (got %d): thread j keeps a column of W_hh in registers", who, Hh);
  return BL_OK;
}

}  // namespace

extern "C" int64_t bl_gru_scan_saved_elems(int32_t B, int32_t L, int32_t Hh) { return (int64_t)2 * B * L * 5 * Hh; }

#define GRU_DISPATCH(Hh_, ...)                           \
  if ((Hh_) == 32) { constexpr int HH = 32; __VA_ARGS__; }   \
  else if ((Hh_) == 64) { constexpr int HH = 64; __VA_ARGS__; } \
  else { constexpr int HH = 128; __VA_ARGS__; }

extern "C" int bl_gru_scan_fwd(const float* gi, int32_t ld_gi, const float* w_hh, const float* b_hh, const int32_t* lens, int32_t B, int32_t L,
                               int32_t Hh, float* out, int32_t ld_out, float* saved, void* stream) {
  const char* who = "bl_gru_scan_fwd";
  if (int rc = check_scan(who, gi, w_hh, b_hh, lens, B, L, Hh)) return rc;
  BL_CHECK_ARG(out && ld_gi >= 6 * Hh && ld_out >= 2 * Hh, "%s: null output or leading dimensions below 6 Hh / 2 Hh", who);
  GRU_DISPATCH(Hh, hipLaunchKernelGGL((gru_scan_fwd_kernel<HH>), dim3(B, 2), dim3(3 * HH), 0, (hipStream_t)stream, gi, ld_gi, w_hh, b_hh, lens, L,
                                      (long long)B * L, out, ld_out, saved));
  BL_LAUNCH_CHECK(who);
  return BL_OK;
}

extern "C" int bl_gru_scan_bwd(const float* g_out, int32_t ld_g, const float* w_hh, const float* saved, const int32_t* lens, int32_t B, int32_t L,
                               int32_t Hh, float* g_gi, int32_t ld_ggi, float* g_gh, void* stream) {
  const char* who = "bl_gru_scan_bwd";
  if (int rc = check_scan(who, g_out, w_hh, saved, lens, B, L, Hh)) return rc;
  BL_CHECK_ARG(g_gi && g_gh && ld_g >= 2 * Hh && ld_ggi >= 6 * Hh, "%s: null output or leading dimensions below 2 Hh / 6 Hh", who);
  GRU_DISPATCH(Hh, hipLaunchKernelGGL((gru_scan_bwd_kernel<HH>), dim3(B, 2), dim3(3 * HH), 0, (hipStream_t)stream, g_out, ld_g, w_hh, saved, lens, L,
                                      (long long)B * L, g_gi, ld_ggi, g_gh));
  BL_LAUNCH_CHECK(who);
  return BL_OK;
}
