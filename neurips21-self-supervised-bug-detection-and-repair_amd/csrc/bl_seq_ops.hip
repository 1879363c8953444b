// Kernels of the `seq-great` / `seq-rat` relational-transformer block that are not GEMMs
// (reference buglab/models/layers/relational_multihead_attention.py, multihead_attention.py,
// relational_transformer.py).  The dense parts -- QKV / output / feed-forward projections, Q.K^T, P.V and
// their gradients -- run on the library's MFMA GEMMs (bl_gemm_rows / bl_gemm_wgrad, grouped by
// (sample, head)); here are the row-wise pieces around them:
//   * add + LayerNorm forward (the backward is bl_layernorm_bwd of bl_graph_ops.hip);
//   * the sparse edge terms of the attention scores ("relational" attention) and their gradients;
//   * masked softmax over the key axis, forward and backward;
//   * the sparse edge value biases of the `rat` variant.
// Layouts: q (pre-scaled by dk^-0.5), k, v, context and their gradients are [B, H, L, dk] (one contiguous
// [L, dk] matrix per (sample, head) = one GEMM group); scores / probabilities are [B * H * L, L].
// The edges of a minibatch arrive as a CSR over QUERY rows (sample, position): entry = (key position, code),
// code = 2 * edge_type + direction (0: the edge's source is the query, 1: its target is).  One wave per query
// row owns that row of every array it writes: no atomics, repeated edges accumulate in list order
// (index_put_(accumulate=True) of the reference, relational_multihead_attention.py:105-109).
#include "bl_common.h"

#define NEG_INF_F (-__builtin_huge_valf())

// ---- y = LayerNorm(x + r) ---------------------------------------------------------------------------
// one wave per row, D <= 1024; z = x + r is written when z_out != NULL (what backward needs)
template <int NV>
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, int nrows, int D,
                                                                float* __restrict__ z_out, float* __restrict__ y,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= nrows) return;
  float v[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int d = lane + 64 * j;
    float t = 0.f;
    if (d < D) {
      t = x[(size_t)row * D + d];
      if (r) t += r[(size_t)row * D + d];
      if (z_out) z_out[(size_t)row * D + d] = t;
    }
    v[j] = t;
    s += t;
  }
  const float mean = bl_wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int d = lane + 64 * j;
    if (d < D) { const float c = v[j] - mean; q += c * c; }
  }
  const float rstd = rsqrtf(bl_wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int d = lane + 64 * j;
    if (d < D) y[(size_t)row * D + d] = (v[j] - mean) * rstd * gamma[d] + beta[d];
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// ---- edge terms of the attention scores ---------------------------------------------------------------
// mode 0 (what `seq-great` runs, relational_multihead_attention.py:135-152): term = <bias[code][h, :], q[b, h, i, :]>
// mode 1 (scalar key bias, :119-134):                                         term = bias[code][h] * sum_d k[b, h, j, d]
// S[(b, h, i), j] += term for every entry (j, code) of query row (b, i).  H * dk <= 512.
template <int NV>
__global__ __launch_bounds__(256) void rel_bias_fwd_kernel(const int* __restrict__ row_ptr, const int* __restrict__ ekey,
                                                           const int* __restrict__ ecode, int B, int L, int H, int dk,
                                                           int mode, const float* __restrict__ qk,  // q (mode 0) or k (mode 1): [B, H, L, dk]
                                                           const float* __restrict__ bias_f, const float* __restrict__ bias_r,
                                                           float* __restrict__ S) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // (b, i)
  if (row >= B * L) return;
  const int beg = row_ptr[row], end = row_ptr[row + 1];
  if (beg == end) return;
  const int b = row / L, i = row - b * L;
  const int HD = H * dk;
  float qv[NV];  // this lane's elements (h, d) = e / dk, e % dk for e = lane + 64 j of the query row (mode 0)
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int e = lane + 64 * j;
    qv[j] = (mode == 0 && e < HD) ? qk[(((size_t)b * H + e / dk) * L + i) * dk + e % dk] : 0.f;
  }
  for (int p = beg; p < end; ++p) {
    const int key = ekey[p], code = ecode[p];
    const float* __restrict__ bt = (code & 1) ? bias_r : bias_f;
    const int t = code >> 1;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int e = lane + 64 * j;
      float part = 0.f;
      if (e < HD) {
        const int h = e / dk;
        if (mode == 0) part = bt[(size_t)t * HD + e] * qv[j];
        else part = bt[(size_t)t * H + h] * qk[(((size_t)b * H + h) * L + key) * dk + e % dk];
      }
      // sum over the dk lanes of one head: dk is a power of two <= 64, a head's elements are dk consecutive lanes
      for (int o = dk >> 1; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
      if (e < HD && (e % dk) == 0) {
        const int h = e / dk;
        S[(((size_t)b * H + h) * L + i) * L + key] += part;  // this wave owns row (b, h, i): plain read-modify-write
      }
    }
  }
}

// gradients of the edge terms.  dS [B*H*L, L].  Row-owned outputs: g_q (mode 0; ADDED to what is there).
// Bias-table gradients go through a per-block LDS table and are flushed with one atomic per element and block.
// mode 1 also has a key-side gradient g_k[b, h, key, :] += dS * bias (atomics: keys belong to other rows).
template <int NV>
__global__ __launch_bounds__(256) void rel_bias_bwd_kernel(const int* __restrict__ row_ptr, const int* __restrict__ ekey,
                                                           const int* __restrict__ ecode, int B, int L, int H, int dk,
                                                           int mode, int T, const float* __restrict__ qk,
                                                           const float* __restrict__ bias_f, const float* __restrict__ bias_r,
                                                           const float* __restrict__ dS, float* __restrict__ g_q,
                                                           float* __restrict__ g_k, float* __restrict__ g_bias_f,
                                                           float* __restrict__ g_bias_r) {
  extern __shared__ float tab[];  // [2 T][W], W = H * dk (mode 0) or H (mode 1)
  const int HD = H * dk, W = mode == 0 ? HD : H;
  for (int x = threadIdx.x; x < 2 * T * W; x += blockDim.x) tab[x] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int nrows = B * L;
  for (int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < nrows; row += gridDim.x * (blockDim.x >> 6)) {
    const int beg = row_ptr[row], end = row_ptr[row + 1];
    if (beg == end) continue;
    const int b = row / L, i = row - b * L;
    float qv[NV], gq[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int e = lane + 64 * j;
      qv[j] = (mode == 0 && e < HD) ? qk[(((size_t)b * H + e / dk) * L + i) * dk + e % dk] : 0.f;
      gq[j] = 0.f;
    }
    for (int p = beg; p < end; ++p) {
      const int key = ekey[p], code = ecode[p];
      const float* __restrict__ bt = (code & 1) ? bias_r : bias_f;
      const int t = code >> 1;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int e = lane + 64 * j;
        const bool ok = e < HD;
        const int h = ok ? e / dk : 0;
        const float g = ok ? dS[(((size_t)b * H + h) * L + i) * L + key] : 0.f;
        if (mode == 0) {
          if (ok) {
            gq[j] += g * bt[(size_t)t * HD + e];
            atomicAdd(&tab[(size_t)code * W + e], g * qv[j]);
          }
        } else {
          float ks = ok ? qk[(((size_t)b * H + h) * L + key) * dk + e % dk] : 0.f;
          for (int o = dk >> 1; o > 0; o >>= 1) ks += __shfl_xor(ks, o, 64);  // every lane takes part
          if (ok) {
            if ((e % dk) == 0) atomicAdd(&tab[(size_t)code * W + h], g * ks);
            unsafeAtomicAdd(&g_k[(((size_t)b * H + h) * L + key) * dk + e % dk], g * bt[(size_t)t * H + h]);
          }
        }
      }
    }
    if (mode == 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int e = lane + 64 * j;
        if (e < HD) g_q[(((size_t)b * H + e / dk) * L + i) * dk + e % dk] += gq[j];
      }
    }
  }
  __syncthreads();
  for (int x = threadIdx.x; x < 2 * T * W; x += blockDim.x) {
    const float v = tab[x];
    if (v != 0.f) {
      const int code = x / W, w = x - code * W;
      unsafeAtomicAdd(((code & 1) ? g_bias_r : g_bias_f) + (size_t)(code >> 1) * W + w, v);
    }
  }
}

// ---- masked softmax over keys ------------------------------------------------------------------------
// rows r = (b, h, i) of S [R, L]; keys >= len[b] are padding (score -inf, probability 0).  In place.  L <= 1024.
// (+ Pd: the probabilities after nn.Dropout, element i = row * L + k of the counter-hash mask -- what bl_dropout_inplace
// would make of a copy of P, without the copy and the second pass)
template <int NV>
__global__ __launch_bounds__(256) void masked_softmax_fwd_kernel(float* __restrict__ S, int R, int L, int rows_per_sample,
                                                                 const int* __restrict__ lens, bl_drop_dev drop,
                                                                 float* __restrict__ Pd) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const int n = lens[row / rows_per_sample];
  float* __restrict__ s = S + (size_t)row * L;
  float v[NV];
  float m = NEG_INF_F;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = lane + 64 * j;
    v[j] = k < n ? s[k] : NEG_INF_F;
    m = fmaxf(m, v[j]);
  }
  m = bl_wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j] = (lane + 64 * j) < n ? expf(v[j] - m) : 0.f;
    sum += v[j];
  }
  sum = bl_wave_sum(sum);
  const float inv = 1.0f / sum;  // n == 0 (a sample without tokens) gives 0/0 like torch.softmax of an all -inf row
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = lane + 64 * j;
    if (k < L) {
      const float pr = v[j] * inv;
      s[k] = pr;
      if (Pd) Pd[(size_t)row * L + k] = bl_keep(drop, (uint32_t)row * (uint32_t)L + (uint32_t)k) ? pr * drop.scale : 0.f;
    }
  }
}

// dS = P * (dP - sum_k P dP), written over dP; DROP: dP arrives as the gradient of the DROPPED probabilities and goes
// through the dropout mask first (bl_dropout_inplace on dP without the extra pass)
template <int NV, bool DROP>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, int R, int L,
                                                          bl_drop_dev drop) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* __restrict__ p = P + (size_t)row * L;
  float* __restrict__ g = dP + (size_t)row * L;
  float pv[NV], gv[NV];
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = lane + 64 * j;
    pv[j] = k < L ? p[k] : 0.f;
    gv[j] = k < L ? g[k] : 0.f;
    if (DROP) gv[j] = bl_keep(drop, (uint32_t)row * (uint32_t)L + (uint32_t)k) ? gv[j] * drop.scale : 0.f;
    dot += pv[j] * gv[j];
  }
  dot = bl_wave_sum(dot);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int k = lane + 64 * j;
    if (k < L) g[k] = pv[j] * (gv[j] - dot);
  }
}

// ---- edge value biases (`rat`, relational_multihead_attention.py:155-178) ------------------------------------
// ctx[b, h, i, :] += P[(b, h, i), key] * vb[code][h, :] for every entry of row (b, i)
template <int NV>
__global__ __launch_bounds__(256) void value_bias_fwd_kernel(const int* __restrict__ row_ptr, const int* __restrict__ ekey,
                                                             const int* __restrict__ ecode, int B, int L, int H, int dk,
                                                             const float* __restrict__ P, const float* __restrict__ vb_f,
                                                             const float* __restrict__ vb_r, float* __restrict__ ctx) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B * L) return;
  const int beg = row_ptr[row], end = row_ptr[row + 1];
  if (beg == end) return;
  const int b = row / L, i = row - b * L, HD = H * dk;
  float acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.f;
  for (int p = beg; p < end; ++p) {
    const int key = ekey[p], code = ecode[p];
    const float* __restrict__ vt = ((code & 1) ? vb_r : vb_f) + (size_t)(code >> 1) * HD;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int e = lane + 64 * j;
      if (e < HD) acc[j] += P[(((size_t)b * H + e / dk) * L + i) * L + key] * vt[e];
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int e = lane + 64 * j;
    if (e < HD) ctx[(((size_t)b * H + e / dk) * L + i) * dk + e % dk] += acc[j];
  }
}

// dP[(b, h, i), key] += <g_ctx[b, h, i, :], vb[code][h, :]>;  g_vb[code][h, :] += P * g_ctx
template <int NV>
__global__ __launch_bounds__(256) void value_bias_bwd_kernel(const int* __restrict__ row_ptr, const int* __restrict__ ekey,
                                                             const int* __restrict__ ecode, int B, int L, int H, int dk, int T,
                                                             const float* __restrict__ P, const float* __restrict__ g_ctx,
                                                             const float* __restrict__ vb_f, const float* __restrict__ vb_r,
                                                             float* __restrict__ dP, float* __restrict__ g_vb_f,
                                                             float* __restrict__ g_vb_r) {
  extern __shared__ float tab[];  // [2 T][H dk]
  const int HD = H * dk;
  for (int x = threadIdx.x; x < 2 * T * HD; x += blockDim.x) tab[x] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int nrows = B * L;
  for (int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < nrows; row += gridDim.x * (blockDim.x >> 6)) {
    const int beg = row_ptr[row], end = row_ptr[row + 1];
    if (beg == end) continue;
    const int b = row / L, i = row - b * L;
    float gc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int e = lane + 64 * j;
      gc[j] = e < HD ? g_ctx[(((size_t)b * H + e / dk) * L + i) * dk + e % dk] : 0.f;
    }
    for (int p = beg; p < end; ++p) {
      const int key = ekey[p], code = ecode[p];
      const float* __restrict__ vt = ((code & 1) ? vb_r : vb_f) + (size_t)(code >> 1) * HD;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int e = lane + 64 * j;
        float part = 0.f;
        if (e < HD) {
          part = gc[j] * vt[e];
          atomicAdd(&tab[(size_t)code * HD + e], P[(((size_t)b * H + e / dk) * L + i) * L + key] * gc[j]);
        }
        for (int o = dk >> 1; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (e < HD && (e % dk) == 0) dP[(((size_t)b * H + e / dk) * L + i) * L + key] += part;
      }
    }
  }
  __syncthreads();
  for (int x = threadIdx.x; x < 2 * T * HD; x += blockDim.x) {
    const float v = tab[x];
    if (v != 0.f) {
      const int code = x / HD, w = x - code * HD;
      unsafeAtomicAdd(((code & 1) ? g_vb_r : g_vb_f) + (size_t)(code >> 1) * HD + w, v);
    }
  }
}

// ---- elementwise counter-hash dropout, in place (attention probabilities, embeddings) --------------------------
__global__ __launch_bounds__(256) void dropout_inplace_kernel(float* __restrict__ x, long long n, bl_drop_dev drop) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = bl_keep(drop, (uint32_t)i) ? x[i] * drop.scale : 0.f;
}

// ================================================================================================
#define SEQ_DISPATCH(D_, call_)                 \
  if ((D_) <= 64) { constexpr int NV = 1; call_; }       \
  else if ((D_) <= 128) { constexpr int NV = 2; call_; } \
  else if ((D_) <= 256) { constexpr int NV = 4; call_; } \
  else if ((D_) <= 512) { constexpr int NV = 8; call_; } \
  else { constexpr int NV = 16; call_; }

extern "C" int bl_add_layernorm_fwd(const float* x, const float* r, const float* gamma, const float* beta, float eps,
                                    int32_t nrows, int32_t D, float* z_out, float* y, float* mean, float* rstd,
                                    void* stream) {
  if (nrows == 0) return BL_OK;
  BL_CHECK_ARG(x && gamma && beta && y && mean && rstd && D > 0 && D <= 1024, "bl_add_layernorm_fwd: null pointer or D outside 1..1024");
  hipStream_t st = (hipStream_t)stream;
  SEQ_DISPATCH(D, hipLaunchKernelGGL((add_layernorm_fwd_kernel<NV>), dim3((nrows + 3) / 4), dim3(256), 0, st, x, r, gamma, beta, eps,
                                     nrows, D, z_out, y, mean, rstd))
  BL_LAUNCH_CHECK("bl_add_layernorm_fwd");
  return BL_OK;
}

static int check_rel(const char* who, const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode, int B, int L, int H, int dk) {
  BL_CHECK_ARG(row_ptr && ekey && ecode, "%s: null edge CSR", who);
  BL_CHECK_ARG(B > 0 && L > 0 && H > 0 && (dk == 8 || dk == 16 || dk == 32 || dk == 64) && H * dk <= 512,
               "%s: head dimension must be 8/16/32/64 and heads x dimension <= 512", who);
  return BL_OK;
}

extern "C" int bl_rel_attn_bias_fwd(const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L,
                                    int32_t H, int32_t dk, int32_t mode, const float* qk, const float* bias_f,
                                    const float* bias_r, float* S, void* stream) {
  int rc = check_rel("bl_rel_attn_bias_fwd", row_ptr, ekey, ecode, B, L, H, dk);
  if (rc != BL_OK) return rc;
  BL_CHECK_ARG(qk && bias_f && bias_r && S && (mode == 0 || mode == 1), "bl_rel_attn_bias_fwd: null pointer or bad mode");
  hipStream_t st = (hipStream_t)stream;
  SEQ_DISPATCH(H * dk, hipLaunchKernelGGL((rel_bias_fwd_kernel<NV>), dim3((B * L + 3) / 4), dim3(256), 0, st, row_ptr, ekey, ecode, B,
                                          L, H, dk, mode, qk, bias_f, bias_r, S))
  BL_LAUNCH_CHECK("bl_rel_attn_bias_fwd");
  return BL_OK;
}

extern "C" int bl_rel_attn_bias_bwd(const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L,
                                    int32_t H, int32_t dk, int32_t mode, int32_t T, const float* qk, const float* bias_f,
                                    const float* bias_r, const float* dS, float* g_q, float* g_k, float* g_bias_f,
                                    float* g_bias_r, void* stream) {
  int rc = check_rel("bl_rel_attn_bias_bwd", row_ptr, ekey, ecode, B, L, H, dk);
  if (rc != BL_OK) return rc;
  BL_CHECK_ARG(qk && bias_f && bias_r && dS && g_bias_f && g_bias_r && (mode == 0 ? g_q != nullptr : g_k != nullptr) && T > 0,
               "bl_rel_attn_bias_bwd: null pointer");
  const size_t lds = (size_t)2 * T * (mode == 0 ? H * dk : H) * sizeof(float);
  BL_CHECK_ARG(lds <= 64 * 1024, "bl_rel_attn_bias_bwd: too many edge types for the LDS table (%d)", T);
  const int blocks = min((B * L + 3) / 4, 1024);
  hipStream_t st = (hipStream_t)stream;
  SEQ_DISPATCH(H * dk, hipLaunchKernelGGL((rel_bias_bwd_kernel<NV>), dim3(blocks), dim3(256), lds, st, row_ptr, ekey, ecode, B, L, H, dk,
                                          mode, T, qk, bias_f, bias_r, dS, g_q, g_k, g_bias_f, g_bias_r))
  BL_LAUNCH_CHECK("bl_rel_attn_bias_bwd");
  return BL_OK;
}

extern "C" int bl_masked_softmax_fwd(float* S, int32_t R, int32_t L, int32_t rows_per_sample, const int32_t* lens, void* stream) {
  if (R == 0) return BL_OK;
  BL_CHECK_ARG(S && lens && L > 0 && L <= 1024 && rows_per_sample > 0, "bl_masked_softmax_fwd: null pointer or L outside 1..1024");
  hipStream_t st = (hipStream_t)stream;
  bl_dropout_t none = {0.f, 0u, 0u};
  SEQ_DISPATCH(L, hipLaunchKernelGGL((masked_softmax_fwd_kernel<NV>), dim3((R + 3) / 4), dim3(256), 0, st, S, R, L, rows_per_sample, lens,
                                     bl_make_drop(none), (float*)nullptr))
  BL_LAUNCH_CHECK("bl_masked_softmax_fwd");
  return BL_OK;
}

extern "C" int bl_masked_softmax_dropout_fwd(float* S, int32_t R, int32_t L, int32_t rows_per_sample, const int32_t* lens,
                                             bl_dropout_t drop, float* Pd, void* stream) {
  if (R == 0) return BL_OK;
  BL_CHECK_ARG(S && lens && L > 0 && L <= 1024 && rows_per_sample > 0, "bl_masked_softmax_dropout_fwd: null pointer or L outside 1..1024");
  BL_CHECK_ARG(drop.p <= 0.f || (Pd && Pd != S && (long long)R * L < (1ll << 32)),
               "bl_masked_softmax_dropout_fwd: dropout needs a second output and fewer than 2^32 elements");
  hipStream_t st = (hipStream_t)stream;
  float* pd = drop.p > 0.f ? Pd : nullptr;
  SEQ_DISPATCH(L, hipLaunchKernelGGL((masked_softmax_fwd_kernel<NV>), dim3((R + 3) / 4), dim3(256), 0, st, S, R, L, rows_per_sample, lens,
                                     bl_make_drop(drop), pd))
  BL_LAUNCH_CHECK("bl_masked_softmax_dropout_fwd");
  return BL_OK;
}

extern "C" int bl_softmax_bwd(const float* P, float* dP, int32_t R, int32_t L, void* stream) {
  bl_dropout_t none = {0.f, 0u, 0u};
  return bl_softmax_dropout_bwd(P, dP, R, L, none, stream);
}

extern "C" int bl_softmax_dropout_bwd(const float* P, float* dP, int32_t R, int32_t L, bl_dropout_t drop, void* stream) {
  if (R == 0) return BL_OK;
  BL_CHECK_ARG(P && dP && L > 0 && L <= 1024, "bl_softmax_bwd: null pointer or L outside 1..1024");
  BL_CHECK_ARG(drop.p <= 0.f || (long long)R * L < (1ll << 32), "bl_softmax_dropout_bwd: more than 2^32 elements");
  hipStream_t st = (hipStream_t)stream;
  if (drop.p > 0.f) {
    SEQ_DISPATCH(L, hipLaunchKernelGGL((softmax_bwd_kernel<NV, true>), dim3((R + 3) / 4), dim3(256), 0, st, P, dP, R, L, bl_make_drop(drop)))
  } else {
    SEQ_DISPATCH(L, hipLaunchKernelGGL((softmax_bwd_kernel<NV, false>), dim3((R + 3) / 4), dim3(256), 0, st, P, dP, R, L, bl_make_drop(drop)))
  }
  BL_LAUNCH_CHECK("bl_softmax_bwd");
  return BL_OK;
}

extern "C" int bl_rel_value_bias_fwd(const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L,
                                     int32_t H, int32_t dk, const float* P, const float* vb_f, const float* vb_r, float* ctx,
                                     void* stream) {
  int rc = check_rel("bl_rel_value_bias_fwd", row_ptr, ekey, ecode, B, L, H, dk);
  if (rc != BL_OK) return rc;
  BL_CHECK_ARG(P && vb_f && vb_r && ctx, "bl_rel_value_bias_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  SEQ_DISPATCH(H * dk, hipLaunchKernelGGL((value_bias_fwd_kernel<NV>), dim3((B * L + 3) / 4), dim3(256), 0, st, row_ptr, ekey, ecode, B,
                                          L, H, dk, P, vb_f, vb_r, ctx))
  BL_LAUNCH_CHECK("bl_rel_value_bias_fwd");
  return BL_OK;
}

extern "C" int bl_rel_value_bias_bwd(const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L,
                                     int32_t H, int32_t dk, int32_t T, const float* P, const float* g_ctx, const float* vb_f,
                                     const float* vb_r, float* dP, float* g_vb_f, float* g_vb_r, void* stream) {
  int rc = check_rel("bl_rel_value_bias_bwd", row_ptr, ekey, ecode, B, L, H, dk);
  if (rc != BL_OK) return rc;
  BL_CHECK_ARG(P && g_ctx && vb_f && vb_r && dP && g_vb_f && g_vb_r && T > 0, "bl_rel_value_bias_bwd: null pointer");
  const size_t lds = (size_t)2 * T * H * dk * sizeof(float);
  BL_CHECK_ARG(lds <= 64 * 1024, "bl_rel_value_bias_bwd: too many edge types for the LDS table (%d)", T);
  const int blocks = min((B * L + 3) / 4, 1024);
  hipStream_t st = (hipStream_t)stream;
  SEQ_DISPATCH(H * dk, hipLaunchKernelGGL((value_bias_bwd_kernel<NV>), dim3(blocks), dim3(256), lds, st, row_ptr, ekey, ecode, B, L, H,
                                          dk, T, P, g_ctx, vb_f, vb_r, dP, g_vb_f, g_vb_r))
  BL_LAUNCH_CHECK("bl_rel_value_bias_bwd");
  return BL_OK;
}

/* x[i] = keep(i) ? x[i] / (1 - p) : 0 with the library's counter-hash mask: the same call is the forward and the
 * backward of a dropout layer (nn.Dropout on the attention probabilities, multihead_attention.py:72). */
extern "C" int bl_dropout_inplace(float* x, int64_t n, bl_dropout_t drop, void* stream) {
  if (n == 0 || drop.p <= 0.f) return BL_OK;
  BL_CHECK_ARG(x && n < (1ll << 32), "bl_dropout_inplace: null pointer or more than 2^32 elements");
  hipLaunchKernelGGL(dropout_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)n,
                     bl_make_drop(drop));
  BL_LAUNCH_CHECK("bl_dropout_inplace");
  return BL_OK;
}
