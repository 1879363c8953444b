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

// A [B, H, L, dk] tensor wherever its rows live (bl_head_view_t): element (b, h, l, d) at p[b sb + h sh + l sl + d].
// Contiguous [B, H, L, dk]: (H L dk, L dk, dk); the q / k / v column blocks of the QKV projection's own output [B L, H 3 dk]
// (per head [q | k | v], multihead_attention.py:46-50): p = qkv + which dk, (L 3 H dk, 3 dk, 3 H dk) -- no permuted copies.
struct HeadView {
  float* p;
  long long sb;
  int sh, sl;
};
__device__ __forceinline__ float* hv_mat(const HeadView& v, int b, int h) { return v.p + (size_t)b * v.sb + (size_t)h * v.sh; }
// ... and the same tensor written in bl_pack_bf16x3's form as columns of a [B L, 3 W] packed matrix (bl_packed_head_view_t):
// element (b, h, l, d), plane pl at p[((b L + l) 3 + pl) W + col0 + h hs + d] -- the attention context as the output projection's
// operand, the gradients of q / k / v as the operand of the QKV projection's gradient GEMMs
struct PackedHeadView {
  uint16_t* p;
  int W, col0, hs;
};
__device__ __forceinline__ void phv_store4(const PackedHeadView& v, size_t grow, int h, int n, float a, float b, float c, float d) {
  uint16_t hh[4], mm[4], ll[4];
  split3(a, hh[0], mm[0], ll[0]);
  split3(b, hh[1], mm[1], ll[1]);
  split3(c, hh[2], mm[2], ll[2]);
  split3(d, hh[3], mm[3], ll[3]);
  uint2* o = reinterpret_cast<uint2*>(v.p + grow * 3 * v.W + v.col0 + h * v.hs + n);
  o[0] = make_uint2((uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16));
  o[v.W >> 2] = make_uint2((uint32_t)mm[0] | ((uint32_t)mm[1] << 16), (uint32_t)mm[2] | ((uint32_t)mm[3] << 16));
  o[v.W >> 1] = make_uint2((uint32_t)ll[0] | ((uint32_t)ll[1] << 16), (uint32_t)ll[2] | ((uint32_t)ll[3] << 16));
}
static inline HeadView hv_contiguous(const float* p, int H, int L, int dk) {
  HeadView v = {const_cast<float*>(p), (long long)H * L * dk, L * dk, dk};
  return v;
}
static inline HeadView hv_from(const bl_head_view_t* a) {
  HeadView v = {a->p, (long long)a->sb, a->sh, a->sl};
  return v;
}

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

// the same with a lane owning 4 consecutive channels (d = 4 lane + 256 j; D % 4 == 0) and, optionally, the result also in
// bl_pack_bf16x3's form y_packed [nrows][3 D] -- the operand of the next Linear's bf16x6 GEMM without a packing pass
// (csrc/bl_great_layer.hip)
template <int NV4>
__global__ __launch_bounds__(256) void add_layernorm_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float eps, int nrows, int D, float* __restrict__ z_out,
                                                                 float* __restrict__ y, float* __restrict__ mean_out,
                                                                 float* __restrict__ rstd_out, uint2* __restrict__ y_packed) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= nrows) return;
  float4 v[NV4];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV4; ++j) {
    const int d = 4 * lane + 256 * j;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d < D) {
      t = *reinterpret_cast<const float4*>(x + (size_t)row * D + d);
      if (r) {
        const float4 rv = *reinterpret_cast<const float4*>(r + (size_t)row * D + d);
        t.x += rv.x; t.y += rv.y; t.z += rv.z; t.w += rv.w;
      }
      if (z_out) *reinterpret_cast<float4*>(z_out + (size_t)row * D + d) = t;
    }
    v[j] = t;
    s += (t.x + t.y) + (t.z + t.w);
  }
  const float mean = bl_wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV4; ++j) {
    if (4 * lane + 256 * j < D) {
      const float c0 = v[j].x - mean, c1 = v[j].y - mean, c2 = v[j].z - mean, c3 = v[j].w - mean;
      q += (c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3);
    }
  }
  const float rstd = rsqrtf(bl_wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int j = 0; j < NV4; ++j) {
    const int d = 4 * lane + 256 * j;
    if (d < D) {
      const float4 gm = *reinterpret_cast<const float4*>(gamma + d), bt = *reinterpret_cast<const float4*>(beta + d);
      float4 o;
      o.x = (v[j].x - mean) * rstd * gm.x + bt.x;
      o.y = (v[j].y - mean) * rstd * gm.y + bt.y;
      o.z = (v[j].z - mean) * rstd * gm.z + bt.z;
      o.w = (v[j].w - mean) * rstd * gm.w + bt.w;
      *reinterpret_cast<float4*>(y + (size_t)row * D + d) = o;
      if (y_packed) {
        uint16_t h[4], m[4], l[4];
        split3(o.x, h[0], m[0], l[0]);
        split3(o.y, h[1], m[1], l[1]);
        split3(o.z, h[2], m[2], l[2]);
        split3(o.w, h[3], m[3], l[3]);
        uint2* op = y_packed + (((size_t)row * 3 * D + d) >> 2);
        op[0] = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
        op[D >> 2] = make_uint2((uint32_t)m[0] | ((uint32_t)m[1] << 16), (uint32_t)m[2] | ((uint32_t)m[3] << 16));
        op[D >> 1] = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
      }
    }
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

// ---- attention probabilities of `seq-great` in one kernel ---------------------------------------------------------
// P = softmax_keys(q.k^T + edge terms (mode 0), padding masked), Pd = dropout(P) for a block of query rows of one
// (sample, head): Q.K^T on the vector units (dk = 32: the grouped MFMA GEMM ran one k-step per tile and wrote a [G L, L]
// score matrix that the edge-term kernel, the softmax and the dropout each read and rewrote -- four passes over 268 MB per
// layer at BASELINE configs[4]; here the scores live in registers and P / Pd are written once).
// A 1024-thread workgroup keeps K^T of its (sample, head) in LDS ([dk][Lp + 4] fp32) and the head's edge-bias vectors
// ([2T][dk + 1]); a wave takes four query rows at a time (every K^T read feeds four rows), a lane owns the keys
// {256 c + 4 lane + x}: 16-byte LDS reads and 16-byte stores of P, one row = contiguous kilobytes.
// Same arithmetic as the separate kernels except the order of the dk products of a score (sequential FMAs here).
#define ATT_ROWS 4      // query rows per wave and step
#define ATT_WG_ROWS 128 // query rows per workgroup
#define ATT_ITERS (ATT_WG_ROWS / (16 * ATT_ROWS))  // steps per wave
template <int DK, int KC>  // KC = 16-byte key chunks per lane: Lp = 256 KC >= L
__global__ __launch_bounds__(1024) void attn_probs_fwd_kernel(const HeadView q, const float q_scale, const HeadView k,
                                                              const int* __restrict__ row_ptr, const int* __restrict__ ekey,
                                                              const int* __restrict__ ecode, int L, int H, int T,
                                                              const float* __restrict__ bias_f, const float* __restrict__ bias_r,
                                                              const int* __restrict__ lens, bl_drop_dev drop,
                                                              float* __restrict__ P, float* __restrict__ Pd) {
  constexpr int LP = 256 * KC, LS = LP + 4, NS = 4 * KC;
  extern __shared__ __attribute__((aligned(16))) float att_lds[];
  float* __restrict__ Kt = att_lds;              // [DK][LS]
  float* __restrict__ bl = att_lds + DK * LS;    // [2 T][DK + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, b = g / H, h = g - b * H, HD = H * DK;
  const int rb = blockIdx.y * ATT_WG_ROWS, re = min(rb + ATT_WG_ROWS, L);
  // the CSR bounds of this wave's rows, requested before the staging of K^T: lane r <= ATT_ROWS of rpv[it] = row_ptr[b L + i0 + r]
  int rpv[ATT_ITERS];
#pragma unroll
  for (int it = 0; it < ATT_ITERS; ++it)
    rpv[it] = (row_ptr && lane <= ATT_ROWS) ? row_ptr[b * L + min(rb + (16 * it + wave) * ATT_ROWS + lane, L)] : 0;
  const float* __restrict__ qg = hv_mat(q, b, h);
  {
    const float* __restrict__ kg = hv_mat(k, b, h);
    for (int x = tid; x < LP * (DK / 4); x += 1024) {  // 16 bytes per thread: a row of the head is DK / 4 of them
      const int j = x / (DK / 4), d = 4 * (x - j * (DK / 4));
      const float4 kv4 = j < L ? *reinterpret_cast<const float4*>(kg + (size_t)j * k.sl + d) : make_float4(0.f, 0.f, 0.f, 0.f);
      Kt[d * LS + j] = kv4.x; Kt[(d + 1) * LS + j] = kv4.y; Kt[(d + 2) * LS + j] = kv4.z; Kt[(d + 3) * LS + j] = kv4.w;
    }
    for (int x = tid; x < 2 * T * DK; x += 1024) {
      const int c = x / DK, d = x - c * DK;
      bl[c * (DK + 1) + d] = ((c & 1) ? bias_r : bias_f)[(size_t)(c >> 1) * HD + h * DK + d];
    }
  }
  __syncthreads();
  const int n = lens[b];
#pragma unroll
  for (int it = 0; it < ATT_ITERS; ++it) {
    const int i0 = rb + (16 * it + wave) * ATT_ROWS;
    if (i0 >= re) break;
    float qv[ATT_ROWS];  // lane d < DK: q[i0 + r][d]
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) qv[r] = lane < DK ? qg[(size_t)min(i0 + r, L - 1) * q.sl + lane] * q_scale : 0.f;
    // the rows' edge entries, up to 64 each, one per lane: requested here, used after the score loop
    int ebeg[ATT_ROWS], ecnt[ATT_ROWS], ek[ATT_ROWS], ec[ATT_ROWS];
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) {
      ebeg[r] = __builtin_amdgcn_readlane(rpv[it], r);
      ecnt[r] = i0 + r < re ? __builtin_amdgcn_readlane(rpv[it], r + 1) - ebeg[r] : 0;
      ek[r] = lane < ecnt[r] ? ekey[ebeg[r] + lane] : 0;
      ec[r] = lane < ecnt[r] ? ecode[ebeg[r] + lane] : 0;
    }
    float s[ATT_ROWS][NS];
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r)
#pragma unroll
      for (int x = 0; x < NS; ++x) s[r][x] = 0.f;
#pragma unroll 2  // (fully unrolled, hipcc hoists every K^T read and query broadcast to the top and spills)
    for (int d = 0; d < DK; ++d) {
      float4 kv[KC];
#pragma unroll
      for (int c = 0; c < KC; ++c) kv[c] = *reinterpret_cast<const float4*>(Kt + d * LS + 256 * c + 4 * lane);
#pragma unroll
      for (int r = 0; r < ATT_ROWS; ++r) {
        const float qs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qv[r]), d));
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          s[r][4 * c + 0] = fmaf(qs, kv[c].x, s[r][4 * c + 0]);
          s[r][4 * c + 1] = fmaf(qs, kv[c].y, s[r][4 * c + 1]);
          s[r][4 * c + 2] = fmaf(qs, kv[c].z, s[r][4 * c + 2]);
          s[r][4 * c + 3] = fmaf(qs, kv[c].w, s[r][4 * c + 3]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) {
      const int i = i0 + r;
      if (i >= re) break;  // wave-uniform
      if (ecnt[r] > 0) {  // edge terms: <bias[code][h, :], q[i, :]> added at the entry's key, entries in list order
        float term = 0.f;  // lane c < 2 T: the term of code c
        if (lane < 2 * T) {
#pragma unroll 4
          for (int d = 0; d < DK; ++d)
            term += bl[lane * (DK + 1) + d] * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qv[r]), d));
        }
        for (int p = 0; p < ecnt[r]; ++p) {  // everything about an entry is wave-uniform: scalar registers and branches
          int key, code;
          if (p < 64) {
            key = __builtin_amdgcn_readlane(ek[r], p);
            code = __builtin_amdgcn_readlane(ec[r], p);
          } else {
            key = __builtin_amdgcn_readfirstlane(ekey[ebeg[r] + p]);
            code = __builtin_amdgcn_readfirstlane(ecode[ebeg[r] + p]);
          }
          const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, term), code));
          const int owner = (key & 255) >> 2, slot = ((key >> 8) << 2) | (key & 3);
          const float tl = lane == owner ? t : 0.f;
#pragma unroll
          for (int x = 0; x < NS; ++x)
            if (slot == x) s[r][x] += tl;
        }
      }
      float m = NEG_INF_F;
#pragma unroll
      for (int x = 0; x < NS; ++x) {
        const int key = 256 * (x >> 2) + 4 * lane + (x & 3);
        s[r][x] = key < n ? s[r][x] : NEG_INF_F;
        m = fmaxf(m, s[r][x]);
      }
      m = bl_wave_max(m);
      float sum = 0.f;
#pragma unroll
      for (int x = 0; x < NS; ++x) {
        // e^(s - m) as one v_exp_f32: 2^((s - m) log2 e) -- within 2e-6 relative of expf for the terms that matter (|s - m| < 30),
        // masked keys are -inf -> 0; an all-masked row (n == 0) has m = -inf -> NaN, like torch.softmax of an all -inf row
        s[r][x] = __builtin_amdgcn_exp2f((s[r][x] - m) * 1.44269504088896340736f);
        sum += s[r][x];
      }
      sum = bl_wave_sum(sum);
      const float inv = 1.0f / sum;
      const size_t row = (size_t)g * L + i;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const int key = 256 * c + 4 * lane;
        if (key < L) {  // (L % 4 == 0)
          float4 pr = make_float4(s[r][4 * c] * inv, s[r][4 * c + 1] * inv, s[r][4 * c + 2] * inv, s[r][4 * c + 3] * inv);
          bl_store_streaming(P + row * L + key, pr);  // (268 MB per layer that only the products after this kernel read: bl_common.h)
          if (Pd) {
            const uint32_t e0 = (uint32_t)row * (uint32_t)L + (uint32_t)key;
            pr.x = bl_keep(drop, e0) ? pr.x * drop.scale : 0.f;
            pr.y = bl_keep(drop, e0 + 1) ? pr.y * drop.scale : 0.f;
            pr.z = bl_keep(drop, e0 + 2) ? pr.z * drop.scale : 0.f;
            pr.w = bl_keep(drop, e0 + 3) ? pr.w * drop.scale : 0.f;
            *reinterpret_cast<float4*>(Pd + row * L + key) = pr;
          }
        }
      }
    }
  }
}

// ... and their backward: dP = dO.V^T on the vector units (V^T in LDS), the dropout mask, the softmax backward against the
// saved P, and the gradients of the edge terms, per query row in registers; dS is written once (the grouped GEMMs for dQ
// and dK read it).  Replaces the grouped dO.V^T GEMM + bl_softmax_dropout_bwd + bl_rel_attn_bias_bwd (mode 0).
// Edge terms of row (b, h, i): coef[c] = sum of dS at the entries with code c;  g_q[i, :] (+)= sum_c coef[c] bias[c][h, :]
// (written to gq_edge, zero elsewhere);  g_bias[c][h, :] += coef[c] q[i, :] -- summed per wave in registers (lane owns the
// elements lane + 64 k of the [2 T][dk] table), then per workgroup in LDS, then one atomic per element and workgroup.
#define ATT_TAB_REGS 16  // 2 T dk <= 1024
template <int DK, int KC>
__global__ __launch_bounds__(1024) void attn_probs_bwd_kernel(const HeadView g_ctx, const HeadView v,
                                                              const float* __restrict__ P, const HeadView q, const float q_scale,
                                                              const int* __restrict__ row_ptr, const int* __restrict__ ekey,
                                                              const int* __restrict__ ecode, int L, int H, int T,
                                                              const float* __restrict__ bias_f, const float* __restrict__ bias_r,
                                                              bl_drop_dev drop, int has_drop, float* __restrict__ dS,
                                                              float* __restrict__ gq_edge, float* __restrict__ g_bias_f,
                                                              float* __restrict__ g_bias_r) {
  constexpr int LP = 256 * KC, LS = LP + 4, NS = 4 * KC;
  extern __shared__ __attribute__((aligned(16))) float att_lds[];
  float* __restrict__ Vt = att_lds;                 // [DK][LS]
  float* __restrict__ bl = att_lds + DK * LS;       // [2 T][DK + 1]
  float* __restrict__ tab = bl + 2 * T * (DK + 1);  // [2 T][DK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, b = g / H, h = g - b * H, HD = H * DK;
  const int ntab = 2 * T * DK;
  const int rb = blockIdx.y * ATT_WG_ROWS, re = min(rb + ATT_WG_ROWS, L);
  int rpv[ATT_ITERS];  // (as in the forward kernel)
#pragma unroll
  for (int it = 0; it < ATT_ITERS; ++it)
    rpv[it] = (row_ptr && lane <= ATT_ROWS) ? row_ptr[b * L + min(rb + (16 * it + wave) * ATT_ROWS + lane, L)] : 0;
  const float* __restrict__ gcg = hv_mat(g_ctx, b, h);
  const float* __restrict__ qg = hv_mat(q, b, h);
  {
    const float* __restrict__ vg = hv_mat(v, b, h);
    for (int x = tid; x < LP * (DK / 4); x += 1024) {
      const int j = x / (DK / 4), d = 4 * (x - j * (DK / 4));
      const float4 vv4 = j < L ? *reinterpret_cast<const float4*>(vg + (size_t)j * v.sl + d) : make_float4(0.f, 0.f, 0.f, 0.f);
      Vt[d * LS + j] = vv4.x; Vt[(d + 1) * LS + j] = vv4.y; Vt[(d + 2) * LS + j] = vv4.z; Vt[(d + 3) * LS + j] = vv4.w;
    }
    if (row_ptr) {
      for (int x = tid; x < ntab; x += 1024) {
        const int c = x / DK, d = x - c * DK;
        bl[c * (DK + 1) + d] = ((c & 1) ? bias_r : bias_f)[(size_t)(c >> 1) * HD + h * DK + d];
        tab[x] = 0.f;
      }
    }
  }
  __syncthreads();
  float tabacc[ATT_TAB_REGS];
#pragma unroll
  for (int kk = 0; kk < ATT_TAB_REGS; ++kk) tabacc[kk] = 0.f;
  bool any_edges = false;
#pragma unroll
  for (int it = 0; it < ATT_ITERS; ++it) {
    const int i0 = rb + (16 * it + wave) * ATT_ROWS;
    if (i0 >= re) break;
    float gv[ATT_ROWS];  // lane d < DK: dO[i0 + r][d]
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) gv[r] = lane < DK ? gcg[(size_t)min(i0 + r, L - 1) * g_ctx.sl + lane] : 0.f;
    int ebeg[ATT_ROWS], ecnt[ATT_ROWS], ek[ATT_ROWS], ec[ATT_ROWS];
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) {
      ebeg[r] = __builtin_amdgcn_readlane(rpv[it], r);
      ecnt[r] = i0 + r < re ? __builtin_amdgcn_readlane(rpv[it], r + 1) - ebeg[r] : 0;
      ek[r] = lane < ecnt[r] ? ekey[ebeg[r] + lane] : 0;
      ec[r] = lane < ecnt[r] ? ecode[ebeg[r] + lane] : 0;
    }
    float pv[NS];  // P of the row in turn: requested one row ahead
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const int key = 256 * c + 4 * lane;
      float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
      if (key < L) pr = *reinterpret_cast<const float4*>(P + ((size_t)g * L + i0) * L + key);
      pv[4 * c] = pr.x; pv[4 * c + 1] = pr.y; pv[4 * c + 2] = pr.z; pv[4 * c + 3] = pr.w;
    }
    float s[ATT_ROWS][NS];
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r)
#pragma unroll
      for (int x = 0; x < NS; ++x) s[r][x] = 0.f;
#pragma unroll 2
    for (int d = 0; d < DK; ++d) {
      float4 kv[KC];
#pragma unroll
      for (int c = 0; c < KC; ++c) kv[c] = *reinterpret_cast<const float4*>(Vt + d * LS + 256 * c + 4 * lane);
#pragma unroll
      for (int r = 0; r < ATT_ROWS; ++r) {
        const float gs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gv[r]), d));
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          s[r][4 * c + 0] = fmaf(gs, kv[c].x, s[r][4 * c + 0]);
          s[r][4 * c + 1] = fmaf(gs, kv[c].y, s[r][4 * c + 1]);
          s[r][4 * c + 2] = fmaf(gs, kv[c].z, s[r][4 * c + 2]);
          s[r][4 * c + 3] = fmaf(gs, kv[c].w, s[r][4 * c + 3]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < ATT_ROWS; ++r) {
      const int i = i0 + r;
      if (i >= re) break;  // wave-uniform
      const size_t row = (size_t)g * L + i;
      float pn[NS];  // the next row's P
#pragma unroll
      for (int x = 0; x < NS; ++x) pn[x] = 0.f;
      if (r + 1 < ATT_ROWS && i + 1 < re) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const int key = 256 * c + 4 * lane;
          if (key < L) {
            const float4 pr = *reinterpret_cast<const float4*>(P + (row + 1) * L + key);
            pn[4 * c] = pr.x; pn[4 * c + 1] = pr.y; pn[4 * c + 2] = pr.z; pn[4 * c + 3] = pr.w;
          }
        }
      }
      float dot = 0.f;
#pragma unroll
      for (int x = 0; x < NS; ++x) {
        if (has_drop) {
          const uint32_t e = (uint32_t)row * (uint32_t)L + (uint32_t)(256 * (x >> 2) + 4 * lane + (x & 3));
          s[r][x] = bl_keep(drop, e) ? s[r][x] * drop.scale : 0.f;
        }
        dot += pv[x] * s[r][x];
      }
      dot = bl_wave_sum(dot);
#pragma unroll
      for (int x = 0; x < NS; ++x) s[r][x] = pv[x] * (s[r][x] - dot);
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const int key = 256 * c + 4 * lane;
        if (key < L) bl_store_streaming(dS + row * L + key, make_float4(s[r][4 * c], s[r][4 * c + 1], s[r][4 * c + 2], s[r][4 * c + 3]));
      }
      if (ecnt[r] > 0) {
        any_edges = true;
        const float qd = qg[(size_t)i * q.sl + (lane & (DK - 1))] * q_scale;
        float coef = 0.f;  // lane c < 2 T: sum of dS over this row's entries with code c
        for (int p = 0; p < ecnt[r]; ++p) {  // (wave-uniform entry: scalar registers and branches)
          int key, code;
          if (p < 64) {
            key = __builtin_amdgcn_readlane(ek[r], p);
            code = __builtin_amdgcn_readlane(ec[r], p);
          } else {
            key = __builtin_amdgcn_readfirstlane(ekey[ebeg[r] + p]);
            code = __builtin_amdgcn_readfirstlane(ecode[ebeg[r] + p]);
          }
          const int owner = (key & 255) >> 2, slot = ((key >> 8) << 2) | (key & 3);
          float pick = 0.f;
#pragma unroll
          for (int x = 0; x < NS; ++x)
            if (slot == x) pick = s[r][x];
          const float gval = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pick), owner));
          coef += lane == code ? gval : 0.f;
        }
        float gq = 0.f;
        for (int c = 0; c < 2 * T; ++c)
          gq = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, coef), c)), bl[c * (DK + 1) + (lane & (DK - 1))], gq);
        if (lane < DK) gq_edge[row * DK + lane] = gq;
#pragma unroll
        for (int kk = 0; kk < ATT_TAB_REGS; ++kk) {
          if (64 * kk < ntab) {  // (uniform)
            const int e = lane + 64 * kk;  // element (code, d) = (e / DK, e % DK) of the table
            const float cv = __shfl(coef, min(e / DK, 63), 64);  // (every lane takes part in the exchange)
            if (e < ntab) tabacc[kk] = fmaf(cv, qd, tabacc[kk]);  // (64 % DK == 0: e % DK == lane % DK)
          }
        }
      } else if (gq_edge && lane < DK) {
        gq_edge[row * DK + lane] = 0.f;  // (every row is written: the caller need not zero the buffer)
      }
#pragma unroll
      for (int x = 0; x < NS; ++x) pv[x] = pn[x];
    }
  }
  if (row_ptr) {
    if (any_edges) {
#pragma unroll
      for (int kk = 0; kk < ATT_TAB_REGS; ++kk) {
        const int e = lane + 64 * kk;
        if (e < ntab && tabacc[kk] != 0.f) atomicAdd(&tab[e], tabacc[kk]);
      }
    }
    __syncthreads();
    for (int x = tid; x < ntab; x += 1024) {
      const float val = tab[x];
      if (val != 0.f) {
        const int c = x / DK, d = x - c * DK;
        unsafeAtomicAdd(((c & 1) ? g_bias_r : g_bias_f) + (size_t)(c >> 1) * HD + h * DK + d, val);
      }
    }
  }
}

// ---- the four tall-and-skinny products of the attention (head dimension 32) on the exact-fp32 matrix cores --------------
// P.V, dS.K (rows of a [G L, L] matrix times the head's [L, 32] matrix) and P^T.dO, dS^T.Q (its transpose times one).  The
// general grouped GEMM stages both operands through LDS tile by tile and had ~48 KB in flight per CU (2 TB/s on a 268 MB
// operand); here the [L, 32] matrix of the (sample, head) sits in LDS whole, and every lane streams its own elements of the
// big operand straight from memory into v_mfma_f32_32x32x2_f32 -- 16 or 32 independent loads in flight per lane, no barrier
// in the loop.  Transposed accumulator (operands swapped): a lane owns one output row and 4 x 4 consecutive columns.
typedef float att_f32x16 __attribute__((ext_vector_type(16)));
#define ATT_MM_WAVES 8

// out[(g, i), :] = (sum_k A[(g, i), k] M[g, k, :] (+ add[(g, i), :])) * scale
// a_drop (thresh != 0): A is read through the counter-hash dropout mask, element index (g L + i) L + k -- the dropped
// probabilities of multihead_attention.py:72 are never stored (bl_rel_attn_probs_fwd's Pd)
__global__ __launch_bounds__(64 * ATT_MM_WAVES, 4) void attn_nn32_kernel(const float* __restrict__ A, const HeadView M, int H, int L,
                                                                      const float* __restrict__ add, float scale,
                                                                      const HeadView out, const bl_drop_dev a_drop,
                                                                      const PackedHeadView outp) {
  extern __shared__ __attribute__((aligned(16))) float att_lds[];  // [L][33]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, half = lane >> 5;
  const int g = blockIdx.x, gb = g / H, gh = g - gb * H;
  {
    const float* __restrict__ mg = hv_mat(M, gb, gh);
    for (int x = tid; x < L * 8; x += 64 * ATT_MM_WAVES) {
      const float4 m4 = *reinterpret_cast<const float4*>(mg + (size_t)(x >> 3) * M.sl + 4 * (x & 7));
      float* o = att_lds + (x >> 3) * 33 + 4 * (x & 7);
      o[0] = m4.x; o[1] = m4.y; o[2] = m4.z; o[3] = m4.w;
    }
  }
  __syncthreads();
  const int r0 = (blockIdx.y * ATT_MM_WAVES + wave) * 32;
  if (r0 >= L) return;
  const int row = min(r0 + li, L - 1);
  const float* __restrict__ arow = A + ((size_t)g * L + row) * L;
  att_f32x16 acc;
#pragma unroll
  for (int x = 0; x < 16; ++x) acc[x] = 0.f;
  for (int k0 = 0; k0 < L; k0 += 64) {
    float4 a[8];
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
      const int k = k0 + 8 * qq + 4 * half;
      a[qq] = k < L ? *reinterpret_cast<const float4*>(arow + k) : make_float4(0.f, 0.f, 0.f, 0.f);  // (L % 4 == 0)
    }
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
      const int k = min(k0 + 8 * qq + 4 * half, L - 4);
      float av[4] = {a[qq].x, a[qq].y, a[qq].z, a[qq].w};
      if (a_drop.thresh) {  // (uniform)
        const uint32_t e0 = (uint32_t)((size_t)g * L + row) * (uint32_t)L + (uint32_t)k;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) av[s2] = bl_keep(a_drop, e0 + s2) ? av[s2] * a_drop.scale : 0.f;
      }
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(att_lds[(k + s2) * 33 + li], av[s2], acc, 0, 0, 0);
    }
  }
  if (r0 + li < L) {
    const size_t o = ((size_t)g * L + row) * 32;
    float* __restrict__ orow = out.p ? hv_mat(out, gb, gh) + (size_t)row * out.sl : nullptr;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int n = 8 * g4 + 4 * half;
      float4 v = make_float4(acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
      if (add) {
        const float4 e = *reinterpret_cast<const float4*>(add + o + n);
        v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
      }
      v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
      if (orow) *reinterpret_cast<float4*>(orow + n) = v;
      if (outp.p) phv_store4(outp, (size_t)gb * L + row, gh, n, v.x, v.y, v.z, v.w);
    }
  }
}

// out[g, key, :] = sum_i A[(g, i), key] Bm[g, i, :]
__global__ __launch_bounds__(64 * ATT_MM_WAVES, 4) void attn_tn32_kernel(const float* __restrict__ A, const HeadView Bm, float bm_scale,
                                                                      int H, int L, const HeadView out, const bl_drop_dev a_drop,
                                                                      const PackedHeadView outp) {
  extern __shared__ __attribute__((aligned(16))) float att_lds[];  // [L][33]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, half = lane >> 5;
  const int g = blockIdx.x, gb = g / H, gh = g - gb * H;
  {
    const float* __restrict__ bg = hv_mat(Bm, gb, gh);
    for (int x = tid; x < L * 8; x += 64 * ATT_MM_WAVES) {
      const float4 b4 = *reinterpret_cast<const float4*>(bg + (size_t)(x >> 3) * Bm.sl + 4 * (x & 7));
      float* o = att_lds + (x >> 3) * 33 + 4 * (x & 7);
      o[0] = b4.x * bm_scale; o[1] = b4.y * bm_scale; o[2] = b4.z * bm_scale; o[3] = b4.w * bm_scale;
    }
  }
  __syncthreads();
  const int key0 = (blockIdx.y * ATT_MM_WAVES + wave) * 32;
  if (key0 >= L) return;
  const int key = min(key0 + li, L - 1);
  const float* __restrict__ ag = A + (size_t)g * L * L;  // (uniform base + one per-lane offset: scalar-base addressing)
  const int voff = half * L + key;
  att_f32x16 acc;
#pragma unroll
  for (int x = 0; x < 16; ++x) acc[x] = 0.f;
  for (int i0 = 0; i0 < L; i0 += 64) {
    float a[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const float* __restrict__ rowp = ag + (size_t)(i0 + 2 * u) * L;
      a[u] = i0 + 2 * u < L ? rowp[voff] : 0.f;  // (L is even: row i0 + 2 u + 1 exists whenever row i0 + 2 u does)
    }
    if (a_drop.thresh) {  // (uniform)
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const uint32_t e = (uint32_t)((size_t)g * L + i0 + 2 * u + half) * (uint32_t)L + (uint32_t)key;
        a[u] = bl_keep(a_drop, e) ? a[u] * a_drop.scale : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int i = min(i0 + 2 * u + half, L - 1);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(att_lds[i * 33 + li], a[u], acc, 0, 0, 0);
    }
  }
  if (key0 + li < L) {
    float* __restrict__ orow = out.p ? hv_mat(out, gb, gh) + (size_t)key * out.sl : nullptr;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      if (orow) *reinterpret_cast<float4*>(orow + 8 * g4 + 4 * half) = make_float4(acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
      if (outp.p) phv_store4(outp, (size_t)gb * L + key, gh, 8 * g4 + 4 * half, acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
    }
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

extern "C" int bl_add_layernorm_fwd_packed(const float* x, const float* r, const float* gamma, const float* beta, float eps,
                                           int32_t nrows, int32_t D, float* z_out, float* y, float* mean, float* rstd,
                                           uint16_t* y_packed, void* stream) {
  if (nrows == 0) return BL_OK;
  BL_CHECK_ARG(x && gamma && beta && y && mean && rstd && D > 0 && D <= 1024 && D % 4 == 0,
               "bl_add_layernorm_fwd_packed: null pointer or D not a multiple of 4 in 4..1024");
  BL_CHECK_ARG(bl_aligned16(x) && (r == nullptr || bl_aligned16(r)) && bl_aligned16(gamma) && bl_aligned16(beta) && bl_aligned16(y) &&
                   (z_out == nullptr || bl_aligned16(z_out)) && (y_packed == nullptr || bl_aligned16(y_packed)),
               "bl_add_layernorm_fwd_packed: 16-byte aligned pointers required");
  hipStream_t st = (hipStream_t)stream;
  uint2* yp = reinterpret_cast<uint2*>(y_packed);
#define ALN4_GO(NV4_) hipLaunchKernelGGL((add_layernorm_fwd4_kernel<NV4_>), dim3((nrows + 3) / 4), dim3(256), 0, st, x, r, gamma, beta, eps, nrows, D, z_out, y, mean, rstd, yp)
  if (D <= 256) ALN4_GO(1);
  else if (D <= 512) ALN4_GO(2);
  else ALN4_GO(4);
#undef ALN4_GO
  BL_LAUNCH_CHECK("bl_add_layernorm_fwd_packed");
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

static size_t att_lds_bytes(int L, int dk, int T) {  // K^T or V^T, the head's bias vectors, (backward) their gradient table
  const int kc = (L + 255) / 256;
  return ((size_t)dk * (256 * kc + 4) + (size_t)2 * T * (dk + 1) + (size_t)2 * T * dk) * 4;
}
// whether bl_rel_attn_probs_fwd handles the shape: K^T of one (sample, head) in 64 KB of LDS, 16-byte rows
extern "C" int32_t bl_rel_attn_probs_ok(int32_t L, int32_t dk, int32_t T) {
  if (L <= 0 || L % 4 != 0 || T <= 0 || 2 * T > 64) return 0;
  const int kc = (L + 255) / 256;
  if (kc != 1 && kc != 2 && kc != 4) return 0;
  if (dk != 16 && dk != 32 && dk != 64) return 0;
  if (2 * T * dk > 64 * ATT_TAB_REGS) return 0;  // the backward keeps the edge-bias gradient table in registers
  return att_lds_bytes(L, dk, T) <= 72 * 1024;
}

static int attn_probs_fwd_impl(HeadView q, float q_scale, HeadView k, const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode,
                               int32_t B, int32_t L, int32_t H, int32_t dk, int32_t T, const float* bias_f, const float* bias_r,
                               const int32_t* lens, bl_dropout_t drop, float* P, float* Pd, void* stream) {
  if (B == 0) return BL_OK;
  BL_CHECK_ARG(q.p && k.p && bias_f && bias_r && lens && P && H > 0, "bl_rel_attn_probs_fwd: null pointer");
  BL_CHECK_ARG((row_ptr == nullptr) == (ekey == nullptr) && (ekey == nullptr) == (ecode == nullptr), "bl_rel_attn_probs_fwd: partial edge CSR");
  BL_CHECK_ARG(bl_rel_attn_probs_ok(L, dk, T), "bl_rel_attn_probs_fwd: unsupported shape L=%d dk=%d T=%d", L, dk, T);
  // (Pd == NULL with dropout: only P is written -- the consumer applies the mask where it reads P, bl_attn_*_times_v's a_drop)
  BL_CHECK_ARG(drop.p <= 0.f || ((Pd == nullptr || Pd != P) && (long long)B * H * L * L < (1ll << 32)),
               "bl_rel_attn_probs_fwd: dropout needs a second output (or none) and fewer than 2^32 scores");
  const int kc = (L + 255) / 256;
  const size_t lds = att_lds_bytes(L, dk, T);
  float* pd = drop.p > 0.f ? Pd : nullptr;
  dim3 grid(B * H, (L + ATT_WG_ROWS - 1) / ATT_WG_ROWS);
  hipStream_t st = (hipStream_t)stream;
  int rc = -1;
#define ATT_CASE(DK_, KC_)                                                                                                        \
  if (dk == DK_ && kc == KC_) {                                                                                                   \
    static bool attr[BL_MAX_DEVICES] = {false}; /* per device: the attribute belongs to the device's copy of the function */   \
    if (bl_raise_lds_limit_once((const void*)attn_probs_fwd_kernel<DK_, KC_>, 72 * 1024, attr) != BL_OK) {                         \
      bl_set_error("bl_rel_attn_probs_fwd: cannot raise the LDS limit");                                                          \
      return BL_EINVAL;                                                                                                           \
    }                                                                                                                             \
    hipLaunchKernelGGL((attn_probs_fwd_kernel<DK_, KC_>), grid, dim3(1024), lds, st, q, q_scale, k, row_ptr, ekey, ecode, L, H, T, bias_f, \
                       bias_r, lens, bl_make_drop(drop), P, pd);                                                                  \
    rc = 0;                                                                                                                       \
  }
  ATT_CASE(32, 1) ATT_CASE(32, 2) ATT_CASE(16, 1) ATT_CASE(16, 2) ATT_CASE(16, 4) ATT_CASE(64, 1)
#undef ATT_CASE
  BL_CHECK_ARG(rc == 0, "bl_rel_attn_probs_fwd: no kernel for L=%d dk=%d", L, dk);
  BL_LAUNCH_CHECK("bl_rel_attn_probs_fwd");
  return BL_OK;
}
extern "C" int bl_rel_attn_probs_fwd(const float* q, const float* k, const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode,
                                     int32_t B, int32_t L, int32_t H, int32_t dk, int32_t T, const float* bias_f, const float* bias_r,
                                     const int32_t* lens, bl_dropout_t drop, float* P, float* Pd, void* stream) {
  BL_CHECK_ARG(drop.p <= 0.f || Pd, "bl_rel_attn_probs_fwd: dropout needs a second output");
  return attn_probs_fwd_impl(hv_contiguous(q, H, L, dk), 1.0f, hv_contiguous(k, H, L, dk), row_ptr, ekey, ecode, B, L, H, dk, T, bias_f, bias_r,
                             lens, drop, P, Pd, stream);
}
static inline HeadView hv_null() {
  HeadView v = {nullptr, 0, 0, 0};
  return v;
}
static inline PackedHeadView phv_none() {
  PackedHeadView v = {nullptr, 0, 0, 0};
  return v;
}
static inline PackedHeadView phv_from(const bl_packed_head_view_t* a) {
  PackedHeadView v = {a->p, a->W, a->col0, a->hs};
  return v;
}
static int check_packed_view(const char* who, const bl_packed_head_view_t* v) {
  BL_CHECK_ARG(v->p && bl_aligned16(v->p) && v->W > 0 && v->W % 8 == 0 && v->col0 % 4 == 0 && v->hs % 4 == 0,
               "%s: a packed head view needs a 16-byte aligned base, W a multiple of 8, col0 and hs multiples of 4", who);
  return BL_OK;
}
static int check_view(const char* who, const bl_head_view_t* v) {
  BL_CHECK_ARG(v && v->p && bl_aligned16(v->p) && v->sb % 4 == 0 && v->sh % 4 == 0 && v->sl % 4 == 0, "%s: a head view needs a 16-byte aligned base and strides that are multiples of 4", who);
  return BL_OK;
}
extern "C" int bl_rel_attn_probs_fwd_v(const bl_head_view_t* q, float q_scale, const bl_head_view_t* k, const int32_t* row_ptr,
                                       const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L, int32_t H, int32_t dk, int32_t T,
                                       const float* bias_f, const float* bias_r, const int32_t* lens, bl_dropout_t drop, float* P, float* Pd,
                                       void* stream) {
  int rc = check_view("bl_rel_attn_probs_fwd_v", q);
  if (rc == BL_OK) rc = check_view("bl_rel_attn_probs_fwd_v", k);
  if (rc != BL_OK) return rc;
  return attn_probs_fwd_impl(hv_from(q), q_scale, hv_from(k), row_ptr, ekey, ecode, B, L, H, dk, T, bias_f, bias_r, lens, drop, P, Pd, stream);
}

static int attn_probs_bwd_impl(HeadView g_ctx, HeadView v, const float* P, HeadView q, float q_scale, const int32_t* row_ptr,
                               const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L, int32_t H, int32_t dk, int32_t T,
                               const float* bias_f, const float* bias_r, bl_dropout_t drop, float* dS, float* gq_edge,
                               float* g_bias_f, float* g_bias_r, void* stream) {
  if (B == 0) return BL_OK;
  BL_CHECK_ARG(g_ctx.p && v.p && P && q.p && bias_f && bias_r && dS && H > 0, "bl_rel_attn_probs_bwd: null pointer");
  BL_CHECK_ARG((row_ptr == nullptr) == (ekey == nullptr) && (ekey == nullptr) == (ecode == nullptr), "bl_rel_attn_probs_bwd: partial edge CSR");
  BL_CHECK_ARG(row_ptr == nullptr || (gq_edge && g_bias_f && g_bias_r), "bl_rel_attn_probs_bwd: edge entries need gq_edge, g_bias_f, g_bias_r");
  BL_CHECK_ARG(bl_rel_attn_probs_ok(L, dk, T), "bl_rel_attn_probs_bwd: unsupported shape L=%d dk=%d T=%d", L, dk, T);
  BL_CHECK_ARG(drop.p <= 0.f || (long long)B * H * L * L < (1ll << 32), "bl_rel_attn_probs_bwd: more than 2^32 scores");
  const int kc = (L + 255) / 256;
  const size_t lds = att_lds_bytes(L, dk, T);
  dim3 grid(B * H, (L + ATT_WG_ROWS - 1) / ATT_WG_ROWS);
  hipStream_t st = (hipStream_t)stream;
  int rc = -1;
#define ATT_CASE(DK_, KC_)                                                                                                        \
  if (dk == DK_ && kc == KC_) {                                                                                                   \
    static bool attr[BL_MAX_DEVICES] = {false}; /* per device: the attribute belongs to the device's copy of the function */   \
    if (bl_raise_lds_limit_once((const void*)attn_probs_bwd_kernel<DK_, KC_>, 72 * 1024, attr) != BL_OK) {                         \
      bl_set_error("bl_rel_attn_probs_bwd: cannot raise the LDS limit");                                                          \
      return BL_EINVAL;                                                                                                           \
    }                                                                                                                             \
    hipLaunchKernelGGL((attn_probs_bwd_kernel<DK_, KC_>), grid, dim3(1024), lds, st, g_ctx, v, P, q, q_scale, row_ptr, ekey, ecode, L, H, T, \
                       bias_f, bias_r, bl_make_drop(drop), drop.p > 0.f ? 1 : 0, dS, gq_edge, g_bias_f, g_bias_r);                \
    rc = 0;                                                                                                                       \
  }
  ATT_CASE(32, 1) ATT_CASE(32, 2) ATT_CASE(16, 1) ATT_CASE(16, 2) ATT_CASE(16, 4) ATT_CASE(64, 1)
#undef ATT_CASE
  BL_CHECK_ARG(rc == 0, "bl_rel_attn_probs_bwd: no kernel for L=%d dk=%d", L, dk);
  BL_LAUNCH_CHECK("bl_rel_attn_probs_bwd");
  return BL_OK;
}
extern "C" int bl_rel_attn_probs_bwd(const float* g_ctx, const float* v, const float* P, const float* q, const int32_t* row_ptr,
                                     const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L, int32_t H, int32_t dk, int32_t T,
                                     const float* bias_f, const float* bias_r, bl_dropout_t drop, float* dS, float* gq_edge,
                                     float* g_bias_f, float* g_bias_r, void* stream) {
  return attn_probs_bwd_impl(hv_contiguous(g_ctx, H, L, dk), hv_contiguous(v, H, L, dk), P, hv_contiguous(q, H, L, dk), 1.0f, row_ptr, ekey, ecode,
                             B, L, H, dk, T, bias_f, bias_r, drop, dS, gq_edge, g_bias_f, g_bias_r, stream);
}
extern "C" int bl_rel_attn_probs_bwd_v(const bl_head_view_t* g_ctx, const bl_head_view_t* v, const float* P, const bl_head_view_t* q,
                                       float q_scale, const int32_t* row_ptr, const int32_t* ekey, const int32_t* ecode, int32_t B, int32_t L,
                                       int32_t H, int32_t dk, int32_t T, const float* bias_f, const float* bias_r, bl_dropout_t drop, float* dS,
                                       float* gq_edge, float* g_bias_f, float* g_bias_r, void* stream) {
  int rc = check_view("bl_rel_attn_probs_bwd_v", g_ctx);
  if (rc == BL_OK) rc = check_view("bl_rel_attn_probs_bwd_v", v);
  if (rc == BL_OK) rc = check_view("bl_rel_attn_probs_bwd_v", q);
  if (rc != BL_OK) return rc;
  return attn_probs_bwd_impl(hv_from(g_ctx), hv_from(v), P, hv_from(q), q_scale, row_ptr, ekey, ecode, B, L, H, dk, T, bias_f, bias_r, drop, dS,
                             gq_edge, g_bias_f, g_bias_r, stream);
}

static int att_mm_lds(const void* fn, int L, const char* who) {
  const size_t lds = (size_t)L * 33 * sizeof(float);
  if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) {
    bl_set_error("%s: cannot raise the LDS limit", who);
    return BL_EINVAL;
  }
  return BL_OK;
}

extern "C" int32_t bl_attn_mm32_ok(int32_t L, int32_t dk) { return dk == 32 && L > 0 && L % 4 == 0 && (size_t)L * 33 * 4 <= 152 * 1024; }

static int attn_rows_times_impl(const float* A, HeadView M, int32_t B, int32_t H, int32_t L, int32_t dk, const float* add, float scale,
                                HeadView out, bl_dropout_t a_drop, PackedHeadView outp, void* stream) {
  if (B * H == 0) return BL_OK;
  BL_CHECK_ARG(A && M.p && (out.p || outp.p), "bl_attn_rows_times: null pointer");
  BL_CHECK_ARG(bl_attn_mm32_ok(L, dk), "bl_attn_rows_times: needs dk == 32, L %% 4 == 0, L <= 1164 (got L=%d dk=%d)", L, dk);
  int rc = att_mm_lds((const void*)attn_nn32_kernel, L, "bl_attn_rows_times");
  if (rc != BL_OK) return rc;
  dim3 grid(B * H, (L + 32 * ATT_MM_WAVES - 1) / (32 * ATT_MM_WAVES));
  BL_CHECK_ARG(a_drop.p <= 0.f || (long long)B * H * L * L < (1ll << 32), "bl_attn_rows_times: more than 2^32 elements under a dropout mask");
  hipLaunchKernelGGL(attn_nn32_kernel, grid, dim3(64 * ATT_MM_WAVES), (size_t)L * 33 * sizeof(float), (hipStream_t)stream, A, M, H, L, add, scale, out,
                     bl_make_drop(a_drop), outp);
  BL_LAUNCH_CHECK("bl_attn_rows_times");
  return BL_OK;
}
extern "C" int bl_attn_rows_times(const float* A, const float* M, int32_t G, int32_t L, int32_t dk, const float* add, float scale,
                                  float* out, void* stream) {
  const bl_dropout_t none = {0.f, 0u, 0u};
  return attn_rows_times_impl(A, hv_contiguous(M, 1, L, dk), G, 1, L, dk, add, scale, hv_contiguous(out, 1, L, dk), none, phv_none(), stream);
}
extern "C" int bl_attn_rows_times_v(const float* A, const bl_head_view_t* M, int32_t B, int32_t H, int32_t L, int32_t dk, const float* add,
                                    float scale, const bl_head_view_t* out, bl_dropout_t a_drop, const bl_packed_head_view_t* out_packed,
                                    void* stream) {
  int rc = check_view("bl_attn_rows_times_v", M);
  if (rc == BL_OK && out) rc = check_view("bl_attn_rows_times_v", out);
  if (rc == BL_OK && out_packed) rc = check_packed_view("bl_attn_rows_times_v", out_packed);
  if (rc != BL_OK) return rc;
  return attn_rows_times_impl(A, hv_from(M), B, H, L, dk, add, scale, out ? hv_from(out) : hv_null(), a_drop,
                              out_packed ? phv_from(out_packed) : phv_none(), stream);
}

static int attn_transposed_times_impl(const float* A, HeadView Bm, float bm_scale, int32_t B, int32_t H, int32_t L, int32_t dk, HeadView out,
                                      bl_dropout_t a_drop, PackedHeadView outp, void* stream) {
  if (B * H == 0) return BL_OK;
  BL_CHECK_ARG(A && Bm.p && (out.p || outp.p), "bl_attn_transposed_times: null pointer");
  BL_CHECK_ARG(bl_attn_mm32_ok(L, dk), "bl_attn_transposed_times: needs dk == 32, L %% 4 == 0, L <= 1164 (got L=%d dk=%d)", L, dk);
  int rc = att_mm_lds((const void*)attn_tn32_kernel, L, "bl_attn_transposed_times");
  if (rc != BL_OK) return rc;
  dim3 grid(B * H, (L + 32 * ATT_MM_WAVES - 1) / (32 * ATT_MM_WAVES));
  BL_CHECK_ARG(a_drop.p <= 0.f || (long long)B * H * L * L < (1ll << 32), "bl_attn_transposed_times: more than 2^32 elements under a dropout mask");
  hipLaunchKernelGGL(attn_tn32_kernel, grid, dim3(64 * ATT_MM_WAVES), (size_t)L * 33 * sizeof(float), (hipStream_t)stream, A, Bm, bm_scale, H, L, out,
                     bl_make_drop(a_drop), outp);
  BL_LAUNCH_CHECK("bl_attn_transposed_times");
  return BL_OK;
}
extern "C" int bl_attn_transposed_times(const float* A, const float* Bm, int32_t G, int32_t L, int32_t dk, float* out, void* stream) {
  const bl_dropout_t none = {0.f, 0u, 0u};
  return attn_transposed_times_impl(A, hv_contiguous(Bm, 1, L, dk), 1.0f, G, 1, L, dk, hv_contiguous(out, 1, L, dk), none, phv_none(), stream);
}
extern "C" int bl_attn_transposed_times_v(const float* A, const bl_head_view_t* Bm, float bm_scale, int32_t B, int32_t H, int32_t L, int32_t dk,
                                          const bl_head_view_t* out, bl_dropout_t a_drop, const bl_packed_head_view_t* out_packed,
                                          void* stream) {
  int rc = check_view("bl_attn_transposed_times_v", Bm);
  if (rc == BL_OK && out) rc = check_view("bl_attn_transposed_times_v", out);
  if (rc == BL_OK && out_packed) rc = check_packed_view("bl_attn_transposed_times_v", out_packed);
  if (rc != BL_OK) return rc;
  return attn_transposed_times_impl(A, hv_from(Bm), bm_scale, B, H, L, dk, out ? hv_from(out) : hv_null(), a_drop,
                                    out_packed ? phv_from(out_packed) : phv_none(), stream);
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
