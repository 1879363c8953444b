// Backward of a message-passing layer's node update (LayerNorm -> Linear -> tanh -> Dropout: ptgnn MlpMessagePassingLayer's
// tail; call site buglab/models/gnnlayerdefs.py:6-23) as ONE kernel per 64-node tile instead of three launches that pass
// [N, Dout] / [N, Dm] rows through HBM five times (act_bwd_kernel -> gemm_rows_x6_kernel<false, 0> -> layernorm_bwd_kernel):
//
//   prologue   g_z = g_out . dropout mask . tanh'(h_out)        computed on the way into LDS (bf16x3 planes = the A operand);
//              the packed g_z also goes to HBM once (the dense weight-gradient GEMM on the side stream reads it), its
//              column sums are the bias gradient
//   GEMM       g_ln = g_z . Wd^T                                bf16x6 on the matrix cores, K = Dout, all Dm columns of a tile's
//                                                               rows inside one workgroup
//   epilogue   LayerNorm backward x activation derivative       row sums in registers (+ one cross-lane, one cross-wave step),
//              at the winners -> gq (fp32 and / or packed)      gamma / beta gradients: per-wave transposing butterfly ->
//                                                               LDS -> one atomic per column and workgroup
//
// Bytes per node at (Dm, Dout) = (128, 128): 1 KB in (g_out, h_out) + 0.75 KB out (packed g_z) + 1 KB in (aggregate,
// activation derivative) + 0.75 KB (+ 0.5 KB fp32) out = 4 KB against 5.9 KB for the three kernels; HBM-bound.
//
// Tile: 64 nodes x Dm columns (Dm = 128 NT, NT = 1 or 2), 4 waves as 2 (32-row halves) x 2 (column halves); a wave owns
// 32 rows x Dm/2 columns = 2 NT accumulator fragments in the transposed form of csrc/bl_gemm_x6.hip (a lane holds 4
// consecutive columns of ONE row, 4 such groups per fragment), so a row's LayerNorm sums are in-register sums + lane ^ 32 +
// the partner wave through LDS.  Workgroups are persistent (grid = resident workgroups) and keep their column sums in LDS
// across tiles: one flush per workgroup.
#include "bl_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define NB_ROWS 64   // nodes per tile
#define NB_XROW 13   // uint4 per LDS row: 4 k-groups x 3 planes + 1 pad (bl_gemm_x6.hip's stage layout)
#define NB_PK(a_, b_) ((uint32_t)(a_) | ((uint32_t)(b_) << 16))

namespace {

// sum over the 32 lanes of each wave half of 16 values per lane; afterwards lane li (and li ^ 16) holds the total of slot
// (bit0 li) 8 + (bit1 li) 4 + (bit2 li) 2 + (bit3 li) in v[0]: 15 + 1 cross-lane moves instead of 16 x 5
__device__ __forceinline__ float colreduce16(float (&v)[16], int li) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int cnt = 16 >> s;
    const bool up = (li >> s) & 1;
#pragma unroll
    for (int i = 0; i < cnt / 2; ++i) {
      float lo = v[i], hi = v[i + cnt / 2];
      // (opaque to the optimiser: it otherwise turns the two selects into v[up ? i : i + cnt / 2], a lane-varying index into a
      // register array = a 16-way compare / select chain per access)
      asm volatile("" : "+v"(lo), "+v"(hi));
      const float send = up ? lo : hi;
      const float keep = up ? hi : lo;
      v[i] = keep + __shfl_xor(send, 1 << s, 64);
    }
  }
  return v[0] + __shfl_xor(v[0], 16, 64);
}
__device__ __forceinline__ int colreduce16_slot(int li) { return ((li & 1) << 3) | ((li & 2) << 1) | ((li & 4) >> 1) | ((li & 8) >> 3); }

// sum of 8 values per lane over the 16 lanes that share (lane & 3) (the 16 rows a wave stages per 8-column group); afterwards
// every lane holds the total of slot (bit2 lane) 4 + (bit3 lane) 2 + (bit4 lane): 7 + 1 cross-lane moves
__device__ __forceinline__ float rowgroup_reduce8(float (&v)[8], int lane) {
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int cnt = 8 >> s;
    const bool up = (lane >> (2 + s)) & 1;
#pragma unroll
    for (int i = 0; i < cnt / 2; ++i) {
      float lo = v[i], hi = v[i + cnt / 2];
      asm volatile("" : "+v"(lo), "+v"(hi));
      const float send = up ? lo : hi;
      const float keep = up ? hi : lo;
      v[i] = keep + __shfl_xor(send, 4 << s, 64);
    }
  }
  return v[0] + __shfl_xor(v[0], 32, 64);
}

template <int NT>
__global__ __launch_bounds__(256, 2) void node_bwd_kernel(
    const float* __restrict__ g_out, const float* __restrict__ h_out, int nrows, int K, bl_drop_dev drop,
    uint4* __restrict__ gz_packed, float* __restrict__ g_bias, const uint4* __restrict__ bp, const float* __restrict__ agg,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
    const float* __restrict__ dact, float* __restrict__ gq_f32, uint32_t* __restrict__ gq_packed, float* __restrict__ g_gamma,
    float* __restrict__ g_beta, int ntiles, float* __restrict__ gq_amax) {
  constexpr int Dm = 128 * NT;
  constexpr int FR = 2 * NT;  // accumulator fragments (32 x 32) per wave
  // operand images of a stage (As: 64 rows, Bs: Dm rows); after the last stage the same memory holds the four waves' result
  // tiles [32 rows][Dm/2 + 4] fp32 on their way out (coalesced row stores instead of the accumulator layout's 32-byte pieces)
  constexpr int ST_LD = Dm / 2 + 4;
  constexpr int AB_BYTES = (NB_ROWS + Dm) * NB_XROW * 16, ST_BYTES = 4 * 32 * ST_LD * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[AB_BYTES > ST_BYTES ? AB_BYTES : ST_BYTES];
  uint4* As = reinterpret_cast<uint4*>(smem);
  uint4* Bs = As + NB_ROWS * NB_XROW;
  __shared__ float bias_s[256];        // column sums of g_z (K <= 256)
  __shared__ float cs_s[2][Dm];        // column sums for g_gamma / g_beta
  __shared__ float rowpart[2][NB_ROWS][2];  // [column half][row][S1, S2]: the partner wave's share of a row's sums
  __shared__ float gamma_s[Dm];        // LayerNorm weights (read per element in both epilogue passes)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, half = lane >> 5;
  const int p_kg = tid & 3, p_row = tid >> 2;  // A loader: one row, one 8-column group of the stage per thread
  const int nk = K >> 5;
  const int kq = K >> 3;  // uint4 per plane of a packed g_z row

  for (int i = tid; i < 256; i += 256) bias_s[i] = 0.f;
  for (int i = tid; i < 2 * Dm; i += 256) (&cs_s[0][0])[i] = 0.f;
  for (int i = tid; i < Dm; i += 256) gamma_s[i] = gamma[i];
  __syncthreads();

  float amx = 0.f;  // max |gq| over what this thread writes (gq_amax: the f16x3 message GEMMs scale gq by it)
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * NB_ROWS;
    const int tr = min(NB_ROWS, nrows - row0);  // rows of this tile
    const bool a_ok = p_row < tr;
    const size_t a_row = (size_t)(row0 + (a_ok ? p_row : 0));

    // staging registers (scalars, not arrays: hipcc keeps small arrays that cross the stage loop in scratch)
    float4 rg0, rg1, ry0, ry1;
    uint4 rb00, rb01, rb02, rb10, rb11, rb12, rb20, rb21, rb22, rb30, rb31, rb32;
    if (NT == 1) { rb20 = rb21 = rb22 = rb30 = rb31 = rb32 = make_uint4(0u, 0u, 0u, 0u); }
#define NB_LOAD_B(j_, tn_, i_, k0_)                                                          \
  {                                                                                          \
    const uint4* bsrc_ = bp + ((size_t)(tn_) * nk + ((k0_) >> 5)) * 1536 + (i_) * 768 + tid; \
    rb##j_##0 = bsrc_[0];                                                                    \
    rb##j_##1 = bsrc_[256];                                                                  \
    rb##j_##2 = bsrc_[512];                                                                  \
  }
#define NB_STORE_B(j_, tn_, i_)                                                              \
  {                                                                                          \
    uint4* d_ = &Bs[((tn_) * 128 + 64 * (i_) + p_row) * NB_XROW + p_kg];                     \
    d_[0] = rb##j_##0;                                                                       \
    d_[4] = rb##j_##1;                                                                       \
    d_[8] = rb##j_##2;                                                                       \
  }
#define NB_LOAD_STAGE(k0_)                                                                   \
  {                                                                                          \
    const float* gp_ = g_out + a_row * K + (k0_) + 8 * p_kg;                                 \
    const float* yp_ = h_out + a_row * K + (k0_) + 8 * p_kg;                                 \
    rg0 = *reinterpret_cast<const float4*>(gp_);                                             \
    rg1 = *reinterpret_cast<const float4*>(gp_ + 4);                                         \
    ry0 = *reinterpret_cast<const float4*>(yp_);                                             \
    ry1 = *reinterpret_cast<const float4*>(yp_ + 4);                                         \
    NB_LOAD_B(0, 0, 0, k0_) NB_LOAD_B(1, 0, 1, k0_)                                          \
    if (NT > 1) { NB_LOAD_B(2, 1, 0, k0_) NB_LOAD_B(3, 1, 1, k0_) }                          \
  }
    // g_z of this thread's 8 elements (act_bwd_kernel's arithmetic and dropout counter: element index row * Dout + column),
    // split into planes: -> LDS (A operand), -> HBM (packed g_z), column sums -> LDS
#define NB_STORE_STAGE(k0_)                                                                  \
  {                                                                                          \
    const int c0_ = (k0_) + 8 * p_kg;                                                        \
    float gg_[8] = {rg0.x, rg0.y, rg0.z, rg0.w, rg1.x, rg1.y, rg1.z, rg1.w};                 \
    const float yy_[8] = {ry0.x, ry0.y, ry0.z, ry0.w, ry1.x, ry1.y, ry1.z, ry1.w};           \
    uint16_t h_[8], m_[8], l_[8];                                                            \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                          \
      float yu_ = yy_[u];                                                                    \
      if (drop.thresh) {                                                                     \
        const bool keep_ = bl_keep(drop, (uint32_t)(row0 + p_row) * (uint32_t)K + (uint32_t)(c0_ + u)); \
        gg_[u] = keep_ ? gg_[u] * drop.scale : 0.f;                                          \
        yu_ = yu_ * (1.0f / drop.scale);                                                     \
      }                                                                                      \
      gg_[u] *= 1.f - yu_ * yu_;                                                             \
      if (!a_ok) gg_[u] = 0.f;                                                               \
      split3(gg_[u], h_[u], m_[u], l_[u]);                                                   \
    }                                                                                        \
    const uint4 ph_ = make_uint4(NB_PK(h_[0], h_[1]), NB_PK(h_[2], h_[3]), NB_PK(h_[4], h_[5]), NB_PK(h_[6], h_[7])); \
    const uint4 pm_ = make_uint4(NB_PK(m_[0], m_[1]), NB_PK(m_[2], m_[3]), NB_PK(m_[4], m_[5]), NB_PK(m_[6], m_[7])); \
    const uint4 pl_ = make_uint4(NB_PK(l_[0], l_[1]), NB_PK(l_[2], l_[3]), NB_PK(l_[4], l_[5]), NB_PK(l_[6], l_[7])); \
    As[p_row * NB_XROW + 0 + p_kg] = ph_;                                                    \
    As[p_row * NB_XROW + 4 + p_kg] = pm_;                                                    \
    As[p_row * NB_XROW + 8 + p_kg] = pl_;                                                    \
    if (a_ok) {                                                                              \
      uint4* o_ = gz_packed + a_row * 3 * kq + (c0_ >> 3);                                   \
      o_[0] = ph_;                                                                           \
      o_[kq] = pm_;                                                                          \
      o_[2 * kq] = pl_;                                                                      \
    }                                                                                        \
    if (g_bias) { /* column sums of the wave's 16 rows (rows past the end hold zeros), then one LDS add */ \
      const float tot_ = rowgroup_reduce8(gg_, lane);                                        \
      if (lane < 32) atomicAdd(&bias_s[c0_ + ((lane >> 2) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 4) & 1)], tot_); \
    }                                                                                        \
    NB_STORE_B(0, 0, 0) NB_STORE_B(1, 0, 1)                                                  \
    if (NT > 1) { NB_STORE_B(2, 1, 0) NB_STORE_B(3, 1, 1) }                                  \
  }

    f32x16 acc[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

#define NB_MFMA_STAGE()                                                                      \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                            \
    const int kg = 2 * s + half;                                                             \
    const uint4* pa = &As[(wm * 32 + li) * NB_XROW + kg];                                    \
    const bf16x8 ah = __builtin_bit_cast(bf16x8, pa[0]);                                     \
    const bf16x8 am = __builtin_bit_cast(bf16x8, pa[4]);                                     \
    const bf16x8 al = __builtin_bit_cast(bf16x8, pa[8]);                                     \
    _Pragma("unroll") for (int f = 0; f < FR; ++f) {                                         \
      const uint4* pb = &Bs[(wn * (Dm / 2) + f * 32 + li) * NB_XROW + kg];                   \
      const bf16x8 bh = __builtin_bit_cast(bf16x8, pb[0]);                                   \
      const bf16x8 bm = __builtin_bit_cast(bf16x8, pb[4]);                                   \
      const bf16x8 bl = __builtin_bit_cast(bf16x8, pb[8]);                                   \
      f32x16 a = acc[f]; /* B fragment in the A slot: transposed accumulator; small terms first (bl_gemm_x6.hip) */ \
      a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm, am, a, 0, 0, 0);                       \
      a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, a, 0, 0, 0);                       \
      a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, a, 0, 0, 0);                       \
      a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm, ah, a, 0, 0, 0);                       \
      a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, am, a, 0, 0, 0);                       \
      a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, a, 0, 0, 0);                       \
      acc[f] = a;                                                                            \
    }                                                                                        \
  }
    // (the staging registers are written outside any branch: hipcc sends small arrays that are written under control flow to scratch)
    NB_LOAD_STAGE(0)
    NB_STORE_STAGE(0)
    __syncthreads();
    for (int kt = 0; kt + 1 < nk; ++kt) {
      NB_LOAD_STAGE((kt + 1) * 32)
      NB_MFMA_STAGE()
      __syncthreads();
      NB_STORE_STAGE((kt + 1) * 32)
      __syncthreads();
    }
    NB_MFMA_STAGE()
    __syncthreads();  // the next tile's first stage overwrites As / Bs

    // ---- epilogue: LayerNorm backward of this lane's row (layernorm_bwd_kernel's arithmetic) ----
    const int m = wm * 32 + li;  // row inside the tile
    const bool r_ok = m < tr;
    const size_t grow = (size_t)(row0 + (r_ok ? m : 0));
    const float mu = mean[grow], rs = rstd[grow];
    const int cbase = wn * (Dm / 2) + 4 * half;  // this lane's first column; + 32 f + 8 gq + u
    // one base pointer per array: the (f, gq) steps are compile-time offsets of the load / store instructions
    const float* agg_r = agg + grow * Dm + cbase;
    const float* dact_r = dact ? dact + grow * Dm + cbase : nullptr;
    float* stage = reinterpret_cast<float*>(smem) + wave * 32 * ST_LD;  // this wave's [32][ST_LD] result tile
    const float* gam_r = &gamma_s[cbase];
    float s1 = 0.f, s2 = 0.f;
    constexpr bool KEEP_X = NT == 1;  // the normalised aggregate stays in registers for the second pass (NT = 2: re-read, L2 hits)
    float xh[KEEP_X ? FR : 1][16];
#pragma unroll
    for (int f = 0; f < FR; ++f)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 x4 = *reinterpret_cast<const float4*>(agg_r + 32 * f + 8 * gq);
        const float4 g4 = *reinterpret_cast<const float4*>(gam_r + 32 * f + 8 * gq);
        const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float gg = acc[f][4 * gq + u] * gv[u];
          const float xn = r_ok ? (xv[u] - mu) * rs : 0.f;
          if (KEEP_X) xh[f][4 * gq + u] = xn;
          s1 += gg;
          s2 += gg * xn;
        }
      }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (half == 0) { rowpart[wn][m][0] = s1; rowpart[wn][m][1] = s2; }
    __syncthreads();
    s1 += rowpart[wn ^ 1][m][0];
    s2 += rowpart[wn ^ 1][m][1];
    const float a_mean = s1 * (1.0f / (float)Dm), b_mean = s2 * (1.0f / (float)Dm);
    const int halfD = Dm >> 1;
    const float* agg2 = agg_r;
    asm volatile("" : "+v"(agg2));  // (a second look at the same addresses, not a value to keep alive from the first pass)
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      float dg[16], db[16];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 g4 = *reinterpret_cast<const float4*>(gam_r + 32 * f + 8 * gq);
        float4 p4 = make_float4(1.f, 1.f, 1.f, 1.f);
        if (dact) p4 = *reinterpret_cast<const float4*>(dact_r + 32 * f + 8 * gq);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, pv[4] = {p4.x, p4.y, p4.z, p4.w};
        float xv[4] = {0.f, 0.f, 0.f, 0.f};
        if (!KEEP_X) {
          const float4 x4 = *reinterpret_cast<const float4*>(agg2 + 32 * f + 8 * gq);
          xv[0] = x4.x; xv[1] = x4.y; xv[2] = x4.z; xv[3] = x4.w;
        }
        float gx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float gy = r_ok ? acc[f][4 * gq + u] : 0.f;
          const float xn = KEEP_X ? xh[KEEP_X ? f : 0][4 * gq + u] : (r_ok ? (xv[u] - mu) * rs : 0.f);
          gx[u] = rs * (gy * gv[u] - a_mean - xn * b_mean) * pv[u];
          dg[4 * gq + u] = gy * xn;
          db[4 * gq + u] = gy;
        }
        *reinterpret_cast<float4*>(stage + li * ST_LD + 32 * f + 8 * gq + 4 * half) = make_float4(gx[0], gx[1], gx[2], gx[3]);
      }
      // column sums of this fragment over the wave's 32 rows -> LDS (the two row halves of the tile add into the same cells)
      const float tg = colreduce16(dg, li), tb = colreduce16(db, li);
      if (!(li & 16)) {
        const int slot = colreduce16_slot(li);  // = 4 gq + u
        const int c = cbase + 32 * f + 8 * (slot >> 2) + (slot & 3);
        atomicAdd(&cs_s[0][c], tg);
        atomicAdd(&cs_s[1][c], tb);
      }
      if (NT > 1) __builtin_amdgcn_sched_barrier(0);  // one fragment's loads at a time (registers)
    }
    // ---- write-out: whole row pieces (Dm/2 columns = 256 / 512 B in fp32, 128 / 256 B per packed plane) per group of lanes ----
    __syncthreads();
    {
      constexpr int LPR = Dm / 8, RP = 64 / LPR;  // lanes per row piece, rows per pass
      const int c4 = lane % LPR, rsub = lane / LPR;
      const int col = wn * (Dm / 2) + 4 * c4;
#pragma unroll 4
      for (int j = 0; j < 32 / RP; ++j) {
        const int r = rsub + RP * j;
        if (wm * 32 + r >= tr) continue;
        const float4 v = *reinterpret_cast<const float4*>(stage + r * ST_LD + 4 * c4);
        const size_t gr = (size_t)(row0 + wm * 32 + r);
        if (gq_f32) *reinterpret_cast<float4*>(gq_f32 + gr * Dm + col) = v;
        amx = fmaxf(amx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        if (gq_packed) {
          const float gx[4] = {v.x, v.y, v.z, v.w};
          uint16_t h[4], mm[4], l[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) split3(gx[u], h[u], mm[u], l[u]);
          uint2* o = reinterpret_cast<uint2*>(gq_packed + gr * 3 * halfD + (col >> 1));  // (col % 4 == 0: 8-byte aligned)
          o[0] = make_uint2(NB_PK(h[0], h[1]), NB_PK(h[2], h[3]));
          o[halfD >> 1] = make_uint2(NB_PK(mm[0], mm[1]), NB_PK(mm[2], mm[3]));
          o[halfD] = make_uint2(NB_PK(l[0], l[1]), NB_PK(l[2], l[3]));
        }
      }
    }
    __syncthreads();  // the next tile's first stage overwrites the staged tiles (and rowpart is written again only after it)
  }

  __syncthreads();
  for (int c = tid; c < Dm; c += 256) {
    unsafeAtomicAdd(&g_gamma[c], cs_s[0][c]);
    unsafeAtomicAdd(&g_beta[c], cs_s[1][c]);
  }
  if (g_bias)
    for (int c = tid; c < K; c += 256) unsafeAtomicAdd(&g_bias[c], bias_s[c]);
  if (gq_amax) {  // one atomic per workgroup (non-negative floats order like their bit patterns; a NaN is not recorded)
    amx = bl_wave_max(amx);
    __syncthreads();
    if (lane == 0) rowpart[0][wave][0] = amx;
    __syncthreads();
    if (tid == 0) {
      const float m = fmaxf(fmaxf(rowpart[0][0][0], rowpart[0][1][0]), fmaxf(rowpart[0][2][0], rowpart[0][3][0]));
      if (m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(gq_amax), __float_as_uint(m));
    }
  }
}

int g_node_bwd_resident[3] = {0, 0, 0};

template <int NT>
int node_bwd_launch(const float* g_out, const float* h_out, int nrows, int K, bl_drop_dev drop, uint16_t* gz_packed, float* g_bias,
                    const uint16_t* wd_packed_bwd, const float* agg, const float* mean, const float* rstd, const float* gamma,
                    const float* dact, float* gq_f32, uint16_t* gq_packed, float* g_gamma, float* g_beta, float* gq_amax, hipStream_t st) {
  int& resident = g_node_bwd_resident[NT];
  if (resident == 0) {
    int per_cu = 0;
    hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, node_bwd_kernel<NT>, 256, 0);
    if (oe != hipSuccess || per_cu <= 0) per_cu = 1;
    resident = per_cu * bl_num_cus();
  }
  const int ntiles = (nrows + NB_ROWS - 1) / NB_ROWS;
  const int grid = ntiles < resident ? ntiles : resident;
  hipLaunchKernelGGL((node_bwd_kernel<NT>), dim3(grid), dim3(256), 0, st, g_out, h_out, nrows, K, drop,
                     reinterpret_cast<uint4*>(gz_packed), g_bias, reinterpret_cast<const uint4*>(wd_packed_bwd), agg, mean, rstd, gamma,
                     dact, gq_f32, reinterpret_cast<uint32_t*>(gq_packed), g_gamma, g_beta, ntiles, gq_amax);
  return BL_OK;
}
}  // namespace

extern "C" int32_t bl_node_update_bwd_ok(int32_t Dm, int32_t Dout) {
  return (Dm == 128 || Dm == 256) && Dout % 32 == 0 && Dout > 0 && Dout <= 256 && !bl_get_deterministic();
}

extern "C" int bl_node_update_bwd(const float* g_out, const float* h_out, int32_t nrows, int32_t Dout, bl_dropout_t drop,
                                  const uint16_t* wd_packed_bwd, const float* agg, const float* mean, const float* rstd,
                                  const float* ln_g, const float* dact, int32_t Dm, uint16_t* g_z_packed, float* g_bias,
                                  float* gq, uint16_t* gq_packed, float* g_ln_g, float* g_ln_b, void* stream) {
  return bl_node_update_bwd_impl(g_out, h_out, nrows, Dout, drop, wd_packed_bwd, agg, mean, rstd, ln_g, dact, Dm, g_z_packed, g_bias, gq,
                                 gq_packed, g_ln_g, g_ln_b, nullptr, stream);
}

// + gq_amax (optional, device float, zeroed by the caller): max |gq| is added to it with one atomic max per workgroup
int bl_node_update_bwd_impl(const float* g_out, const float* h_out, int32_t nrows, int32_t Dout, bl_dropout_t drop,
                            const uint16_t* wd_packed_bwd, const float* agg, const float* mean, const float* rstd, const float* ln_g,
                            const float* dact, int32_t Dm, uint16_t* g_z_packed, float* g_bias, float* gq, uint16_t* gq_packed,
                            float* g_ln_g, float* g_ln_b, float* gq_amax, void* stream) {
  if (nrows == 0) return BL_OK;
  BL_CHECK_ARG(g_out && h_out && wd_packed_bwd && agg && mean && rstd && ln_g && g_z_packed && (gq || gq_packed) && g_ln_g && g_ln_b,
               "bl_node_update_bwd: null pointer");
  BL_CHECK_ARG(bl_node_update_bwd_ok(Dm, Dout), "bl_node_update_bwd: needs Dm 128 or 256, Dout a multiple of 32 up to 256, deterministic mode off");
  BL_CHECK_ARG(bl_aligned16(g_out) && bl_aligned16(h_out) && bl_aligned16(wd_packed_bwd) && bl_aligned16(agg) && bl_aligned16(ln_g) &&
                   bl_aligned16(g_z_packed) && (dact == nullptr || bl_aligned16(dact)) && (gq == nullptr || bl_aligned16(gq)) &&
                   (gq_packed == nullptr || bl_aligned16(gq_packed)),
               "bl_node_update_bwd: misaligned pointer");
  BL_CHECK_ARG((uint64_t)nrows * (uint64_t)Dout < (1ull << 32) || drop.p <= 0.f, "bl_node_update_bwd: dropout index space is 32 bit");
  const bl_drop_dev d = bl_make_drop(drop);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (Dm == 128)
    rc = node_bwd_launch<1>(g_out, h_out, nrows, Dout, d, g_z_packed, g_bias, wd_packed_bwd, agg, mean, rstd, ln_g, dact, gq, gq_packed,
                            g_ln_g, g_ln_b, gq_amax, st);
  else
    rc = node_bwd_launch<2>(g_out, h_out, nrows, Dout, d, g_z_packed, g_bias, wd_packed_bwd, agg, mean, rstd, ln_g, dact, gq, gq_packed,
                            g_ln_g, g_ln_b, gq_amax, st);
  BL_LAUNCH_CHECK("bl_node_update_bwd");
  return rc;
}
