// fp32-accurate GEMMs on the fp16 matrix cores ("f16x3"): the message-passing GEMMs' second operand split.
//
// csrc/bl_gemm_x6.hip splits an fp32 operand into THREE bf16 planes and evaluates a product as six bf16 MFMA terms.  fp16 has
// 11 significant bits where bf16 has 8: TWO planes
//      x s = hi + lo (+ r, |r| <= 2^-24 |x s|),    hi = fp16(x s),  lo = fp16(x s - hi)          (s: a power of two, below)
// carry the same 22+ bits, and a product needs three terms
//      a b = (a_h b_h + a_h b_l + a_l b_h) / (s_a s_b)                                   (fp32 accumulate; a_l b_l < 2^-22 |ab| dropped)
// -- half the matrix-pipe work of bf16x6 and 4 instead of 6 bytes per packed element (the row GEMMs are co-limited by the
// delivery of the gathered operand).  Timing proxy before it was built (two planes / three terms of the
// bf16 kernels, results wrong; profiles/r06e_planes2_proxy.log): message GEMM 0.257 -> 0.167 ms, routed input-gradient GEMM
// 0.336 -> 0.242, weight gradient (128 x 128 tile) 0.234 -> 0.160 at the c2 layer shape.
//
// What fp16 does not have is bf16's exponent range (5 bits: 6e-8 ... 65504), so every packed tensor carries a power-of-two
// scale s that puts it into the upper part of the range, where BOTH planes are normal numbers:
//   * values with |x s| >= 2^-3 keep 22 bits (relative error 2^-24: fp32's own rounding unit);
//   * below that, lo becomes subnormal and the error is ABSOLUTE: <= 2^-25 / s.  With |x s| <= 2^15 that floor is 2^-40 of the
//     tensor's largest representable magnitude -- eight decades below fp32's relative precision at the top of the range.
//   * |x s| > 65504 saturates (finite values never become inf; +-inf and NaN propagate as NaN like in the bf16 split).
// Scales: layer inputs (tanh x dropout outputs, |h| <= 1.25; embedding rows) 2^8, weights 2^6 (|w| < 512) -- fixed, host-known
// constants of the call; gradient tensors have no bound known in advance: their packer takes the tensor's amax from DEVICE
// memory (written by the producing kernel with one atomic max per workgroup) and derives s = 2^(14 - ceil(log2 amax)); the
// consuming GEMM reads the same number and multiplies its result by 1 / (s_a s_b).  Power-of-two scaling commutes with every
// rounding involved, so results do not depend on s as long as nothing saturates or falls below the absolute floor.
// Error against fp64 on c2-like operands (emulation, tools/experiments/README.md "f16x3"): rms 2.4e-8 where a plain fp32
// matmul has 1.0e-7 and bf16x6 2.2e-9 -- below fp32 accumulation noise; parity tests keep their 1e-4 bound.
//
// Kernel shapes are bl_gemm_x6.hip's: 128 x 128 tile, 4 waves 2 x 2, 32 k's per LDS stage with a register prefetch, swapped MFMA
// operands (transposed accumulators), LDS-staged epilogue, XCD-aware work order (bl_x6_locate.h).  LDS stage row:
// [plane (2)][k-group slot (4)] x 16 B + 16 B of padding = 144 B (36 r mod 64 walks all sixteen 4-bank groups over 16 rows:
// fragment reads conflict-free).
#include <float.h>
#include <stdio.h>
#include <stdlib.h>

#include "bl_common.h"
#include "bl_x6_locate.h"
#include "bl_h3_image.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define HBM 128
#define HBN 128
#define HROW 9  // uint4 per LDS stage row: 2 planes x 4 k-groups + 1 pad

// ---- packing ------------------------------------------------------------------------------------
// scale from a device-resident amax (gradient tensors): s = 2^(14 - e) with 2^(e-1) < amax <= 2^e; amax == 0 (or non-finite) -> 1
__device__ __forceinline__ float h3_scale_from_amax(float amax) {
  if (!(amax > 0.f) || amax > FLT_MAX) return 1.f;
  int e;
  (void)frexpf(amax, &e);  // amax = m 2^e, m in [0.5, 1)
  int k = 14 - e;
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  return ldexpf(1.f, k);
}

// rows: out[r][plane][kg][j] = plane(x[r, 8 kg + j] * scale), planes back to back (row = 2 D halves)
// (kg_total, kg_off) as in pack_rows_kernel: a ConcatResidual pair is packed without a concatenated copy
__global__ __launch_bounds__(256) void pack_rows_h_kernel(const float* __restrict__ x, int ld, long long R, int D, uint4* __restrict__ out,
                                                          int kg_total, int kg_off, float scale, const float* __restrict__ amax_dev,
                                                          unsigned* __restrict__ sat_counter) {
  const int kgs = D >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * kgs) return;
  if (amax_dev) scale *= h3_scale_from_amax(*amax_dev);
  const long long r = t / kgs;
  const int kgn = kg_total;
  const int kg = (int)(t % kgs) + kg_off;
  x -= 8 * kg_off;
  const float4 a = *reinterpret_cast<const float4*>(x + r * ld + 8 * kg);
  const float4 b = *reinterpret_cast<const float4*>(x + r * ld + 8 * kg + 4);
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint16_t h[8], l[8];
  bool sat = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) split2h(v[j] * scale, h[j], l[j], sat);
  if (sat && sat_counter) atomicAdd(sat_counter, 1u);  // (rare by construction: the scales leave 2^6 - 2^8 of headroom)
  uint4* o = out + r * 2 * kgn + kg;
#define PK(a_, b_) ((uint32_t)(a_) | ((uint32_t)(b_) << 16))
  o[0] = make_uint4(PK(h[0], h[1]), PK(h[2], h[3]), PK(h[4], h[5]), PK(h[6], h[7]));
  o[kgn] = make_uint4(PK(l[0], l[1]), PK(l[2], l[3]), PK(l[4], l[5]), PK(l[6], l[7]));
}

__global__ __launch_bounds__(256) void pack_weights_h_kernel(const float* __restrict__ w, int G, int K, int N, int w_is_kn,
                                                             uint4* __restrict__ out, float scale, unsigned* __restrict__ sat_counter) {
  pack_weights_h_thread(w, G, K, N, w_is_kn, out, (long long)blockIdx.x * blockDim.x + threadIdx.x, scale, sat_counter);
}

// ---- row GEMM --------------------------------------------------------------------------------------
// C[rows of g] = out_scale * rows(a) . B_g, rows gathered from <= 3 packed sources; MASKED: the routed left operand (one source,
// winner bitmask) of the input-gradient GEMM.  out_scale = 1 / (s_a s_b) (host part) x *out_scale_dev (when the left operand's
// scale lives in device memory: the reciprocal of h3_scale_from_amax(*amax)).
// ONE: the reduced-precision form behind `train.py --amp` (bl_set_msg_gemm_mode(2)): the high planes only -- one fp16 MFMA term with
// fp32 accumulation, what torch.cuda.amp.autocast makes of a Linear -- from the same packed images (the low planes are not read).
template <bool MASKED, bool ONE>
__global__ __launch_bounds__(256, MASKED ? 2 : 3) void gemm_rows_h3_kernel(
    const uint4* __restrict__ xp0, const uint4* __restrict__ xp1, const uint4* __restrict__ xp2,
    const int* __restrict__ idx0, const int* __restrict__ idx1, const int* __restrict__ idx2, int w0, int w1, int w2,
    int koff1, int koff2, int nsrc, const uint32_t* __restrict__ win_bits, int ld_bits, const uint4* __restrict__ bp,
    long long strideB, const int* __restrict__ group_ptr, const int* __restrict__ group_w, int G, int M, int N, int K,
    float* __restrict__ c, int ldc, int xcd_remap, float out_scale, const float* __restrict__ a_amax_dev) {
  __shared__ uint4 ABs[(HBM + HBN) * HROW];  // 36 KB; after the last stage the waves' result tiles are staged in it (4 x 8.5 KB)
  uint4* As = ABs;
  uint4* Bs = ABs + HBM * HROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g, row0, nrows, tile_y;
  if (!x6_locate(group_ptr, G, M, HBM, xcd_remap, tile_y, g, row0, nrows)) return;
  const int n0 = tile_y * HBN;
  const int wsel = group_w ? group_w[g] : g;
  const uint4* __restrict__ Bt = bp + (long long)wsel * strideB + (size_t)tile_y * (K >> 5) * 1024 + tid;

  const int p_kg = tid & 3, p_row0 = tid >> 2;  // rows p_row0 and p_row0 + 64
  int gr0[2], gr1[2], gr2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = row0 + min(p_row0 + 64 * i, nrows - 1);
    gr0[i] = idx0 ? idx0[r] : r;
    gr1[i] = nsrc > 1 ? (idx1 ? idx1[r] : r) : 0;
    gr2[i] = nsrc > 2 ? (idx2 ? idx2[r] : r) : 0;
  }
  uint4 ra[2][2], rb[2][2];
  uint32_t ma[2];
  const int nk = (K + 31) / 32;

#define H3_LOAD_STAGE(k0_)                                                                                    \
  {                                                                                                           \
    const int k_ = (k0_) + 8 * p_kg;                                                                          \
    const int kc_ = k_ < K ? k_ : 0;                                                                          \
    int j_ = 0;                                                                                               \
    if (nsrc > 1 && kc_ >= koff1) j_ = 1;                                                                     \
    if (nsrc > 2 && kc_ >= koff2) j_ = 2;                                                                     \
    const int kl_ = kc_ - (j_ == 0 ? 0 : (j_ == 1 ? koff1 : koff2));                                          \
    const uint4* base_ = j_ == 0 ? xp0 : (j_ == 1 ? xp1 : xp2);                                               \
    const int wj_ = j_ == 0 ? w0 : (j_ == 1 ? w1 : w2);                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                           \
      const int row_ = p_row0 + 64 * i;                                                                       \
      const int gr_ = j_ == 0 ? gr0[i] : (j_ == 1 ? gr1[i] : gr2[i]);                                         \
      const uint4* src_ = base_ + (size_t)gr_ * 2 * (wj_ >> 3) + (kl_ >> 3);                                  \
      ra[i][0] = src_[0];                                                                                     \
      if (!ONE) ra[i][1] = src_[wj_ >> 3];                                                                    \
      if (MASKED) ma[i] = win_bits[(size_t)(row0 + min(row_, nrows - 1)) * ld_bits + (kc_ >> 5)];              \
      const uint4* bsrc_ = Bt + (size_t)((k0_) >> 5) * 1024 + i * 512;                                        \
      rb[i][0] = bsrc_[0];                                                                                    \
      if (!ONE) rb[i][1] = bsrc_[256];                                                                        \
    }                                                                                                         \
  }
#define H3_STORE_STAGE(k0_)                                                                                   \
  {                                                                                                           \
    const bool kok_ = (k0_) + 8 * p_kg < K;                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                           \
      const int row_ = p_row0 + 64 * i;                                                                       \
      uint4 keep_ = make_uint4(~0u, ~0u, ~0u, ~0u);                                                           \
      if (MASKED) keep_ = keep_from_bits(ma[i] >> (8 * p_kg)); /* k0 is a multiple of 32 */                  \
      if (!kok_) keep_ = make_uint4(0u, 0u, 0u, 0u);                                                          \
      const bool nok_ = kok_ && (n0 + row_ < N);                                                              \
      _Pragma("unroll") for (int p = 0; p < (ONE ? 1 : 2); ++p) {                                             \
        uint4 a_ = ra[i][p];                                                                                  \
        a_.x &= keep_.x; a_.y &= keep_.y; a_.z &= keep_.z; a_.w &= keep_.w;                                   \
        As[row_ * HROW + p * 4 + p_kg] = a_;                                                                  \
        Bs[row_ * HROW + p * 4 + p_kg] = nok_ ? rb[i][p] : make_uint4(0u, 0u, 0u, 0u);                        \
      }                                                                                                       \
    }                                                                                                         \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, half = lane >> 5;

  H3_LOAD_STAGE(0)
  H3_STORE_STAGE(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) H3_LOAD_STAGE((kt + 1) * 32)
#pragma unroll
    for (int s = 0; s < 2; ++s) {  // two 16-k MFMA steps per stage; this lane's 8 k's = group 2s + half
      const int kg = 2 * s + half;
      f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        const uint4* p = &As[(wm * 64 + ti * 32 + li) * HROW + kg];
        ah[ti] = __builtin_bit_cast(f16x8, p[0]);
        if (!ONE) al[ti] = __builtin_bit_cast(f16x8, p[4]);
      }
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        const uint4* p = &Bs[(wn * 64 + tj * 32 + li) * HROW + kg];
        bh[tj] = __builtin_bit_cast(f16x8, p[0]);
        if (!ONE) bl[tj] = __builtin_bit_cast(f16x8, p[4]);
      }
      // swapped operands (B fragment in the A slot): the accumulator holds the transposed tile; small terms first
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          f32x16 a = acc[ti][tj];
          if (!ONE) {
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[tj], ah[ti], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[tj], al[ti], a, 0, 0, 0);
          }
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[tj], ah[ti], a, 0, 0, 0);
          acc[ti][tj] = a;
        }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      H3_STORE_STAGE((kt + 1) * 32)
      __syncthreads();
    }
  }

  if (a_amax_dev) out_scale /= h3_scale_from_amax(*a_amax_dev);
  // result tile through LDS (the operand images are dead), leaving as whole 256-byte row pieces (see bl_gemm_x6.hip)
  float* stage = reinterpret_cast<float*>(ABs) + wave * (32 * 68);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<float4*>(stage + li * 68 + tj * 32 + 8 * gq + 4 * half) =
            make_float4(acc[ti][tj][4 * gq + 0] * out_scale, acc[ti][tj][4 * gq + 1] * out_scale, acc[ti][tj][4 * gq + 2] * out_scale,
                        acc[ti][tj][4 * gq + 3] * out_scale);
    const int c4 = lane & 15, n = n0 + wn * 64 + 4 * c4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = (lane >> 4) + 4 * j;
      const int mm = wm * 64 + ti * 32 + r;
      const float4 v = *reinterpret_cast<const float4*>(stage + r * 68 + 4 * c4);
      if (mm < nrows && n < N) bl_store_streaming(c + (size_t)(row0 + mm) * ldc + n, v);
    }
  }
}


// ---- weight-gradient GEMM (128 x 128 tile) ---------------------------------------------------------
// gW_g[i, n] += out_scale * sum_{e in group g} A[e, i] * Gr[e, n]   (bl_gemm_x6.hip::gemm_wgrad_x6_kernel with two planes:
// operands stored in LDS as they arrive, [plane][message][feature] rows of 320 B, fragments by ds_read_b64_tr_b16)
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define HWRS 160
#define HWPLANE (32 * HWRS)
#define HWOPER (2 * HWPLANE)

__device__ __forceinline__ f16x8 h3_tr_frag(const short* p) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * HWRS));
  return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <bool ROUTED, bool ONE>
__global__ __launch_bounds__(256, 2) void gemm_wgrad_h3_kernel(
    const uint4* __restrict__ xp0, const uint4* __restrict__ xp1, const uint4* __restrict__ xp2,
    const int* __restrict__ idx0, const int* __restrict__ idx1, const int* __restrict__ idx2, int w0, int w1, int w2,
    int koff1, int koff2, int nsrc, const uint4* __restrict__ gp, const int* __restrict__ g_idx,
    const uint32_t* __restrict__ win_bits, int ld_bits, const int* __restrict__ group_ptr, const int* __restrict__ group_w, int G,
    int M, int N, int K, int kchunk, float* __restrict__ gw_base, long long strideW, int ldw, int ntiles_n, int xcd_remap,
    unsigned* __restrict__ order_ctr, float out_scale, const float* __restrict__ g_amax_dev) {
  __shared__ __attribute__((aligned(16))) short As[HWOPER];
  __shared__ __attribute__((aligned(16))) short Bs[HWOPER];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g, e0, ne, tile_y;
  if (!x6_locate(group_ptr, G, M, kchunk, xcd_remap, tile_y, g, e0, ne)) return;
  const int e1 = e0 + ne;
  const int i0 = (tile_y / ntiles_n) * HBM;
  const int n0 = (tile_y % ntiles_n) * HBN;
  const int wsel = group_w ? group_w[g] : g;

  const int fg = tid & 15, msg0 = tid >> 4;  // messages msg0 and msg0 + 16
  const int fi = i0 + 8 * fg, nn = n0 + 8 * fg;
  const bool a_ok = fi < K, b_ok = nn < N;
  const int fic = a_ok ? fi : 0, nnc = b_ok ? nn : 0;
  int aj = 0;
  if (nsrc > 1 && fic >= koff1) aj = 1;
  if (nsrc > 2 && fic >= koff2) aj = 2;
  const uint4* __restrict__ abase = (aj == 0 ? xp0 : (aj == 1 ? xp1 : xp2)) + ((fic - (aj == 0 ? 0 : (aj == 1 ? koff1 : koff2))) >> 3);
  const int* __restrict__ aidx = aj == 0 ? idx0 : (aj == 1 ? idx1 : idx2);
  const int awg = (aj == 0 ? w0 : (aj == 1 ? w1 : w2)) >> 3;  // uint4 per plane of an A row
  const int gwg = N >> 3;
  const uint4* __restrict__ gbase = gp + (nnc >> 3);
  const uint32_t* __restrict__ mbase = ROUTED ? win_bits + (nnc >> 5) : nullptr;
  const int mshift = nnc & 31;

  uint4 ra[2][2], rb[2][2];
  uint32_t mk[2];
  int arow[2], grow[2], mrow[2];

#define HW_LOAD_IDX(k0_)                                           \
  {                                                                \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                \
      const int e_ = (k0_) + msg0 + 16 * i;                        \
      const int ec_ = e_ < e1 ? e_ : e0;                           \
      arow[i] = aidx ? aidx[ec_] : ec_;                            \
      grow[i] = g_idx ? g_idx[ec_] : ec_;                          \
      mrow[i] = ec_;                                               \
    }                                                              \
  }
#define HW_LOAD_STAGE()                                                                          \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                              \
      const uint4* a_ = abase + (size_t)arow[i] * 2 * awg;                                       \
      ra[i][0] = a_[0];                                                                          \
      if (!ONE) ra[i][1] = a_[awg];                                                              \
      const uint4* g_ = gbase + (size_t)grow[i] * 2 * gwg;                                       \
      rb[i][0] = g_[0];                                                                          \
      if (!ONE) rb[i][1] = g_[gwg];                                                              \
      mk[i] = ROUTED ? mbase[(size_t)mrow[i] * ld_bits] : 0u;                                    \
    }                                                                                            \
  }
#define HW_STORE_STAGE(k0_)                                                                      \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                              \
      const int eid_ = (k0_) + msg0 + 16 * i;                                                    \
      const bool eok_ = eid_ < e1;                                                               \
      uint4 keep_ = ROUTED ? keep_from_bits(mk[i] >> mshift) : make_uint4(~0u, ~0u, ~0u, ~0u);   \
      if (!(eok_ && b_ok)) keep_ = make_uint4(0u, 0u, 0u, 0u);                                   \
      const int slot_ = (msg0 + 16 * i) * HWRS + 8 * fg;                                         \
      _Pragma("unroll") for (int p = 0; p < (ONE ? 1 : 2); ++p) {                                \
        *reinterpret_cast<uint4*>(&As[p * HWPLANE + slot_]) = (eok_ && a_ok) ? ra[i][p] : make_uint4(0u, 0u, 0u, 0u); \
        uint4 b_ = rb[i][p];                                                                     \
        b_.x &= keep_.x; b_.y &= keep_.y; b_.z &= keep_.z; b_.w &= keep_.w;                      \
        *reinterpret_cast<uint4*>(&Bs[p * HWPLANE + slot_]) = b_;                                \
      }                                                                                          \
    }                                                                                            \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, half = lane >> 5;
  const int l16 = lane & 15, grp = lane >> 4;
  const int tr_off = ((grp >> 1) * 8 + (l16 >> 2)) * HWRS + (grp & 1) * 16 + 4 * (l16 & 3);
  const short* a_tr = As + tr_off + wm * 64;
  const short* b_tr = Bs + tr_off + wn * 64;
  const int nk = (ne + 31) / 32;

  HW_LOAD_IDX(e0)
  HW_LOAD_STAGE()
  HW_LOAD_IDX(e0 + 32)
  HW_STORE_STAGE(e0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      HW_LOAD_STAGE()
      HW_LOAD_IDX(e0 + (kt + 2) * 32)
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {  // two 16-message MFMA steps per stage
      f16x8 af[2][2], bf[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < (ONE ? 1 : 2); ++p) {
          af[t][p] = h3_tr_frag(a_tr + p * HWPLANE + s * 16 * HWRS + t * 32);
          bf[t][p] = h3_tr_frag(b_tr + p * HWPLANE + s * 16 * HWRS + t * 32);
        }
#define HW_TERM(pa_, pb_)                                                                             \
  _Pragma("unroll") for (int ti = 0; ti < 2; ++ti) _Pragma("unroll") for (int tj = 0; tj < 2; ++tj)   \
      acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ti][pa_], bf[tj][pb_], acc[ti][tj], 0, 0, 0);
      if (!ONE) { HW_TERM(1, 0) HW_TERM(0, 1) }
      HW_TERM(0, 0)
    }
    __syncthreads();
    if (kt + 1 < nk) {
      HW_STORE_STAGE(e0 + (kt + 1) * 32)
      __syncthreads();
    }
  }

  if (g_amax_dev) out_scale /= h3_scale_from_amax(*g_amax_dev);
  float* __restrict__ gw = gw_base + (long long)wsel * strideW;
  unsigned* ctr = order_ctr ? order_ctr + (size_t)g * gridDim.y + tile_y : nullptr;
  const unsigned turn = (unsigned)((e0 - (group_ptr ? group_ptr[g] : 0)) / kchunk);
  bl_ordered_enter(ctr, turn);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int n = n0 + wn * 64 + tj * 32 + li;
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = i0 + wm * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (f < K) unsafeAtomicAdd(&gw[(size_t)f * ldw + n], acc[ti][tj][r] * out_scale);
      }
    }
  bl_ordered_leave(ctr, turn);
}

// ---- amax of a tensor into device memory (gradient operands) ----------------------------------------
// *amax = max(*amax, max |x|): non-negative floats order like their bit patterns; the caller zeroes *amax first
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long n4, float* __restrict__ amax) {
  __shared__ float part[4];
  float m = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  m = bl_wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  // ONE atomic per workgroup, few workgroups: same-address atomics serialise at the L2 (~40 ns each)
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m));
  }
}

// ---- saturation events ----------------------------------------------------------------------------------
// One unsigned per device, allocated on first use and never freed (like the ordered-flush ring of bl_core.hip): a packing thread
// that had to clamp a finite value to +-65504 adds 1.  The fixed scales leave 2^6 - 2^8 of headroom over every bounded tensor, so a
// non-zero count means a layer input beyond +-255.9 or a weight beyond +-1023 -- a diverging run, or a model this split does not fit.
unsigned* bl_h3_sat_counter() {
  static unsigned* ctr[BL_MAX_DEVICES] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BL_MAX_DEVICES) return nullptr;
  if (ctr[dev] == nullptr) {
    if (hipMalloc((void**)&ctr[dev], sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(ctr[dev], 0, sizeof(unsigned)) != hipSuccess) return nullptr;
  }
  return ctr[dev];
}

// number of saturation events on the current device since the last reset (synchronises the device); -1 if unavailable
extern "C" int64_t bl_h3_saturation_events(int32_t reset) {
  unsigned* c = bl_h3_sat_counter();
  unsigned v = 0;
  if (c == nullptr || hipMemcpy(&v, c, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (reset && hipMemset(c, 0, sizeof(unsigned)) != hipSuccess) return -1;
  return (int64_t)v;
}

// ================================================================================================
extern "C" int bl_pack_f16x2(const float* x, int32_t ld, int64_t R, int32_t D, int32_t D_total, int32_t col_off, float scale,
                             const float* amax_dev, uint16_t* out, void* stream) {
  if (R == 0) return BL_OK;
  BL_CHECK_ARG(x && out && bl_aligned16(x) && bl_aligned16(out), "bl_pack_f16x2: null or misaligned pointer");
  BL_CHECK_ARG(D > 0 && D % 8 == 0 && ld % 4 == 0 && D_total % 8 == 0 && col_off % 8 == 0 && col_off >= 0 && col_off + D <= D_total,
               "bl_pack_f16x2: widths / offset must be multiples of 8 with col_off + D <= D_total");
  BL_CHECK_ARG(scale > 0.f, "bl_pack_f16x2: scale must be positive (a power of two)");
  const long long total = (long long)R * (D / 8);
  hipLaunchKernelGGL(pack_rows_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld, (long long)R, D,
                     reinterpret_cast<uint4*>(out), D_total / 8, col_off / 8, scale, amax_dev, bl_h3_sat_counter());
  BL_LAUNCH_CHECK("bl_pack_f16x2");
  return BL_OK;
}

extern "C" int bl_amax(const float* x, int64_t n, float* amax_dev, void* stream) {
  if (n == 0) return BL_OK;
  BL_CHECK_ARG(x && amax_dev && bl_aligned16(x) && n % 4 == 0, "bl_amax: aligned pointer and a multiple of 4 elements required");
  const long long n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 < (long long)bl_num_cus() * 2 ? (n4 + 255) / 256 : (long long)bl_num_cus() * 2);
  hipLaunchKernelGGL(amax_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n4, amax_dev);
  BL_LAUNCH_CHECK("bl_amax");
  return BL_OK;
}

extern "C" int64_t bl_packed_weight_elems_h3(int32_t G, int32_t K, int32_t N) { return (int64_t)G * ((N + 127) / 128) * (K / 32) * 8192; }

extern "C" int bl_pack_weights_h3(const float* w, int32_t G, int32_t K, int32_t N, int32_t w_is_kn, float scale, uint16_t* out,
                                  void* stream) {
  if (G == 0) return BL_OK;
  BL_CHECK_ARG(w && out && bl_aligned16(out), "bl_pack_weights_h3: null or misaligned pointer");
  BL_CHECK_ARG(K > 0 && K % 32 == 0 && N > 0 && scale > 0.f, "bl_pack_weights_h3: K must be a multiple of 32 (got %d), scale positive", K);
  const long long total = (long long)G * ((N + 127) / 128) * (K / 32) * 512;
  hipLaunchKernelGGL(pack_weights_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, G, K, N, w_is_kn,
                     reinterpret_cast<uint4*>(out), scale, bl_h3_sat_counter());
  BL_LAUNCH_CHECK("bl_pack_weights_h3");
  return BL_OK;
}

// 1: the f16x3 GEMMs of this file evaluate the high-plane term only (bl_set_msg_gemm_mode(2), `train.py --amp`); see the ONE forms
bool g_h3_one_term = false;

extern "C" int bl_gemm_rows_h3(const bl_rows_packed_t* a, const uint32_t* win_bits, int32_t ld_bits, const uint16_t* bp,
                               int64_t b_group_stride, const int32_t* group_ptr, const int32_t* group_w, int32_t G, int32_t M, int32_t N,
                               int32_t K, float out_scale, const float* a_amax_dev, float* c, int32_t ldc, void* stream) {
  const char* who = "bl_gemm_rows_h3";
  if (M == 0) return BL_OK;
  BL_CHECK_ARG(a && a->nsrc >= 1 && a->nsrc <= 3, "%s: rows descriptor needs 1..3 sources", who);
  int off = 0, koff[3] = {0, 0, 0};
  for (int j = 0; j < a->nsrc; ++j) {
    BL_CHECK_ARG(a->xp[j] && bl_aligned16(a->xp[j]) && a->width[j] > 0 && a->width[j] % 32 == 0,
                 "%s: source %d: packed pointer 16-byte aligned and width a multiple of 32 required", who, j);
    koff[j] = off;
    off += a->width[j];
  }
  BL_CHECK_ARG(off == K, "%s: K (%d) != sum of source widths (%d)", who, K, off);
  BL_CHECK_ARG(M > 0 && N > 0 && N % 4 == 0 && ldc % 4 == 0 && bp && c && bl_aligned16(bp) && bl_aligned16(c),
               "%s: N/ldc multiples of 4, aligned pointers required", who);
  BL_CHECK_ARG(b_group_stride % 8 == 0 && (G <= 1 || b_group_stride >= bl_packed_weight_elems_h3(1, K, N)),
               "%s: packed group stride must cover one group's tiled weights (bl_pack_weights_h3)", who);
  BL_CHECK_ARG(win_bits == nullptr || (a->nsrc == 1 && a->idx[0] && ld_bits * 32 >= K),
               "%s: the routed form needs exactly one gathered source and ld_bits >= K / 32", who);
  BL_CHECK_ARG(out_scale > 0.f, "%s: out_scale must be positive", who);
  dim3 grid((M + HBM - 1) / HBM + (group_ptr ? G : 0), (N + HBN - 1) / HBN);
  const uint4* x0 = reinterpret_cast<const uint4*>(a->xp[0]);
  const uint4* x1 = a->nsrc > 1 ? reinterpret_cast<const uint4*>(a->xp[1]) : nullptr;
  const uint4* x2 = a->nsrc > 2 ? reinterpret_cast<const uint4*>(a->xp[2]) : nullptr;
#define H3_ARGS                                                                                                          \
  x0, x1, x2, a->idx[0], a->nsrc > 1 ? a->idx[1] : nullptr, a->nsrc > 2 ? a->idx[2] : nullptr, a->width[0],              \
      a->nsrc > 1 ? a->width[1] : 0, a->nsrc > 2 ? a->width[2] : 0, koff[1], koff[2], a->nsrc, win_bits, ld_bits,        \
      reinterpret_cast<const uint4*>(bp), (long long)(b_group_stride / 8), group_ptr, group_w, G, M, N, K, c, ldc, 1, out_scale, a_amax_dev
  if (g_h3_one_term) {
    if (win_bits)
      hipLaunchKernelGGL((gemm_rows_h3_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, H3_ARGS);
    else
      hipLaunchKernelGGL((gemm_rows_h3_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, H3_ARGS);
  } else if (win_bits)
    hipLaunchKernelGGL((gemm_rows_h3_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, H3_ARGS);
  else
    hipLaunchKernelGGL((gemm_rows_h3_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, H3_ARGS);
  BL_LAUNCH_CHECK(who);
  return BL_OK;
}

int g_h3_kchunk_cap = 4096;  // rows per workgroup flush; follows bl_set_wgrad_kchunk_cap (csrc/bl_gemm_x6.hip)
namespace {
template <bool ROUTED>
int wgrad_h3_resident() {
  static int resident = 0;
  if (resident == 0) {
    int per_cu = 0;
    hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_wgrad_h3_kernel<ROUTED, false>, 256, 0);
    if (oe != hipSuccess || per_cu <= 0) per_cu = 2;
    resident = per_cu * bl_num_cus();
  }
  return resident;
}
}  // namespace

extern "C" int bl_gemm_wgrad_h3(const bl_rows_packed_t* a, const uint16_t* g_packed, const int32_t* g_idx, const uint32_t* win_bits,
                                int32_t ld_bits, const int32_t* group_ptr, const int32_t* group_w, int32_t G, int32_t M, int32_t N,
                                int32_t K, float out_scale, const float* g_amax_dev, float* gw, int64_t gw_group_stride, int32_t ld_gw,
                                void* stream) {
  const char* who = "bl_gemm_wgrad_h3";
  if (M == 0) return BL_OK;
  BL_CHECK_ARG(a && a->nsrc >= 1 && a->nsrc <= 3, "%s: rows descriptor needs 1..3 sources", who);
  int off = 0, koff[3] = {0, 0, 0};
  for (int j = 0; j < a->nsrc; ++j) {
    BL_CHECK_ARG(a->xp[j] && bl_aligned16(a->xp[j]) && a->width[j] > 0 && a->width[j] % 32 == 0,
                 "%s: source %d: packed pointer 16-byte aligned and width a multiple of 32 required", who, j);
    koff[j] = off;
    off += a->width[j];
  }
  BL_CHECK_ARG(off == K, "%s: K (%d) != sum of source widths (%d)", who, K, off);
  BL_CHECK_ARG(M > 0 && N > 0 && N % 32 == 0 && g_packed && gw && bl_aligned16(g_packed) && out_scale > 0.f,
               "%s: N a multiple of 32, aligned pointers and a positive out_scale required", who);
  const bool routed = win_bits != nullptr;
  BL_CHECK_ARG(!routed || (g_idx && ld_bits * 32 >= N), "%s: the routed form needs g_idx and ld_bits >= N / 32", who);
  const int resident = routed ? wgrad_h3_resident<true>() : wgrad_h3_resident<false>();
  const int ntiles_n = (N + HBN - 1) / HBN;
  const int ntiles_all = ((K + HBM - 1) / HBM) * ntiles_n;
  const int extra = (group_ptr ? G : 0) * ntiles_all;
  int kchunk = 256;
  for (int rounds = 1; rounds <= 64; ++rounds) {
    const long long slots = (long long)resident * rounds - extra;
    if (slots <= 0) continue;
    const long long kc = ((long long)M * ntiles_all + slots - 1) / slots;
    if (kc <= g_h3_kchunk_cap || rounds == 64) {
      kchunk = (int)((kc + 31) / 32 * 32);
      break;
    }
  }
  if (kchunk < 256) kchunk = 256;
  dim3 grid((M + kchunk - 1) / kchunk + (group_ptr ? G : 0), ntiles_all);
  unsigned* order_ctr = group_w ? nullptr : bl_order_counters((group_ptr ? G : 1) * ntiles_all, stream);
  const int xcd = order_ctr ? 0 : 1;
#define HW_ARGS                                                                                                                \
  reinterpret_cast<const uint4*>(a->xp[0]), a->nsrc > 1 ? reinterpret_cast<const uint4*>(a->xp[1]) : nullptr,                  \
      a->nsrc > 2 ? reinterpret_cast<const uint4*>(a->xp[2]) : nullptr, a->idx[0], a->nsrc > 1 ? a->idx[1] : nullptr,          \
      a->nsrc > 2 ? a->idx[2] : nullptr, a->width[0], a->nsrc > 1 ? a->width[1] : 0, a->nsrc > 2 ? a->width[2] : 0, koff[1],   \
      koff[2], a->nsrc, reinterpret_cast<const uint4*>(g_packed), g_idx, win_bits, ld_bits, group_ptr, group_w, G, M, N, K,     \
      kchunk, gw, (long long)gw_group_stride, ld_gw, ntiles_n, xcd, order_ctr, out_scale, g_amax_dev
  if (g_h3_one_term) {
    if (routed)
      hipLaunchKernelGGL((gemm_wgrad_h3_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, HW_ARGS);
    else
      hipLaunchKernelGGL((gemm_wgrad_h3_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, HW_ARGS);
  } else if (routed)
    hipLaunchKernelGGL((gemm_wgrad_h3_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, HW_ARGS);
  else
    hipLaunchKernelGGL((gemm_wgrad_h3_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, HW_ARGS);
  BL_LAUNCH_CHECK(who);
  return BL_OK;
}
