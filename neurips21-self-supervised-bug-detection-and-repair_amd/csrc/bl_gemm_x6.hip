// fp32-accurate GEMMs on the bf16 matrix cores ("bf16x6").
//
// The exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 MFMA rate, and at the
// ~1.9 GHz the chip sustains under this load the message / input-gradient GEMMs are pinned at
// 85-110 TF/s.  Here every fp32 operand x is split ONCE into three bf16 terms
//      x = hi + mid + lo  (+ r, |r| <= 2^-27 |x|),   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
// and a product a*b is evaluated as the six bf16 x bf16 MFMA terms
//      a_h b_h + a_h b_m + a_m b_h + a_h b_l + a_l b_h + a_m b_m          (fp32 accumulate),
// dropping only terms below 2^-26 |ab| -- smaller than fp32's own rounding unit (2^-24): results
// are fp32-equivalent (the parity tests keep the same 1e-4 bound and pass with ~1e-6).  Six
// v_mfma_f32_32x32x16_bf16 (32 cycles each) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each)
// per 16 k's: 2.67x fewer matrix-pipe cycles.
//
// Packed operand layout ("bf16x3 packed", produced by bl_pack_bf16x3*): every row is three bf16
// planes back to back, [hi x D | mid x D | lo x D] (6 D bytes): a row's plane is contiguous, which
// is what both the k-contiguous loads of this file's row GEMM (64 B per row, plane and 32-k stage)
// and the feature-contiguous loads of the weight-gradient GEMM want.  In LDS a stage row of the row
// GEMM is [plane][k-group][8] (row stride 208 B: b128 reads conflict-free, the 8-lane write groups
// overlap in one of four slots), an MFMA fragment (8 k's of one row) is three ds_read_b128 64 B apart.
//
// Kernel shape is the fp32 one's (csrc/bl_gemm.hip): 128 x 128 tile, 4 waves 2 x 2, transposed
// accumulators -> float4 epilogue stores, wave-cooperative group lookup, optional routed
// (winner-masked) left operand for the input-gradient GEMM.
#include <stdio.h>
#include <stdlib.h>

#include "bl_common.h"
#include "bl_x6_locate.h"
#include "bl_x6w_image.h"
#include "bl_h3_image.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define XBM 128
#define XBN 128
// LDS row of a stage image: [plane (3)][k-group slot (4)] x 16 B, in one of two layouts:
//   padded    208-byte rows (13 uint4): fragment reads conflict-free, the staging stores' 16-lane groups overlap in 4 of 64 banks
//             (SQ_LDS_BANK_CONFLICT a third of SQ_LDS_IDX_ACTIVE, profiles/r04z_fwd_gemm_pmc.json);
//   swizzled  192-byte rows, k-group kg of a row in slot kg ^ ((row >> 2) & 3): four consecutive rows' 64-byte plane segments tile
//             the 64 banks and the reads of 16 consecutive rows hit 16 different (segment, slot) pairs -- conflict-free both ways.
// Measured (profiles/r04y_swizzle.log): the routed form gains 2 % (H = 128 layer) / 4.3 % (concat layer), the plain form loses 2 %
// at the H = 128 layer (equal at the concat layer).  So: swizzled for the routed form, padded for the plain one (X6_SWZ: 0 = padded
// everywhere, 1 = swizzled everywhere, 2 = as measured).
#define X6_SWZ 2
#define X6_SWIZZLED(masked_) (X6_SWZ == 1 || (X6_SWZ == 2 && (masked_)))
#define XROW_MAX 13


// ---- packing ------------------------------------------------------------------------------------
// rows: out[r][kg][plane][j] = plane(x[r, 8 kg + j])
// (kg_total, kg_off): the packed row is kg_total k-groups wide per plane and this source fills the groups
// kg_off .. kg_off + D/8 -- how [stash ; current] of a ConcatResidual layer is packed without a concatenated copy
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ x, int ld, long long R, int D,
                                                        uint4* __restrict__ out, int kg_total, int kg_off) {
  const int kgs = D >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * kgs) return;
  const long long r = t / kgs;
  const int kgn = kg_total;
  const int kg = (int)(t % kgs) + kg_off;
  x -= 8 * kg_off;  // column index below is relative to the packed row
  const float4 a = *reinterpret_cast<const float4*>(x + r * ld + 8 * kg);
  const float4 b = *reinterpret_cast<const float4*>(x + r * ld + 8 * kg + 4);
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint16_t h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3(v[j], h[j], m[j], l[j]);
  uint4* o = out + r * 3 * kgn + kg;
#define PK(a_, b_) ((uint32_t)(a_) | ((uint32_t)(b_) << 16))
  o[0] = make_uint4(PK(h[0], h[1]), PK(h[2], h[3]), PK(h[4], h[5]), PK(h[6], h[7]));
  o[kgn] = make_uint4(PK(m[0], m[1]), PK(m[2], m[3]), PK(m[4], m[5]), PK(m[6], m[7]));
  o[2 * kgn] = make_uint4(PK(l[0], l[1]), PK(l[2], l[3]), PK(l[4], l[5]), PK(l[6], l[7]));
}

// The B operand (weights) is packed TILED: for every group, 128-column tile and 32-k stage one
// contiguous 24 KB block, ordered [i (2)][plane (3)][row_lo (64)][k-group (4)] x 8 bf16, column
// n = 64 i + row_lo of the tile.  Thread t of the GEMM reads uint4s t, t + 256, ... of the block: every
// wave-load is 1 KB contiguous, and the block's 24 KB spread over all L2 channels.  With the row-major
// packed form ([n][plane][K], 16 x 64-byte pieces per wave-load, the same few lines requested by every
// workgroup of an edge type at once) the B loads cost 1.8x more (load-only microbenchmark, c2 shapes:
// 0.082 -> 0.045 ms per GEMM).  Columns past N are zero.
// w is [G][K][N] (w_is_kn = 1: the forward weights, transposed on the fly) or [G][N][K] (w_is_kn = 0).
__device__ __forceinline__ void pack_weights_thread(const float* __restrict__ w, int G, int K, int N, int w_is_kn,
                                                    uint4* __restrict__ out, long long t) {
  const int nst = K >> 5, ntn = (N + 127) >> 7;  // t = one (g, tile, stage, i, row_lo, kg)
  if (t >= (long long)G * ntn * nst * 512) return;
  int r = (int)(t & 511);
  const long long blk = t >> 9;
  const int st = (int)(blk % nst), tile = (int)((blk / nst) % ntn), g = (int)(blk / ((long long)nst * ntn));
  // consecutive threads -> consecutive n when the source is [K][N] (coalesced), consecutive k-groups otherwise
  int i, row_lo, kg;
  if (w_is_kn) { row_lo = r & 63; i = (r >> 6) & 1; kg = r >> 7; }
  else { kg = r & 3; row_lo = (r >> 2) & 63; i = r >> 8; }
  const int n = tile * 128 + 64 * i + row_lo, k0 = st * 32 + 8 * kg;
  uint16_t h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = 0.f;
    if (n < N) v = w_is_kn ? w[((size_t)g * K + k0 + j) * N + n] : w[((size_t)g * N + n) * K + k0 + j];
    split3(v, h[j], m[j], l[j]);
  }
  uint4* o = out + (size_t)blk * 1536 + i * 768 + row_lo * 4 + kg;
  o[0] = make_uint4(PK(h[0], h[1]), PK(h[2], h[3]), PK(h[4], h[5]), PK(h[6], h[7]));
  o[256] = make_uint4(PK(m[0], m[1]), PK(m[2], m[3]), PK(m[4], m[5]), PK(m[6], m[7]));
  o[512] = make_uint4(PK(l[0], l[1]), PK(l[2], l[3]), PK(l[4], l[5]), PK(l[6], l[7]));
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int G, int K, int N, int w_is_kn,
                                                           uint4* __restrict__ out) {
  pack_weights_thread(w, G, K, N, w_is_kn, out, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// Every operand copy a training step needs of the (just updated) weights in ONE launch: a table of jobs in device memory
// (built once per model by the caller), job j owning the workgroups first_block[j] .. first_block[j+1].
//   kind 0 / 1: bl_pack_weights_x6 with w_is_kn = kind;  kind 2: out[g][n][k] = w[g][k][n] in fp32 (W transposed, the
//   operand of bl_routed_dgrad_nodes);  kind 3 / 4: bl_pack_weights_x6w (the wide row GEMM's image) with w_is_kn = kind - 3;
//   kind 5 / 6: bl_pack_weights_h3 (the f16x3 image) with w_is_kn = kind - 5.
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const bl_pack_job_t* __restrict__ jobs, int njobs,
                                                                 unsigned* __restrict__ h3_sat_counter) {
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;  // (a few dozen jobs: a linear walk of a cached table)
  const bl_pack_job_t job = jobs[j];
  const long long t = (long long)((int)blockIdx.x - job.first_block) * 256 + threadIdx.x;
  if (job.kind <= 1) {
    pack_weights_thread(job.w, job.G, job.K, job.N, job.kind, reinterpret_cast<uint4*>(job.out), t);
  } else if (job.kind >= 5) {  // 5 / 6: bl_pack_weights_h3 (the f16x3 image, scale BL_H3_W_SCALE) with w_is_kn = kind - 5
    pack_weights_h_thread(job.w, job.G, job.K, job.N, job.kind - 5, reinterpret_cast<uint4*>(job.out), t, BL_H3_W_SCALE, h3_sat_counter);
  } else if (job.kind >= 3) {
    pack_weights_wide_thread(job.w, job.G, job.K, job.N, job.kind - 3, reinterpret_cast<uint4*>(job.out), t);
  } else {
    const long long per = (long long)job.K * job.N;
    if (t >= per * job.G) return;
    const int g = (int)(t / per);
    const long long r = t - (long long)g * per;
    const int n = (int)(r / job.K), k = (int)(r - (long long)n * job.K);  // consecutive threads -> consecutive k of the output row
    reinterpret_cast<float*>(job.out)[t] = job.w[((size_t)g * job.K + k) * job.N + n];
  }
}

// ---- GEMM -----------------------------------------------------------------------------------------
// (work-item lookup x6_find_piece / x6_locate, XCD-contiguous tile order, routing masks: bl_x6_locate.h)
// optional epilogue of the row GEMM: C = drop(act(A . B + bias)) -- the dense node update of the message-passing layer
// (ptgnn MlpMessagePassingLayer's Linear -> tanh -> Dropout tail; call site buglab/models/gnnlayerdefs.py:6-23)
struct X6Epi {
  const float* bias;  // [N] or nullptr
  int act;            // BL_ACT_*
  uint32_t drop_key, drop_thresh;
  float drop_scale;
  // the extended forms (bl_gemm_rows_x6_epi2: the Linear layers of the relational transformer block, csrc/bl_great_layer.hip)
  int form;                // BL_X6_EPI_*
  const float* res;        // RES: c = A . B + res[row, n]
  int ld_res;
  const uint2* himask;     // MASK: the packed forward output y [M][3 N] -- c = (y's hi plane != 0) ? A . B x mask_scale : 0
  float mask_scale;
  float* colsum;           // MASK: [N] += column sums of c (the bias gradient)
  uint2* c_packed;         // PACK / MASK: the result in bl_pack_bf16x3's form [M][3 N] instead of fp32
};

// The plain form fits three workgroups per CU (3 x 52 KB of LDS, <= 168 registers); the routed form
// keeps its routing bytes and masks in registers and runs two.
// EPI: -1 = no epilogue, else the activation code (a template parameter: with a run-time switch the compiler evaluates
// every activation's libm call for every element -- measured 0.12 vs 0.05 ms on the c2 dense shape)
#define X6_MASKED_WGS 2  // workgroups per CU the routed form is compiled for (3: 168 registers with 11 of them in scratch)
#define X6_PLAIN_WGS 3
// (ablation builds of this kernel -- rows not gathered, no MFMAs, no result stores, term-major MFMA order, direct stores -- are made
// from tools/experiments/bl_gemm_x6_switches.hip; their numbers are in tools/experiments/README.md)
template <bool MASKED, int EPI>
__global__ __launch_bounds__(256, MASKED ? X6_MASKED_WGS : X6_PLAIN_WGS) void gemm_rows_x6_kernel(
    const uint4* __restrict__ xp0, const uint4* __restrict__ xp1, const uint4* __restrict__ xp2,
    const int* __restrict__ idx0, const int* __restrict__ idx1, const int* __restrict__ idx2, int w0, int w1, int w2,
    int koff1, int koff2, int nsrc, const uint32_t* __restrict__ win_bits, int ld_bits, const uint4* __restrict__ bp,
    long long strideB, const int* __restrict__ group_ptr, const int* __restrict__ group_w, int G, int M, int N, int K,
    float* __restrict__ c, int ldc, int xcd_remap, X6Epi epi) {
  // one array: after the last stage the four waves' result tiles are staged in it on their way out (see the epilogue)
  // EPI: -1 none; 0..15 = activation code (bias / activation / dropout, fp32 result); 16 + code = the same, result PACKED only;
  // 32 = + residual (fp32 result); 64 = masked by the packed forward output, column sums, result packed only
  constexpr bool E_ACT = EPI >= 0 && EPI < 32, E_PACK = EPI >= 16 && (EPI < 32 || EPI == 64), E_RES = EPI == 32, E_MASK = EPI == 64;
  constexpr int ACT = EPI >= 0 ? (EPI & 15) : 0;
  constexpr bool SWZ = X6_SWIZZLED(MASKED);
  constexpr int XROW = SWZ ? 12 : 13;
#define XSLOT(row_, kg_) (SWZ ? ((kg_) ^ (((row_) >> 2) & 3)) : (kg_))
  __shared__ uint4 ABs[(XBM + XBN) * XROW];
  uint4* As = ABs;
  uint4* Bs = ABs + XBM * XROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g, row0, nrows, tile_y;
  if (!x6_locate(group_ptr, G, M, XBM, xcd_remap, tile_y, g, row0, nrows)) return;
  const int n0 = tile_y * XBN;
  const int wsel = group_w ? group_w[g] : g;
  // tiled packed weights: this workgroup's 24 KB stage blocks, thread t reads uint4s t + 256 q
  const uint4* __restrict__ Bt = bp + (long long)wsel * strideB + (size_t)tile_y * (K >> 5) * 1536 + tid;

  // loader mapping: (row, k-group) pairs, 2 per thread; 4 consecutive lanes cover one row's 64-byte
  // plane segment.  Gathered row ids live in registers (one per piece and source).
  const int p_kg = tid & 3, p_row0 = tid >> 2;  // rows p_row0 and p_row0 + 64
  int gr0[2], gr1[2], gr2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = row0 + min(p_row0 + 64 * i, nrows - 1);
    gr0[i] = idx0 ? idx0[r] : r;
    gr1[i] = nsrc > 1 ? (idx1 ? idx1[r] : r) : 0;
    gr2[i] = nsrc > 2 ? (idx2 ? idx2[r] : r) : 0;
  }
  uint4 ra[2][3], rb[2][3];
  uint32_t ma[2];
  const int nk = (K + 31) / 32;

#define X6_LOAD_STAGE(k0_)                                                                                    \
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
      const uint4* src_ = base_ + (size_t)gr_ * 3 * (wj_ >> 3) + (kl_ >> 3);                                  \
      ra[i][0] = src_[0];                                                                                     \
      ra[i][1] = src_[wj_ >> 3];                                                                              \
      ra[i][2] = src_[2 * (wj_ >> 3)];                                                                        \
      if (MASKED) ma[i] = win_bits[(size_t)(row0 + min(row_, nrows - 1)) * ld_bits + (kc_ >> 5)];              \
      const uint4* bsrc_ = Bt + (size_t)((k0_) >> 5) * 1536 + i * 768;                                        \
      rb[i][0] = bsrc_[0];                                                                                    \
      rb[i][1] = bsrc_[256];                                                                                  \
      rb[i][2] = bsrc_[512];                                                                                  \
    }                                                                                                         \
  }
#define X6_STORE_STAGE(k0_)                                                                                   \
  {                                                                                                           \
    const bool kok_ = (k0_) + 8 * p_kg < K;                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                           \
      const int row_ = p_row0 + 64 * i;                                                                       \
      uint4 keep_ = make_uint4(~0u, ~0u, ~0u, ~0u);                                                           \
      if (MASKED) keep_ = keep_from_bits(ma[i] >> (8 * p_kg)); /* k0 is a multiple of 32 */                  \
      if (!kok_) keep_ = make_uint4(0u, 0u, 0u, 0u);                                                          \
      const bool nok_ = kok_ && (n0 + row_ < N);                                                              \
      _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                         \
        uint4 a_ = ra[i][p];                                                                                  \
        a_.x &= keep_.x; a_.y &= keep_.y; a_.z &= keep_.z; a_.w &= keep_.w;                                   \
        As[row_ * XROW + p * 4 + XSLOT(row_, p_kg)] = a_;                                                     \
        Bs[row_ * XROW + p * 4 + XSLOT(row_, p_kg)] = nok_ ? rb[i][p] : make_uint4(0u, 0u, 0u, 0u);           \
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

  X6_LOAD_STAGE(0)
  X6_STORE_STAGE(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) X6_LOAD_STAGE((kt + 1) * 32)
#pragma unroll
    for (int s = 0; s < 2; ++s) {  // two 16-k MFMA steps per stage; this lane's 8 k's = group 2s + half
      const int kg = 2 * s + half;
      bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        const uint4* p = &As[(wm * 64 + ti * 32 + li) * XROW + XSLOT(li, kg)];
        ah[ti] = __builtin_bit_cast(bf16x8, p[0]);
        am[ti] = __builtin_bit_cast(bf16x8, p[4]);
        al[ti] = __builtin_bit_cast(bf16x8, p[8]);
      }
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        const uint4* p = &Bs[(wn * 64 + tj * 32 + li) * XROW + XSLOT(li, kg)];
        bh[tj] = __builtin_bit_cast(bf16x8, p[0]);
        bm[tj] = __builtin_bit_cast(bf16x8, p[4]);
        bl[tj] = __builtin_bit_cast(bf16x8, p[8]);
      }
      // swapped operands (B fragment in the A slot): the accumulator holds the transposed tile, so
      // a lane owns 4 consecutive columns of one row.  Small terms first.  The six terms of one accumulator are written back to
      // back (hipcc alternates between two accumulators); letting the four accumulators take turns instead measured equal
      // (profiles/r04y_term_major.log): a dependent MFMA two issue slots later does not stall
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          f32x16 a = acc[ti][tj];
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm[tj], am[ti], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[tj], ah[ti], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[tj], al[ti], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm[tj], ah[ti], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[tj], am[ti], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[tj], ah[ti], a, 0, 0, 0);
          acc[ti][tj] = a;
        }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      X6_STORE_STAGE((kt + 1) * 32)
      __syncthreads();
    }
  }

  // The accumulator layout gives a lane 4 consecutive columns of one row, a wave-wide store 64 pieces of 16 B on 32 different
  // rows: 32-byte segments.  The tile goes through LDS instead (per wave [32 rows][64 + 4] fp32, the operand images are dead
  // after the last stage's barrier) and leaves as whole 256-byte row pieces, 16 lanes per piece: measured on the node
  // update's backward kernel (same layout, csrc/bl_node_bwd.hip), the direct form cost 2-3x the time of its bytes.
  float* stage = reinterpret_cast<float*>(ABs) + wave * (32 * 68);
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int m = wm * 64 + ti * 32 + li;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = n0 + wn * 64 + tj * 32 + 8 * gq + 4 * half;
        float v[4] = {acc[ti][tj][4 * gq + 0], acc[ti][tj][4 * gq + 1], acc[ti][tj][4 * gq + 2], acc[ti][tj][4 * gq + 3]};
        if (E_ACT) {
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (epi.bias && n < N) bv = *reinterpret_cast<const float4*>(epi.bias + n);
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            v[u] = bl_act(ACT, v[u]);
            if (epi.drop_thresh) {  // same counter as the fp32 row GEMM: element index row * N + column
              const uint32_t idx = (uint32_t)(row0 + m) * (uint32_t)N + (uint32_t)(n + u);
              v[u] = ((bl_lowbias32(idx + epi.drop_key) >> 8) >= epi.drop_thresh) ? v[u] * epi.drop_scale : 0.f;
            }
          }
        }
        *reinterpret_cast<float4*>(stage + li * 68 + tj * 32 + 8 * gq + 4 * half) = make_float4(v[0], v[1], v[2], v[3]);
      }
    // (a wave reads back only what it wrote itself; its LDS operations execute in order)
    const int c4 = lane & 15, n = n0 + wn * 64 + 4 * c4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = (lane >> 4) + 4 * j;
      const int mm = wm * 64 + ti * 32 + r;
      float4 v = *reinterpret_cast<const float4*>(stage + r * 68 + 4 * c4);
      if (mm < nrows && n < N) {
        const size_t grow = (size_t)(row0 + mm);
        if (E_RES) {
          const float4 rv = *reinterpret_cast<const float4*>(epi.res + grow * epi.ld_res + n);
          v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (E_MASK) {  // y = drop(relu(z)) != 0  <=>  kept and z > 0: the hi plane of a non-zero fp32 is non-zero
          const uint2 hm = epi.himask[(grow * 3 * N + n) >> 2];
          v.x = (hm.x & 0x7fffu) ? v.x * epi.mask_scale : 0.f;
          v.y = (hm.x & 0x7fff0000u) ? v.y * epi.mask_scale : 0.f;
          v.z = (hm.y & 0x7fffu) ? v.z * epi.mask_scale : 0.f;
          v.w = (hm.y & 0x7fff0000u) ? v.w * epi.mask_scale : 0.f;
          csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w;
        }
        if (E_PACK) {
          uint16_t h[4], m[4], l[4];
          split3(v.x, h[0], m[0], l[0]);
          split3(v.y, h[1], m[1], l[1]);
          split3(v.z, h[2], m[2], l[2]);
          split3(v.w, h[3], m[3], l[3]);
          uint2* o = epi.c_packed + ((grow * 3 * N + n) >> 2);
          o[0] = make_uint2(PK(h[0], h[1]), PK(h[2], h[3]));
          o[N >> 2] = make_uint2(PK(m[0], m[1]), PK(m[2], m[3]));
          o[N >> 1] = make_uint2(PK(l[0], l[1]), PK(l[2], l[3]));
        } else {
          *reinterpret_cast<float4*>(c + grow * ldc + n) = v;
        }
      }
    }
  }
  if (E_MASK) {  // column sums of this wave's 64 x 64 block: over the four row groups of the lanes, then one atomic per column
    csum.x += __shfl_xor(csum.x, 16, 64); csum.y += __shfl_xor(csum.y, 16, 64); csum.z += __shfl_xor(csum.z, 16, 64); csum.w += __shfl_xor(csum.w, 16, 64);
    csum.x += __shfl_xor(csum.x, 32, 64); csum.y += __shfl_xor(csum.y, 32, 64); csum.z += __shfl_xor(csum.z, 32, 64); csum.w += __shfl_xor(csum.w, 32, 64);
    const int n = n0 + wn * 64 + 4 * (lane & 15);
    if (lane < 16 && n < N && epi.colsum) {
      unsafeAtomicAdd(epi.colsum + n, csum.x);
      unsafeAtomicAdd(epi.colsum + n + 1, csum.y);
      unsafeAtomicAdd(epi.colsum + n + 2, csum.z);
      unsafeAtomicAdd(epi.colsum + n + 3, csum.w);
    }
  }
}

#undef XSLOT

// ---- weight-gradient GEMM -------------------------------------------------------------------------
// gW_g[i, n] += sum_{e in group g} A[e, i] * Gr[e, n]     A = gathered packed rows (h[src] | h[tgt]),
//                                                          Gr[e, :] = g_node[g_idx[e], :] where winner == e
// The contraction runs over MESSAGES, but the bf16 MFMA wants 8 consecutive k's of one row in a
// lane: the operands have to be transposed on the way.  Both tiles are stored in LDS exactly as
// they arrive -- [plane][message][feature], feature-contiguous rows of 320 B -- and the fragments are
// read with ds_read_b64_tr_b16, gfx950's transposing LDS read: a 16-lane group reads a
// [4 messages][16 features] block (lane 4j+q supplies the address of message j, features 4q..4q+3)
// and lane i receives the 4 messages of feature i.  Two reads = the 8 k's of one MFMA operand.
// Row stride 320 B puts the 4 message rows of a block 16 banks apart: conflict-free.
// Global loads are full 256-byte plane rows (16 lanes x 16 B per message and plane).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define WRS 160                 // shorts per LDS message row (128 features + 32 pad)
#define WPLANE (32 * WRS)       // shorts per plane (32 messages)
#define WOPER (3 * WPLANE)      // shorts per operand image

__device__ __forceinline__ bf16x8 tr_frag(const short* p) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * WRS));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <bool ROUTED>
__global__ __launch_bounds__(256, 2) void gemm_wgrad_x6_kernel(
    const uint4* __restrict__ xp0, const uint4* __restrict__ xp1, const uint4* __restrict__ xp2,
    const int* __restrict__ idx0, const int* __restrict__ idx1, const int* __restrict__ idx2, int w0, int w1, int w2,
    int koff1, int koff2, int nsrc, const uint4* __restrict__ gp, const int* __restrict__ g_idx,
    const uint32_t* __restrict__ win_bits, int ld_bits, const int* __restrict__ group_ptr, const int* __restrict__ group_w, int G,
    int M, int N, int K, int kchunk, float* __restrict__ gw_base, long long strideW, int ldw, int ntiles_n, int xcd_remap,
    unsigned* __restrict__ order_ctr) {
  __shared__ __attribute__((aligned(16))) short As[WOPER];
  __shared__ __attribute__((aligned(16))) short Bs[WOPER];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g, e0, ne, tile_y;
  if (!x6_locate(group_ptr, G, M, kchunk, xcd_remap, tile_y, g, e0, ne)) return;
  const int e1 = e0 + ne;
  const int i0 = (tile_y / ntiles_n) * XBM;
  const int n0 = (tile_y % ntiles_n) * XBN;
  const int wsel = group_w ? group_w[g] : g;

  // loader: units (message, 8-feature group); unit u = tid + 256 i -> message u >> 4, group u & 15
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
  const int gwg = N >> 3;                                      // uint4 per plane of a G row
  const uint4* __restrict__ gbase = gp + (nnc >> 3);
  const uint32_t* __restrict__ mbase = ROUTED ? win_bits + (nnc >> 5) : nullptr;
  const int mshift = nnc & 31;

  uint4 ra[2][3], rb[2][3];
  uint32_t mk[2];
  int arow[2], grow[2], mrow[2];  // gathered rows / message ids of the NEXT stage to load

#define WX6_LOAD_IDX(k0_)                                          \
  {                                                                \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                \
      const int e_ = (k0_) + msg0 + 16 * i;                        \
      const int ec_ = e_ < e1 ? e_ : e0;                           \
      arow[i] = aidx ? aidx[ec_] : ec_;                            \
      grow[i] = g_idx ? g_idx[ec_] : ec_;                          \
      mrow[i] = ec_;                                               \
    }                                                              \
  }
#define WX6_LOAD_STAGE()                                                                         \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                              \
      const uint4* a_ = abase + (size_t)arow[i] * 3 * awg;                                       \
      ra[i][0] = a_[0];                                                                          \
      ra[i][1] = a_[awg];                                                                        \
      ra[i][2] = a_[2 * awg];                                                                    \
      const uint4* g_ = gbase + (size_t)grow[i] * 3 * gwg;                                       \
      rb[i][0] = g_[0];                                                                          \
      rb[i][1] = g_[gwg];                                                                        \
      rb[i][2] = g_[2 * gwg];                                                                    \
      mk[i] = ROUTED ? mbase[(size_t)mrow[i] * ld_bits] : 0u;                                    \
    }                                                                                            \
  }
#define WX6_STORE_STAGE(k0_)                                                                     \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                              \
      const int eid_ = (k0_) + msg0 + 16 * i;                                                    \
      const bool eok_ = eid_ < e1;                                                               \
      uint4 keep_ = ROUTED ? keep_from_bits(mk[i] >> mshift) : make_uint4(~0u, ~0u, ~0u, ~0u);   \
      if (!(eok_ && b_ok)) keep_ = make_uint4(0u, 0u, 0u, 0u);                                   \
      const int slot_ = (msg0 + 16 * i) * WRS + 8 * fg;                                          \
      _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                            \
        *reinterpret_cast<uint4*>(&As[p * WPLANE + slot_]) = (eok_ && a_ok) ? ra[i][p] : make_uint4(0u, 0u, 0u, 0u); \
        uint4 b_ = rb[i][p];                                                                     \
        b_.x &= keep_.x; b_.y &= keep_.y; b_.z &= keep_.z; b_.w &= keep_.w;                      \
        *reinterpret_cast<uint4*>(&Bs[p * WPLANE + slot_]) = b_;                                 \
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
  // transposing-read address of this lane inside a [32 messages][WRS] plane, tile feature base 0
  const int l16 = lane & 15, grp = lane >> 4;
  const int tr_off = ((grp >> 1) * 8 + (l16 >> 2)) * WRS + (grp & 1) * 16 + 4 * (l16 & 3);
  const short* a_tr = As + tr_off + wm * 64;
  const short* b_tr = Bs + tr_off + wn * 64;
  const int nk = (ne + 31) / 32;

  WX6_LOAD_IDX(e0)
  WX6_LOAD_STAGE()
  WX6_LOAD_IDX(e0 + 32)
  WX6_STORE_STAGE(e0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      WX6_LOAD_STAGE()
      WX6_LOAD_IDX(e0 + (kt + 2) * 32)
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {  // two 16-message MFMA steps per stage
      bf16x8 af[2][3], bf[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          af[t][p] = tr_frag(a_tr + p * WPLANE + s * 16 * WRS + t * 32);
          bf[t][p] = tr_frag(b_tr + p * WPLANE + s * 16 * WRS + t * 32);
        }
#define WX6_TERM(pa_, pb_)                                                                            \
  _Pragma("unroll") for (int ti = 0; ti < 2; ++ti) _Pragma("unroll") for (int tj = 0; tj < 2; ++tj)   \
      acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ti][pa_], bf[tj][pb_], acc[ti][tj], 0, 0, 0);
      WX6_TERM(1, 1) WX6_TERM(2, 0) WX6_TERM(0, 2) WX6_TERM(1, 0) WX6_TERM(0, 1) WX6_TERM(0, 0)
    }
    __syncthreads();
    if (kt + 1 < nk) {
      WX6_STORE_STAGE(e0 + (kt + 1) * 32)
      __syncthreads();
    }
  }

  float* __restrict__ gw = gw_base + (long long)wsel * strideW;
  // deterministic mode (launched without the XCD remap): the message chunks of one (group, tile) add in chunk order
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
        if (f < K) unsafeAtomicAdd(&gw[(size_t)f * ldw + n], acc[ti][tj][r]);
      }
    }
  bl_ordered_leave(ctr, turn);
}

// ---- weight-gradient GEMM, wide tile ---------------------------------------------------------------
// Same contraction, 256 (features of A) x 128 (channels) per workgroup: 8 waves, each the 64 x 64 block of the kernel above.
// What the PMC counters of the 128 x 128 kernel say (profiles/r04b_wgrad_pmc.json): its matrix pipe is 39 % busy with every
// operand L2-resident or not (0.254 vs 0.269 ms), its waves spend a quarter of their cycles issuing ~6 non-MFMA instructions
// per MFMA (staging: address arithmetic, routing masks, zero-selects, LDS stores) and half of them stalled -- the two waves
// of a SIMD run their staging phases and their MFMA phases at the same time more often than not.  So this kernel
//  * stages the routed operand -- the packed node gradient of the messages' targets -- ONCE per message instead of once
//    per 128-feature half (a quarter less staging work per MFMA),
//  * owns the CU (one 512-thread workgroup, 2 x 72 KB of LDS: double-buffered stages, ONE barrier per 32-message stage),
//  * and runs its two wave groups in opposite phases: waves 0-3 do the stage's 48 MFMAs first and their share of the
//    staging (LDS stores of stage k+1, global loads of stage k+2) after them, waves 4-7 -- the SIMDs' second waves -- do
//    their staging first and the MFMAs after it, so that a SIMD's matrix pipe sees one wave's MFMA stream while the other
//    wave stages.
// LDS image of a stage: [operand: A half 0, A half 1, B][plane][message (32)][128 features], rows of 256 B without padding;
// the four 64-byte granules of a row are XOR-swizzled with (message & 3), which puts the four message rows of a
// transposing read's [4 messages][32 features] block on four different bank groups (conflict-free, SQ_LDS_BANK_CONFLICT 0)
// and keeps the 16-byte stores of 8 consecutive lanes on 32 different banks.
// Used when K is a multiple of 256, every source's width a multiple of 128 and the operands are either all gathered or
// all direct (the message weight gradients of every shipped configuration); everything else runs on the kernel above.
// A rows past the end of the chunk are NOT zeroed: their B rows are (the keep mask), and the clamped row they load is a
// real message of the chunk, so a non-finite value there is in the exact result as well.
#define WW_ROW 128                      // shorts per LDS message row
#define WW_PLANE (32 * WW_ROW)          // shorts per plane (32 messages)
#define WW_OPER (3 * WW_PLANE)          // shorts per operand image
#define WW_STAGE (3 * WW_OPER)          // shorts per stage (A half 0, A half 1, B): 36 864 shorts = 72 KB

__device__ __forceinline__ bf16x8 tr_frag_w(const short* p) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * WW_ROW));  // messages +4: same (message & 3)
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// routing byte -> AND-masks, three VALU per dword (two sign-extending bit-field extracts + one bit-field insert)
__device__ __forceinline__ uint4 keep_from_bits_bfi(uint32_t b) {
  uint4 k;
#define KB_(i_) (uint32_t)__builtin_amdgcn_sbfe(b, i_, 1)
  k.x = (KB_(0) & 0x0000FFFFu) | (KB_(1) & 0xFFFF0000u);
  k.y = (KB_(2) & 0x0000FFFFu) | (KB_(3) & 0xFFFF0000u);
  k.z = (KB_(4) & 0x0000FFFFu) | (KB_(5) & 0xFFFF0000u);
  k.w = (KB_(6) & 0x0000FFFFu) | (KB_(7) & 0xFFFF0000u);
#undef KB_
  return k;
}

// GATHER: every operand row is addressed through an index array (idx0/1/2, g_idx all non-null); otherwise none is
template <bool ROUTED, bool GATHER>
__global__ __launch_bounds__(512) void gemm_wgrad_x6_wide_kernel(
    const uint4* __restrict__ xp0, const uint4* __restrict__ xp1, const uint4* __restrict__ xp2,
    const int* __restrict__ idx0, const int* __restrict__ idx1, const int* __restrict__ idx2, int w0, int w1, int w2,
    int koff1, int koff2, int nsrc, const uint4* __restrict__ gp, const int* __restrict__ g_idx,
    const uint32_t* __restrict__ win_bits, int ld_bits, const int* __restrict__ group_ptr, const int* __restrict__ group_w, int G,
    int M, int N, int K, int kchunk, float* __restrict__ gw_base, long long strideW, int ldw, int ntiles_n, int xcd_remap,
    unsigned* __restrict__ order_ctr) {
  __shared__ __attribute__((aligned(16))) short Ls[2 * WW_STAGE];  // the ONE LDS object of this kernel (144 KB)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int g, e0, ne, tile_y;
  if (!x6_locate(group_ptr, G, M, kchunk, xcd_remap, tile_y, g, e0, ne)) return;
  const int e1 = e0 + ne;
  const int i0 = (tile_y / ntiles_n) * 256;
  const int n0 = (tile_y % ntiles_n) * XBN;
  const int wsel = group_w ? group_w[g] : g;

  // loader: thread -> (message tid >> 4 of the stage, 16-byte chunk tid & 15 = 8 features) of each of the three operand rows
  const int fg = tid & 15, msg = tid >> 4;
  const int nn = n0 + 8 * fg;
  const bool b_ok = nn < N;
  const int nnc = b_ok ? nn : 0;
  // per 128-feature half of the A tile: source (wave-uniform: a half lies inside one source), row length, this thread's chunk.
  // Row offsets are 32-bit in units of 16 bytes (a packed operand below 64 GB).
#define WW_HALF(h_, src_, idx_, rowq_, planeq_, colq_)                                                                 \
  const int fi##h_ = i0 + 128 * h_;                                                                                    \
  const int aj##h_ = (nsrc > 2 && fi##h_ >= koff2) ? 2 : ((nsrc > 1 && fi##h_ >= koff1) ? 1 : 0);                      \
  const uint4* __restrict__ src_ = aj##h_ == 0 ? xp0 : (aj##h_ == 1 ? xp1 : xp2);                                      \
  const int* __restrict__ idx_ = aj##h_ == 0 ? idx0 : (aj##h_ == 1 ? idx1 : idx2);                                     \
  const uint32_t planeq_ = (uint32_t)((aj##h_ == 0 ? w0 : (aj##h_ == 1 ? w1 : w2)) >> 3); /* uint4 per plane */        \
  const uint32_t rowq_ = 3u * planeq_;                                                                                 \
  const uint32_t colq_ = (uint32_t)((fi##h_ - (aj##h_ == 0 ? 0 : (aj##h_ == 1 ? koff1 : koff2))) >> 3) + (uint32_t)fg;
  WW_HALF(0, asrc0, aidx0, arowq0, aplq0, acolq0)
  WW_HALF(1, asrc1, aidx1, arowq1, aplq1, acolq1)
  const uint32_t gplq = (uint32_t)(N >> 3), growq = 3u * gplq, gcolq = (uint32_t)(nnc >> 3);
  const uint32_t* __restrict__ mbase = ROUTED ? win_bits + (nnc >> 5) : nullptr;
  const int mshift = nnc & 31;
  // this thread's slot in an operand plane (shorts): row `msg`, granule (fg >> 2) ^ (msg & 3), chunk fg & 3
  const int slot = msg * WW_ROW + ((((fg >> 2) ^ (msg & 3)) << 2) | (fg & 3)) * 8;

  // Two register sets (A, B) of staged operands, alternating per stage: set X receives stage k+2 while set Y, loaded one
  // iteration earlier, is masked and stored to LDS as stage k+1.  (scalars, not arrays: hipcc sends small arrays that are
  // written under control flow to scratch)
  uint4 ra00A, ra01A, ra02A, ra10A, ra11A, ra12A, rb0A, rb1A, rb2A, ra00B, ra01B, ra02B, ra10B, ra11B, ra12B, rb0B, rb1B, rb2B;
  uint32_t mkA = 0u, mkB = 0u;
  uint32_t ar0A, ar1A, grA, mrA, ar0B, ar1B, grB, mrB;  // gathered rows / message id of a stage still to be loaded

  // rows of the stage that starts at message k0_ -> index set S_   (past the end of the chunk: the chunk's first message --
  // a valid row whose B row gets a zero keep mask)
#define WW_IDX(k0_, S_)                                      \
  {                                                          \
    const int e_ = (k0_) + msg;                              \
    const uint32_t ec_ = (uint32_t)(e_ < e1 ? e_ : e0);      \
    ar0##S_ = GATHER ? (uint32_t)aidx0[ec_] : ec_;           \
    ar1##S_ = GATHER ? (uint32_t)aidx1[ec_] : ec_;           \
    gr##S_ = GATHER ? (uint32_t)g_idx[ec_] : ec_;            \
    mr##S_ = ec_;                                            \
  }
#define WW_LOAD_A0(S_)                                       \
  {                                                          \
    const uint32_t o_ = ar0##S_ * arowq0 + acolq0;           \
    ra00##S_ = asrc0[o_];                                    \
    ra01##S_ = asrc0[o_ + aplq0];                            \
    ra02##S_ = asrc0[o_ + 2u * aplq0];                       \
  }
#define WW_LOAD_A1(S_)                                       \
  {                                                          \
    const uint32_t o_ = ar1##S_ * arowq1 + acolq1;           \
    ra10##S_ = asrc1[o_];                                    \
    ra11##S_ = asrc1[o_ + aplq1];                            \
    ra12##S_ = asrc1[o_ + 2u * aplq1];                       \
  }
#define WW_LOAD_B(S_)                                                    \
  {                                                                      \
    const uint32_t o_ = gr##S_ * growq + gcolq;                          \
    rb0##S_ = gp[o_];                                                    \
    rb1##S_ = gp[o_ + gplq];                                             \
    rb2##S_ = gp[o_ + 2u * gplq];                                        \
    if (ROUTED) mk##S_ = mbase[(size_t)mr##S_ * (uint32_t)ld_bits];      \
  }
#define WW_ST1(off_, v_) *reinterpret_cast<uint4*>(wr_ + (off_)) = (v_);
#define WW_STB(off_, v_)                                                 \
  {                                                                      \
    uint4 b_ = (v_);                                                     \
    b_.x &= keep_.x; b_.y &= keep_.y; b_.z &= keep_.z; b_.w &= keep_.w;  \
    *reinterpret_cast<uint4*>(wr_ + 2 * WW_OPER + (off_)) = b_;          \
  }
#define WW_KEEP(S_, k0_)                                                                             \
  uint4 keep_ = ROUTED ? keep_from_bits_bfi(mk##S_ >> mshift) : make_uint4(~0u, ~0u, ~0u, ~0u);      \
  if (!((k0_) + msg < e1 && b_ok)) keep_ = make_uint4(0u, 0u, 0u, 0u);
#define WW_STORE_A0(S_) WW_ST1(0, ra00##S_) WW_ST1(WW_PLANE, ra01##S_) WW_ST1(2 * WW_PLANE, ra02##S_)
#define WW_STORE_A1(S_) WW_ST1(WW_OPER, ra10##S_) WW_ST1(WW_OPER + WW_PLANE, ra11##S_) WW_ST1(WW_OPER + 2 * WW_PLANE, ra12##S_)
#define WW_STORE_B(S_) WW_STB(0, rb0##S_) WW_STB(WW_PLANE, rb1##S_) WW_STB(2 * WW_PLANE, rb2##S_)

  f32x16 acc[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  // wave -> (A half, 64-feature block of the half, 64-channel block of B)
  const int wq = wave & 3, wn = wave >> 2, li = lane & 31, half = lane >> 5;
  const int ahalf = wq >> 1, afb = (wq & 1) * 64, bfb = wn * 64;
  // transposing read of this lane: message (grp >> 1) * 8 + (l16 >> 2) (+ 4, + 16 s), features base + 32 t + 16 (grp & 1) + 4 (l16 & 3)
  const int l16 = lane & 15, grp = lane >> 4;
  const int tr_row = ((grp >> 1) * 8 + (l16 >> 2)) * WW_ROW;
  const int tr_in = (grp & 1) * 16 + 4 * (l16 & 3);  // inside the 32-feature granule
  const int sw = (l16 >> 2) & 3;                      // (message & 3) of every row this lane reads
  int a_off[2], b_off[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    a_off[t] = ahalf * WW_OPER + tr_row + ((((afb >> 5) + t) ^ sw) << 5) + tr_in;
    b_off[t] = 2 * WW_OPER + tr_row + ((((bfb >> 5) + t) ^ sw) << 5) + tr_in;
  }
  const int nk = (ne + 31) / 32;
  const int nk2 = (nk + 1) & ~1;  // the loop runs stage pairs; a phantom last stage has an all-zero B image

#define WW_FRAGS(rd_, s_)                                                                              \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) _Pragma("unroll") for (int p = 0; p < 3; ++p) {        \
    af[t][p] = tr_frag_w((rd_) + (s_) * 16 * WW_ROW + a_off[t] + p * WW_PLANE);                        \
    bf[t][p] = tr_frag_w((rd_) + (s_) * 16 * WW_ROW + b_off[t] + p * WW_PLANE);                        \
  }
  // One loop iteration = one 32-message stage, straight-line (no branch: loads past the chunk's end are clamped to valid
  // rows, stores past it fill a buffer nobody reads), the staging of the NEXT stages cut in pieces and placed between the
  // six-MFMA-term groups of this stage, so that a wave never leaves the matrix pipe for a whole staging phase: in the
  // phase-structured form (profiles/r04c_trace2.log) a stage took ~7 000 cycles of which each wave spent ~3 400 staging
  // (500-1 000 waiting for its loads, 1 100 on masks + LDS stores, 1 500-1 800 on address arithmetic + issuing 13 loads into
  // a busy texture path) with the matrix pipe idle for half of the stage.
  //   X: the register set that receives stage KT + 2 (rows from index set X, loaded one iteration ago)
  //   Y: the set that holds stage KT + 1 (loaded one iteration ago), masked and stored to the other LDS buffer now;
  //      index set Y receives the rows of stage KT + 3
// scheduling fence: VALU, SALU and LDS reads may move across it; MFMAs, global loads and LDS stores may not -- the staging pieces
// stay where they are written (left alone, hipcc sinks the loads to the end of the iteration and hoists the stores)
#define WW_PIN() __builtin_amdgcn_sched_barrier(0x106);
// (WW_S staging piece, WW_T six-term MFMA group, WW_F fragment reads: the ablation / trace builds that blank them out one at a
// time are made from tools/experiments/bl_gemm_x6_switches.hip, profiles/r04d_wgrad_ablation.log)
#define WW_S(x_) x_
#define WW_T(pa_, pb_) WX6_TERM(pa_, pb_)
#define WW_F(rd_, s_) WW_FRAGS(rd_, s_)
#define WW_ITER(X, Y, KT)                                                                              \
  {                                                                                                    \
    const int kt_ = (KT);                                                                              \
    const short* rd_ = Ls + (kt_ & 1) * WW_STAGE;                                                      \
    short* wr_ = Ls + ((kt_ + 1) & 1) * WW_STAGE + slot;                                               \
    WW_F(rd_, 0)                                                                                       \
    WW_S(WW_IDX(e0 + (kt_ + 3) * 32, Y))                                                               \
    WW_PIN()                                                                                           \
    WW_T(1, 1)                                                                                         \
    WW_PIN()                                                                                           \
    WW_S(WW_LOAD_A0(X))                                                                                \
    WW_PIN()                                                                                           \
    WW_T(2, 0)                                                                                         \
    WW_PIN()                                                                                           \
    WW_S(WW_LOAD_A1(X))                                                                                \
    WW_PIN()                                                                                           \
    WW_T(0, 2)                                                                                         \
    WW_PIN()                                                                                           \
    WW_S(WW_LOAD_B(X))                                                                                 \
    WW_PIN()                                                                                           \
    WW_T(1, 0)                                                                                         \
    WW_S(WW_KEEP(Y, e0 + (kt_ + 1) * 32))                                                              \
    WW_T(0, 1)                                                                                         \
    WW_PIN()                                                                                           \
    WW_S(WW_STORE_A0(Y))                                                                               \
    WW_PIN()                                                                                           \
    WW_T(0, 0)                                                                                         \
    WW_F(rd_, 1)                                                                                       \
    WW_T(1, 1)                                                                                         \
    WW_PIN()                                                                                           \
    WW_S(WW_STORE_A1(Y))                                                                               \
    WW_PIN()                                                                                           \
    WW_T(2, 0)                                                                                         \
    WW_PIN()                                                                                           \
    WW_S(WW_STORE_B(Y))                                                                                \
    WW_PIN()                                                                                           \
    WW_T(0, 2) WW_T(1, 0) WW_T(0, 1) WW_T(0, 0)                                                        \
    __syncthreads();                                                                                   \
  }

  // prologue: stage 0 through set A into buffer 0, stage 1 into set B, rows of stage 2 into index set A
  WW_IDX(e0, A)
  WW_LOAD_A0(A) WW_LOAD_A1(A) WW_LOAD_B(A)
  WW_IDX(e0 + 32, B)
  {
    short* wr_ = Ls + slot;
    WW_KEEP(A, e0)
    WW_STORE_A0(A) WW_STORE_A1(A) WW_STORE_B(A)
  }
  WW_LOAD_A0(B) WW_LOAD_A1(B) WW_LOAD_B(B)
  WW_IDX(e0 + 64, A)
  __syncthreads();
  bf16x8 af[2][3], bf[2][3];
  for (int kt = 0; kt < nk2; kt += 2) {
    WW_ITER(A, B, kt)
    WW_ITER(B, A, kt + 1)
  }

  float* __restrict__ gw = gw_base + (long long)wsel * strideW;
  unsigned* ctr = order_ctr ? order_ctr + (size_t)g * gridDim.y + tile_y : nullptr;
  const unsigned turn = (unsigned)((e0 - (group_ptr ? group_ptr[g] : 0)) / kchunk);
  bl_ordered_enter(ctr, turn);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int n = n0 + bfb + tj * 32 + li;
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = i0 + 128 * ahalf + afb + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        unsafeAtomicAdd(&gw[(size_t)f * ldw + n], acc[ti][tj][r]);  // (K is a multiple of 256: every row of the tile exists)
      }
    }
  bl_ordered_leave(ctr, turn);
}

// ================================================================================================
extern "C" int bl_pack_bf16x3(const float* x, int32_t ld, int64_t R, int32_t D, uint16_t* out, void* stream) {
  if (R == 0) return BL_OK;
  BL_CHECK_ARG(x && out && bl_aligned16(x) && bl_aligned16(out), "bl_pack_bf16x3: null or misaligned pointer");
  BL_CHECK_ARG(D > 0 && D % 8 == 0 && ld % 4 == 0, "bl_pack_bf16x3: D must be a multiple of 8 (got %d)", D);
  const long long total = (long long)R * (D / 8);
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld,
                     (long long)R, D, reinterpret_cast<uint4*>(out), D / 8, 0);
  BL_LAUNCH_CHECK("bl_pack_bf16x3");
  return BL_OK;
}

extern "C" int bl_pack_bf16x3_cols(const float* x, int32_t ld, int64_t R, int32_t D, int32_t D_total, int32_t col_off,
                                   uint16_t* out, void* stream) {
  if (R == 0) return BL_OK;
  BL_CHECK_ARG(x && out && bl_aligned16(x) && bl_aligned16(out), "bl_pack_bf16x3_cols: null or misaligned pointer");
  BL_CHECK_ARG(D > 0 && D % 8 == 0 && ld % 4 == 0 && D_total % 8 == 0 && col_off % 8 == 0 && col_off >= 0 && col_off + D <= D_total,
               "bl_pack_bf16x3_cols: widths / offset must be multiples of 8 with col_off + D <= D_total");
  const long long total = (long long)R * (D / 8);
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld,
                     (long long)R, D, reinterpret_cast<uint4*>(out), D_total / 8, col_off / 8);
  BL_LAUNCH_CHECK("bl_pack_bf16x3_cols");
  return BL_OK;
}

extern "C" int bl_pack_weights_x6(const float* w, int32_t G, int32_t K, int32_t N, int32_t w_is_kn, uint16_t* out,
                                  void* stream) {
  if (G == 0) return BL_OK;
  BL_CHECK_ARG(w && out && bl_aligned16(out), "bl_pack_weights_x6: null or misaligned pointer");
  BL_CHECK_ARG(K > 0 && K % 32 == 0 && N > 0, "bl_pack_weights_x6: K must be a multiple of 32 (got %d)", K);
  const long long total = (long long)G * ((N + 127) / 128) * (K / 32) * 512;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, G, K,
                     N, w_is_kn, reinterpret_cast<uint4*>(out));
  BL_LAUNCH_CHECK("bl_pack_weights_x6");
  return BL_OK;
}

extern "C" int64_t bl_pack_job_blocks(int32_t kind, int32_t G, int32_t K, int32_t N) {
  if (kind <= 1 || kind >= 5) return ((int64_t)G * ((N + 127) / 128) * (K / 32) * 512 + 255) / 256;
  if (kind >= 3) return ((int64_t)G * ((N + WBN - 1) / WBN) * (K / 32) * (WBN * 4) + 255) / 256;
  return ((int64_t)G * K * N + 255) / 256;
}

extern "C" int bl_pack_weights_multi(const bl_pack_job_t* jobs_device, int32_t njobs, int32_t total_blocks, void* stream) {
  if (njobs == 0 || total_blocks == 0) return BL_OK;
  BL_CHECK_ARG(jobs_device && njobs > 0 && total_blocks > 0, "bl_pack_weights_multi: bad arguments");
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, njobs,
                     bl_h3_sat_counter());
  BL_LAUNCH_CHECK("bl_pack_weights_multi");
  return BL_OK;
}

namespace {
int gemm_rows_x6_impl(const char* who, const bl_rows_packed_t* a, const uint32_t* win_bits, int32_t ld_bits, const uint16_t* bp,
                      int64_t b_group_stride, const int32_t* group_ptr, const int32_t* group_w, int32_t G, int32_t M, int32_t N,
                      int32_t K, const X6Epi* epi, float* c, int32_t ldc, void* stream) {
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
  BL_CHECK_ARG(b_group_stride % 8 == 0 && (G <= 1 || b_group_stride >= (int64_t)((N + 127) / 128) * (K / 32) * 12288),
               "%s: packed group stride must cover one group's tiled weights (bl_pack_weights_x6)", who);
  BL_CHECK_ARG(win_bits == nullptr || (a->nsrc == 1 && a->idx[0] && ld_bits * 32 >= K),
               "%s: the routed form needs exactly one gathered source and ld_bits >= K / 32", who);
  BL_CHECK_ARG(!(win_bits && epi), "%s: the routed form has no epilogue", who);
  dim3 grid((M + XBM - 1) / XBM + (group_ptr ? G : 0), (N + XBN - 1) / XBN);
  const int xcd = 1;  // XCD-contiguous tile order (x6_locate)
  const uint4* x0 = reinterpret_cast<const uint4*>(a->xp[0]);
  const uint4* x1 = a->nsrc > 1 ? reinterpret_cast<const uint4*>(a->xp[1]) : nullptr;
  const uint4* x2 = a->nsrc > 2 ? reinterpret_cast<const uint4*>(a->xp[2]) : nullptr;
  X6Epi e = {nullptr, BL_ACT_NONE, 0u, 0u, 1.f, 0, nullptr, 0, nullptr, 1.f, nullptr, nullptr};
  if (epi) e = *epi;
#define X6_ARGS                                                                                                          \
  x0, x1, x2, a->idx[0], a->nsrc > 1 ? a->idx[1] : nullptr, a->nsrc > 2 ? a->idx[2] : nullptr, a->width[0],              \
      a->nsrc > 1 ? a->width[1] : 0, a->nsrc > 2 ? a->width[2] : 0, koff[1], koff[2], a->nsrc, win_bits, ld_bits,        \
      reinterpret_cast<const uint4*>(bp), (long long)(b_group_stride / 8), group_ptr, group_w, G, M, N, K, c, ldc, xcd, e
  if (win_bits)
    hipLaunchKernelGGL((gemm_rows_x6_kernel<true, -1>), grid, dim3(256), 0, (hipStream_t)stream, X6_ARGS);
  else if (epi && e.form == BL_X6_EPI_ACT_PACK && e.act == BL_ACT_RELU)
    hipLaunchKernelGGL((gemm_rows_x6_kernel<false, 16 + BL_ACT_RELU>), grid, dim3(256), 0, (hipStream_t)stream, X6_ARGS);
  else if (epi && e.form == BL_X6_EPI_RES)
    hipLaunchKernelGGL((gemm_rows_x6_kernel<false, 32>), grid, dim3(256), 0, (hipStream_t)stream, X6_ARGS);
  else if (epi && e.form == BL_X6_EPI_MASK_PACK)
    hipLaunchKernelGGL((gemm_rows_x6_kernel<false, 64>), grid, dim3(256), 0, (hipStream_t)stream, X6_ARGS);
  else if (epi && e.form != BL_X6_EPI_ACT) {
    bl_set_error("%s: epilogue form %d with activation %d is not built (packed result: relu only)", who, e.form, e.act);
    return BL_EINVAL;
  } else if (epi == nullptr)
    hipLaunchKernelGGL((gemm_rows_x6_kernel<false, -1>), grid, dim3(256), 0, (hipStream_t)stream, X6_ARGS);
  else if (e.act == BL_ACT_TANH)
    hipLaunchKernelGGL((gemm_rows_x6_kernel<false, BL_ACT_TANH>), grid, dim3(256), 0, (hipStream_t)stream, X6_ARGS);
  else if (e.act == BL_ACT_RELU)
    hipLaunchKernelGGL((gemm_rows_x6_kernel<false, BL_ACT_RELU>), grid, dim3(256), 0, (hipStream_t)stream, X6_ARGS);
  else if (e.act == BL_ACT_SIGMOID)
    hipLaunchKernelGGL((gemm_rows_x6_kernel<false, BL_ACT_SIGMOID>), grid, dim3(256), 0, (hipStream_t)stream, X6_ARGS);
  else if (e.act == BL_ACT_NONE)
    hipLaunchKernelGGL((gemm_rows_x6_kernel<false, BL_ACT_NONE>), grid, dim3(256), 0, (hipStream_t)stream, X6_ARGS);
  else {
    bl_set_error("%s: activation %d has no bf16x6 epilogue (none / relu / sigmoid / tanh)", who, e.act);
    return BL_EINVAL;
  }
  BL_LAUNCH_CHECK(who);
  return BL_OK;
}

}  // namespace
extern int g_h3_kchunk_cap;  // csrc/bl_gemm_h3.hip
namespace {
bool g_wgrad_wide = true;  // bl_set_wgrad_tile: A/B switch between the 256 x 128 and the 128 x 128 weight-gradient tile
// Largest number of rows one workgroup reduces before it flushes its output tile (bl_set_wgrad_kchunk_cap).  Every flush is
// tile-size fp32 atomics, and the chip retires ~312 G of those per second whatever the addresses (tools/atomic_bench.py):
// at c2's layer shape (E = 640 000, K = 256, N = 128) the 864-row chunks of the old cap (1024) were 97 MB = 24 M atomics per
// launch, ~0.08 ms of a 0.25-ms kernel; the cap trades that against the balance of the last round of workgroups.  Measured
// (profiles/r04e_kcap_*.log, same box): H = 128 layer 0.254 / 0.232 / 0.210 / 0.220 / 0.217 ms at 1024 / 2048 / 3072 / 4096 /
// 8192, concat layer 0.921 / 0.863 / 0.824 / 0.829 / 0.908 ms at 1024 / 2048 / 4096 / 8192 / 16384.
int g_wgrad_kchunk_cap = 4096;

template <bool ROUTED>
int wgrad_x6_resident() {
  static int resident = 0;
  if (resident == 0) {
    int per_cu = 0;
    hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_wgrad_x6_kernel<ROUTED>, 256, 0);
    if (oe != hipSuccess || per_cu <= 0) per_cu = 2;
    resident = per_cu * bl_num_cus();
  }
  return resident;
}

int gemm_wgrad_x6_impl(const char* who, const bl_rows_packed_t* a, const uint16_t* g_packed, const int32_t* g_idx,
                       const uint32_t* win_bits, int32_t ld_bits, const int32_t* group_ptr, const int32_t* group_w, int32_t G,
                       int32_t M, int32_t N, int32_t K, float* gw, int64_t gw_group_stride, int32_t ld_gw, void* stream) {
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
  BL_CHECK_ARG(M > 0 && N > 0 && N % 32 == 0 && g_packed && gw && bl_aligned16(g_packed),
               "%s: N a multiple of 32 and aligned pointers required", who);
  const bool routed = win_bits != nullptr;
  BL_CHECK_ARG(!routed || (g_idx && ld_bits * 32 >= N), "%s: the routed form needs g_idx and ld_bits >= N / 32", who);
  // the 256 x 128 tile (8 waves, routed operand staged once per message) when every 128-feature half lies in one source
  bool wide = g_wgrad_wide && K % 256 == 0;
  int nidx = g_idx ? 1 : 0;
  for (int j = 0; j < a->nsrc; ++j) {
    wide = wide && a->width[j] % 128 == 0;
    nidx += a->idx[j] ? 1 : 0;
  }
  const bool gather = nidx == a->nsrc + 1;
  wide = wide && (gather || nidx == 0) && (gather || !routed);
  // plain (direct-row) weight gradients with few rows -- the sequence models' Linears, 16 384 token rows -- are faster on the
  // 128 x 128 tile (two workgroups per CU): 58 vs 64 us per launch at seq-great's shapes (profiles/r03q / r04m seq kernel stats)
  if (!gather && M < 65536) wide = false;
  // rows reduced by one workgroup: an integer number of rounds of resident workgroups (see bl_gemm.hip)
  const int resident = wide ? bl_num_cus() : (routed ? wgrad_x6_resident<true>() : wgrad_x6_resident<false>());
  const int ntiles_n = (N + XBN - 1) / XBN;
  const int ntiles_all = ((K + (wide ? 255 : XBM - 1)) / (wide ? 256 : XBM)) * ntiles_n;
  const int extra = (group_ptr ? G : 0) * ntiles_all;
  int kchunk = 256;
  for (int rounds = 1; rounds <= 64; ++rounds) {
    const long long slots = (long long)resident * rounds - extra;
    if (slots <= 0) continue;
    const long long kc = ((long long)M * ntiles_all + slots - 1) / slots;
    if (kc <= g_wgrad_kchunk_cap || rounds == 64) {
      kchunk = (int)((kc + 31) / 32 * 32);
      break;
    }
  }
  if (kchunk < 256) kchunk = 256;
  dim3 grid((M + kchunk - 1) / kchunk + (group_ptr ? G : 0), ntiles_all);
  unsigned* order_ctr = group_w ? nullptr : bl_order_counters((group_ptr ? G : 1) * ntiles_all, stream);
  const int xcd = order_ctr ? 0 : 1;  // ordered flushes want "lower chunk = lower workgroup id"
#define WX6_ARGS                                                                                                               \
  reinterpret_cast<const uint4*>(a->xp[0]), a->nsrc > 1 ? reinterpret_cast<const uint4*>(a->xp[1]) : nullptr,                  \
      a->nsrc > 2 ? reinterpret_cast<const uint4*>(a->xp[2]) : nullptr, a->idx[0], a->nsrc > 1 ? a->idx[1] : nullptr,          \
      a->nsrc > 2 ? a->idx[2] : nullptr, a->width[0], a->nsrc > 1 ? a->width[1] : 0, a->nsrc > 2 ? a->width[2] : 0, koff[1],   \
      koff[2], a->nsrc, reinterpret_cast<const uint4*>(g_packed), g_idx, win_bits, ld_bits, group_ptr, group_w, G, M, N, K,     \
      kchunk, gw, (long long)gw_group_stride, ld_gw, ntiles_n, xcd, order_ctr
  if (wide && routed)
    hipLaunchKernelGGL((gemm_wgrad_x6_wide_kernel<true, true>), grid, dim3(512), 0, (hipStream_t)stream, WX6_ARGS);
  else if (wide && gather)
    hipLaunchKernelGGL((gemm_wgrad_x6_wide_kernel<false, true>), grid, dim3(512), 0, (hipStream_t)stream, WX6_ARGS);
  else if (wide)
    hipLaunchKernelGGL((gemm_wgrad_x6_wide_kernel<false, false>), grid, dim3(512), 0, (hipStream_t)stream, WX6_ARGS);
  else if (routed)
    hipLaunchKernelGGL((gemm_wgrad_x6_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, WX6_ARGS);
  else
    hipLaunchKernelGGL((gemm_wgrad_x6_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, WX6_ARGS);
  BL_LAUNCH_CHECK(who);
  return BL_OK;
}
}  // namespace

// 256: the wide (256 x 128, 8-wave) weight-gradient tile where it applies (default); 128: the 128 x 128 tile everywhere.
// A measurement switch (tools/gemm_bench.py); returns the previous setting.
extern "C" int32_t bl_set_wgrad_tile(int32_t rows) {
  const int32_t prev = g_wgrad_wide ? 256 : 128;
  g_wgrad_wide = rows != 128;
  return prev;
}

// Rows per workgroup of the bf16x6 weight-gradient GEMMs: the chunk is the smallest one that fills an integer number of rounds
// of resident workgroups and is <= cap rows.  Returns the previous cap.
extern "C" int32_t bl_set_wgrad_kchunk_cap(int32_t rows) {
  const int32_t prev = g_wgrad_kchunk_cap;
  if (rows >= 256) g_wgrad_kchunk_cap = g_h3_kchunk_cap = rows;  // (the f16x3 weight gradient follows the same cap)
  return prev;
}

extern "C" int bl_gemm_rows_x6(const bl_rows_packed_t* a, const uint32_t* win_bits, int32_t ld_bits, const uint16_t* bp,
                               int64_t b_group_stride, const int32_t* group_ptr, const int32_t* group_w, int32_t G,
                               int32_t M, int32_t N, int32_t K, float* c, int32_t ldc, void* stream) {
  return gemm_rows_x6_impl("bl_gemm_rows_x6", a, win_bits, ld_bits, bp, b_group_stride, group_ptr, group_w, G, M, N, K, nullptr, c, ldc,
                           stream);
}

extern "C" int bl_gemm_rows_x6_epi(const bl_rows_packed_t* a, const uint16_t* bp, int64_t b_group_stride, const int32_t* group_ptr,
                                   const int32_t* group_w, int32_t G, int32_t M, int32_t N, int32_t K, const float* bias,
                                   int32_t act, bl_dropout_t drop, float* c, int32_t ldc, void* stream) {
  BL_CHECK_ARG((uint64_t)M * (uint64_t)N < (1ull << 32) || drop.p <= 0.f, "bl_gemm_rows_x6_epi: dropout index space is 32 bit");
  BL_CHECK_ARG(bias == nullptr || bl_aligned16(bias), "bl_gemm_rows_x6_epi: misaligned bias");
  const bl_drop_dev d = bl_make_drop(drop);
  X6Epi e = {bias, act, d.key, d.thresh, d.scale, 0, nullptr, 0, nullptr, 1.f, nullptr, nullptr};
  return gemm_rows_x6_impl("bl_gemm_rows_x6_epi", a, nullptr, 0, bp, b_group_stride, group_ptr, group_w, G, M, N, K, &e, c, ldc, stream);
}

extern "C" int bl_gemm_rows_x6_epi2(const bl_rows_packed_t* a, const uint16_t* bp, int32_t M, int32_t N, int32_t K, const bl_x6_epi_t* epi,
                                    float* c, int32_t ldc, void* stream) {
  BL_CHECK_ARG(epi, "bl_gemm_rows_x6_epi2: null epilogue");
  BL_CHECK_ARG((uint64_t)M * (uint64_t)N < (1ull << 32) || epi->drop.p <= 0.f, "bl_gemm_rows_x6_epi2: dropout index space is 32 bit");
  BL_CHECK_ARG(epi->bias == nullptr || bl_aligned16(epi->bias), "bl_gemm_rows_x6_epi2: misaligned bias");
  const bool packs = epi->form == BL_X6_EPI_ACT_PACK || epi->form == BL_X6_EPI_MASK_PACK;
  BL_CHECK_ARG(!packs || (epi->c_packed && bl_aligned16(epi->c_packed) && N % 8 == 0),
               "bl_gemm_rows_x6_epi2: the packed result needs a 16-byte aligned buffer and N %% 8 == 0");
  BL_CHECK_ARG(epi->form != BL_X6_EPI_RES || (epi->res && bl_aligned16(epi->res) && epi->ld_res % 4 == 0 && epi->ld_res >= N),
               "bl_gemm_rows_x6_epi2: the residual needs an aligned [M, ld_res >= N] matrix");
  BL_CHECK_ARG(epi->form != BL_X6_EPI_MASK_PACK || (epi->y_packed && bl_aligned16(epi->y_packed)),
               "bl_gemm_rows_x6_epi2: the masked form needs the packed forward output");
  BL_CHECK_ARG(epi->form != BL_X6_EPI_MASK_PACK || epi->colsum == nullptr || !bl_get_deterministic(),
               "bl_gemm_rows_x6_epi2: the column sums are flushed by unordered atomics (deterministic mode is on)");
  const bl_drop_dev d = bl_make_drop(epi->drop);
  X6Epi e = {epi->bias, epi->act, d.key, d.thresh, d.scale, epi->form, epi->res, epi->ld_res,
             reinterpret_cast<const uint2*>(epi->y_packed), epi->mask_scale, epi->colsum, reinterpret_cast<uint2*>(epi->c_packed)};
  // (the packed forms never touch c: any non-null aligned pointer passes the common checks)
  float* cc = packs ? reinterpret_cast<float*>(epi->c_packed) : c;
  return gemm_rows_x6_impl("bl_gemm_rows_x6_epi2", a, nullptr, 0, bp, 0, nullptr, nullptr, 1, M, N, K, &e, cc, packs ? N : ldc, stream);
}

extern "C" int bl_gemm_wgrad_routed_x6(const bl_rows_packed_t* a, const uint16_t* g_node_packed, const int32_t* g_idx,
                                       const uint32_t* win_bits, int32_t ld_bits, const int32_t* group_ptr,
                                       const int32_t* group_w, int32_t G, int32_t M, int32_t N, int32_t K, float* gw,
                                       int64_t gw_group_stride, int32_t ld_gw, void* stream) {
  BL_CHECK_ARG(M == 0 || (g_idx && win_bits), "bl_gemm_wgrad_routed_x6: needs g_idx and the winner bitmask");
  return gemm_wgrad_x6_impl("bl_gemm_wgrad_routed_x6", a, g_node_packed, g_idx, win_bits, ld_bits, group_ptr, group_w, G, M, N, K, gw,
                            gw_group_stride, ld_gw, stream);
}

extern "C" int bl_gemm_wgrad_x6(const bl_rows_packed_t* a, const uint16_t* g_packed, const int32_t* g_idx, const int32_t* group_ptr,
                                const int32_t* group_w, int32_t G, int32_t M, int32_t N, int32_t K, float* gw,
                                int64_t gw_group_stride, int32_t ld_gw, void* stream) {
  return gemm_wgrad_x6_impl("bl_gemm_wgrad_x6", a, g_packed, g_idx, nullptr, 0, group_ptr, group_w, G, M, N, K, gw, gw_group_stride, ld_gw,
                            stream);
}
