// Grouped, gathered fp32 GEMMs on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// bitwise an fmaf chain, 157 TF/s chip peak).  Two kernels cover every dense contraction on the
// BugLab hot path:
//
//   gemm_rows_kernel   C[r, :] = act(rows(A)[r, :] . B_g + bias)   one B_g per row group (edge type)
//                      (forward messages, dense node update, heads; with B used transposed it is
//                       the input-gradient GEMM)
//   gemm_wgrad_kernel  gW_g += rows(A)[rows of g]^T . G[rows of g]  (reduction over rows, split
//                      in chunks, fp32 atomics)
//
// Tiling (both): 128 x 128 output tile per 256-thread workgroup = 4 waves in a 2 x 2 grid, each
// wave 64 x 64 = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs); BK k's per LDS stage with a
// register prefetch of the next stage while the current one is multiplied.  NBUF = 1: one LDS
// buffer, two barriers per stage; NBUF = 2: two LDS buffers, ONE barrier per stage.
// At fp32 one MFMA occupies its SIMD for 64 cycles, so LDS/L2 traffic per flop is tiny compared
// with a bf16 kernel: the design goal is simply to keep all four matrix pipes issuing.
//
// Operand images in LDS:
//   "MN-major"  [128 rows][BK + 4]: rows are gathered global rows (k contiguous); fragment = ONE
//               ds_read_b128 per lane = 4 MFMA k-steps; the +4 pad makes both the b128 writes
//               (contiguous lane groups) and the b128 reads (rows distinct mod 16) conflict-free.
//   "K-major"   [BK k][128]: global rows are k-lines (n contiguous); fragment = 4 ds_read_b32,
//               lanes 0-31 read 32 consecutive floats -> conflict-free.
// MFMA k-assignment inside a group of 8 k's: lanes 0-31 take k = 8q+s, lanes 32-63 take k = 8q+4+s
// (s = 0..3 are four consecutive MFMAs); A and B use the same assignment so the contraction is
// complete whatever the order.
#include <stdlib.h>

#include "bl_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 128

// Row sources are passed as individual scalar kernel parameters (macro below) and selected with
// value selects: arrays (or select-of-adjacent-fields) inside a by-value kernel argument make hipcc
// copy the whole argument block to scratch and index it there.
struct RowsDev {  // host-side staging only
  const float *x0, *x1, *x2;
  const int *idx0, *idx1, *idx2;
  int ld0, ld1, ld2;
  int koff1, koff2;  // first k of source 1 / 2 (source 0 starts at 0)
  int nsrc;
};
#define ROWS_PARAMS                                                                                         \
  const float *__restrict__ x0, const float *__restrict__ x1, const float *__restrict__ x2,                \
      const int *__restrict__ idx0, const int *__restrict__ idx1, const int *__restrict__ idx2, int ld0,   \
      int ld1, int ld2, int koff1, int koff2, int nsrc
#define ROWS_ARGS(d) d.x0, d.x1, d.x2, d.idx0, d.idx1, d.idx2, d.ld0, d.ld1, d.ld2, d.koff1, d.koff2, d.nsrc

// Locate the (group, first row, row count) of piece `t` when every group is cut in pieces of
// `piece` rows.  Wave-cooperative: 64 group extents per load + a shuffle prefix sum, so the cost is
// one memory latency instead of G dependent scalar loads.
__device__ __forceinline__ bool find_piece(const int* __restrict__ group_ptr, int G, int M, int piece, int t, int& g,
                                           int& row0, int& nrows) {
  if (group_ptr == nullptr) {
    g = 0;
    row0 = t * piece;
    if (row0 >= M) return false;
    nrows = min(piece, M - row0);
    return true;
  }
  const int lane = threadIdx.x & 63;
  int base = 0;
  for (int g0 = 0; g0 < G; g0 += 64) {
    const int gi = g0 + lane;
    const int lo = gi < G ? group_ptr[gi] : 0;
    const int hi = gi < G ? group_ptr[gi + 1] : 0;
    const int nt = (hi - lo + piece - 1) / piece;
    int incl = nt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    const int excl = base + incl - nt;
    const unsigned long long hit = __ballot(t >= excl && t < excl + nt);
    if (hit) {
      const int src = __ffsll((long long)hit) - 1;
      g = g0 + src;
      const int lo_s = __shfl(lo, src, 64), hi_s = __shfl(hi, src, 64), ex_s = __shfl(excl, src, 64);
      row0 = lo_s + (t - ex_s) * piece;
      nrows = min(piece, hi_s - row0);
      return true;
    }
    base += __shfl(incl, 63, 64);
  }
  return false;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 sel4(bool ok, float4 v) { return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f); }

// one MFMA k-group (8 k's) for the 2 x 2 tiles of a wave.  With SWAP the operands are exchanged (B
// fragment in the A slot): the accumulator then holds the TRANSPOSED tile, i.e. lane l owns row
// m = l & 31 of the C tile and registers 4g..4g+3 are the four consecutive columns
// n = 8g + 4*(l >> 5) + 0..3, so the epilogue stores float4s: 16 instead of 64 store instructions per
// wave (measured: message GEMM 0.487 -> 0.462 ms, input-gradient GEMM 0.566 -> 0.510 ms at c2 shapes).
// The weight-gradient kernel keeps the plain layout (col = lane & 31): its fp32 atomics then hit 32
// consecutive floats of one row per instruction.
template <bool SWAP, int TI>
__device__ __forceinline__ void mfma_group(const float (&a_)[TI][4], const float (&b_)[2][4], f32x16 (&acc)[TI][2]) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
        acc[ti][tj] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b_[tj][s], a_[ti][s], acc[ti][tj], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x2f32(a_[ti][s], b_[tj][s], acc[ti][tj], 0, 0, 0);
}

template <bool B_NK, int ACT, int BK, int NBUF, int MINW, bool MASKED, int TBM>
__global__ __launch_bounds__(256, MINW) void gemm_rows_kernel(ROWS_PARAMS, const int* __restrict__ mask_arg,
                                                              int mask_ld, const float* __restrict__ b,
                                                              long long strideB, int ldb,
                                                              const float* __restrict__ bias,
                                                              const int* __restrict__ group_ptr,
                                                              const int* __restrict__ group_w, int G, int M, int N,
                                                              int K, uint32_t drop_key, uint32_t drop_thresh,
                                                              float drop_scale, float* __restrict__ c, int ldc) {
  constexpr int LDS_MN = BK + 4;        // row stride of an MN-major image
  constexpr int LDS_K = 128;            // row stride of a K-major image
  constexpr int TI = TBM / 64;          // 32-row MFMA tiles per wave (waves are 2 x 2 over TBM x 128)
  constexpr int A_SZ = TBM * LDS_MN;    // floats per A buffer
  constexpr int B_SZ = B_NK ? BN * LDS_MN : BK * LDS_K;
  constexpr int NLD = BK / 8;           // float4 loads per thread per stage, B operand (128 x BK)
  constexpr int NLDA = (TBM * BK / 4) / 256;  // ... A operand (TBM x BK)
  constexpr int A_LPR = BK / 4;         // lanes per MN-major line
  constexpr int A_LSTEP = 256 / A_LPR;  // MN-major lines covered by one pass of the block
  __shared__ __attribute__((aligned(16))) float As[NBUF * A_SZ];
  __shared__ __attribute__((aligned(16))) float Bs[NBUF * B_SZ];
  __shared__ int rowidx[3][TBM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g, row0, nrows;
  if (!find_piece(group_ptr, G, M, TBM, blockIdx.x, g, row0, nrows)) return;
  const int n0 = blockIdx.y * BN;
  const int wsel = group_w ? group_w[g] : g;
  const float* __restrict__ Bg = b + (long long)wsel * strideB;

  if (tid < TBM) {
    const int r = row0 + min(tid, nrows - 1);  // rows past the group end re-read its last row (never stored)
    rowidx[0][tid] = idx0 ? idx0[r] : r;
    if (nsrc > 1) rowidx[1][tid] = idx1 ? idx1[r] : r;
    if (nsrc > 2) rowidx[2][tid] = idx2 ? idx2[r] : r;
  }
  __syncthreads();

  float4 ra[NLDA], rb[NLD];
  int4 ia[MASKED ? NLDA : 1];  // winner ids of the routing mask (source 0), compared at LDS-store time
  const int a_c4 = tid % A_LPR, a_line0 = tid / A_LPR;  // MN-major loader
  const int k_c4 = tid & 31, k_line0 = tid >> 5;         // K-major loader: 32 lanes per 512-byte line
  const int nk = (K + BK - 1) / BK;

#define ROWS_LOAD_STAGE(k0_)                                                                              \
  {                                                                                                       \
    const int k_ = (k0_) + 4 * a_c4;                                                                      \
    const bool kok_ = k_ < K;                                                                             \
    const int kc_ = kok_ ? k_ : 0;                                                                        \
    int j_ = 0;                                                                                           \
    if (nsrc > 1 && kc_ >= koff1) j_ = 1;                                                                 \
    if (nsrc > 2 && kc_ >= koff2) j_ = 2;                                                                 \
    const int kl_ = kc_ - (j_ == 0 ? 0 : (j_ == 1 ? koff1 : koff2));                                      \
    const float* base_ = j_ == 0 ? x0 : (j_ == 1 ? x1 : x2);                                              \
    const int ld_ = j_ == 0 ? ld0 : (j_ == 1 ? ld1 : ld2);                                                \
    _Pragma("unroll") for (int i = 0; i < NLDA; ++i) {                                                    \
      const int line_ = a_line0 + A_LSTEP * i;                                                            \
      ra[i] = ld4(base_ + (size_t)rowidx[j_][line_] * ld_ + kl_);                                         \
      if (MASKED)                                                                                         \
        ia[i] = *reinterpret_cast<const int4*>(mask_arg + (size_t)rowidx[0][line_] * mask_ld + kc_);      \
    }                                                                                                     \
    if (!B_NK) {                                                                                          \
      const int n_ = n0 + 4 * k_c4;                                                                       \
      const int nc_ = n_ < N ? n_ : 0;                                                                    \
      _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                   \
        const int kk_ = (k0_) + k_line0 + 8 * i;                                                          \
        const int kkc_ = kk_ < K ? kk_ : 0;                                                               \
        rb[i] = ld4(Bg + (size_t)kkc_ * ldb + nc_);                                                       \
      }                                                                                                   \
    } else {                                                                                              \
      _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                   \
        const int n_ = n0 + a_line0 + A_LSTEP * i;                                                        \
        const int nc_ = n_ < N ? n_ : 0;                                                                  \
        rb[i] = ld4(Bg + (size_t)nc_ * ldb + kc_);                                                        \
      }                                                                                                   \
    }                                                                                                     \
  }
// The zero-fill predicates are applied HERE, not at the load: a select right behind the load makes
// hipcc wait vmcnt(0) before the MFMAs of the current stage and serialises load latency with compute.
#define ROWS_STORE_STAGE(buf_, k0_)                                                                          \
  {                                                                                                          \
    float* As_w = As + (buf_) * A_SZ;                                                                        \
    float* Bs_w = Bs + (buf_) * B_SZ;                                                                        \
    const bool kok_ = (k0_) + 4 * a_c4 < K;                                                                  \
    _Pragma("unroll") for (int i = 0; i < NLDA; ++i) {                                                       \
      float4 av_ = ra[i];                                                                                    \
      if (MASKED) { /* routing mask: keep element k of row r only where r is the recorded winner */          \
        const int rid_ = row0 + a_line0 + A_LSTEP * i;                                                       \
        av_.x = ia[i].x == rid_ ? av_.x : 0.f; av_.y = ia[i].y == rid_ ? av_.y : 0.f;                        \
        av_.z = ia[i].z == rid_ ? av_.z : 0.f; av_.w = ia[i].w == rid_ ? av_.w : 0.f;                        \
      }                                                                                                      \
      *reinterpret_cast<float4*>(&As_w[(a_line0 + A_LSTEP * i) * LDS_MN + 4 * a_c4]) = sel4(kok_, av_);      \
    }                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                        \
      if (!B_NK)                                                                                             \
        *reinterpret_cast<float4*>(&Bs_w[(k_line0 + 8 * i) * LDS_K + 4 * k_c4]) =                             \
            sel4(n0 + 4 * k_c4 < N && (k0_) + k_line0 + 8 * i < K, rb[i]);                                    \
      else                                                                                                   \
        *reinterpret_cast<float4*>(&Bs_w[(a_line0 + A_LSTEP * i) * LDS_MN + 4 * a_c4]) =                     \
            sel4(n0 + a_line0 + A_LSTEP * i < N && kok_, rb[i]);                                              \
    }                                                                                                        \
  }

  f32x16 acc[TI][2];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, half = lane >> 5;

  ROWS_LOAD_STAGE(0)
  ROWS_STORE_STAGE(0, 0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = NBUF == 2 ? (kt & 1) : 0;
    if (kt + 1 < nk) ROWS_LOAD_STAGE((kt + 1) * BK)
    const float* As_ = As + cur * A_SZ;
    const float* Bs_ = Bs + cur * B_SZ;
    // fragments of k-group q+1 are read from LDS while the 16 MFMAs of group q issue
    float fa[2][TI][4], fb[2][2][4];
#define ROWS_READ_FRAGS(q_, slot_)                                                                                      \
  {                                                                                                                     \
    _Pragma("unroll") for (int ti = 0; ti < TI; ++ti) {                                                                 \
      const float4 v = *reinterpret_cast<const float4*>(&As_[(wm * 32 * TI + ti * 32 + li) * LDS_MN + 8 * (q_) + 4 * half]); \
      fa[slot_][ti][0] = v.x; fa[slot_][ti][1] = v.y; fa[slot_][ti][2] = v.z; fa[slot_][ti][3] = v.w;                   \
    }                                                                                                                   \
    _Pragma("unroll") for (int tj = 0; tj < 2; ++tj) {                                                                  \
      if (!B_NK) {                                                                                                      \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                   \
            fb[slot_][tj][s] = Bs_[(8 * (q_) + 4 * half + s) * LDS_K + wn * 64 + tj * 32 + li];                         \
      } else {                                                                                                          \
        const float4 v = *reinterpret_cast<const float4*>(&Bs_[(wn * 64 + tj * 32 + li) * LDS_MN + 8 * (q_) + 4 * half]); \
        fb[slot_][tj][0] = v.x; fb[slot_][tj][1] = v.y; fb[slot_][tj][2] = v.z; fb[slot_][tj][3] = v.w;                  \
      }                                                                                                                 \
    }                                                                                                                   \
  }
    ROWS_READ_FRAGS(0, 0)
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      if (q + 1 < BK / 8) ROWS_READ_FRAGS(q + 1, (q + 1) & 1)
      mfma_group<true, TI>(fa[q & 1], fb[q & 1], acc);
    }
    if (NBUF == 1) {
      __syncthreads();
      if (kt + 1 < nk) {
        ROWS_STORE_STAGE(0, (kt + 1) * BK)
        __syncthreads();
      }
    } else {
      // the other buffer was last read in stage kt-1, before the barrier that ended it
      if (kt + 1 < nk) ROWS_STORE_STAGE(cur ^ 1, (kt + 1) * BK)
      __syncthreads();
    }
  }

  // epilogue (transposed accumulator, see mfma_group): lane owns row m = li of each 32x32 tile and,
  // per register group g, the 4 consecutive columns n = 8g + 4*half + 0..3 -> one float4 store each
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    const int m = wm * 32 * TI + ti * 32 + li;
    if (m >= nrows) continue;
    float* __restrict__ crow = c + (size_t)(row0 + m) * ldc;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = n0 + wn * 64 + tj * 32 + 8 * gq + 4 * half;
        if (n >= N) continue;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4*>(bias + n);
        float v[4] = {acc[ti][tj][4 * gq + 0] + bv.x, acc[ti][tj][4 * gq + 1] + bv.y, acc[ti][tj][4 * gq + 2] + bv.z,
                      acc[ti][tj][4 * gq + 3] + bv.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[u] = bl_act(ACT, v[u]);
          if (drop_thresh) {
            const uint32_t idx = (uint32_t)(row0 + m) * (uint32_t)N + (uint32_t)(n + u);
            v[u] = ((bl_lowbias32(idx + drop_key) >> 8) >= drop_thresh) ? v[u] * drop_scale : 0.f;
          }
        }
        *reinterpret_cast<float4*>(crow + n) = make_float4(v[0], v[1], v[2], v[3]);
      }
  }
}

template <int BK, int NBUF, int MINW, bool GMASK>
__global__ __launch_bounds__(256, MINW) void gemm_wgrad_kernel(ROWS_PARAMS, const float* __restrict__ gc, int ldg,
                                                               const int* __restrict__ g_idx,
                                                               const int* __restrict__ g_mask, int ld_mask,
                                                               const int* __restrict__ group_ptr,
                                                               const int* __restrict__ group_w, int G, int M, int N,
                                                               int K, int kchunk, float* __restrict__ gw_base,
                                                               long long strideW, int ldw, int ntiles_n,
                                                               unsigned* __restrict__ order_ctr) {
  constexpr int LDS_K = 128;
  constexpr int T_SZ = BK * LDS_K;
  constexpr int NLD = BK / 8;
  __shared__ __attribute__((aligned(16))) float As[NBUF * T_SZ];
  __shared__ __attribute__((aligned(16))) float Bs[NBUF * T_SZ];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g, e0, ne;
  if (!find_piece(group_ptr, G, M, kchunk, blockIdx.x, g, e0, ne)) return;
  const int e1 = e0 + ne;
  const int i0 = (blockIdx.y / ntiles_n) * BM;
  const int n0 = (blockIdx.y % ntiles_n) * BN;
  const int wsel = group_w ? group_w[g] : g;

  float4 ra[NLD], rb[NLD];
  int4 ib[GMASK ? NLD : 1];
  const int c4 = tid & 31, line0 = tid >> 5;
  const int fi = i0 + 4 * c4;  // this thread's feature columns of the A rows
  const int nn = n0 + 4 * c4;  // this thread's columns of the G rows
  const bool a_ok = fi < K, b_ok = nn < N;
  const int fic = a_ok ? fi : 0, nnc = b_ok ? nn : 0;
  int aj = 0;
  if (nsrc > 1 && fic >= koff1) aj = 1;
  if (nsrc > 2 && fic >= koff2) aj = 2;
  const int akl = fic - (aj == 0 ? 0 : (aj == 1 ? koff1 : koff2));
  const float* __restrict__ abase = aj == 0 ? x0 : (aj == 1 ? x1 : x2);
  const int* __restrict__ aidx = aj == 0 ? idx0 : (aj == 1 ? idx1 : idx2);
  const int ald = aj == 0 ? ld0 : (aj == 1 ? ld1 : ld2);

#define WGRAD_LOAD_STAGE(k0_)                                                       \
  {                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                               \
      const int e_ = (k0_) + line0 + 8 * i;                                         \
      const bool eok_ = e_ < e1;                                                    \
      const int ec_ = eok_ ? e_ : e0;                                               \
      const int row_ = aidx ? aidx[ec_] : ec_;                                      \
      ra[i] = ld4(abase + (size_t)row_ * ald + akl);                                \
      const int grow_ = g_idx ? g_idx[ec_] : ec_;                                   \
      rb[i] = ld4(gc + (size_t)grow_ * ldg + nnc);                                  \
      if (GMASK) ib[i] = *reinterpret_cast<const int4*>(g_mask + (size_t)grow_ * ld_mask + nnc); \
    }                                                                               \
  }
#define WGRAD_STORE_STAGE(buf_, k0_)                                                                               \
  {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                              \
      const bool eok_ = (k0_) + line0 + 8 * i < e1;                                                                \
      *reinterpret_cast<float4*>(&As[(buf_) * T_SZ + (line0 + 8 * i) * LDS_K + 4 * c4]) = sel4(eok_ && a_ok, ra[i]); \
      float4 bv_ = rb[i];                                                                                          \
      if (GMASK) {                                                                                                 \
        const int eid_ = (k0_) + line0 + 8 * i;                                                                    \
        bv_.x = ib[i].x == eid_ ? bv_.x : 0.f; bv_.y = ib[i].y == eid_ ? bv_.y : 0.f;                              \
        bv_.z = ib[i].z == eid_ ? bv_.z : 0.f; bv_.w = ib[i].w == eid_ ? bv_.w : 0.f;                              \
      }                                                                                                            \
      *reinterpret_cast<float4*>(&Bs[(buf_) * T_SZ + (line0 + 8 * i) * LDS_K + 4 * c4]) = sel4(eok_ && b_ok, bv_);  \
    }                                                                                                              \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, half = lane >> 5;
  const int nk = (ne + BK - 1) / BK;

  WGRAD_LOAD_STAGE(e0)
  WGRAD_STORE_STAGE(0, e0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = NBUF == 2 ? (kt & 1) : 0;
    if (kt + 1 < nk) WGRAD_LOAD_STAGE(e0 + (kt + 1) * BK)
    const float* As_ = As + cur * T_SZ;
    const float* Bs_ = Bs + cur * T_SZ;
    float fa[2][2][4], fb[2][2][4];
#define WGRAD_READ_FRAGS(q_, slot_)                                                                     \
  {                                                                                                     \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                     \
      const int kk = 8 * (q_) + 4 * half + s;                                                           \
      _Pragma("unroll") for (int ti = 0; ti < 2; ++ti) fa[slot_][ti][s] = As_[kk * LDS_K + wm * 64 + ti * 32 + li]; \
      _Pragma("unroll") for (int tj = 0; tj < 2; ++tj) fb[slot_][tj][s] = Bs_[kk * LDS_K + wn * 64 + tj * 32 + li]; \
    }                                                                                                   \
  }
    WGRAD_READ_FRAGS(0, 0)
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      if (q + 1 < BK / 8) WGRAD_READ_FRAGS(q + 1, (q + 1) & 1)
      mfma_group<false, 2>(fa[q & 1], fb[q & 1], acc);
    }
    if (NBUF == 1) {
      __syncthreads();
      if (kt + 1 < nk) {
        WGRAD_STORE_STAGE(0, e0 + (kt + 1) * BK)
        __syncthreads();
      }
    } else {
      if (kt + 1 < nk) WGRAD_STORE_STAGE(cur ^ 1, e0 + (kt + 1) * BK)
      __syncthreads();
    }
  }

  float* __restrict__ gw = gw_base + (long long)wsel * strideW;
  // deterministic mode: the row chunks of one (group, tile) add in chunk order
  unsigned* ctr = order_ctr ? order_ctr + (size_t)g * gridDim.y + blockIdx.y : nullptr;
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

// ------------------------------------------------------------------------------------------------
static int fill_rows(const bl_rows_t* a, RowsDev& d, int& K, const char* who) {
  BL_CHECK_ARG(a != nullptr && a->nsrc >= 1 && a->nsrc <= 3, "%s: rows descriptor needs 1..3 sources", who);
  int off = 0, koff[4] = {0, 0, 0, 0};
  for (int j = 0; j < a->nsrc; ++j) {
    BL_CHECK_ARG(a->x[j] != nullptr && bl_aligned16(a->x[j]), "%s: source %d null or not 16-byte aligned", who, j);
    BL_CHECK_ARG(a->width[j] > 0 && a->width[j] % 4 == 0 && a->ld[j] % 4 == 0 && a->ld[j] >= a->width[j],
                 "%s: source %d width/ld must be multiples of 4 floats (width %d ld %d)", who, j, a->width[j], a->ld[j]);
    koff[j] = off;
    off += a->width[j];
  }
  d.x0 = a->x[0];
  d.idx0 = a->idx[0];
  d.ld0 = a->ld[0];
  d.x1 = a->nsrc > 1 ? a->x[1] : nullptr;
  d.idx1 = a->nsrc > 1 ? a->idx[1] : nullptr;
  d.ld1 = a->nsrc > 1 ? a->ld[1] : 0;
  d.x2 = a->nsrc > 2 ? a->x[2] : nullptr;
  d.idx2 = a->nsrc > 2 ? a->idx[2] : nullptr;
  d.ld2 = a->nsrc > 2 ? a->ld[2] : 0;
  d.koff1 = koff[1];
  d.koff2 = koff[2];
  d.nsrc = a->nsrc;
  K = off;
  return BL_OK;
}

// 32-k stages, one LDS buffer; four workgroups per CU (<= 128 VGPRs): the node-level GEMMs have ~1000 row tiles, which
// is one round at 4 x 256 resident workgroups and 1.3 rounds at 3 x 256.  The routed form needs its registers: two.
#define ROWS_CFG_DEFAULT 32, 1, 4
#define ROWS_CFG_ROUTED 32, 1, 2

static int gemm_rows_impl(const bl_rows_t* a, const int32_t* mask_arg, int32_t mask_ld, const float* b, int64_t b_group_stride, int32_t ldb, int32_t b_is_nk,
                            const float* bias, const int32_t* group_ptr, const int32_t* group_w, int32_t G, int32_t M,
                            int32_t N, int32_t K, int32_t act, bl_dropout_t drop, float* c, int32_t ldc, void* stream) {
  if (M == 0) return BL_OK;
  RowsDev d;
  int Ksum = 0;
  int rc = fill_rows(a, d, Ksum, "bl_gemm_rows");
  if (rc) return rc;
  BL_CHECK_ARG(Ksum == K, "bl_gemm_rows: K (%d) != sum of source widths (%d)", K, Ksum);
  BL_CHECK_ARG(M > 0 && N > 0 && N % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0, "bl_gemm_rows: M>0, N/ldb/ldc multiples of 4 required");
  BL_CHECK_ARG(b && c && bl_aligned16(b) && bl_aligned16(c), "bl_gemm_rows: b/c null or misaligned");
  BL_CHECK_ARG(group_ptr == nullptr || G >= 1, "bl_gemm_rows: G must be >= 1 with group_ptr");
  BL_CHECK_ARG((uint64_t)M * (uint64_t)N < (1ull << 32) || drop.p <= 0.f, "bl_gemm_rows: dropout index space is 32 bit");
  const bl_drop_dev dd = bl_make_drop(drop);
  dim3 grid((M + BM - 1) / BM + (group_ptr ? G : 0), (N + BN - 1) / BN);
  hipStream_t st = (hipStream_t)stream;
#define ROWS_LAUNCH ROWS_ARGS(d), mask_arg, mask_ld, b, (long long)b_group_stride, ldb, bias, group_ptr, group_w, G, M, N, K, dd.key, dd.thresh, dd.scale, c, ldc
  // Few row tiles (the scoring heads: 10^2 - 10^3 candidate rows): a handful of workgroups walk K in 32-wide stages, each a full
  // global-load round trip -- the kernel's time is stages x latency (40 - 70 us for 0.1 GFLOP).  Those launches take 64-row
  // tiles and 128-wide stages instead (one workgroup per CU, 100 KB of LDS): K = 128 ... 384 in 1 - 3 round trips.
  const bool small = !mask_arg && (long long)grid.x * grid.y * 2 <= bl_num_cus();
  if (small) grid.x = (M + 63) / 64 + (group_ptr ? G : 0);
#define ROWS_GO(NK_, ACT_)                                                                                                          \
  do {                                                                                                                              \
    if (small) hipLaunchKernelGGL((gemm_rows_kernel<NK_, ACT_, 128, 1, 1, false, 64>), grid, dim3(256), 0, st, ROWS_LAUNCH);        \
    else hipLaunchKernelGGL((gemm_rows_kernel<NK_, ACT_, ROWS_CFG_DEFAULT, false, BM>), grid, dim3(256), 0, st, ROWS_LAUNCH);       \
  } while (0)
  if (b_is_nk) {
    BL_CHECK_ARG(act == BL_ACT_NONE, "bl_gemm_rows: the transposed-B (input gradient) form takes no activation");
    if (mask_arg)
      hipLaunchKernelGGL((gemm_rows_kernel<true, BL_ACT_NONE, ROWS_CFG_ROUTED, true, BM>), grid, dim3(256), 0, st, ROWS_LAUNCH);
    else
      ROWS_GO(true, BL_ACT_NONE);
  } else {
    BL_CHECK_ARG(mask_arg == nullptr, "bl_gemm_rows_masked: only the transposed-B (input gradient) form takes a routing mask");
    switch (act) {
      case BL_ACT_NONE: ROWS_GO(false, BL_ACT_NONE); break;
      case BL_ACT_RELU: ROWS_GO(false, BL_ACT_RELU); break;
      case BL_ACT_SIGMOID: ROWS_GO(false, BL_ACT_SIGMOID); break;
      case BL_ACT_TANH: ROWS_GO(false, BL_ACT_TANH); break;
      case BL_ACT_GELU: ROWS_GO(false, BL_ACT_GELU); break;
      default: BL_CHECK_ARG(false, "bl_gemm_rows: unknown activation %d", act);
    }
  }
  BL_LAUNCH_CHECK("bl_gemm_rows");
  return BL_OK;
}

extern "C" int bl_gemm_rows(const bl_rows_t* a, const float* b, int64_t b_group_stride, int32_t ldb, int32_t b_is_nk,
                            const float* bias, const int32_t* group_ptr, const int32_t* group_w, int32_t G, int32_t M,
                            int32_t N, int32_t K, int32_t act, bl_dropout_t drop, float* c, int32_t ldc, void* stream) {
  return gemm_rows_impl(a, nullptr, 0, b, b_group_stride, ldb, b_is_nk, bias, group_ptr, group_w, G, M, N, K, act, drop, c,
                        ldc, stream);
}

extern "C" int bl_gemm_rows_routed(const bl_rows_t* a, const int32_t* winner, int32_t ld_winner, const float* b,
                                   int64_t b_group_stride, int32_t ldb, const int32_t* group_ptr, const int32_t* group_w,
                                   int32_t G, int32_t M, int32_t N, int32_t K, float* c, int32_t ldc, void* stream) {
  BL_CHECK_ARG(a && a->nsrc == 1 && a->idx[0] != nullptr && winner != nullptr && ld_winner % 4 == 0,
               "bl_gemm_rows_routed: needs exactly one gathered source and a winner table");
  bl_dropout_t nodrop = {0.f, 0u, 0u};
  return gemm_rows_impl(a, winner, ld_winner, b, b_group_stride, ldb, 1, nullptr, group_ptr, group_w, G, M, N, K,
                        BL_ACT_NONE, nodrop, c, ldc, stream);
}

static int gemm_wgrad_impl(const bl_rows_t* a, const float* g_c, int32_t ld_g, const int32_t* g_idx,
                           const int32_t* g_mask, int32_t ld_mask, const int32_t* group_ptr,
                             const int32_t* group_w, int32_t G, int32_t M, int32_t N, int32_t K, float* gw,
                             int64_t gw_group_stride, int32_t ld_gw, void* stream) {
  if (M == 0) return BL_OK;
  RowsDev d;
  int Ksum = 0;
  int rc = fill_rows(a, d, Ksum, "bl_gemm_wgrad");
  if (rc) return rc;
  BL_CHECK_ARG(Ksum == K, "bl_gemm_wgrad: K (%d) != sum of source widths (%d)", K, Ksum);
  BL_CHECK_ARG(M > 0 && N > 0 && N % 4 == 0 && ld_g % 4 == 0, "bl_gemm_wgrad: N/ld_g multiples of 4 required");
  BL_CHECK_ARG(g_c && gw && bl_aligned16(g_c), "bl_gemm_wgrad: g_c/gw null or misaligned");
  // chunk of rows reduced by one workgroup: large enough to amortise the 128x128 atomic epilogue,
  // small enough that >= ~1000 workgroups exist at minibatch sizes
  // Rows reduced by one workgroup.  The workgroup count should fill an INTEGER number of rounds of
  // resident workgroups (1.24 rounds at the old fixed chunk cost 38 % of this kernel in tail), while
  // staying >= 256 rows so that the 128x128 atomic flush is amortised.
  static int resident_plain = 0, resident_masked = 0;
  int& resident = g_mask ? resident_masked : resident_plain;
  if (resident == 0) {
    int per_cu = 0;
    hipError_t oe = g_mask ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_wgrad_kernel<32, 1, 2, true>, 256, 0)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_wgrad_kernel<32, 1, 2, false>, 256, 0);
    if (oe != hipSuccess || per_cu <= 0) per_cu = 3;
    resident = per_cu * bl_num_cus();
  }
  const int ntiles_all = ((K + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int extra = (group_ptr ? G : 0) * ntiles_all;  // partial last pieces of the groups
  int kchunk = 256;
  for (int rounds = 1; rounds <= 64; ++rounds) {
    const long long slots = (long long)resident * rounds - extra;
    if (slots <= 0) continue;
    const long long kc = ((long long)M * ntiles_all + slots - 1) / slots;
    if (kc <= 1024 || rounds == 64) {
      kchunk = (int)((kc + 31) / 32 * 32);
      break;
    }
  }
  if (kchunk < 256) kchunk = 256;
  const int ntiles_n = (N + BN - 1) / BN;
  dim3 grid((M + kchunk - 1) / kchunk + (group_ptr ? G : 0), ((K + BM - 1) / BM) * ntiles_n);
  // (groups that share a weight slice through group_w have no defined order among themselves: not ordered)
  unsigned* order_ctr = group_w ? nullptr : bl_order_counters((group_ptr ? G : 1) * (int)grid.y, stream);
#define WGRAD_GO(...)                                                                                                      \
  {                                                                                                                        \
    if (g_mask)                                                                                                            \
      hipLaunchKernelGGL((gemm_wgrad_kernel<__VA_ARGS__, true>), grid, dim3(256), 0, (hipStream_t)stream, ROWS_ARGS(d), g_c, \
                         ld_g, g_idx, g_mask, ld_mask, group_ptr, group_w, G, M, N, K, kchunk, gw,                           \
                         (long long)gw_group_stride, ld_gw, ntiles_n, order_ctr);                                            \
    else                                                                                                                   \
      hipLaunchKernelGGL((gemm_wgrad_kernel<__VA_ARGS__, false>), grid, dim3(256), 0, (hipStream_t)stream, ROWS_ARGS(d),   \
                         g_c, ld_g, g_idx, g_mask, ld_mask, group_ptr, group_w, G, M, N, K, kchunk, gw,                      \
                         (long long)gw_group_stride, ld_gw, ntiles_n, order_ctr);                                            \
  }
  // (128-row stages for launches with few rows, as in bl_gemm_rows, were measured slower here: 37 - 45 vs 28 - 43 us per launch
  // at the heads' shapes, profiles/r04r_trace.csv)
  WGRAD_GO(32, 1, 2)
  BL_LAUNCH_CHECK("bl_gemm_wgrad");
  return BL_OK;
}

extern "C" int bl_gemm_wgrad(const bl_rows_t* a, const float* g_c, int32_t ld_g, const int32_t* group_ptr,
                             const int32_t* group_w, int32_t G, int32_t M, int32_t N, int32_t K, float* gw,
                             int64_t gw_group_stride, int32_t ld_gw, void* stream) {
  return gemm_wgrad_impl(a, g_c, ld_g, nullptr, nullptr, 0, group_ptr, group_w, G, M, N, K, gw, gw_group_stride, ld_gw, stream);
}

extern "C" int bl_gemm_wgrad_routed(const bl_rows_t* a, const float* g_node, int32_t ld_g, const int32_t* g_idx,
                                    const int32_t* winner, int32_t ld_winner, const int32_t* group_ptr,
                                    const int32_t* group_w, int32_t G, int32_t M, int32_t N, int32_t K, float* gw,
                                    int64_t gw_group_stride, int32_t ld_gw, void* stream) {
  BL_CHECK_ARG(g_idx != nullptr && winner != nullptr && ld_winner % 4 == 0, "bl_gemm_wgrad_routed: needs g_idx and a winner table");
  return gemm_wgrad_impl(a, g_node, ld_g, g_idx, winner, ld_winner, group_ptr, group_w, G, M, N, K, gw, gw_group_stride, ld_gw,
                         stream);
}
