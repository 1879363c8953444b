// bf16x6 row GEMM, third structure: loader waves + MFMA waves (workgroup-level specialisation).
//
// What the first two structures showed (tools/gemm_bench.py, tools/v2_trace.py on MI355X):
//   * v1 (128 x 128, every wave loads, stores to LDS and multiplies): the three phases add up -- all
//     workgroups of a CU run the same phase at the same time;
//   * v2 (operands DMA'd into a double-buffered LDS image): a wave's 9 `global_load_lds` take 900-4000
//     cycles to ISSUE (the CU's address path is the bottleneck, 64-byte gather pieces), and they sit in the
//     same instruction stream as the MFMAs, so each SIMD's matrix pipe is busy one third of a stage;
//   * making every operand row L2-resident changes the time by < 20 %: the limit is inside the CU.
// Here the two jobs run on DIFFERENT waves of one workgroup, so the time is max(load, MFMA), not the sum:
//   * 4 loader waves (one per SIMD): gather the A rows as FP32 -- one full 128-byte line per row and 32-k
//     stage instead of three 64-byte pieces of a bf16x3-packed row (1/3 of the requests, 2/3 of the bytes, and
//     no packing pass over the node states at all) -- split them into the three bf16 planes in registers
//     (v_cvt_pk_bf16_f32, ~4.5 VALU per element, on a SIMD whose matrix pipe is busy anyway), apply the
//     routing mask of the input-gradient GEMM, and write the swizzled LDS stage image; two stages of A loads
//     are in flight per loader thread.  The weights stay pre-packed (bl_pack_weights_x6v2: the block in memory
//     IS the LDS image, six contiguous 16-byte loads per thread).
//   * 8 MFMA waves (two per SIMD, 4 x 2, each 64 x 64 of a 256 x 128 tile): ds_read_b128 fragments + 48 MFMAs
//     per stage, nothing else; they never wait for memory, only for the stage barrier.
//   * two LDS stage buffers (2 x 72 KB), ONE workgroup barrier per 32-k stage.
// Up to four row sources (x_j[idx_sel(j)[r]], concatenated along k): [stash ; cur][src] ; [stash ; cur][tgt]
// of a ConcatResidual layer is read in place -- no concatenated copy of the node states.
// Accumulation order per 16-k step is the first structure's: results are bit-identical to bl_gemm_rows_x6.
#include <stdio.h>
#include <stdlib.h>

#include "bl_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define V3BM 256
#define V3BN 128
#define V3_A_UINT4 (V3BM * 12)                 // uint4 per A stage image: 3 planes x 256 rows x 4 k-groups
#define V3_STAGE_UINT4 ((V3BM + V3BN) * 12)    // A + B
#define V3_LOADERS 256                          // loader threads (4 waves); 512 MFMA threads follow

struct v3_src_t {
  const float* x[4];
  int ld[4];     // row stride (floats)
  int sel[4];    // which index array gathers this source (0 / 1)
  int koff[5];   // first k of source j; koff[nsrc] = K
  int nsrc;
};

__device__ __forceinline__ bool v3_find_piece(const int* __restrict__ group_ptr, int G, int M, int piece, int t, int& g,
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

// 4 fp32 -> three bf16 planes (2 dwords each), x = hi + mid + lo up to 2^-27 |x|  (same split as bl_common.h::split3)
__device__ __forceinline__ void v3_split4(const float4 v, uint2& h, uint2& m, uint2& l) {
  const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
  const f32x2 ra = a - __builtin_convertvector(ha, f32x2), rb = b - __builtin_convertvector(hb, f32x2);
  const bf16x2 ma = __builtin_convertvector(ra, bf16x2), mb = __builtin_convertvector(rb, bf16x2);
  const f32x2 sa = ra - __builtin_convertvector(ma, f32x2), sb = rb - __builtin_convertvector(mb, f32x2);
  const bf16x2 la = __builtin_convertvector(sa, bf16x2), lb = __builtin_convertvector(sb, bf16x2);
  h = make_uint2(__builtin_bit_cast(uint32_t, ha), __builtin_bit_cast(uint32_t, hb));
  m = make_uint2(__builtin_bit_cast(uint32_t, ma), __builtin_bit_cast(uint32_t, mb));
  l = make_uint2(__builtin_bit_cast(uint32_t, la), __builtin_bit_cast(uint32_t, lb));
}

template <bool MASKED>
__global__ __launch_bounds__(768) void gemm_rows_x6v3_kernel(
    const v3_src_t S, const int* __restrict__ idxA, const int* __restrict__ idxB, const uint32_t* __restrict__ win_bits,
    int ld_bits, const uint4* __restrict__ bp, long long strideB, const int* __restrict__ group_ptr,
    const int* __restrict__ group_w, int G, int M, int N, int K, float* __restrict__ c, int ldc, int xcd_remap,
    long long* __restrict__ dbg, int ablate) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];  // [2 buffers][A: 3 x 256 x 4 | B: 3 x 128 x 4]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int g, row0, nrows, tile_y;
  {
    int tx = blockIdx.x;
    tile_y = blockIdx.y;
    if (xcd_remap) {  // every XCD gets one contiguous range of (row tile, column tile) work items
      const int lin = blockIdx.x + blockIdx.y * gridDim.x, total = gridDim.x * gridDim.y;
      const int q = total >> 3, r = total & 7, cx = lin & 7;
      const int v = cx * q + min(cx, r) + (lin >> 3);
      tx = v / gridDim.y;
      tile_y = v - tx * gridDim.y;
    }
    if (!v3_find_piece(group_ptr, G, M, V3BM, tx, g, row0, nrows)) return;
  }
  const int nk = K >> 5;
  const bool trace = dbg != nullptr && blockIdx.x == 300 && blockIdx.y == 0 && lane == 0;
#define V3_STAMP(stage_, slot_) if (trace && (stage_) < 8) dbg[(wave * 8 + (stage_)) * 8 + (slot_)] = clock64();

  if (wave < 4) {
    // ======================================= loader waves =======================================
    if (!(ablate & 16)) __builtin_amdgcn_s_setprio(3);  // their few VALU / VMEM / LDS instructions go ahead of the partner waves' MFMA stream
    const int wsel = group_w ? group_w[g] : g;
    const uint4* __restrict__ Bt = bp + (long long)wsel * strideB + (size_t)tile_y * nk * 1536 + tid;
    // A items of this thread: rows rbase + 32 j (j < 8), four k's 4q .. 4q+3 of the stage
    const int q = tid & 7, rbase = tid >> 3;
    int ga[8], gb[8];  // gathered row ids through the two index arrays
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = row0 + min(rbase + 32 * j, nrows - 1);
      ga[j] = idxA ? idxA[r] : r;
      gb[j] = MASKED ? 0 : (idxB ? idxB[r] : r);  // the routed form has one source
    }
    // LDS byte offset of item (row, q) inside the A image: plane p adds 16 KB, row j adds 2 KB
    const int a_off = rbase * 64 + (((q >> 1) ^ ((rbase >> 2) & 3)) << 4) + ((q & 1) << 3);
    const int mshift = 8 * (q >> 1) + 4 * (q & 1);
    const uint32_t* __restrict__ mbase = MASKED ? win_bits + (size_t)row0 * ld_bits : nullptr;

    float4 ra0[8], ra1[8];
    if (ablate & 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) ra0[j] = ra1[j] = make_float4(1.f + j, 2.f, 3.f, 4.f + tid);
    }
    uint32_t rm0[8], rm1[8];
    uint4 rb0, rb1, rb2, rb3, rb4, rb5;
    if (ablate & 2) rb0 = rb1 = rb2 = rb3 = rb4 = rb5 = make_uint4(tid, 1, 2, 3);

#define V3_ISSUE_A(s_, ra_, rm_)                                                                          \
  {                                                                                                       \
    const int k0_ = (s_) * 32;                                                                            \
    /* source of this stage: explicit selects (indexing the by-value descriptor would put it in scratch) */ \
    const float* xs_ = S.x[0];                                                                            \
    int ld_ = S.ld[0], ko_ = 0, sl_ = S.sel[0];                                                           \
    if (S.nsrc > 1 && k0_ >= S.koff[1]) { xs_ = S.x[1]; ld_ = S.ld[1]; ko_ = S.koff[1]; sl_ = S.sel[1]; } \
    if (S.nsrc > 2 && k0_ >= S.koff[2]) { xs_ = S.x[2]; ld_ = S.ld[2]; ko_ = S.koff[2]; sl_ = S.sel[2]; } \
    if (S.nsrc > 3 && k0_ >= S.koff[3]) { xs_ = S.x[3]; ld_ = S.ld[3]; ko_ = S.koff[3]; sl_ = S.sel[3]; } \
    const float* __restrict__ xb_ = xs_ + (k0_ - ko_) + 4 * q;                                            \
    const bool useb_ = !MASKED && sl_ != 0;                                                                        \
    if (!(ablate & 1)) _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                    \
      ra_[j] = *reinterpret_cast<const float4*>(xb_ + (size_t)(useb_ ? gb[j] : ga[j]) * ld_);             \
      if (MASKED) rm_[j] = mbase[min(rbase + 32 * j, nrows - 1) * ld_bits + (s_)];                        \
    }                                                                                                     \
  }
#define V3_ISSUE_B(s_)                                                                                    \
  {                                                                                                       \
    const uint4* __restrict__ b_ = Bt + (size_t)(s_) * 1536;                                              \
    if (!(ablate & 2)) { rb0 = b_[0]; rb1 = b_[256]; rb2 = b_[512]; rb3 = b_[768]; rb4 = b_[1024]; rb5 = b_[1280]; } \
  }
#define V3_WRITE(buf_, ra_, rm_)                                                                          \
  {                                                                                                       \
    char* As_ = reinterpret_cast<char*>(smem + (buf_) * V3_STAGE_UINT4);                                  \
    uint4* Bs_ = smem + (buf_) * V3_STAGE_UINT4 + V3_A_UINT4 + tid;                                       \
    if (!(ablate & 8)) { Bs_[0] = rb0; Bs_[256] = rb1; Bs_[512] = rb2; Bs_[768] = rb3; Bs_[1024] = rb4; Bs_[1280] = rb5; } \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                       \
      uint2 h_, m_, l_;                                                                                   \
      v3_split4(ra_[j], h_, m_, l_);                                                                      \
      if (MASKED) {                                                                                       \
        const uint32_t b_ = rm_[j] >> mshift;                                                             \
        const uint32_t k0_ = (__builtin_amdgcn_sbfe(b_, 0, 1) & 0x0000FFFFu) | (__builtin_amdgcn_sbfe(b_, 1, 1) & 0xFFFF0000u); \
        const uint32_t k1_ = (__builtin_amdgcn_sbfe(b_, 2, 1) & 0x0000FFFFu) | (__builtin_amdgcn_sbfe(b_, 3, 1) & 0xFFFF0000u); \
        h_.x &= k0_; m_.x &= k0_; l_.x &= k0_;                                                            \
        h_.y &= k1_; m_.y &= k1_; l_.y &= k1_;                                                            \
      }                                                                                                   \
      char* d_ = As_ + a_off + j * 2048;                                                                  \
      if (!(ablate & 8)) {                                                                                \
      *reinterpret_cast<uint2*>(d_) = h_;                                                                 \
      *reinterpret_cast<uint2*>(d_ + 16384) = m_;                                                         \
      *reinterpret_cast<uint2*>(d_ + 32768) = l_;                                                         \
      } else if (h_.x == 0x12345678u) *reinterpret_cast<uint2*>(d_) = l_;                                 \
    }                                                                                                     \
  }

    // Software pipeline (nk even, >= 4): A loads run THREE stages ahead of the MFMA waves (two register sets),
    // B loads two.  Step t (while the MFMA waves compute stage t): write stage t+1 from the set that holds it,
    // then issue B(t+2) and A(t+3) into the registers just freed, then the stage barrier.  The body is
    // straight-line (main loop + an unrolled four-step tail): with branches in it hipcc's wait-count pass merges
    // the pending-load state of all paths and drains every load before the next issue.
    V3_ISSUE_B(0)
    V3_ISSUE_A(0, ra0, rm0)
    V3_ISSUE_A(1, ra1, rm1)
    V3_WRITE(0, ra0, rm0)
    V3_ISSUE_B(1)
    V3_ISSUE_A(2, ra0, rm0)
    __syncthreads();
    int t = 0;
    for (; t < nk - 4; t += 2) {
      V3_STAMP(t, 0)
      V3_WRITE(1, ra1, rm1)            // stage t+1
      V3_STAMP(t, 1)
      V3_ISSUE_B(t + 2)
      V3_ISSUE_A(t + 3, ra1, rm1)
      V3_STAMP(t, 2)
      __syncthreads();
      V3_STAMP(t, 3)
      V3_STAMP(t + 1, 0)
      V3_WRITE(0, ra0, rm0)            // stage t+2
      V3_STAMP(t + 1, 1)
      V3_ISSUE_B(t + 3)
      V3_ISSUE_A(t + 4, ra0, rm0)
      V3_STAMP(t + 1, 2)
      __syncthreads();
      V3_STAMP(t + 1, 3)
    }
    // tail: t = nk - 4
    V3_WRITE(1, ra1, rm1)              // stage nk-3
    V3_ISSUE_B(t + 2)
    V3_ISSUE_A(t + 3, ra1, rm1)        // the last stage
    __syncthreads();
    V3_WRITE(0, ra0, rm0)              // stage nk-2
    V3_ISSUE_B(t + 3)
    __syncthreads();
    V3_WRITE(1, ra1, rm1)              // stage nk-1
    __syncthreads();
    __syncthreads();                   // the MFMA waves' last stage
    return;
  }

  // ========================================= MFMA waves =========================================
  const int cw = wave - 4;
  const int wm = cw >> 1, wn = cw & 1, li = lane & 31, half = lane >> 5;
  const int swz = (li >> 2) & 3;  // rows wm * 64 + ti * 32 + li: (row >> 2) & 3 == (li >> 2) & 3
  const int n0 = tile_y * V3BN;
  const bool pace = (ablate & 64) != 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  __syncthreads();  // stage 0 is in buffer 0
  for (int kt = 0; kt < nk; ++kt) {
    V3_STAMP(kt, 0)
    const uint4* As = smem + (kt & 1) * V3_STAGE_UINT4;
    const uint4* Bs = As + V3_A_UINT4;
#pragma unroll
    for (int s = 0; s < 2; ++s) {  // two 16-k MFMA steps per stage; this lane's 8 k's = group 2s + half
      const int kg = (2 * s + half) ^ swz;
      bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        const uint4* p = As + (wm * 64 + ti * 32 + li) * 4 + kg;
        ah[ti] = __builtin_bit_cast(bf16x8, p[0]);
        am[ti] = __builtin_bit_cast(bf16x8, p[V3BM * 4]);
        al[ti] = __builtin_bit_cast(bf16x8, p[2 * V3BM * 4]);
      }
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        const uint4* p = Bs + (wn * 64 + tj * 32 + li) * 4 + kg;
        bh[tj] = __builtin_bit_cast(bf16x8, p[0]);
        bm[tj] = __builtin_bit_cast(bf16x8, p[V3BN * 4]);
        bl[tj] = __builtin_bit_cast(bf16x8, p[2 * V3BN * 4]);
      }
      // swapped operands (B fragment in the A slot): the accumulator holds the transposed tile, a lane owns
      // 4 consecutive columns of one row -> float4 epilogue stores.  Small terms first.
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          f32x16 a = acc[ti][tj];
#define V3_PACE if (pace) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(1); __builtin_amdgcn_sched_barrier(0); }
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm[tj], am[ti], a, 0, 0, 0);
          V3_PACE
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[tj], ah[ti], a, 0, 0, 0);
          V3_PACE
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[tj], al[ti], a, 0, 0, 0);
          V3_PACE
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm[tj], ah[ti], a, 0, 0, 0);
          V3_PACE
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[tj], am[ti], a, 0, 0, 0);
          V3_PACE
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[tj], ah[ti], a, 0, 0, 0);
          V3_PACE
          acc[ti][tj] = a;
        }
    }
    V3_STAMP(kt, 1)
    __syncthreads();  // stage kt+1 is complete in the other buffer; this one may be overwritten
    V3_STAMP(kt, 2)
  }

#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int m = wm * 64 + ti * 32 + li;
    if (m >= nrows) continue;
    float* __restrict__ crow = c + (size_t)(row0 + m) * ldc;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = n0 + wn * 64 + tj * 32 + 8 * gq + 4 * half;
        if (n >= N) continue;
        *reinterpret_cast<float4*>(crow + n) =
            make_float4(acc[ti][tj][4 * gq + 0], acc[ti][tj][4 * gq + 1], acc[ti][tj][4 * gq + 2], acc[ti][tj][4 * gq + 3]);
      }
  }
}

static int v3_ablate = 0;
extern "C" void bl_v3_set_ablate(int a) { v3_ablate = a; }
static long long* g_v3_dbg = nullptr;
extern "C" void bl_v3_set_trace(long long* p) { g_v3_dbg = p; }

// SIMD placement probe (tools/): which SIMD does wave w of a 768-thread workgroup run on?
__global__ __launch_bounds__(768) void v3_simd_probe_kernel(int* out) {
  const int hw = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);  // HW_REG_HW_ID bits [5:4] = SIMD_ID
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 12 + (threadIdx.x >> 6)] = hw;
}

extern "C" int bl_v3_simd_probe(int32_t* out, int32_t nblocks, void* stream) {
  hipLaunchKernelGGL(v3_simd_probe_kernel, dim3(nblocks), dim3(768), 0, (hipStream_t)stream, out);
  BL_LAUNCH_CHECK("bl_v3_simd_probe");
  return BL_OK;
}

// ================================================================================================
/* C[r, :] = concat_j(x_j[idx[sel_j][r]]) . B_g   (FP32 row sources, weights packed by bl_pack_weights_x6v2).
 * win_bits != NULL: the routed form (one source; bit d of row r keeps k = d). */
extern "C" int bl_gemm_rows_x6v3(const bl_rows4_t* a, const uint32_t* win_bits, int32_t ld_bits, const uint16_t* bp,
                                 int64_t b_group_stride, const int32_t* group_ptr, const int32_t* group_w, int32_t G,
                                 int32_t M, int32_t N, int32_t K, float* c, int32_t ldc, void* stream) {
  if (M == 0) return BL_OK;
  BL_CHECK_ARG(a && a->nsrc >= 1 && a->nsrc <= 4, "bl_gemm_rows_x6v3: rows descriptor needs 1..4 sources");
  v3_src_t S;
  int off = 0;
  for (int j = 0; j < 4; ++j) {
    S.x[j] = nullptr; S.ld[j] = 0; S.sel[j] = 0; S.koff[j] = 0;
  }
  for (int j = 0; j < a->nsrc; ++j) {
    BL_CHECK_ARG(a->x[j] && bl_aligned16(a->x[j]) && a->width[j] > 0 && a->width[j] % 32 == 0 && a->ld[j] % 4 == 0 &&
                     a->ld[j] >= a->width[j] && (a->sel[j] == 0 || a->sel[j] == 1),
                 "bl_gemm_rows_x6v3: source %d: 16-byte aligned pointer, width a multiple of 32, ld a multiple of 4, sel 0/1", j);
    S.x[j] = a->x[j]; S.ld[j] = a->ld[j]; S.sel[j] = a->sel[j]; S.koff[j] = off;
    off += a->width[j];
  }
  for (int j = a->nsrc; j < 5; ++j) S.koff[j] = off;
  S.nsrc = a->nsrc;
  BL_CHECK_ARG(off == K, "bl_gemm_rows_x6v3: K (%d) != sum of source widths (%d)", K, off);
  BL_CHECK_ARG(K % 64 == 0 && K >= 128, "bl_gemm_rows_x6v3: K must be a multiple of 64 and >= 128 (got %d)", K);
  BL_CHECK_ARG(M > 0 && N > 0 && N % 4 == 0 && ldc % 4 == 0 && bp && c && bl_aligned16(bp) && bl_aligned16(c),
               "bl_gemm_rows_x6v3: N/ldc multiples of 4, aligned pointers required");
  BL_CHECK_ARG(b_group_stride % 8 == 0 && (G <= 1 || b_group_stride >= (int64_t)((N + 127) / 128) * (K / 32) * 12288),
               "bl_gemm_rows_x6v3: packed group stride must cover one group's tiled weights (bl_pack_weights_x6v2)");
  BL_CHECK_ARG(win_bits == nullptr || (a->nsrc == 1 && ld_bits * 32 >= K),
               "bl_gemm_rows_x6v3: the routed form needs exactly one source and ld_bits >= K / 32");
  const size_t lds = (size_t)2 * V3_STAGE_UINT4 * sizeof(uint4);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e1 = hipFuncSetAttribute((const void*)gemm_rows_x6v3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e2 = hipFuncSetAttribute((const void*)gemm_rows_x6v3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      bl_set_error("bl_gemm_rows_x6v3: cannot reserve %zu bytes of LDS", lds);
      return (int)(e1 != hipSuccess ? e1 : e2);
    }
    attr_set = true;
  }
  static const int xcd = getenv("BL_XCD_REMAP") ? atoi(getenv("BL_XCD_REMAP")) : 1;
  dim3 grid((M + V3BM - 1) / V3BM + (group_ptr ? G : 0), (N + V3BN - 1) / V3BN);
  if (win_bits)
    hipLaunchKernelGGL((gemm_rows_x6v3_kernel<true>), grid, dim3(768), lds, (hipStream_t)stream, S, a->idx[0], a->idx[1], win_bits,
                       ld_bits, reinterpret_cast<const uint4*>(bp), (long long)(b_group_stride / 8), group_ptr, group_w, G, M, N, K,
                       c, ldc, xcd, g_v3_dbg, v3_ablate);
  else
    hipLaunchKernelGGL((gemm_rows_x6v3_kernel<false>), grid, dim3(768), lds, (hipStream_t)stream, S, a->idx[0], a->idx[1], win_bits,
                       ld_bits, reinterpret_cast<const uint4*>(bp), (long long)(b_group_stride / 8), group_ptr, group_w, G, M, N, K,
                       c, ldc, xcd, g_v3_dbg, v3_ablate);
  BL_LAUNCH_CHECK("bl_gemm_rows_x6v3");
  return BL_OK;
}
