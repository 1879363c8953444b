// bf16x6 row GEMM, fourth structure: the second structure's data movement (operands DMA'd into a swizzled,
// double-buffered LDS image, 256 x 128 tile, 8 waves) under a PING-PONG schedule: the two waves that share a
// SIMD (waves w and w + 4) alternate between a "load segment" (fragment ds_reads + the next stage's DMA
// pieces) and an "MFMA segment" (12 MFMAs of one 32 x 32 quadrant), kept in lock step by two s_barriers per
// phase and one barrier of initial offset -- the matrix pipe always has one wave streaming MFMAs while its
// partner's loads are in flight.  Four phases per 32-k stage (quadrants 00, 01, 11, 10 of the wave's 64 x 64).
//
// hipcc keeps every ds_read behind every LDS-DMA in flight (s_waitcnt vmcnt(0)); here the fragment reads are
// inline asm (the compiler does not see an LDS read), waited for by hand (lgkmcnt(0) before the MFMAs), the DMA
// by a hand-placed vmcnt(0) in phase 3, and the barriers are raw s_barrier.
// Accumulation order per accumulator is the first structure's: results are bit-identical to bl_gemm_rows_x6.
#include <stdio.h>
#include <stdlib.h>

#include "bl_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef V4BM
#define V4BM 256
#endif
#define V4BN (384 - V4BM)   // 256 x 128 (round 2) or 128 x 256 (round 5: the GATHERED operand staged once per 256 columns)
#define V4NA (V4BM / 128)    // 16-row blocks of the A image a wave fills per stage
#define V4NB (V4BN * 3 / 128) // 1 KB chunks of the weight block a wave fills per stage
#define V4BLK (V4BN * 12)    // uint4 per (group, column tile, stage) weight block
#define V4_STAGE_UINT4 ((V4BM + V4BN) * 12)  // uint4 per stage buffer: (256 + 128) rows x 3 planes x 4 k-groups
#define V4_STAGE_BYTES (V4_STAGE_UINT4 * 16)  // 73728
#define V4_A_PLANE_BYTES (V4BM * 64)          // 16384
#define V4_B_OFF_BYTES (V4BM * 12 * 16)       // 49152: B image behind the A image
#define V4_B_PLANE_BYTES (V4BN * 64)          // 8192

__device__ __forceinline__ void glds16(const uint4* g, uint4* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// v2 weight image (the LDS stage image of the B operand): slot (plane p, column n, k-group kg) at uint4 index
// (p * 128 + n) * 4 + (kg ^ ((n >> 2) & 3)) of the 24 KB block of (group, 128-column tile, 32-k stage).
__global__ __launch_bounds__(256) void pack_weights_v4_kernel(const float* __restrict__ w, int G, int K, int N, int w_is_kn,
                                                              uint4* __restrict__ out) {
  const int nst = K >> 5, ntn = (N + V4BN - 1) / V4BN;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)G * ntn * nst * (V4BN * 4)) return;
  const int r = (int)(t % (V4BN * 4));
  const long long blk = t / (V4BN * 4);
  const int st = (int)(blk % nst), tile = (int)((blk / nst) % ntn), g = (int)(blk / ((long long)nst * ntn));
  int n_lo, kg;
  if (w_is_kn) { n_lo = r % V4BN; kg = r / V4BN; }
  else { kg = r & 3; n_lo = r >> 2; }
  const int n = tile * V4BN + n_lo, k0 = st * 32 + 8 * kg;
  uint16_t h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = 0.f;
    if (n < N) v = w_is_kn ? w[((size_t)g * K + k0 + j) * N + n] : w[((size_t)g * N + n) * K + k0 + j];
    split3(v, h[j], m[j], l[j]);
  }
#define PK(a_, b_) ((uint32_t)(a_) | ((uint32_t)(b_) << 16))
  uint4* o = out + (size_t)blk * V4BLK + n_lo * 4 + (kg ^ ((n_lo >> 2) & 3));
  o[0] = make_uint4(PK(h[0], h[1]), PK(h[2], h[3]), PK(h[4], h[5]), PK(h[6], h[7]));
  o[V4BN * 4] = make_uint4(PK(m[0], m[1]), PK(m[2], m[3]), PK(m[4], m[5]), PK(m[6], m[7]));
  o[V4BN * 8] = make_uint4(PK(l[0], l[1]), PK(l[2], l[3]), PK(l[4], l[5]), PK(l[6], l[7]));
}

__device__ __forceinline__ bool v4_find_piece(const int* __restrict__ group_ptr, int G, int M, int piece, int t, int& g,
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

// one ds_read_b128 the compiler does not know about: LDS byte address in a VGPR + a literal offset
#define LDS_RD(dst_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(off_) : "memory")
// all fragment reads issued so far have landed; the operands tie the MFMAs behind the wait
#define LDS_WAIT6(a_, b_, c_, d_, e_, f_) \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_), "+v"(b_), "+v"(c_), "+v"(d_), "+v"(e_), "+v"(f_)::"memory")

#define MF(b_, a_, acc_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b_), __builtin_bit_cast(bf16x8, a_), acc_, 0, 0, 0)
// the six terms of one 16-k step, small terms first (A fragment planes a0/a1/a2 = hi/mid/lo, same for B)
#define SIX(acc_, A_, B_, s_)            \
  MF(B_[s_][1], A_[s_][1], acc_);        \
  MF(B_[s_][2], A_[s_][0], acc_);        \
  MF(B_[s_][0], A_[s_][2], acc_);        \
  MF(B_[s_][1], A_[s_][0], acc_);        \
  MF(B_[s_][0], A_[s_][1], acc_);        \
  MF(B_[s_][0], A_[s_][0], acc_);

template <bool PINGPONG, bool NODMA>
__global__ __launch_bounds__(512, 2) void gemm_rows_x6v4_kernel(
    const uint4* __restrict__ xp0, const uint4* __restrict__ xp1, const int* __restrict__ idx0, const int* __restrict__ idx1,
    int w0, int w1, int koff1, int nsrc, const uint4* __restrict__ bp, long long strideB, const int* __restrict__ group_ptr,
    const int* __restrict__ group_w, int G, int M, int N, int K, float* __restrict__ c, int ldc, long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];  // [2 buffers][A: 3 x 256 x 4 | B: 3 x 128 x 4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool trace = dbg != nullptr && (blockIdx.x % 97) == 5 && blockIdx.y == 0 && lane == 0;  // a few workgroups across the grid
  long long* tr = dbg + ((blockIdx.x / 97) * 8 + wave) * 8;
  if (trace) tr[0] = wall_clock64();
  int g, row0, nrows, tile_y;
  {
    const int lin = blockIdx.x + blockIdx.y * gridDim.x, total = gridDim.x * gridDim.y;
    const int q = total >> 3, r = total & 7, cx = lin & 7;  // every XCD gets one contiguous range of work items
    const int v = cx * q + min(cx, r) + (lin >> 3);
    const int tx = v / gridDim.y;
    tile_y = v - tx * gridDim.y;
    if (!v4_find_piece(group_ptr, G, M, V4BM, tx, g, row0, nrows)) return;
  }
  const int n0 = tile_y * V4BN;
  const int wsel = group_w ? group_w[g] : g;
  const int nk = K >> 5;
  const uint4* __restrict__ Bt = bp + (long long)wsel * strideB + (size_t)tile_y * nk * V4BLK;

  // DMA mapping (as the second structure): wave w fills row blocks 2w, 2w+1 (16 rows each) of the three A planes,
  // lane l -> row (2w+i) * 16 + (l >> 2), physical slot l & 3 = logical k-group (l & 3) ^ ((row >> 2) & 3);
  // and chunks 3w .. 3w+2 (1 KB each) of the 24 KB weight block.
  const int a_kg = (lane & 3) ^ ((lane >> 4) & 3);
  int gr0[V4NA], gr1[V4NA];
#pragma unroll
  for (int i = 0; i < V4NA; ++i) {
    const int r = row0 + min((V4NA * wave + i) * 16 + (lane >> 2), nrows - 1);
    gr0[i] = idx0 ? idx0[r] : r;
    gr1[i] = nsrc > 1 ? (idx1 ? idx1[r] : r) : 0;
  }
  auto dma_a = [&](int kt, uint4* As, int i) {  // 3 pieces: row block 2w+i, planes 0..2
    const int k0 = kt * 32;
    const bool second = nsrc > 1 && k0 >= koff1;
    const uint4* __restrict__ base = second ? xp1 : xp0;
    const int wq = (second ? w1 : w0) >> 3;
    const int kq = ((second ? k0 - koff1 : k0) >> 3) + a_kg;
    const uint4* src = base + (size_t)(second ? gr1[i] : gr0[i]) * 3 * wq + kq;
#pragma unroll
    for (int p = 0; p < 3; ++p) glds16(src + p * wq, As + (p * V4BM + (V4NA * wave + i) * 16) * 4);
  };
  auto dma_b = [&](int kt, uint4* As, int q0) {  // 3 pieces: chunks q0 .. q0 + 2 of this wave's V4NB
    const uint4* bsrc = Bt + (size_t)kt * V4BLK + lane;
    uint4* Bs = As + V4BM * 12;
#pragma unroll
    for (int q = q0; q < q0 + 3; ++q) glds16(bsrc + (wave * V4NB + q) * 64, Bs + (wave * V4NB + q) * 64);
  };

  const int wm = wave & (V4BM / 64 - 1), wn = wave / (V4BM / 64);  // partners on a SIMD (w, w + 4) share the row block, not the columns
  const int li = lane & 31, half = lane >> 5, swz = (li >> 2) & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // byte addresses (buffer 0) of this lane's fragments: A row wm*64 + ti*32 + li, B column wn*64 + tj*32 + li, k-step s
  uint32_t aa[2][2][2], ab[2][2][2];  // [buffer][tile][k-step]; a literal offset selects the plane (16-bit field)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kg = (2 * s + half) ^ swz;
      aa[0][t][s] = lds0 + ((wm * 64 + t * 32 + li) * 4 + kg) * 16;
      ab[0][t][s] = lds0 + V4_B_OFF_BYTES + ((wn * 64 + t * 32 + li) * 4 + kg) * 16;
      aa[1][t][s] = aa[0][t][s] + V4_STAGE_BYTES;
      ab[1][t][s] = ab[0][t][s] + V4_STAGE_BYTES;
    }

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc00[r] = acc01[r] = acc10[r] = acc11[r] = 0.f;

  // prologue: stage 0 lands before anything else
  if (trace) tr[1] = wall_clock64();
#pragma unroll
  for (int i = 0; i < V4NA; ++i) dma_a(0, smem, i);
#pragma unroll
  for (int q = 0; q < V4NB; q += 3) dma_b(0, smem, q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (trace) tr[2] = wall_clock64();
  const bool late = PINGPONG && wave >= 4;
  if (late) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0

  u32x4 A0[2][3], A1[2][3], B0[2][3], B1[2][3];  // [k-step][plane]

#define RD_A(dst_, t_, BUF_)                                                    \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                               \
    LDS_RD(dst_[s][0], aa[BUF_][t_][s], 0);                                     \
    LDS_RD(dst_[s][1], aa[BUF_][t_][s], V4_A_PLANE_BYTES);                      \
    LDS_RD(dst_[s][2], aa[BUF_][t_][s], 2 * V4_A_PLANE_BYTES);                  \
  }
#define RD_B(dst_, t_, BUF_)                                                    \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                               \
    LDS_RD(dst_[s][0], ab[BUF_][t_][s], 0);                                     \
    LDS_RD(dst_[s][1], ab[BUF_][t_][s], V4_B_PLANE_BYTES);                      \
    LDS_RD(dst_[s][2], ab[BUF_][t_][s], 2 * V4_B_PLANE_BYTES);                  \
  }
#define WAIT_FRAG(F_) LDS_WAIT6(F_[0][0], F_[0][1], F_[0][2], F_[1][0], F_[1][1], F_[1][2])
#define SEG_MFMA(acc_, A_, B_)                \
  __builtin_amdgcn_s_barrier();               \
  WAIT_FRAG(A_);                              \
  WAIT_FRAG(B_);                              \
  __builtin_amdgcn_sched_barrier(0);          \
  __builtin_amdgcn_s_setprio(1);              \
  SIX(acc_, A_, B_, 0)                        \
  SIX(acc_, A_, B_, 1)                        \
  __builtin_amdgcn_s_setprio(0);              \
  __builtin_amdgcn_sched_barrier(0);          \
  __builtin_amdgcn_s_barrier();

  // BUF_ = buffer this stage reads (literal 0/1); the next stage's pieces go to the other one
#define STAGE(kt_, BUF_)                                                         \
  {                                                                              \
    uint4* nxt_ = smem + (1 - (BUF_)) * V4_STAGE_UINT4;                          \
    const bool more_ = !NODMA && (kt_) + 1 < nk;                                           \
    /* phase 0: quadrant 00 */                                                   \
    RD_A(A0, 0, BUF_)                                                            \
    RD_B(B0, 0, BUF_)                                                            \
    if (more_) dma_a((kt_) + 1, nxt_, 0);                                        \
    __builtin_amdgcn_sched_barrier(0);                                           \
    SEG_MFMA(acc00, A0, B0)                                                      \
    /* phase 1: quadrant 01 */                                                   \
    RD_B(B1, 1, BUF_)                                                            \
    if (more_) { if (V4NA > 1) dma_a((kt_) + 1, nxt_, V4NA - 1); else dma_b((kt_) + 1, nxt_, 0); }                                        \
    __builtin_amdgcn_sched_barrier(0);                                           \
    SEG_MFMA(acc01, A0, B1)                                                      \
    /* phase 2: quadrant 11 */                                                   \
    RD_A(A1, 1, BUF_)                                                            \
    if (more_) dma_b((kt_) + 1, nxt_, V4NB - 3);                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    SEG_MFMA(acc11, A1, B1)                                                      \
    /* phase 3: quadrant 10; the next stage has landed before anybody passes this phase's first barrier */ \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             \
    __builtin_amdgcn_sched_barrier(0);                                           \
    SEG_MFMA(acc10, A1, B0)                                                      \
  }

  for (int kt = 0; kt < nk; kt += 2) {
    STAGE(kt, 0)
    STAGE(kt + 1, 1)
  }
  if (PINGPONG && !late) __builtin_amdgcn_s_barrier();
  if (trace) tr[3] = wall_clock64();

#define STORE_ROW(ti_, accA_, accB_)                                                                         \
  {                                                                                                          \
    const int m = wm * 64 + (ti_) * 32 + li;                                                                 \
    if (m < nrows) {                                                                                         \
      float* __restrict__ crow = c + (size_t)(row0 + m) * ldc;                                               \
      _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                     \
        const int na = n0 + wn * 64 + 8 * gq + 4 * half, nb = na + 32;                                       \
        if (na < N) *reinterpret_cast<float4*>(crow + na) = make_float4(accA_[4 * gq], accA_[4 * gq + 1], accA_[4 * gq + 2], accA_[4 * gq + 3]); \
        if (nb < N) *reinterpret_cast<float4*>(crow + nb) = make_float4(accB_[4 * gq], accB_[4 * gq + 1], accB_[4 * gq + 2], accB_[4 * gq + 3]); \
      }                                                                                                      \
    }                                                                                                        \
  }
  STORE_ROW(0, acc00, acc01)
  STORE_ROW(1, acc10, acc11)
  if (trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tr[4] = wall_clock64(); }
}

// ================================================================================================
extern "C" int bl_pack_weights_x6v4(const float* w, int32_t G, int32_t K, int32_t N, int32_t w_is_kn, uint16_t* out,
                                    void* stream) {
  if (G == 0) return 0;
  const long long total = (long long)G * ((N + V4BN - 1) / V4BN) * (K / 32) * (V4BN * 4);
  hipLaunchKernelGGL(pack_weights_v4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, G, K,
                     N, w_is_kn, reinterpret_cast<uint4*>(out));
  return (int)hipGetLastError();
}

static long long* g_v4_dbg = nullptr;
extern "C" void bl_v4_set_trace(long long* p) { g_v4_dbg = p; }

typedef struct {
  const uint16_t* xp[3];
  const int32_t* idx[3];
  int32_t width[3];
  int32_t nsrc;
} rows_packed_t;

// Same contract as bl_gemm_rows_x6 (no routing mask) with the weights packed by bl_pack_weights_x6v4; at most two row
// sources, K a multiple of 64.  pingpong = 0: both wave groups in step (A/B of the schedule alone).
extern "C" int bl_gemm_rows_x6v4(const rows_packed_t* a, const uint16_t* bp, int64_t b_group_stride, const int32_t* group_ptr,
                                 const int32_t* group_w, int32_t G, int32_t M, int32_t N, int32_t K, float* c, int32_t ldc,
                                 int32_t pingpong, void* stream) {
  if (M == 0) return 0;
  if (K % 64 != 0 || a->nsrc < 1 || a->nsrc > 2) return -1;
  const size_t lds = (size_t)2 * V4_STAGE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_rows_x6v4_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_rows_x6v4_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_rows_x6v4_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_rows_x6v4_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid((M + V4BM - 1) / V4BM + (group_ptr ? G : 0), (N + V4BN - 1) / V4BN);
#define V4_ARGS                                                                                                              \
  reinterpret_cast<const uint4*>(a->xp[0]), a->nsrc > 1 ? reinterpret_cast<const uint4*>(a->xp[1]) : nullptr, a->idx[0],       \
      a->nsrc > 1 ? a->idx[1] : nullptr, a->width[0], a->nsrc > 1 ? a->width[1] : 0, a->nsrc > 1 ? a->width[0] : 0, a->nsrc,   \
      reinterpret_cast<const uint4*>(bp), (long long)(b_group_stride / 8), group_ptr, group_w, G, M, N, K, c, ldc, g_v4_dbg
  switch (pingpong) {
    case 1: hipLaunchKernelGGL((gemm_rows_x6v4_kernel<true, false>), grid, dim3(512), lds, (hipStream_t)stream, V4_ARGS); break;
    case 0: hipLaunchKernelGGL((gemm_rows_x6v4_kernel<false, false>), grid, dim3(512), lds, (hipStream_t)stream, V4_ARGS); break;
    case 3: hipLaunchKernelGGL((gemm_rows_x6v4_kernel<true, true>), grid, dim3(512), lds, (hipStream_t)stream, V4_ARGS); break;
    default: hipLaunchKernelGGL((gemm_rows_x6v4_kernel<false, true>), grid, dim3(512), lds, (hipStream_t)stream, V4_ARGS); break;
  }
  return (int)hipGetLastError();
}
