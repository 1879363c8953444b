// bf16x6 row GEMM, second structure: 256-row tiles, operands DMA'd straight into LDS, two LDS stage
// buffers, ONE barrier per 32-k stage.
//
// Why (measured on the first structure, csrc/bl_gemm_x6.hip: 128 x 128 tile, register-staged, single
// LDS buffer, two barriers per stage): its three cost centres -- operand loads, LDS staging writes,
// MFMA -- ADD UP instead of overlapping (0.12 + 0.08 + 0.125 ms = the 0.32 ms kernel), because every
// workgroup of a CU runs the same phase at the same time.  Here
//   * `global_load_lds_dwordx4` writes the stage image without passing through registers: no staging
//     VGPRs, no ds_write pass, and the loads of stage t+1 are in flight during all of stage t's MFMAs;
//   * the LDS image is [plane][row][k-group ^ swizzle(row)] x 16 B, so that (a) four consecutive lanes
//     still fetch one row's contiguous 64-byte plane segment, (b) a wave's DMA destination is lane-linear
//     (base + 16 * lane, what the instruction requires) and (c) the MFMA fragment reads (`ds_read_b128`,
//     16-lane groups on distinct rows) are bank-conflict free without padding;
//   * a 256-row tile halves the weight-operand traffic per MFMA; 8 waves (4 x 2, each 64 x 64) give two
//     waves per SIMD;
//   * the routing mask of the input-gradient GEMM (bit d of message e = "e won channel d") is applied
//     to the A fragments after the LDS read (12 v_and per fragment set, hidden behind 24 MFMAs).
// Accumulation order per 16-k step is the first structure's, so results are bit-identical to it.
#include <stdio.h>
#include <stdlib.h>

#include "bl_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define V2BN 128
#define V2_STAGE_UINT4(BM) (((BM) + V2BN) * 12)  // uint4 per stage buffer: (BM + 128) rows x 3 planes x 4 k-groups

__device__ __forceinline__ void glds16(const uint4* g, uint4* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// ---- packed weights, v2 image: per (group, 128-column tile, 32-k stage) one 24 KB block that IS the
// LDS stage image of the B operand: slot (plane p, column n, k-group kg) at uint4 index
// (p * 128 + n) * 4 + (kg ^ ((n >> 2) & 3)).  Columns past N are zero.
__global__ __launch_bounds__(256) void pack_weights_v2_kernel(const float* __restrict__ w, int G, int K, int N, int w_is_kn,
                                                              uint4* __restrict__ out) {
  const int nst = K >> 5, ntn = (N + 127) >> 7;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // one (g, tile, stage, n_lo, kg)
  if (t >= (long long)G * ntn * nst * 512) return;
  const int r = (int)(t & 511);
  const long long blk = t >> 9;
  const int st = (int)(blk % nst), tile = (int)((blk / nst) % ntn), g = (int)(blk / ((long long)nst * ntn));
  int n_lo, kg;
  if (w_is_kn) { n_lo = r & 127; kg = r >> 7; }   // consecutive threads -> consecutive n (source [K][N])
  else { kg = r & 3; n_lo = r >> 2; }             // consecutive threads -> consecutive k-groups (source [N][K])
  const int n = tile * 128 + n_lo, k0 = st * 32 + 8 * kg;
  uint16_t h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = 0.f;
    if (n < N) v = w_is_kn ? w[((size_t)g * K + k0 + j) * N + n] : w[((size_t)g * N + n) * K + k0 + j];
    split3(v, h[j], m[j], l[j]);
  }
#define PK(a_, b_) ((uint32_t)(a_) | ((uint32_t)(b_) << 16))
  uint4* o = out + (size_t)blk * 1536 + n_lo * 4 + (kg ^ ((n_lo >> 2) & 3));
  o[0] = make_uint4(PK(h[0], h[1]), PK(h[2], h[3]), PK(h[4], h[5]), PK(h[6], h[7]));
  o[512] = make_uint4(PK(m[0], m[1]), PK(m[2], m[3]), PK(m[4], m[5]), PK(m[6], m[7]));
  o[1024] = make_uint4(PK(l[0], l[1]), PK(l[2], l[3]), PK(l[4], l[5]), PK(l[6], l[7]));
}

// wave-cooperative lookup of the row piece a work item covers (same scheme as bl_gemm_x6.hip)
__device__ __forceinline__ bool v2_find_piece(const int* __restrict__ group_ptr, int G, int M, int piece, int t, int& g,
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

__device__ __forceinline__ uint4 v2_keep_from_bits(uint32_t b) {
  uint4 k;
  k.x = (__builtin_amdgcn_sbfe(b, 0, 1) & 0x0000FFFFu) | (__builtin_amdgcn_sbfe(b, 1, 1) & 0xFFFF0000u);
  k.y = (__builtin_amdgcn_sbfe(b, 2, 1) & 0x0000FFFFu) | (__builtin_amdgcn_sbfe(b, 3, 1) & 0xFFFF0000u);
  k.z = (__builtin_amdgcn_sbfe(b, 4, 1) & 0x0000FFFFu) | (__builtin_amdgcn_sbfe(b, 5, 1) & 0xFFFF0000u);
  k.w = (__builtin_amdgcn_sbfe(b, 6, 1) & 0x0000FFFFu) | (__builtin_amdgcn_sbfe(b, 7, 1) & 0xFFFF0000u);
  return k;
}

__device__ __forceinline__ bf16x8 v2_and(uint4 v, uint4 k) {
  v.x &= k.x; v.y &= k.y; v.z &= k.z; v.w &= k.w;
  return __builtin_bit_cast(bf16x8, v);
}

// BM = 256: 512 threads, waves 4 (rows) x 2 (columns);  BM = 128: 256 threads, waves 2 x 2.
template <int BM, bool MASKED>
__global__ __launch_bounds__(2 * BM, BM == 256 ? 2 : 1) void gemm_rows_x6v2_kernel(
    const uint4* __restrict__ xp0, const uint4* __restrict__ xp1, const int* __restrict__ idx0, const int* __restrict__ idx1,
    int w0, int w1, int koff1, int nsrc, const uint32_t* __restrict__ win_bits, int ld_bits, const uint4* __restrict__ bp,
    long long strideB, const int* __restrict__ group_ptr, const int* __restrict__ group_w, int G, int M, int N, int K,
    float* __restrict__ c, int ldc, int xcd_remap, long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];  // [2 buffers][A: 3 x BM x 4 | B: 3 x 128 x 4]
  constexpr int NW = BM / 32;                                    // waves per workgroup
  constexpr int STAGE = V2_STAGE_UINT4(BM);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    if (!v2_find_piece(group_ptr, G, M, BM, tx, g, row0, nrows)) return;
  }
  const int n0 = tile_y * V2BN;
  const int wsel = group_w ? group_w[g] : g;
  const int nk = K >> 5;
  const uint4* __restrict__ Bt = bp + (long long)wsel * strideB + (size_t)tile_y * nk * 1536;

  // ---- DMA mapping.  A: wave w fills row blocks 2w and 2w+1 (16 rows each) of all three planes: lane l
  // -> row (2w+i) * 16 + (l >> 2), physical slot l & 3, i.e. logical k-group (l & 3) ^ ((row >> 2) & 3).
  const int a_kg = (lane & 3) ^ ((lane >> 4) & 3);
  int gr0[2], gr1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = row0 + min((2 * wave + i) * 16 + (lane >> 2), nrows - 1);
    gr0[i] = idx0 ? idx0[r] : r;
    gr1[i] = nsrc > 1 ? (idx1 ? idx1[r] : r) : 0;
  }
  // B: 24 chunks of 1 KB per stage (the packed block is the LDS image): wave w copies chunks w * CPW ..
  constexpr int CPW = 24 / NW;

  auto stage_load = [&](int kt, int buf) {
    uint4* As = smem + buf * STAGE;
    uint4* Bs = As + BM * 12;
    const int k0 = kt * 32;
    const bool second = nsrc > 1 && k0 >= koff1;
    const uint4* __restrict__ base = second ? xp1 : xp0;
    const int wq = (second ? w1 : w0) >> 3;  // uint4 per plane of a packed row
    const int kq = ((second ? k0 - koff1 : k0) >> 3) + a_kg;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint4* src = base + (size_t)(second ? gr1[i] : gr0[i]) * 3 * wq + kq;
#pragma unroll
      for (int p = 0; p < 3; ++p) glds16(src + p * wq, As + (p * BM + (2 * wave + i) * 16) * 4);  // LDS side: wave base; the DMA adds 16 * lane
    }
    const uint4* bsrc = Bt + (size_t)kt * 1536 + lane;
#pragma unroll
    for (int q = 0; q < CPW; ++q) glds16(bsrc + (wave * CPW + q) * 64, Bs + (wave * CPW + q) * 64);
  };

  constexpr int WN = 2;
  const int wm = wave / WN, wn = wave % WN, li = lane & 31, half = lane >> 5;
  const int swz = (li >> 2) & 3;  // rows wm * 64 + ti * 32 + li: (row >> 2) & 3 == (li >> 2) & 3

  // routing words of this lane's two fragment rows, one 32-bit word per stage, fetched a stage ahead
  uint32_t mw_next[2] = {0u, 0u};
  size_t mrow[2] = {0, 0};
  if (MASKED) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      mrow[ti] = (size_t)(row0 + min(wm * 64 + ti * 32 + li, nrows - 1)) * ld_bits;
      mw_next[ti] = win_bits[mrow[ti]];
    }
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

  stage_load(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool trace = dbg != nullptr && blockIdx.x == 300 && blockIdx.y == 0 && lane == 0 && kt < 8;
    long long* tr = dbg + (wave * 8 + kt) * 8;
    if (trace) tr[0] = clock64();
    uint32_t mw[2] = {0u, 0u};
    if (MASKED) {
      mw[0] = mw_next[0];
      mw[1] = mw_next[1];
    }
    // (1) ALL fragment reads of this stage first: hipcc orders every ds_read behind every LDS-DMA that is in
    // flight (s_waitcnt vmcnt(0) in front of the read), so a read issued after the next stage's DMA would wait
    // for that DMA.  24 x ds_read_b128 -> 96 VGPRs of fragments, then the DMA, then 48 MFMAs that cover it.
    const uint4* As = smem + cur * STAGE;
    const uint4* Bs = As + BM * 12;
    uint4 fa[2][2][3], fb[2][2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {  // two 16-k MFMA steps per stage; this lane's 8 k's = group 2s + half
      const int kg = (2 * s + half) ^ swz;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        const uint4* p = As + (wm * 64 + ti * 32 + li) * 4 + kg;
        fa[s][ti][0] = p[0];
        fa[s][ti][1] = p[BM * 4];
        fa[s][ti][2] = p[2 * BM * 4];
      }
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        const uint4* p = Bs + (wn * 64 + tj * 32 + li) * 4 + kg;
        fb[s][tj][0] = p[0];
        fb[s][tj][1] = p[V2BN * 4];
        fb[s][tj][2] = p[2 * V2BN * 4];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (trace) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tr[1] = clock64(); }
    // (2) next stage's DMA (and routing words) go out while this stage computes
    if (kt + 1 < nk) {
      stage_load(kt + 1, cur ^ 1);
      if (MASKED) {
        mw_next[0] = win_bits[mrow[0] + kt + 1];
        mw_next[1] = win_bits[mrow[1] + kt + 1];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (trace) tr[2] = clock64();
    // (3) MFMAs.  Swapped operands (B fragment in the A slot): the accumulator holds the transposed tile, a
    // lane owns 4 consecutive columns of one row -> float4 epilogue stores.  Small terms first.
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        if (MASKED) {
          const uint4 keep = v2_keep_from_bits(mw[ti] >> (8 * (2 * s + half)));
          ah[ti] = v2_and(fa[s][ti][0], keep);
          am[ti] = v2_and(fa[s][ti][1], keep);
          al[ti] = v2_and(fa[s][ti][2], keep);
        } else {
          ah[ti] = __builtin_bit_cast(bf16x8, fa[s][ti][0]);
          am[ti] = __builtin_bit_cast(bf16x8, fa[s][ti][1]);
          al[ti] = __builtin_bit_cast(bf16x8, fa[s][ti][2]);
        }
      }
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        bh[tj] = __builtin_bit_cast(bf16x8, fb[s][tj][0]);
        bm[tj] = __builtin_bit_cast(bf16x8, fb[s][tj][1]);
        bl[tj] = __builtin_bit_cast(bf16x8, fb[s][tj][2]);
      }
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
    __builtin_amdgcn_sched_barrier(0);  // keep the barrier (and its vmcnt(0)) BEHIND the MFMAs: hipcc hoists it otherwise
    if (trace) { tr[3] = clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tr[4] = clock64(); }
    __syncthreads();  // stage kt+1 has landed (the barrier's vmcnt(0)) and nobody reads buffer `cur` any more
    __builtin_amdgcn_sched_barrier(0);
    if (trace) tr[5] = clock64();
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

// ================================================================================================
extern "C" int bl_pack_weights_x6v2(const float* w, int32_t G, int32_t K, int32_t N, int32_t w_is_kn, uint16_t* out,
                                    void* stream) {
  if (G == 0) return BL_OK;
  BL_CHECK_ARG(w && out && bl_aligned16(out), "bl_pack_weights_x6v2: null or misaligned pointer");
  BL_CHECK_ARG(K > 0 && K % 32 == 0 && N > 0, "bl_pack_weights_x6v2: K must be a multiple of 32 (got %d)", K);
  const long long total = (long long)G * ((N + 127) / 128) * (K / 32) * 512;
  hipLaunchKernelGGL(pack_weights_v2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, G, K,
                     N, w_is_kn, reinterpret_cast<uint4*>(out));
  BL_LAUNCH_CHECK("bl_pack_weights_x6v2");
  return BL_OK;
}

static long long* g_v2_dbg = nullptr;
extern "C" void bl_v2_set_trace(long long* p) { g_v2_dbg = p; }

template <int BM, bool MASKED>
static int launch_v2(const bl_rows_packed_t* a, const uint32_t* win_bits, int32_t ld_bits, const uint16_t* bp,
                     int64_t b_group_stride, const int32_t* group_ptr, const int32_t* group_w, int32_t G, int32_t M, int32_t N,
                     int32_t K, float* c, int32_t ldc, void* stream, long long* dbg) {
  static bool attr_set = false;
  const size_t lds = (size_t)2 * V2_STAGE_UINT4(BM) * sizeof(uint4);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_rows_x6v2_kernel<BM, MASKED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      bl_set_error("bl_gemm_rows_x6v2: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  static const int xcd = getenv("BL_XCD_REMAP") ? atoi(getenv("BL_XCD_REMAP")) : 1;
  dim3 grid((M + BM - 1) / BM + (group_ptr ? G : 0), (N + V2BN - 1) / V2BN);
  hipLaunchKernelGGL((gemm_rows_x6v2_kernel<BM, MASKED>), grid, dim3(2 * BM), lds, (hipStream_t)stream,
                     reinterpret_cast<const uint4*>(a->xp[0]), a->nsrc > 1 ? reinterpret_cast<const uint4*>(a->xp[1]) : nullptr,
                     a->idx[0], a->nsrc > 1 ? a->idx[1] : nullptr, a->width[0], a->nsrc > 1 ? a->width[1] : 0,
                     a->nsrc > 1 ? a->width[0] : 0, a->nsrc, win_bits, ld_bits, reinterpret_cast<const uint4*>(bp),
                     (long long)(b_group_stride / 8), group_ptr, group_w, G, M, N, K, c, ldc, xcd, dbg);
  BL_LAUNCH_CHECK("bl_gemm_rows_x6v2");
  return BL_OK;
}

// Same contract as bl_gemm_rows_x6 with the weights packed by bl_pack_weights_x6v2; at most two row sources.
// tile_rows: 256 (default) or 128.
extern "C" int bl_gemm_rows_x6v2(const bl_rows_packed_t* a, const uint32_t* win_bits, int32_t ld_bits, const uint16_t* bp,
                                 int64_t b_group_stride, const int32_t* group_ptr, const int32_t* group_w, int32_t G,
                                 int32_t M, int32_t N, int32_t K, float* c, int32_t ldc, int32_t tile_rows, void* stream) {
  if (M == 0) return BL_OK;
  BL_CHECK_ARG(a && a->nsrc >= 1 && a->nsrc <= 2, "bl_gemm_rows_x6v2: rows descriptor needs 1 or 2 sources");
  int off = 0;
  for (int j = 0; j < a->nsrc; ++j) {
    BL_CHECK_ARG(a->xp[j] && bl_aligned16(a->xp[j]) && a->width[j] > 0 && a->width[j] % 32 == 0,
                 "bl_gemm_rows_x6v2: source %d: packed pointer 16-byte aligned and width a multiple of 32 required", j);
    off += a->width[j];
  }
  BL_CHECK_ARG(off == K, "bl_gemm_rows_x6v2: K (%d) != sum of source widths (%d)", K, off);
  BL_CHECK_ARG(M > 0 && N > 0 && N % 4 == 0 && ldc % 4 == 0 && bp && c && bl_aligned16(bp) && bl_aligned16(c),
               "bl_gemm_rows_x6v2: N/ldc multiples of 4, aligned pointers required");
  BL_CHECK_ARG(b_group_stride % 8 == 0 && (G <= 1 || b_group_stride >= (int64_t)((N + 127) / 128) * (K / 32) * 12288),
               "bl_gemm_rows_x6v2: packed group stride must cover one group's tiled weights (bl_pack_weights_x6v2)");
  BL_CHECK_ARG(win_bits == nullptr || (a->nsrc == 1 && a->idx[0] && ld_bits * 32 >= K),
               "bl_gemm_rows_x6v2: the routed form needs exactly one gathered source and ld_bits >= K / 32");
  BL_CHECK_ARG(tile_rows == 256 || tile_rows == 128 || tile_rows == 0, "bl_gemm_rows_x6v2: tile_rows must be 128 or 256");
#define V2_GO(BM_, MK_) return launch_v2<BM_, MK_>(a, win_bits, ld_bits, bp, b_group_stride, group_ptr, group_w, G, M, N, K, c, ldc, stream, g_v2_dbg)
  if (tile_rows == 128) {
    if (win_bits) V2_GO(128, true);
    V2_GO(128, false);
  }
  if (win_bits) V2_GO(256, true);
  V2_GO(256, false);
}
