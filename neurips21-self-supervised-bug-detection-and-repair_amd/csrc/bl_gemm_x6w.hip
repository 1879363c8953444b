// bf16x6 row GEMM, wide form: 128 x 256 output tile, 8 waves, operands DMA'd (global_load_lds_dwordx4) into a
// double-buffered, XOR-swizzled LDS image, the two waves of a SIMD in a ping-pong schedule.
//
// Same contract and the same accumulation order per output element as gemm_rows_x6_kernel (bl_gemm_x6.hip): results are
// BIT-IDENTICAL to bl_gemm_rows_x6 -- this is a second schedule of one computation, chosen by shape.  What it is for: the
// message GEMMs with long K and >= 256 output columns (the ConcatResidual layers of the hidden-128 model, every layer of
// the hidden-256 configurations; ptgnn MlpMessagePassingLayer's per-type Linear, call site
// buglab/models/gnnlayerdefs.py:6-23, and its backward).  Measured on MI355X against the 128 x 128 register-staged kernel
// (profiles/r05b_v4_longk.log: forward, E = 640 000 / 320 000 messages):
//     K = 512,  N = 256 (c2 concat layer)   1.000 -> 0.906 ms
//     K = 512,  N = 256 (c3 plain layer)    0.522 -> 0.465 ms
//     K = 1024, N = 512 (c3 concat layer)   1.787 -> 1.571 ms
// and no gain at K = 256 / N = 128 (8 stages per tile: prologue and epilogue are as long as the loop and nothing overlaps
// them at one workgroup per CU) -- the hidden-128 layers stay on bl_gemm_rows_x6.
//
// Structure (round 2's fourth experiment, tools/experiments/README.md, turned by 90 degrees: the GATHERED operand is the
// narrow side of the tile, so a message row is fetched once per 256 output columns instead of once per 128):
//   * stage = 32 k's: A image [plane 3][row 128][slot 4] x 16 B, B image [plane 3][column 256][slot 4] x 16 B, k-group kg of
//     row r in slot kg ^ ((r >> 2) & 3): DMA writes (lane-linear 1 KB pieces) and fragment reads (ds_read_b128, 32 rows x
//     one k-group per half-wave) are both bank-conflict free.  Two stage buffers = 144 KB of LDS: one workgroup per CU.
//   * the weights are packed in exactly that image (bl_pack_weights_x6w): a 48 KB block per (group, 256-column tile, stage),
//     copied by 48 lane-linear DMA pieces (6 per wave).
//   * plain form: the A rows are DMA'd too (wave w gathers rows 16 w .. 16 w + 15: four lanes per 64-byte plane segment).
//     Routed form (input gradient: the left operand is the node gradient of the message's target, AND-masked by the
//     channels the message won): the DMA cannot mask, so A goes global -> registers -> mask -> ds_write_b128 (one
//     (row, k-group) per thread and stage), the weights stay on the DMA.
//   * schedule: waves w and w + 4 share a SIMD; a stage is four phases (the quadrants 00, 01, 11, 10 of the wave's 64 x 64
//     tile, 12 MFMAs each); in every phase one of the two issues fragment reads and its share of the next stage's DMA while
//     the other streams MFMAs; two s_barrier per phase keep them in step.  The fragment reads are inline asm: hipcc orders
//     every compiler-visible LDS read behind every LDS-DMA in flight (s_waitcnt vmcnt(0)), which serialises the pipeline.
//   * epilogue: the result tile leaves through LDS as whole 256-byte row pieces, like bl_gemm_rows_x6's (5 % against stores from
//     the accumulator layout).
// Measured and not kept (commit 488c744, profiles/r05h_rows_wide_schedules.log): a schedule with ONE barrier per stage (every
// wave software-pipelines its own four quadrants with counted lgkmcnt waits, the gathered operand through registers two stages
// ahead) is bit-identical and 1 - 4 % slower in the plain form, 7 - 9 % in the routed one (254 - 256 registers).  Back to back these
// GEMMs hold the package at its 1 400 W limit at 1.80 - 1.82 GHz (profiles/r05i_clock_probe.log); 7.6 % more clock makes them
// 2.9 % faster (tools/gemm_bench.py --mixed, profiles/r05x_mixed_clock.log): co-limited by the core and by the gathered
// operand's delivery -- which is what this tile halves -- not by issue slots.
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "bl_common.h"
#include "bl_x6_locate.h"
#include "bl_x6w_image.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void pack_weights_wide_kernel(const float* __restrict__ w, int G, int K, int N, int w_is_kn,
                                                                uint4* __restrict__ out) {
  pack_weights_wide_thread(w, G, K, N, w_is_kn, out, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- GEMM -------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const uint4* g, uint4* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16,
                                   0, 0);
}

// one ds_read_b128 the compiler does not know about: LDS byte address in a VGPR + a literal offset
#define LDS_RD(dst_, addr_, off_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(off_) : "memory")
// all fragment reads issued so far have landed; the operands tie the MFMAs behind the wait
#define LDS_WAIT6(a_, b_, c_, d_, e_, f_) \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_), "+v"(b_), "+v"(c_), "+v"(d_), "+v"(e_), "+v"(f_)::"memory")

#define MF(b_, a_, acc_) \
  acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b_), __builtin_bit_cast(bf16x8, a_), acc_, 0, 0, 0)
// the six terms of one 16-k step, small terms first (fragment planes 0 / 1 / 2 = hi / mid / lo): bl_gemm_rows_x6's order
#define SIX(acc_, A_, B_, s_)     \
  MF(B_[s_][1], A_[s_][1], acc_); \
  MF(B_[s_][2], A_[s_][0], acc_); \
  MF(B_[s_][0], A_[s_][2], acc_); \
  MF(B_[s_][1], A_[s_][0], acc_); \
  MF(B_[s_][0], A_[s_][1], acc_); \
  MF(B_[s_][0], A_[s_][0], acc_);

template <bool MASKED>
__global__ __launch_bounds__(512, 2) void gemm_rows_x6w_kernel(
    const uint4* __restrict__ xp0, const uint4* __restrict__ xp1, const uint4* __restrict__ xp2, const int* __restrict__ idx0,
    const int* __restrict__ idx1, const int* __restrict__ idx2, int w0, int w1, int w2, int koff1, int koff2, int nsrc,
    const uint32_t* __restrict__ win_bits, int ld_bits, const uint4* __restrict__ bp, long long strideB,
    const int* __restrict__ group_ptr, const int* __restrict__ group_w, int G, int M, int N, int K, float* __restrict__ c,
    int ldc) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];  // [2 buffers][A: 3 x 128 x 4 | B: 3 x 256 x 4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int g, row0, nrows, tile_y;
  if (!x6_locate(group_ptr, G, M, WBM, 1, tile_y, g, row0, nrows)) return;
  const int n0 = tile_y * WBN;
  const int wsel = group_w ? group_w[g] : g;
  const int nk = K >> 5;
  const uint4* __restrict__ Bt = bp + (long long)wsel * strideB + (size_t)tile_y * nk * W_BLK;

  // plain form, A by DMA: wave w fills row block w (16 rows) of the three planes; lane l -> row 16 w + (l >> 2), physical slot
  // l & 3 = logical k-group (l & 3) ^ ((row >> 2) & 3).
  // routed form, A through registers: thread t -> row t >> 2, k-group t & 3.
  const int a_row = MASKED ? (tid >> 2) : (wave * 16 + (lane >> 2));
  const int a_kg = MASKED ? (tid & 3) : ((lane & 3) ^ ((lane >> 4) & 3));
  const int a_grow = row0 + min(a_row, nrows - 1);
  const int gr0 = idx0 ? idx0[a_grow] : a_grow;
  const int gr1 = nsrc > 1 ? (idx1 ? idx1[a_grow] : a_grow) : 0;
  const int gr2 = nsrc > 2 ? (idx2 ? idx2[a_grow] : a_grow) : 0;
  // this thread's / lane's 16 bytes of plane 0 at k = 0 of every source (sources change at multiples of 32 k: the choice is
  // uniform per stage; written as selects -- an index into {xp0, xp1, xp2} puts the table into scratch)
  const int wq0 = w0 >> 3, wq1 = w1 >> 3, wq2 = w2 >> 3;
  const uint4* __restrict__ ap0 = xp0 + (size_t)gr0 * 3 * wq0 + a_kg;
  const uint4* __restrict__ ap1 = nsrc > 1 ? xp1 + (size_t)gr1 * 3 * wq1 + a_kg : ap0;
  const uint4* __restrict__ ap2 = nsrc > 2 ? xp2 + (size_t)gr2 * 3 * wq2 + a_kg : ap0;
#define W_A_SRC(kt_, src_, wq_)                                                        \
  const int k0_ = (kt_) * 32;                                                          \
  const bool s1_ = nsrc > 1 && k0_ >= koff1, s2_ = nsrc > 2 && k0_ >= koff2;           \
  const uint4* src_ = (s2_ ? ap2 : (s1_ ? ap1 : ap0)) + ((k0_ - (s2_ ? koff2 : (s1_ ? koff1 : 0))) >> 3); \
  const int wq_ = s2_ ? wq2 : (s1_ ? wq1 : wq0);
#define W_DMA_A(kt_, As_) /* 3 pieces: row block `wave`, planes 0..2 */                                 \
  {                                                                                                     \
    W_A_SRC(kt_, src_, wq_)                                                                             \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) glds16(src_ + p * wq_, (As_) + (p * WBM + wave * 16) * 4); \
  }
#define W_DMA_B(kt_, As_, q0_) /* 3 of this wave's 6 pieces of the 48 KB weight block */                 \
  {                                                                                                     \
    const uint4* bsrc_ = Bt + (size_t)(kt_) * W_BLK + lane;                                             \
    uint4* Bs_ = (As_) + WBM * 12;                                                                      \
    _Pragma("unroll") for (int q = (q0_); q < (q0_) + 3; ++q) glds16(bsrc_ + (wave * 6 + q) * 64, Bs_ + (wave * 6 + q) * 64); \
  }
  uint4 ra0, ra1, ra2;  // routed form: the next stage's A piece and its routing word
  uint32_t ma = 0;
  const uint32_t* __restrict__ mrow = MASKED ? win_bits + (size_t)a_grow * ld_bits : nullptr;
#define W_LOAD_A(kt_)                  \
  {                                    \
    W_A_SRC(kt_, src_, wq_)            \
    ra0 = src_[0];                     \
    ra1 = src_[wq_];                   \
    ra2 = src_[2 * wq_];               \
    ma = mrow[kt_];                    \
  }
#define W_STORE_A(As_)                                                              \
  {                                                                                 \
    const uint4 keep_ = keep_from_bits(ma >> (8 * a_kg));                           \
    uint4* dst_ = (As_) + a_row * 4 + (a_kg ^ ((a_row >> 2) & 3));                  \
    dst_[0] = make_uint4(ra0.x & keep_.x, ra0.y & keep_.y, ra0.z & keep_.z, ra0.w & keep_.w);           \
    dst_[WBM * 4] = make_uint4(ra1.x & keep_.x, ra1.y & keep_.y, ra1.z & keep_.z, ra1.w & keep_.w);     \
    dst_[WBM * 8] = make_uint4(ra2.x & keep_.x, ra2.y & keep_.y, ra2.z & keep_.z, ra2.w & keep_.w);     \
  }

  const int wm = wave & 1, wn = wave >> 1;  // partners on a SIMD (w, w + 4) share the row block, not the columns
  const int li = lane & 31, half = lane >> 5, swz = (li >> 2) & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // byte addresses of this lane's fragments: A row wm*64 + t*32 + li, B column wn*64 + t*32 + li, k-step s
  uint32_t aa[2][2][2], ab[2][2][2];  // [buffer][tile][k-step]; a literal offset selects the plane (16-bit field)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kg = (2 * s + half) ^ swz;
      aa[0][t][s] = lds0 + ((wm * 64 + t * 32 + li) * 4 + kg) * 16;
      ab[0][t][s] = lds0 + W_B_OFF_BYTES + ((wn * 64 + t * 32 + li) * 4 + kg) * 16;
      aa[1][t][s] = aa[0][t][s] + W_STAGE_BYTES;
      ab[1][t][s] = ab[0][t][s] + W_STAGE_BYTES;
    }

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc00[r] = acc01[r] = acc10[r] = acc11[r] = 0.f;

  // prologue: stage 0 lands before anything else
  if constexpr (MASKED) {
    W_LOAD_A(0)
  } else {
    W_DMA_A(0, smem)
  }
  W_DMA_B(0, smem, 0)
  W_DMA_B(0, smem, 3)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (MASKED) {
    W_STORE_A(smem)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  const bool late = wave >= 4;
  if (late) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0

  u32x4 A0[2][3], A1[2][3], B0[2][3], B1[2][3];  // [k-step][plane]

#define RD_A(dst_, t_, BUF_)                                  \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {             \
    LDS_RD(dst_[s][0], aa[BUF_][t_][s], 0);                   \
    LDS_RD(dst_[s][1], aa[BUF_][t_][s], W_A_PLANE_BYTES);     \
    LDS_RD(dst_[s][2], aa[BUF_][t_][s], 2 * W_A_PLANE_BYTES); \
  }
#define RD_B(dst_, t_, BUF_)                                  \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {             \
    LDS_RD(dst_[s][0], ab[BUF_][t_][s], 0);                   \
    LDS_RD(dst_[s][1], ab[BUF_][t_][s], W_B_PLANE_BYTES);     \
    LDS_RD(dst_[s][2], ab[BUF_][t_][s], 2 * W_B_PLANE_BYTES); \
  }
#define WAIT_FRAG(F_) LDS_WAIT6(F_[0][0], F_[0][1], F_[0][2], F_[1][0], F_[1][1], F_[1][2])
#define SEG_MFMA(acc_, A_, B_)       \
  __builtin_amdgcn_s_barrier();      \
  WAIT_FRAG(A_);                     \
  WAIT_FRAG(B_);                     \
  __builtin_amdgcn_sched_barrier(0); \
  __builtin_amdgcn_s_setprio(1);     \
  SIX(acc_, A_, B_, 0)               \
  SIX(acc_, A_, B_, 1)               \
  __builtin_amdgcn_s_setprio(0);     \
  __builtin_amdgcn_sched_barrier(0); \
  __builtin_amdgcn_s_barrier();

  // BUF_ = buffer this stage reads (literal 0 / 1); the next stage's pieces go to the other one
#define STAGE(kt_, BUF_)                                                                                   \
  {                                                                                                        \
    uint4* nxt_ = smem + (1 - (BUF_)) * W_STAGE_UINT4;                                                     \
    const bool more_ = (kt_) + 1 < nk;                                                                     \
    /* phase 0: quadrant 00 */                                                                             \
    RD_A(A0, 0, BUF_)                                                                                      \
    RD_B(B0, 0, BUF_)                                                                                      \
    if (more_) {                                                                                           \
      if constexpr (MASKED) W_LOAD_A((kt_) + 1) else W_DMA_A((kt_) + 1, nxt_)                                         \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    SEG_MFMA(acc00, A0, B0)                                                                                \
    /* phase 1: quadrant 01 */                                                                             \
    RD_B(B1, 1, BUF_)                                                                                      \
    if (more_) W_DMA_B((kt_) + 1, nxt_, 0)                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    SEG_MFMA(acc01, A0, B1)                                                                                \
    /* phase 2: quadrant 11 */                                                                             \
    RD_A(A1, 1, BUF_)                                                                                      \
    if (more_) W_DMA_B((kt_) + 1, nxt_, 3)                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    SEG_MFMA(acc11, A1, B1)                                                                                \
    /* phase 3: quadrant 10; the next stage has landed before anybody passes this phase's barriers: a wave's DMA by its   \
       vmcnt(0), the routed form's register-staged A piece by the lgkmcnt(0) behind its stores (the other wave group is one \
       barrier ahead and starts reading the next buffer right after this phase's second barrier) */                         \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
    if constexpr (MASKED) if (more_) {                                                                     \
      W_STORE_A(nxt_)                                                                                      \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    SEG_MFMA(acc10, A1, B0)                                                                                \
  }

  for (int kt = 0; kt < nk; kt += 2) {  // (K is a multiple of 64: bl_gemm_rows_x6w_ok)
    STAGE(kt, 0)
    STAGE(kt + 1, 1)
  }
  if (!late) __builtin_amdgcn_s_barrier();  // both groups have left the loop: the stage buffers are dead

  // epilogue: a wave's 64 x 64 part leaves through LDS (per wave [32 rows][64 + 4] fp32) as whole 256-byte row pieces
  float* stage = reinterpret_cast<float*>(smem) + wave * (32 * 68);
#define W_STORE_HALF(ti_, accA_, accB_)                                                                                  \
  {                                                                                                                      \
    _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                                   \
      *reinterpret_cast<float4*>(stage + li * 68 + 8 * gq + 4 * half) =                                                  \
          make_float4(accA_[4 * gq], accA_[4 * gq + 1], accA_[4 * gq + 2], accA_[4 * gq + 3]);                           \
      *reinterpret_cast<float4*>(stage + li * 68 + 32 + 8 * gq + 4 * half) =                                             \
          make_float4(accB_[4 * gq], accB_[4 * gq + 1], accB_[4 * gq + 2], accB_[4 * gq + 3]);                           \
    }                                                                                                                    \
    /* (a wave reads back only what it wrote itself; its LDS operations execute in order) */                             \
    const int c4_ = lane & 15, n_ = n0 + wn * 64 + 4 * c4_;                                                              \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                      \
      const int r_ = (lane >> 4) + 4 * j;                                                                                \
      const int mm_ = wm * 64 + (ti_) * 32 + r_;                                                                         \
      const float4 v_ = *reinterpret_cast<const float4*>(stage + r_ * 68 + 4 * c4_);                                     \
      if (mm_ < nrows && n_ < N) *reinterpret_cast<float4*>(c + (size_t)(row0 + mm_) * ldc + n_) = v_;                   \
    }                                                                                                                    \
  }
  // (direct float4 stores from the accumulator layout instead of the LDS-staged row pieces: measured slower, profiles/r05f_rows_wide.log)
  W_STORE_HALF(0, acc00, acc01)
  W_STORE_HALF(1, acc10, acc11)
}


// ================================================================================================
extern "C" int bl_pack_weights_x6w(const float* w, int32_t G, int32_t K, int32_t N, int32_t w_is_kn, uint16_t* out, void* stream) {
  if (G == 0) return BL_OK;
  BL_CHECK_ARG(w && out && bl_aligned16(out), "bl_pack_weights_x6w: null or misaligned pointer");
  BL_CHECK_ARG(K > 0 && K % 32 == 0 && N > 0, "bl_pack_weights_x6w: K must be a multiple of 32 (got %d)", K);
  const long long total = (long long)G * ((N + WBN - 1) / WBN) * (K / 32) * (WBN * 4);
  hipLaunchKernelGGL(pack_weights_wide_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, G, K, N,
                     w_is_kn, reinterpret_cast<uint4*>(out));
  BL_LAUNCH_CHECK("bl_pack_weights_x6w");
  return BL_OK;
}

extern "C" int64_t bl_packed_weight_elems_x6w(int32_t G, int32_t K, int32_t N) {
  return (int64_t)G * ((N + WBN - 1) / WBN) * (K / 32) * (W_BLK * 8);
}

// Shapes the wide form takes.  The switch (bl_set_rows_tile) is a measurement aid like bl_set_wgrad_tile.
static std::atomic<bool> g_rows_wide{true};
extern "C" int32_t bl_set_rows_tile(int32_t cols) { return g_rows_wide.exchange(cols != 128) ? 256 : 128; }

// Can the CURRENT device run the wide kernel (two 72 KB stage images of dynamic LDS)?  Cached per device.  Without a visible device
// (the build container's header / export checks) the shape rule alone answers, so that packing and dispatch agree everywhere:
// bl_mp_layer_weight_image, the weight packers and the layer calls all go through bl_gemm_rows_x6w_ok, and a device that cannot
// hold the image falls back to the 128 x 128 kernel's weight image in all three places instead of failing inside the layer call.
static bool x6w_device_ok() {
  static std::atomic<int8_t> cache[BL_MAX_DEVICES];  // 0 unknown, 1 yes, 2 no
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BL_MAX_DEVICES) return true;
  int8_t c = cache[dev].load(std::memory_order_relaxed);
  if (c == 0) {
    c = bl_max_lds_per_block() >= 2 * W_STAGE_BYTES ? 1 : 2;
    cache[dev].store(c, std::memory_order_relaxed);
  }
  return c == 1;
}
// (the kernel itself takes any K >= 64 that is a multiple of 64; below 8 stages per tile its prologue and epilogue cost as much as
// the loop and the 128 x 128 kernel's three workgroups per CU hide them better -- measured equal at K = 256, so the line is there)
extern "C" int32_t bl_gemm_rows_x6w_ok(int32_t N, int32_t K) {
  return g_rows_wide.load(std::memory_order_relaxed) && N > 0 && N % WBN == 0 && K >= 256 && K % 64 == 0 && x6w_device_ok();
}

extern "C" int bl_gemm_rows_x6w(const bl_rows_packed_t* a, const uint32_t* win_bits, int32_t ld_bits, const uint16_t* bp,
                                int64_t b_group_stride, const int32_t* group_ptr, const int32_t* group_w, int32_t G, int32_t M,
                                int32_t N, int32_t K, float* c, int32_t ldc, void* stream) {
  const char* who = "bl_gemm_rows_x6w";
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
  BL_CHECK_ARG(N > 0 && N % WBN == 0 && K >= 64 && K % 64 == 0, "%s: N must be a multiple of 256 and K of 64 (bl_gemm_rows_x6w_ok)", who);
  BL_CHECK_ARG(M > 0 && ldc % 4 == 0 && bp && c && bl_aligned16(bp) && bl_aligned16(c), "%s: ldc a multiple of 4, aligned pointers required", who);
  BL_CHECK_ARG(b_group_stride % 8 == 0 && (G <= 1 || b_group_stride >= bl_packed_weight_elems_x6w(1, K, N)),
               "%s: packed group stride must cover one group's weight image (bl_pack_weights_x6w)", who);
  BL_CHECK_ARG(win_bits == nullptr || (a->nsrc == 1 && a->idx[0] && ld_bits * 32 >= K),
               "%s: the routed form needs exactly one gathered source and ld_bits >= K / 32", who);
  const size_t lds = (size_t)2 * W_STAGE_BYTES;
  static bool attr_plain[BL_MAX_DEVICES] = {false}, attr_routed[BL_MAX_DEVICES] = {false};  // (calls come from one thread per process)
  BL_CHECK_ARG(bl_max_lds_per_block() >= (int)lds, "%s: the device offers %d B of LDS per workgroup, %d needed (bl_gemm_rows_x6w_ok)", who,
               bl_max_lds_per_block(), (int)lds);
  if (bl_raise_lds_limit_once((const void*)gemm_rows_x6w_kernel<false>, (int)lds, attr_plain) != BL_OK ||
      bl_raise_lds_limit_once((const void*)gemm_rows_x6w_kernel<true>, (int)lds, attr_routed) != BL_OK) {
    bl_set_error("%s: cannot raise the dynamic LDS limit to %d B", who, (int)lds);
    return BL_EINVAL;
  }
  dim3 grid((M + WBM - 1) / WBM + (group_ptr ? G : 0), N / WBN);
#define X6W_ARGS                                                                                                              \
  reinterpret_cast<const uint4*>(a->xp[0]), a->nsrc > 1 ? reinterpret_cast<const uint4*>(a->xp[1]) : nullptr,                   \
      a->nsrc > 2 ? reinterpret_cast<const uint4*>(a->xp[2]) : nullptr, a->idx[0], a->nsrc > 1 ? a->idx[1] : nullptr,           \
      a->nsrc > 2 ? a->idx[2] : nullptr, a->width[0], a->nsrc > 1 ? a->width[1] : 0, a->nsrc > 2 ? a->width[2] : 0, koff[1],    \
      koff[2], a->nsrc, win_bits, ld_bits, reinterpret_cast<const uint4*>(bp), (long long)(b_group_stride / 8), group_ptr,      \
      group_w, G, M, N, K, c, ldc
  if (win_bits)
    hipLaunchKernelGGL((gemm_rows_x6w_kernel<true>), grid, dim3(512), lds, (hipStream_t)stream, X6W_ARGS);
  else
    hipLaunchKernelGGL((gemm_rows_x6w_kernel<false>), grid, dim3(512), lds, (hipStream_t)stream, X6W_ARGS);
  BL_LAUNCH_CHECK(who);
  return BL_OK;
}
