// Routed input gradient of a message-passing layer from the NON-ZEROS of the message gradient, on the vector units.
// Library form of tools/experiments/bl_routed_dgrad_vec2.hip (see there and tools/experiments/README.md for the
// measurements: 0.304 ms against 0.392 ms for the routed bf16x6 GEMM at the c2 H=128 layer shape) with the entry points
// bl_routed_dgrad_vec_integration.patch expects.  To integrate: copy to csrc/bl_routed_dgrad.hip, add to the Makefile's
// SRCS, apply the patch's other hunks.
#include "bl_common.h"

namespace {
constexpr int V2_THREADS = 1024;
constexpr int V2_WAVES = V2_THREADS / 64;
constexpr int V2_GROUP = 4;  // messages whose operands a wave requests together

__device__ __forceinline__ bool v2_find_piece(const int* __restrict__ type_ptr, int T, int piece, int t, int& g, int& row0, int& nrows) {
  const int lane = threadIdx.x & 63;
  int base = 0;
  for (int g0 = 0; g0 < T; g0 += 64) {
    const int gi = g0 + lane;
    const int lo = gi < T ? type_ptr[gi] : 0;
    const int hi = gi < T ? type_ptr[gi + 1] : 0;
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

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Pair {
  int row_off;  // channel * NOUT (floats): offset of the channel's weight row in the LDS block
  float g;
};

// OPL = output columns per lane (NOUT = 64 * OPL); NG = Dm / 64
template <int OPL, int NG>
__global__ __launch_bounds__(V2_THREADS, 1) void routed_dgrad_vec2_kernel(
    const float* __restrict__ gq, int ld_gq, const int* __restrict__ msg_tgt, const uint32_t* __restrict__ win_bits, int ld_bits,
    const int* __restrict__ type_ptr, int T, const float* __restrict__ wt, int K2, int piece, float* __restrict__ g_a, int ld_ga) {
  constexpr int NOUT = 64 * OPL, Dm = 64 * NG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;                                                    // [Dm][NOUT]
  Pair* lists = reinterpret_cast<Pair*>(smem + Dm * NOUT);             // [V2_WAVES][Dm + 4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t, e0, ne;
  if (!v2_find_piece(type_ptr, T, piece, blockIdx.x, t, e0, ne)) return;
  {
    const float* __restrict__ src = wt + ((size_t)t * Dm) * K2;
    constexpr int V4_PER_ROW = NOUT / 4;
    for (int i = tid; i < Dm * V4_PER_ROW; i += V2_THREADS) {
      const int d = i / V4_PER_ROW, q = i - d * V4_PER_ROW;
      *reinterpret_cast<float4*>(wl + d * NOUT + 4 * q) = *reinterpret_cast<const float4*>(src + (size_t)d * K2 + 4 * q);
    }
  }
  __syncthreads();
  Pair* mine = lists + wave * (Dm + 4);  // 16-byte aligned: Dm + 4 pairs of 8 bytes
  const float* __restrict__ wl_lane = wl + lane * OPL;
  const int e1 = e0 + ne;
  for (int base = e0 + wave * V2_GROUP; base < e1; base += V2_WAVES * V2_GROUP) {
    uint32_t w_g[V2_GROUP][NG];  // the routing word that holds this lane's channel lane + 64 j
    float gq_g[V2_GROUP][NG];
    int v_g[V2_GROUP];
#pragma unroll
    for (int p = 0; p < V2_GROUP; ++p) v_g[p] = msg_tgt[min(base + p, e1 - 1)];
#pragma unroll
    for (int p = 0; p < V2_GROUP; ++p) {
      const int ep = min(base + p, e1 - 1);
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        w_g[p][j] = win_bits[(size_t)ep * ld_bits + 2 * j + (lane >> 5)];
        gq_g[p][j] = gq[(size_t)v_g[p] * ld_gq + lane + 64 * j];
      }
    }
#pragma unroll
    for (int p = 0; p < V2_GROUP; ++p) {
      const int e = base + p;
      if (e >= e1) break;
      // (1) compact this message's non-zeros into the wave's list
      int total = 0;
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        const bool set = (w_g[p][j] >> (lane & 31)) & 1u;
        const unsigned long long m = __ballot(set);
        const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (set) {
          Pair pr;
          pr.row_off = (lane + 64 * j) * NOUT;
          pr.g = gq_g[p][j];
          mine[total + pos] = pr;
        }
        total += __popcll(m);
      }
      // three zero-weight entries behind the list: the multiply loop reads whole groups of four without index clamps
      if (lane < 3) {
        Pair z;
        z.row_off = 0;
        z.g = 0.f;
        mine[total + lane] = z;
      }
      // (2) multiply: four non-zeros per trip; the pair reads are wave-uniform (LDS broadcast), two pairs per 16-byte read
      f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll 2  // (the compiler's own choice of 4 runs into the 128-register cap of a 16-wave workgroup and spills)
      for (int i = 0; i < total; i += 4) {
        Pair pr[4];
        const int4 lo = *reinterpret_cast<const int4*>(mine + i), hi = *reinterpret_cast<const int4*>(mine + i + 2);
        pr[0].row_off = lo.x; pr[0].g = __builtin_bit_cast(float, lo.y);
        pr[1].row_off = lo.z; pr[1].g = __builtin_bit_cast(float, lo.w);
        pr[2].row_off = hi.x; pr[2].g = __builtin_bit_cast(float, hi.y);
        pr[3].row_off = hi.z; pr[3].g = __builtin_bit_cast(float, hi.w);
        // explicit packed FMAs: left to itself hipcc turns half of these into separate v_mul + v_pk_add with a dozen v_mov
        // shuffles per trip (8 VALU instructions per non-zero instead of 3)
        if (OPL == 4) {
          f32x4 r[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q] = *reinterpret_cast<const f32x4*>(wl_lane + pr[q].row_off);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 g2 = {pr[q].g, pr[q].g};
            a01 = __builtin_elementwise_fma(g2, __builtin_shufflevector(r[q], r[q], 0, 1), a01);
            a23 = __builtin_elementwise_fma(g2, __builtin_shufflevector(r[q], r[q], 2, 3), a23);
          }
        } else if (OPL == 2) {
          f32x2 r[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q] = *reinterpret_cast<const f32x2*>(wl_lane + pr[q].row_off);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 g2 = {pr[q].g, pr[q].g};
            a01 = __builtin_elementwise_fma(g2, r[q], a01);
          }
        } else {
          float r[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q] = wl_lane[pr[q].row_off];
#pragma unroll
          for (int q = 0; q < 4; ++q) a01.x = __builtin_fmaf(pr[q].g, r[q], a01.x);
        }
      }
      float* __restrict__ out = g_a + (size_t)e * ld_ga + lane * OPL;
      if (OPL == 4) *reinterpret_cast<float4*>(out) = make_float4(a01.x, a01.y, a23.x, a23.y);
      else if (OPL == 2) *reinterpret_cast<float2*>(out) = make_float2(a01.x, a01.y);
      else out[0] = a01.x;
    }
  }
}

template <int OPL, int NG>
int v2_launch(const float* gq, int ld_gq, const int* msg_tgt, const uint32_t* win_bits, int ld_bits, const int* type_ptr, int T,
              const float* wt, int E, int K2, float* g_a, int ld_ga, hipStream_t st) {
  constexpr int NOUT = 64 * OPL, Dm = 64 * NG;
  const size_t lds = (size_t)Dm * NOUT * sizeof(float) + (size_t)V2_WAVES * (Dm + 4) * sizeof(Pair);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)routed_dgrad_vec2_kernel<OPL, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  int dev = 0, ncu = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
  int piece = (int)(((long long)E + 2LL * ncu - 1) / (2LL * ncu));
  piece = ((piece < 512 ? 512 : piece) + V2_WAVES * V2_GROUP - 1) / (V2_WAVES * V2_GROUP) * (V2_WAVES * V2_GROUP);
  dim3 grid((E + piece - 1) / piece + T);
  hipLaunchKernelGGL((routed_dgrad_vec2_kernel<OPL, NG>), grid, dim3(V2_THREADS), lds, st, gq, ld_gq, msg_tgt, win_bits, ld_bits, type_ptr, T, wt, K2,
                     piece, g_a, ld_ga);
  return (int)hipGetLastError();
}
}  // namespace

// 1 if the vector form handles (Dm, K2): W[t]^T must fit the 128 KB LDS block in one piece
extern "C" int32_t bl_routed_dgrad_vec_ok(int32_t Dm, int32_t K2) {
  return (Dm == 64 || Dm == 128) && (K2 == 64 || K2 == 128 || K2 == 256) && Dm * K2 <= 32768;
}

// gq [*, Dm] fp32 node gradients (at the winners' pre-activations), wt [T][Dm][K2] = the layer's per-type weights transposed,
// win_bits the routing bitmask of bl_segment_max_fwd, messages type-major (type_ptr [T+1]).  Writes every row of g_a [E, K2].
extern "C" int bl_routed_dgrad_vec(const float* gq, int32_t ld_gq, const int32_t* msg_tgt, const uint32_t* win_bits, int32_t ld_bits,
                                   const int32_t* type_ptr, int32_t T, const float* wt, int32_t E, int32_t Dm, int32_t K2, float* g_a,
                                   int32_t ld_ga, void* stream) {
  if (E == 0) return BL_OK;
  BL_CHECK_ARG(gq && msg_tgt && win_bits && type_ptr && wt && g_a, "bl_routed_dgrad_vec: null pointer");
  BL_CHECK_ARG(bl_routed_dgrad_vec_ok(Dm, K2), "bl_routed_dgrad_vec: unsupported shape Dm=%d K2=%d", Dm, K2);
  BL_CHECK_ARG(ld_bits * 32 >= Dm && ld_ga == K2 && bl_aligned16(wt) && bl_aligned16(g_a), "bl_routed_dgrad_vec: ld_bits / ld_ga / alignment");
  hipStream_t st = (hipStream_t)stream;
  int rc = -1;
#define V2_CASE(DM_, K2_, OPL_, NG_) \
  if (Dm == DM_ && K2 == K2_) rc = v2_launch<OPL_, NG_>(gq, ld_gq, msg_tgt, win_bits, ld_bits, type_ptr, T, wt, E, K2, g_a, ld_ga, st);
  V2_CASE(128, 256, 4, 2)
  V2_CASE(128, 128, 2, 2)
  V2_CASE(128, 64, 1, 2)
  V2_CASE(64, 256, 4, 1)
  V2_CASE(64, 128, 2, 1)
  V2_CASE(64, 64, 1, 1)
#undef V2_CASE
  if (rc != 0) {
    bl_set_error("bl_routed_dgrad_vec: launch failed (%d)", rc);
    return rc;
  }
  return BL_OK;
}
