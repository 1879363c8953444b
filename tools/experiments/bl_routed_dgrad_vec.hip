// Routed input gradient of a message-passing layer on the VECTOR units.
//
//   g_a[e, k] = sum_d  g_m[e, d] * W[t(e)][k, d],      g_m[e, d] = gq[tgt(e), d] if message e won channel d at its target, else 0
//
// The max aggregation routes every (node, channel) gradient to exactly ONE message, so g_m has N*Dm non-zeros among
// E*Dm entries (20 % at the usual ~5 messages per node).  The matrix-core form (bl_gemm_rows_x6 with win_bits) multiplies
// the zeros too: 2*E*Dm*K2 FLOP as six bf16 MFMA terms.  Here only the non-zeros are touched, in exact fp32:
// 2*N*Dm*K2 FLOP, i.e. 5x fewer at the headline shapes, with W[t]^T resident in LDS.
//
// One 1024-thread workgroup = one chunk of one edge type's messages x one block of NOUT output columns.  It loads
// Wt[t][:, block] ([Dm][NOUT] fp32, 128 KB) into LDS once; then each of its 16 waves walks messages: the routing words
// and the target's gq row sit in registers (eight messages' worth requested together), the set bits are peeled off with scalar
// s_ff1 / v_readlane, and every set bit costs one conflict-free LDS row read (ds_read_b128 per lane) and OPL fused
// multiply-adds per lane (four bits per trip, so that four row reads are in flight).  Bound: LDS bandwidth (one
// 4-clock row read per non-zero), ~26 non-zeros per message at Dm = 128.
#include "bl_common.h"

namespace {
constexpr int RD_THREADS = 1024;
constexpr int RD_WAVES = RD_THREADS / 64;
constexpr int RD_LDS_FLOATS = 32768;  // 128 KB of the CU's 160 KB
constexpr int RD_GROUP = 8;           // messages whose operands a wave has in flight together

__device__ __forceinline__ bool rd_find_piece(const int* __restrict__ type_ptr, int T, int piece, int t, int& g, int& row0, int& nrows) {
  // wave-cooperative: pieces are numbered type by type (same scheme as the GEMMs' find_piece)
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

// OPL = output columns per lane (NOUT = 64 * OPL); NG = Dm / 64 channel registers per lane
template <int OPL, int NG>
__global__ __launch_bounds__(RD_THREADS, 1) void routed_dgrad_vec_kernel(
    const float* __restrict__ gq, int ld_gq, const int* __restrict__ msg_tgt, const uint32_t* __restrict__ win_bits, int ld_bits,
    const int* __restrict__ type_ptr, int T, const float* __restrict__ wt, int K2, int piece, float* __restrict__ g_a, int ld_ga) {
  constexpr int NOUT = 64 * OPL, Dm = 64 * NG, WPR = Dm / 32;
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [Dm][NOUT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t, e0, ne;
  if (!rd_find_piece(type_ptr, T, piece, blockIdx.x, t, e0, ne)) return;
  const int c0 = blockIdx.y * NOUT;
  {  // W[t]^T block -> LDS: rows of NOUT floats, float4 per thread
    const float* __restrict__ src = wt + ((size_t)t * Dm) * K2 + c0;
    constexpr int V4_PER_ROW = NOUT / 4;
    for (int i = tid; i < Dm * V4_PER_ROW; i += RD_THREADS) {
      const int d = i / V4_PER_ROW, q = i - d * V4_PER_ROW;
      *reinterpret_cast<float4*>(wl + d * NOUT + 4 * q) = *reinterpret_cast<const float4*>(src + (size_t)d * K2 + 4 * q);
    }
  }
  __syncthreads();

  const int e1 = e0 + ne;
  // A wave takes RD_GROUP consecutive messages at a time: all their operands (target id -> gq row, routing words) are
  // requested before the first one is computed, so RD_GROUP dependent two-step fetches overlap instead of one
  // (measured with one message ahead: the wave sat out a full memory round trip per message).
  for (int base = e0 + wave * RD_GROUP; base < e1; base += RD_WAVES * RD_GROUP) {
    uint32_t bits_g[RD_GROUP];
    float gq_g[RD_GROUP][NG];
    int v_g[RD_GROUP];
#pragma unroll
    for (int p = 0; p < RD_GROUP; ++p) v_g[p] = msg_tgt[min(base + p, e1 - 1)];
#pragma unroll
    for (int p = 0; p < RD_GROUP; ++p) {
      const int ep = min(base + p, e1 - 1);
      bits_g[p] = lane < WPR ? win_bits[(size_t)ep * ld_bits + lane] : 0u;
#pragma unroll
      for (int j = 0; j < NG; ++j) gq_g[p][j] = gq[(size_t)v_g[p] * ld_gq + lane + 64 * j];
    }
#pragma unroll
    for (int p = 0; p < RD_GROUP; ++p) {
      const int e = base + p;
      if (e >= e1) break;
      const uint32_t bits = bits_g[p];
      float acc[OPL];
#pragma unroll
      for (int u = 0; u < OPL; ++u) acc[u] = 0.f;
#pragma unroll
      for (int j = 0; j < NG; ++j) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)bits, 2 * j + half);  // wave-uniform
          const float* __restrict__ rows = wl + (64 * j + 32 * half) * NOUT + lane * OPL;
          const int gq_bits = __builtin_bit_cast(int, gq_g[p][j]);
          // four set bits per trip: their LDS row reads are independent and in flight together.  Missing bits of the last
          // trip repeat bit 0 with weight 0.
          while (word) {
            int b[4];
            float g[4];
            b[0] = __builtin_ctz(word);
            g[0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(gq_bits, 32 * half + b[0]));
            word &= word - 1u;
#pragma unroll
            for (int q = 1; q < 4; ++q) {
              const bool has = word != 0u;
              b[q] = has ? __builtin_ctz(word) : b[0];
              g[q] = has ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(gq_bits, 32 * half + b[q])) : 0.f;
              word &= word - 1u;  // 0 stays 0
            }
            if (OPL == 4) {
              float4 r[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) r[q] = *reinterpret_cast<const float4*>(rows + b[q] * NOUT);
#pragma unroll
              for (int q = 0; q < 4; ++q) { acc[0] += g[q] * r[q].x; acc[1] += g[q] * r[q].y; acc[2] += g[q] * r[q].z; acc[3] += g[q] * r[q].w; }
            } else if (OPL == 2) {
              float2 r[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) r[q] = *reinterpret_cast<const float2*>(rows + b[q] * NOUT);
#pragma unroll
              for (int q = 0; q < 4; ++q) { acc[0] += g[q] * r[q].x; acc[1] += g[q] * r[q].y; }
            } else {
              float r[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) r[q] = rows[b[q] * NOUT];
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[0] += g[q] * r[q];
            }
          }
        }
      }
      float* __restrict__ out = g_a + (size_t)e * ld_ga + c0 + lane * OPL;
      if (OPL == 4) *reinterpret_cast<float4*>(out) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      else if (OPL == 2) *reinterpret_cast<float2*>(out) = make_float2(acc[0], acc[1]);
      else out[0] = acc[0];
    }
  }
}

template <int OPL, int NG>
int rd_launch(const float* gq, int ld_gq, const int* msg_tgt, const uint32_t* win_bits, int ld_bits, const int* type_ptr, int T,
              const float* wt, int E, int K2, float* g_a, int ld_ga, hipStream_t st) {
  constexpr int NOUT = 64 * OPL, Dm = 64 * NG;
  const size_t lds = (size_t)Dm * NOUT * sizeof(float);
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)routed_dgrad_vec_kernel<OPL, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      bl_set_error("bl_routed_dgrad_vec: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
    attr = true;
  }
  // chunk of messages per workgroup: ~2 rounds of workgroups over the chip's CUs, at least 512 messages (the 128 KB weight
  // block is loaded once per workgroup)
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const int passes = K2 / NOUT;
  int piece = (int)(((long long)E * passes + 2LL * ncu - 1) / (2LL * ncu));
  piece = ((piece < 512 ? 512 : piece) + RD_WAVES * RD_GROUP - 1) / (RD_WAVES * RD_GROUP) * (RD_WAVES * RD_GROUP);
  dim3 grid((E + piece - 1) / piece + T, passes);
  hipLaunchKernelGGL((routed_dgrad_vec_kernel<OPL, NG>), grid, dim3(RD_THREADS), lds, st, gq, ld_gq, msg_tgt, win_bits, ld_bits, type_ptr, T, wt,
                     K2, piece, g_a, ld_ga);
  BL_LAUNCH_CHECK("bl_routed_dgrad_vec");
  return BL_OK;
}
}  // namespace

// 1 if (Dm, K2) is a shape the vector form handles: Dm in {64, 128, 256, 512} channels and K2 a multiple of the column block
extern "C" int32_t bl_routed_dgrad_vec_ok(int32_t Dm, int32_t K2) {
  if (Dm != 64 && Dm != 128 && Dm != 256 && Dm != 512) return 0;
  const int nout = RD_LDS_FLOATS / Dm > 256 ? 256 : RD_LDS_FLOATS / Dm;  // 256 / 256 / 128 / 64
  return K2 > 0 && K2 % nout == 0;
}

// gq [*, Dm] fp32 node gradients (at the winners' pre-activations), wt [T][Dm][K2] = the layer's per-type weights transposed,
// win_bits the routing bitmask of bl_segment_max_fwd, messages type-major (type_ptr [T+1]).  Writes every row of g_a [E, K2].
extern "C" int bl_routed_dgrad_vec(const float* gq, int32_t ld_gq, const int32_t* msg_tgt, const uint32_t* win_bits, int32_t ld_bits,
                                   const int32_t* type_ptr, int32_t T, const float* wt, int32_t E, int32_t Dm, int32_t K2, float* g_a,
                                   int32_t ld_ga, void* stream) {
  if (E == 0) return BL_OK;
  BL_CHECK_ARG(gq && msg_tgt && win_bits && type_ptr && wt && g_a, "bl_routed_dgrad_vec: null pointer");
  BL_CHECK_ARG(bl_routed_dgrad_vec_ok(Dm, K2), "bl_routed_dgrad_vec: unsupported shape Dm=%d K2=%d", Dm, K2);
  BL_CHECK_ARG(ld_bits * 32 >= Dm && ld_ga % 4 == 0 && bl_aligned16(wt) && bl_aligned16(g_a), "bl_routed_dgrad_vec: ld_bits / alignment");
  hipStream_t st = (hipStream_t)stream;
  switch (Dm) {
    case 64: return rd_launch<4, 1>(gq, ld_gq, msg_tgt, win_bits, ld_bits, type_ptr, T, wt, E, K2, g_a, ld_ga, st);
    case 128: return rd_launch<4, 2>(gq, ld_gq, msg_tgt, win_bits, ld_bits, type_ptr, T, wt, E, K2, g_a, ld_ga, st);
    case 256: return rd_launch<2, 4>(gq, ld_gq, msg_tgt, win_bits, ld_bits, type_ptr, T, wt, E, K2, g_a, ld_ga, st);
    default: return rd_launch<1, 8>(gq, ld_gq, msg_tgt, win_bits, ld_bits, type_ptr, T, wt, E, K2, g_a, ld_ga, st);
  }
}
