// Routed input gradient of a message-passing layer from the NON-ZEROS of the message gradient, on the vector units,
// with the node-gradient sums fused in.
//
// The max aggregation hands the gradient of every (node, channel) to exactly ONE incoming message (torch_scatter's
// scatter_max backward behind ptgnn's MlpMessagePassingLayer; call site /root/reference/buglab/models/gnnlayerdefs.py:6-23),
// so the [E, Dm] message gradient has N * Dm non-zeros -- 20 % at five messages per node.  A 1024-thread workgroup keeps
// W[t]^T ([Dm][2 Din] fp32, <= 128 KB) in LDS and walks a contiguous piece of type t's messages; per message the lanes
// test their own channel's routing bit, `ballot` + `mbcnt` compact the set lanes into a per-wave (weight row, gradient)
// list in LDS, and the multiply loop reads the pairs back wave-uniformly (LDS broadcast) and does one row read + packed
// FMAs per non-zero: 2 * N * Dm * 2Din FLOP in exact fp32 instead of 2 * E * Dm * 2Din on the matrix cores.
//
// Two output forms:
//  * bl_routed_dgrad_nodes (default): nothing per message ever reaches memory.  A message's SOURCE half is added straight
//    into g_h[src(e)]; the TARGET half is kept in registers across the run of messages that share a target (messages are
//    target-sorted inside a type) and added into g_h[tgt] once per run.  fp32 atomics, one 256-byte row segment per
//    instruction; the caller zeroes g_h.  Replaces the [E, 2 Din] per-message gradient (written once, re-read once by
//    bl_mp_scatter_grad: 1.3 GB per H = 128 layer at BASELINE config c2) and that kernel.
//  * bl_routed_dgrad_vec: writes the per-message rows g_a [E, 2 Din] like the routed matrix-core GEMM does (bit-reproducible;
//    used by the deterministic mode in front of bl_mp_scatter_grad).
// LDS image of W[t]^T: row d holds, for lane l, the OPL = 2 Din / 64 columns {l, l + 64, ...} back to back, so that a lane's
// row read is one ds_read_b64 / b128 and every atomic / store instruction of the wave covers 64 consecutive floats.
#include "bl_common.h"

namespace {
constexpr int RD_THREADS = 1024;
constexpr int RD_WAVES = RD_THREADS / 64;
constexpr int RD_GROUP = 4;  // messages whose operands a wave requests together

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct RdPair {  // one non-zero of a message's routed gradient; both members int so that the list is written and read as ints
  int row_off;   // offset of the weight row in the LDS image
  int g_bits;    // the gradient value's bit pattern
};

// which (type, first message, message count) piece workgroup `t` works on
__device__ __forceinline__ bool rd_find_piece(const int* __restrict__ type_ptr, int T, int piece, int t, int& g, int& row0, int& nrows) {
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

struct RdOut {        // where column c of a node's gradient row lives: c < split -> lo, else hi (a folded ConcatResidual)
  float* lo; int ld_lo;
  float* hi; int ld_hi;
  int split;
};

// OPL = output columns per lane (2 Din = 64 * OPL, OPL in {2, 4}); NG = Dm / 64; FUSED: node sums by atomics
template <int OPL, int NG, bool FUSED>
__global__ __launch_bounds__(RD_THREADS, 1) void routed_dgrad_kernel(
    const float* __restrict__ gq, int ld_gq, const int* __restrict__ msg_src, const int* __restrict__ msg_tgt,
    const uint32_t* __restrict__ win_bits, int ld_bits, const int* __restrict__ type_ptr, int T, const float* __restrict__ wt,
    int piece, float* __restrict__ g_a, int ld_ga, RdOut out) {
  constexpr int NOUT = 64 * OPL, Dm = 64 * NG, HALF = OPL / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;                                                      // [Dm][64][OPL]
  RdPair* lists = reinterpret_cast<RdPair*>(smem + Dm * NOUT);           // [RD_WAVES][Dm + 4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t, e0, ne;
  if (!rd_find_piece(type_ptr, T, piece, blockIdx.x, t, e0, ne)) return;
  {  // W[t]^T -> LDS, columns regrouped per lane; global reads are 256-byte row segments
    const float* __restrict__ src = wt + ((size_t)t * Dm) * NOUT;
    for (int i = tid; i < Dm * 64; i += RD_THREADS) {
      const int d = i >> 6, l = i & 63;
      float v[OPL];
#pragma unroll
      for (int k = 0; k < OPL; ++k) v[k] = src[(size_t)d * NOUT + l + 64 * k];
      if (OPL == 4) *reinterpret_cast<float4*>(wl + d * NOUT + 4 * l) = make_float4(v[0], v[1], v[2], v[3]);
      else *reinterpret_cast<float2*>(wl + d * NOUT + 2 * l) = make_float2(v[0], v[1]);
    }
  }
  __syncthreads();
  RdPair* mine = lists + wave * (Dm + 4);  // 16-byte aligned: Dm + 4 pairs of 8 bytes
  const float* __restrict__ wl_lane = wl + lane * OPL;
  // a wave takes a CONTIGUOUS share of the piece, so that runs of equal targets stay inside one wave
  int per = (ne + RD_WAVES - 1) / RD_WAVES;
  const int wb = e0 + wave * per, we = min(wb + per, e0 + ne);

  // per-lane output columns: k < HALF -> source half column lane + 64 k, else target half column lane + 64 (k - HALF)
  float* colp[HALF];
  int cold[HALF];
  if (FUSED) {
#pragma unroll
    for (int k = 0; k < HALF; ++k) {
      const int c = lane + 64 * k;
      const bool lo = c < out.split;
      colp[k] = lo ? out.lo + c : out.hi + (c - out.split);
      cold[k] = lo ? out.ld_lo : out.ld_hi;
    }
  }
  f32x2 acc[OPL / 2];  // [0 .. HALF/2) source half (HALF == 1: .x source, .y target), rest target half
#pragma unroll
  for (int q = 0; q < OPL / 2; ++q) acc[q] = f32x2{0.f, 0.f};
  int run_v = -1;  // target of the run whose target-half sum is being accumulated (FUSED)

  for (int base = wb; base < we; base += RD_GROUP) {
    uint32_t w_g[RD_GROUP][NG];  // the routing word that holds this lane's channel lane + 64 j
    float gq_g[RD_GROUP][NG];
    int v_g[RD_GROUP], s_g[RD_GROUP];
#pragma unroll
    for (int p = 0; p < RD_GROUP; ++p) {
      const int ep = min(base + p, we - 1);
      v_g[p] = msg_tgt[ep];
      s_g[p] = FUSED ? msg_src[ep] : 0;
    }
#pragma unroll
    for (int p = 0; p < RD_GROUP; ++p) {
      const int ep = min(base + p, we - 1);
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        w_g[p][j] = win_bits[(size_t)ep * ld_bits + 2 * j + (lane >> 5)];
        gq_g[p][j] = gq[(size_t)v_g[p] * ld_gq + lane + 64 * j];
      }
    }
#pragma unroll
    for (int p = 0; p < RD_GROUP; ++p) {
      const int e = base + p;
      if (e >= we) break;
      if (FUSED) {
        if (v_g[p] != run_v) {  // wave-uniform
          if (run_v >= 0) {
#pragma unroll
            for (int k = 0; k < HALF; ++k) {
              const float val = HALF == 1 ? acc[0].y : (k & 1 ? acc[(HALF + k) >> 1].y : acc[(HALF + k) >> 1].x);
              unsafeAtomicAdd(colp[k] + (size_t)run_v * cold[k], val);
            }
          }
          run_v = v_g[p];
          if (HALF == 1) acc[0].y = 0.f;
          else {
#pragma unroll
            for (int q = HALF / 2; q < OPL / 2; ++q) acc[q] = f32x2{0.f, 0.f};
          }
        }
        if (HALF == 1) acc[0].x = 0.f;
        else {
#pragma unroll
          for (int q = 0; q < HALF / 2; ++q) acc[q] = f32x2{0.f, 0.f};
        }
      } else {
#pragma unroll
        for (int q = 0; q < OPL / 2; ++q) acc[q] = f32x2{0.f, 0.f};
      }
      // (1) compact this message's non-zeros into the wave's list
      int total = 0;
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        const bool set = (w_g[p][j] >> (lane & 31)) & 1u;
        const unsigned long long m = __ballot(set);
        const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (set) {
          RdPair pr;
          pr.row_off = (lane + 64 * j) * NOUT;
          pr.g_bits = __builtin_bit_cast(int, gq_g[p][j]);
          mine[total + pos] = pr;
        }
        total += __popcll(m);
      }
      if (total == 0) {
        if (!FUSED || g_a != nullptr) {
          float* __restrict__ o = g_a + (size_t)e * ld_ga + lane;
#pragma unroll
          for (int k = 0; k < (FUSED ? HALF : OPL); ++k) o[64 * k] = 0.f;
        }
        continue;
      }
      // three zero-weight entries behind the list: the multiply loop reads whole groups of four without index clamps
      if (lane < 3) {
        RdPair z;
        z.row_off = 0;
        z.g_bits = 0;
        mine[total + lane] = z;
      }
      // the list is written by some lanes and read back by all: same wave, so program order + this fence is the ordering
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // (2) multiply: four non-zeros per trip; the pair reads are wave-uniform (LDS broadcast), two pairs per 16-byte read
#pragma unroll 2  // (4 runs into the 128-register cap of a 16-wave workgroup and spills)
      for (int i = 0; i < total; i += 4) {
        struct { int row_off; float g; } pr[4];
        const int4 lo = *reinterpret_cast<const int4*>(mine + i), hi = *reinterpret_cast<const int4*>(mine + i + 2);
        pr[0].row_off = lo.x; pr[0].g = __builtin_bit_cast(float, lo.y);
        pr[1].row_off = lo.z; pr[1].g = __builtin_bit_cast(float, lo.w);
        pr[2].row_off = hi.x; pr[2].g = __builtin_bit_cast(float, hi.y);
        pr[3].row_off = hi.z; pr[3].g = __builtin_bit_cast(float, hi.w);
        // explicit packed FMAs: left to itself hipcc turns half of these into v_mul + v_pk_add + a dozen v_mov per trip
        if (OPL == 4) {
          f32x4 r[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q] = *reinterpret_cast<const f32x4*>(wl_lane + pr[q].row_off);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 g2 = {pr[q].g, pr[q].g};
            acc[0] = __builtin_elementwise_fma(g2, __builtin_shufflevector(r[q], r[q], 0, 1), acc[0]);
            acc[1] = __builtin_elementwise_fma(g2, __builtin_shufflevector(r[q], r[q], 2, 3), acc[1]);
          }
        } else {
          f32x2 r[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q] = *reinterpret_cast<const f32x2*>(wl_lane + pr[q].row_off);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 g2 = {pr[q].g, pr[q].g};
            acc[0] = __builtin_elementwise_fma(g2, r[q], acc[0]);
          }
        }
      }
      if (FUSED && g_a != nullptr) {  // source half as a plain row of g_a [E, Din]: summed per node by bl_mp_scatter_grad afterwards
        float* __restrict__ o = g_a + (size_t)e * ld_ga + lane;
#pragma unroll
        for (int k = 0; k < HALF; ++k) o[64 * k] = HALF == 1 ? acc[0].x : (k & 1 ? acc[k >> 1].y : acc[k >> 1].x);
      } else if (FUSED) {  // source half of this message -> its source node
#pragma unroll
        for (int k = 0; k < HALF; ++k) {
          const float val = HALF == 1 ? acc[0].x : (k & 1 ? acc[k >> 1].y : acc[k >> 1].x);
          unsafeAtomicAdd(colp[k] + (size_t)s_g[p] * cold[k], val);
        }
      } else {
        float* __restrict__ o = g_a + (size_t)e * ld_ga + lane;
#pragma unroll
        for (int k = 0; k < OPL; ++k) o[64 * k] = k & 1 ? acc[k >> 1].y : acc[k >> 1].x;
      }
    }
  }
  if (FUSED && run_v >= 0) {
#pragma unroll
    for (int k = 0; k < HALF; ++k) {
      const float val = HALF == 1 ? acc[0].y : (k & 1 ? acc[(HALF + k) >> 1].y : acc[(HALF + k) >> 1].x);
      unsafeAtomicAdd(colp[k] + (size_t)run_v * cold[k], val);
    }
  }
}

template <int OPL, int NG, bool FUSED>
int rd_launch(const float* gq, int ld_gq, const int* msg_src, const int* msg_tgt, const uint32_t* win_bits, int ld_bits,
              const int* type_ptr, int T, const float* wt, int E, float* g_a, int ld_ga, RdOut out, hipStream_t st) {
  constexpr int NOUT = 64 * OPL, Dm = 64 * NG;
  const size_t lds = (size_t)Dm * NOUT * sizeof(float) + (size_t)RD_WAVES * (Dm + 4) * sizeof(RdPair);
  {  // per device, not per process: set on every launch (a host-side table write)
    hipError_t e = hipFuncSetAttribute((const void*)routed_dgrad_kernel<OPL, NG, FUSED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  // one workgroup per CU is resident (LDS): two rounds of pieces; every type adds at most one short piece
  const int ncu = bl_num_cus();
  const int slots = 2 * ncu > 2 * T ? 2 * ncu - T : ncu;
  int piece = (int)(((long long)E + slots - 1) / slots);
  piece = ((piece < 256 ? 256 : piece) + RD_WAVES * RD_GROUP - 1) / (RD_WAVES * RD_GROUP) * (RD_WAVES * RD_GROUP);
  dim3 grid((E + piece - 1) / piece + T);
  hipLaunchKernelGGL((routed_dgrad_kernel<OPL, NG, FUSED>), grid, dim3(RD_THREADS), lds, st, gq, ld_gq, msg_src, msg_tgt, win_bits,
                     ld_bits, type_ptr, T, wt, piece, g_a, ld_ga, out);
  return (int)hipGetLastError();
}

template <bool FUSED>
int rd_dispatch(const float* gq, int ld_gq, const int* msg_src, const int* msg_tgt, const uint32_t* win_bits, int ld_bits,
                const int* type_ptr, int T, const float* wt, int E, int Dm, int K2, float* g_a, int ld_ga, RdOut out, hipStream_t st) {
  int rc = -1;
#define RD_CASE(DM_, K2_, OPL_, NG_) \
  if (Dm == DM_ && K2 == K2_) rc = rd_launch<OPL_, NG_, FUSED>(gq, ld_gq, msg_src, msg_tgt, win_bits, ld_bits, type_ptr, T, wt, E, g_a, ld_ga, out, st);
  RD_CASE(128, 256, 4, 2)
  RD_CASE(128, 128, 2, 2)
  RD_CASE(64, 256, 4, 1)
  RD_CASE(64, 128, 2, 1)
#undef RD_CASE
  return rc;
}
}  // namespace

// 1 if the vector form handles (Dm, K2 = 2 Din): W[t]^T must fit the 128 KB LDS block in one piece
extern "C" int32_t bl_routed_dgrad_vec_ok(int32_t Dm, int32_t K2) {
  if (!((Dm == 64 || Dm == 128) && (K2 == 128 || K2 == 256) && Dm * K2 <= 32768)) return 0;
  // W[t]^T (Dm x K2 floats) + the per-wave non-zero lists must fit what this device lets one workgroup declare
  const size_t lds = (size_t)Dm * K2 * sizeof(float) + (size_t)RD_WAVES * (Dm + 4) * sizeof(RdPair);
  return (size_t)bl_max_lds_per_block() >= lds;
}

extern "C" int bl_routed_dgrad_vec(const float* gq, int32_t ld_gq, const int32_t* msg_tgt, const uint32_t* win_bits, int32_t ld_bits,
                                   const int32_t* type_ptr, int32_t T, const float* wt, int32_t E, int32_t Dm, int32_t K2, float* g_a,
                                   int32_t ld_ga, void* stream) {
  if (E == 0) return BL_OK;
  BL_CHECK_ARG(gq && msg_tgt && win_bits && type_ptr && wt && g_a, "bl_routed_dgrad_vec: null pointer");
  BL_CHECK_ARG(bl_routed_dgrad_vec_ok(Dm, K2), "bl_routed_dgrad_vec: unsupported shape Dm=%d K2=%d", Dm, K2);
  BL_CHECK_ARG(ld_bits * 32 >= Dm && ld_ga >= K2 && ld_gq >= Dm, "bl_routed_dgrad_vec: ld_bits / ld_ga / ld_gq");
  RdOut none = {nullptr, 0, nullptr, 0, 0};
  const int rc = rd_dispatch<false>(gq, ld_gq, nullptr, msg_tgt, win_bits, ld_bits, type_ptr, T, wt, E, Dm, K2, g_a, ld_ga, none, (hipStream_t)stream);
  if (rc != 0) {
    bl_set_error("bl_routed_dgrad_vec: launch failed (%d)", rc);
    return rc;
  }
  return BL_OK;
}

extern "C" int bl_routed_dgrad_nodes(const float* gq, int32_t ld_gq, const int32_t* msg_src, const int32_t* msg_tgt,
                                     const uint32_t* win_bits, int32_t ld_bits, const int32_t* type_ptr, int32_t T, const float* wt,
                                     int32_t E, int32_t Dm, int32_t Din, int32_t split, float* g_h_lo, int32_t ld_lo, float* g_h_hi,
                                     int32_t ld_hi, void* stream) {
  return bl_routed_dgrad_nodes_rows(gq, ld_gq, msg_src, msg_tgt, win_bits, ld_bits, type_ptr, T, wt, E, Dm, Din, split, g_h_lo, ld_lo,
                                    g_h_hi, ld_hi, nullptr, 0, stream);
}

extern "C" int bl_routed_dgrad_nodes_rows(const float* gq, int32_t ld_gq, const int32_t* msg_src, const int32_t* msg_tgt,
                                          const uint32_t* win_bits, int32_t ld_bits, const int32_t* type_ptr, int32_t T,
                                          const float* wt, int32_t E, int32_t Dm, int32_t Din, int32_t split, float* g_h_lo,
                                          int32_t ld_lo, float* g_h_hi, int32_t ld_hi, float* g_src, int32_t ld_src, void* stream) {
  if (E == 0) return BL_OK;
  BL_CHECK_ARG(g_src == nullptr || ld_src >= Din, "bl_routed_dgrad_nodes_rows: ld_src");
  BL_CHECK_ARG(gq && msg_src && msg_tgt && win_bits && type_ptr && wt && g_h_lo, "bl_routed_dgrad_nodes: null pointer");
  BL_CHECK_ARG(bl_routed_dgrad_vec_ok(Dm, 2 * Din), "bl_routed_dgrad_nodes: unsupported shape Dm=%d Din=%d", Dm, Din);
  BL_CHECK_ARG(ld_bits * 32 >= Dm && ld_gq >= Dm, "bl_routed_dgrad_nodes: ld_bits / ld_gq");
  BL_CHECK_ARG((split == Din && ld_lo >= Din) || (split > 0 && split < Din && g_h_hi && ld_lo >= split && ld_hi >= Din - split),
               "bl_routed_dgrad_nodes: split must be Din (one output) or inside (0, Din) with a second output");
  RdOut out = {g_h_lo, ld_lo, g_h_hi, ld_hi, split};
  const int rc = rd_dispatch<true>(gq, ld_gq, msg_src, msg_tgt, win_bits, ld_bits, type_ptr, T, wt, E, Dm, 2 * Din, g_src, ld_src, out, (hipStream_t)stream);
  if (rc != 0) {
    bl_set_error("bl_routed_dgrad_nodes: launch failed (%d)", rc);
    return rc;
  }
  return BL_OK;
}
