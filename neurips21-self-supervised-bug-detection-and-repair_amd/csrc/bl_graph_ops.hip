// HBM-bound graph kernels of the message-passing layer: subtoken embedder, segmented max (+argmax,
// +fused LayerNorm), its gather-form backward, LayerNorm backward, activation/dropout backward and
// the deterministic segmented sums that turn per-message input gradients into node gradients.
//
// Common shape: ONE WAVE (64 lanes) PER SEGMENT / ROW.  A row of D <= 512 floats is spread over the
// lanes as d = lane + 64 * j (j < NV): every load instruction of the wave covers 256 contiguous
// bytes, row statistics are wave reductions (no LDS, no barriers), and the CSR item ids of a
// segment are fetched 64 at a time with one coalesced load and handed out with v_readlane.
#include "bl_common.h"

#define NEG_INF (-__builtin_huge_valf())
// strict '>' keeps the first of equal values (torch_scatter's tie rule); a NaN message wins once and then
// sticks (nothing compares greater than it, and the second clause needs a non-NaN incumbent), so a
// diverged run shows up in the aggregate like it does with torch's amax instead of being dropped
#define BL_MAX_WINS(t, best) ((t) > (best) || ((t) != (t) && (best) == (best)))

// ------------------------------------------------------------------------------------------------
// M0 embedder
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ table, int H,
                                                        const int* __restrict__ ids, const int* __restrict__ lens,
                                                        int N, int S, bl_drop_dev drop, float* __restrict__ out,
                                                        int ld_out, int8_t* __restrict__ argsub, int drop_before_pool, int comb) {
  const int h4n = H >> 2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * h4n) return;
  const int n = (int)(t / h4n), h = (int)(t % h4n) * 4;
  int len = lens[n];
  len = len < 1 ? 1 : (len > S ? S : len);
  // drop_before_pool: dropout on the EMBEDDED SUBTOKENS [N, S, H] (mask index (n S + s) H + h) and the max over what is left
  // (a dropped element is a 0 that can win); default: dropout on the pooled [N, H] rows (the frozen spec)
  const bool pre = drop_before_pool && drop.thresh;
  float4 best = *reinterpret_cast<const float4*>(table + (size_t)ids[(size_t)n * S] * H + h);
  if (pre) {
    const uint32_t i = (uint32_t)n * (uint32_t)S * (uint32_t)H + (uint32_t)h;
    best.x = bl_keep(drop, i) ? best.x * drop.scale : 0.f;
    best.y = bl_keep(drop, i + 1) ? best.y * drop.scale : 0.f;
    best.z = bl_keep(drop, i + 2) ? best.z * drop.scale : 0.f;
    best.w = bl_keep(drop, i + 3) ? best.w * drop.scale : 0.f;
  }
  int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int s = 1; s < len; ++s) {
    float4 v = *reinterpret_cast<const float4*>(table + (size_t)ids[(size_t)n * S + s] * H + h);
    if (pre) {
      const uint32_t i = ((uint32_t)n * (uint32_t)S + (uint32_t)s) * (uint32_t)H + (uint32_t)h;
      v.x = bl_keep(drop, i) ? v.x * drop.scale : 0.f;
      v.y = bl_keep(drop, i + 1) ? v.y * drop.scale : 0.f;
      v.z = bl_keep(drop, i + 2) ? v.z * drop.scale : 0.f;
      v.w = bl_keep(drop, i + 3) ? v.w * drop.scale : 0.f;
    }
    if (comb != 0) {  // sum / mean over the subtokens (subtoken_combination of the node model, modelregistry.py:65-66)
      best.x += v.x; best.y += v.y; best.z += v.z; best.w += v.w;
      continue;
    }
    if (v.x > best.x) { best.x = v.x; a0 = s; }
    if (v.y > best.y) { best.y = v.y; a1 = s; }
    if (v.z > best.z) { best.z = v.z; a2 = s; }
    if (v.w > best.w) { best.w = v.w; a3 = s; }
  }
  if (comb == 2) {
    const float inv = 1.0f / (float)len;
    best.x *= inv; best.y *= inv; best.z *= inv; best.w *= inv;
  }
  if (drop.thresh && !pre) {
    const uint32_t i = (uint32_t)n * (uint32_t)H + (uint32_t)h;
    best.x = bl_keep(drop, i) ? best.x * drop.scale : 0.f;
    best.y = bl_keep(drop, i + 1) ? best.y * drop.scale : 0.f;
    best.z = bl_keep(drop, i + 2) ? best.z * drop.scale : 0.f;
    best.w = bl_keep(drop, i + 3) ? best.w * drop.scale : 0.f;
  }
  *reinterpret_cast<float4*>(out + (size_t)n * ld_out + h) = best;
  if (argsub == nullptr) return;  // (sum / mean: there is no winner)
  char4 a;
  a.x = (char)a0; a.y = (char)a1; a.z = (char)a2; a.w = (char)a3;
  *reinterpret_cast<char4*>(argsub + (size_t)n * H + h) = a;
}

// index of the dropout mask bit that scales the gradient of (node n, channel h) whose winning subtoken slot is s
__device__ __forceinline__ uint32_t embed_mask_index(int n, int s, int h, int S, int H, int drop_before_pool) {
  return drop_before_pool ? ((uint32_t)n * (uint32_t)S + (uint32_t)s) * (uint32_t)H + (uint32_t)h : (uint32_t)n * (uint32_t)H + (uint32_t)h;
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ g_out, int ld_g,
                                                        const int* __restrict__ ids,
                                                        const int8_t* __restrict__ argsub, int N, int S, int H,
                                                        bl_drop_dev drop, float* __restrict__ g_table, int drop_before_pool,
                                                        int comb, const int* __restrict__ lens) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * H) return;
  const int n = (int)(t / H), h = (int)(t % H);
  float g = g_out[(size_t)n * ld_g + h];
  if (comb != 0) {  // sum / mean: every slot s < len receives the (scaled) gradient
    int len = lens[n];
    len = len < 1 ? 1 : (len > S ? S : len);
    if (comb == 2) g *= 1.0f / (float)len;
    for (int s = 0; s < len; ++s) {
      const float gs = drop.thresh ? (bl_keep(drop, embed_mask_index(n, s, h, S, H, drop_before_pool)) ? g * drop.scale : 0.f) : g;
      if (gs != 0.f) unsafeAtomicAdd(&g_table[(size_t)ids[(size_t)n * S + s] * H + h], gs);
    }
    return;
  }
  const int s = argsub[(size_t)n * H + h];
  if (drop.thresh) g = bl_keep(drop, embed_mask_index(n, s, h, S, H, drop_before_pool)) ? g * drop.scale : 0.f;
  if (g != 0.f) {
    unsafeAtomicAdd(&g_table[(size_t)ids[(size_t)n * S + s] * H + h], g);
  }
}

// deterministic mode: thread h owns column h and walks the nodes in order (slow; the token-sorted kernel is the fast path)
__global__ __launch_bounds__(256) void embed_bwd_serial_kernel(const float* __restrict__ g_out, int ld_g, const int* __restrict__ ids,
                                                               const int8_t* __restrict__ argsub, int N, int S, int H, bl_drop_dev drop,
                                                               float* __restrict__ g_table, int drop_before_pool, int comb,
                                                               const int* __restrict__ lens) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  for (int n = 0; n < N; ++n) {
    float g = g_out[(size_t)n * ld_g + h];
    if (comb != 0) {
      int len = lens[n];
      len = len < 1 ? 1 : (len > S ? S : len);
      if (comb == 2) g *= 1.0f / (float)len;
      for (int s = 0; s < len; ++s) {
        const float gs = drop.thresh ? (bl_keep(drop, embed_mask_index(n, s, h, S, H, drop_before_pool)) ? g * drop.scale : 0.f) : g;
        if (gs != 0.f) g_table[(size_t)ids[(size_t)n * S + s] * H + h] += gs;
      }
      continue;
    }
    const int s = argsub[(size_t)n * H + h];
    if (drop.thresh) g = bl_keep(drop, embed_mask_index(n, s, h, S, H, drop_before_pool)) ? g * drop.scale : 0.f;
    if (g != 0.f) g_table[(size_t)ids[(size_t)n * S + s] * H + h] += g;
  }
}

// The same gradient from a token-sorted occurrence list (built by the collator): one wave per chunk of
// <= 256 occurrences (node, slot) of ONE token, channel sums kept in registers, one atomic per channel
// and chunk.  Subtoken frequencies are Zipfian: with one atomic per (node, channel) the hottest table
// rows receive ~10^4 same-address atomics, which serialise at the L2 (0.59 ms at c2); here the hottest
// row gets ~10^2.
template <int NV>
__global__ __launch_bounds__(256) void embed_bwd_sorted_kernel(const float* __restrict__ g_out, int ld_g,
                                                               const int* __restrict__ occ,
                                                               const int* __restrict__ chunk_ptr,
                                                               const int* __restrict__ chunk_tok, int nchunks,
                                                               const int8_t* __restrict__ argsub, int S, int H,
                                                               bl_drop_dev drop, float* __restrict__ g_table, int drop_before_pool,
                                                               int comb, const int* __restrict__ lens) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= nchunks) return;
  const int beg = chunk_ptr[c], end = chunk_ptr[c + 1];
  float acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.f;
  for (int base = beg; base < end; base += 64) {
    const int cnt = min(64, end - base);
    const int mine = lane < cnt ? occ[base + lane] : 0;
    for (int i0 = 0; i0 < cnt; i0 += 4) {
      float g[4][NV], inv[4];
      int a[4][NV], n[4], sl[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pos = __shfl(mine, min(i0 + u, cnt - 1), 64);
        n[u] = pos / S;
        sl[u] = i0 + u < cnt ? pos - n[u] * S : -2;  // -2 never equals an argsub value
        inv[u] = 1.f;
        if (comb == 2) {  // mean: the occurrence's share of its node's gradient
          int len = lens[n[u]];
          len = len < 1 ? 1 : (len > S ? S : len);
          inv[u] = 1.0f / (float)len;
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const int h = lane + 64 * j;
          g[u][j] = h < H ? g_out[(size_t)n[u] * ld_g + h] : 0.f;
          a[u][j] = comb != 0 ? (h < H ? sl[u] : -1) : (h < H ? (int)argsub[(size_t)n[u] * H + h] : -1);  // (sum / mean: every listed slot counts)
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          float v = g[u][j];
          // (only the slot that won contributes, so the mask bit of (n, sl, h) is the winner's when it matters)
          if (drop.thresh) v = bl_keep(drop, embed_mask_index(n[u], sl[u] < 0 ? 0 : sl[u], lane + 64 * j, S, H, drop_before_pool)) ? v * drop.scale : 0.f;
          if (a[u][j] == sl[u] && sl[u] >= 0) acc[j] += v * inv[u];
        }
    }
  }
  const size_t row = (size_t)chunk_tok[c] * H;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int h = lane + 64 * j;
    if (h < H && acc[j] != 0.f) unsafeAtomicAdd(&g_table[row + h], acc[j]);
  }
}

// ------------------------------------------------------------------------------------------------
// M2 (+ LayerNorm of M3): segmented max with argmax, one wave per segment.
// GELU (the message activation in front of the max) is not monotonic -- it falls from 0- at -inf to its minimum at
// x* = -0.7518 and rises from there -- but it is monotonic on either side of x*: the largest gelu(x) of a segment is
// gelu(largest x) or gelu(smallest x).  The GELU form therefore tracks both extremes of the RAW messages and evaluates
// two erf per (segment, channel) instead of one per (message, channel), and it has the winner's raw message at hand for
// the derivative.  (A -inf message still gives gelu(-inf) = NaN, like the eager op.)
// Segments longer than SEGMAX_HUB items among the first hub_slots entries of seg_order are "hubs": a whole workgroup
// each, at the front of the same launch (the waves scan contiguous shares, segment_max_kernel below).
#define SEGMAX_HUB 40  // longer segments (among the hub candidates) get a 4-wave workgroup
// (measured at BASELINE config c4, hidden 256: one wave per segment 3.06 ms per step; 4-wave hubs 2.39; 16-wave hubs 2.75;
// 16 waves above 128 items + 4 waves below 2.76 -- the 1024-thread workgroups cost more to dispatch than their shorter
// chains save)
#define SEGMAX_HUB_SLOTS 4096  // hub_slots unknown to the caller: the first so many slots of seg_order are looked at

// extremes of the items [beg, end) of one segment: best = largest (activated, unless GELU2) value and its item, low = smallest raw value
template <int NV, bool GELU2>
__device__ __forceinline__ void segmax_scan(const float* __restrict__ x, int ldx, const int* __restrict__ seg_items, int beg, int end, int D,
                                            int act, float (&best)[NV], int (&barg)[NV], float (&low)[NV], int (&larg)[NV]) {
  constexpr int U = NV <= 4 ? 8 : 4;  // rows in flight per step
  const int lane = threadIdx.x & 63;
  for (int base = beg; base < end; base += 64) {
    const int cnt = min(64, end - base);
    const int mine = lane < cnt ? (seg_items ? seg_items[base + lane] : base + lane) : 0;
    // U rows in flight per step (independent loads; a short last step re-reads the last row and ignores it, so that a
    // segment of up to U items is ONE round trip), compared in item order (ties -> first item)
    for (int i = 0; i < cnt; i += U) {
      int e[U];
      float v[U][NV];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        e[u] = __shfl(mine, min(i + u, cnt - 1), 64);
        const float* __restrict__ row = x + (size_t)e[u] * ldx;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const int d = lane + 64 * j;
          v[u][j] = d < D ? row[d] : NEG_INF;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i + u >= cnt) break;  // wave-uniform
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          if (lane + 64 * j >= D) continue;  // padding lanes keep barg = -1
          float t = v[u][j];
          if (GELU2) {
            if (t < low[j]) { low[j] = t; larg[j] = e[u]; }
          } else if (act == BL_ACT_GELU) {
            t = bl_gelu(t);
          }
          if (BL_MAX_WINS(t, best[j])) { best[j] = t; barg[j] = e[u]; }
        }
      }
    }
  }
}

// GELU form: decide between the two extremes; raw = the winner's message before the activation
// (werf: erf(raw / sqrt 2) of the winner -- the derivative at the winner needs it again: bl_gelu_grad without its erff)
template <int NV, bool GELU2>
__device__ __forceinline__ void segmax_pick(float (&best)[NV], int (&barg)[NV], const float (&low)[NV], const int (&larg)[NV], float (&raw)[NV],
                                            float (&werf)[NV]) {
  if (GELU2) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      raw[j] = best[j];
      werf[j] = 0.f;
      if (barg[j] < 0) continue;  // empty segment / padding lane
      const float eh = erff(best[j] * 0.70710678118654752440f), el = erff(low[j] * 0.70710678118654752440f);
      const float gh = 0.5f * best[j] * (1.0f + eh), gl = 0.5f * low[j] * (1.0f + el);  // = bl_gelu; a NaN message sits in best[j] and stays NaN
      best[j] = gh;
      werf[j] = eh;
      if (gl > gh || gl != gl) { best[j] = gl; barg[j] = larg[j]; raw[j] = low[j]; werf[j] = el; }
    }
  }
}

// per ITEM bitmask of the channels it won (bit d of row `item`) for the items [beg, end): the routed bf16x6 GEMMs of the
// backward pass read these 4 bytes per 32 channels instead of 128 bytes of the arg table
template <int NV>
__device__ __forceinline__ void segmax_winbits(const int* __restrict__ seg_items, int beg, int end, int D, const int (&barg)[NV],
                                               uint32_t* __restrict__ winbits) {
  const int lane = threadIdx.x & 63;
  const int wpr = (D + 31) >> 5;
  for (int base = beg; base < end; base += 64) {
    const int cnt = min(64, end - base);
    const int mine = lane < cnt ? (seg_items ? seg_items[base + lane] : base + lane) : 0;
    for (int i = 0; i < cnt; ++i) {
      const int e = __shfl(mine, i, 64);
      uint32_t word = 0;  // lane w ends up holding word w of the item's mask: one store per item
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const unsigned long long b = __ballot(barg[j] == e);  // padding lanes hold -1
        if (lane == 2 * j) word = (uint32_t)b;
        if (lane == 2 * j + 1) word = (uint32_t)(b >> 32);
      }
      if (lane < wpr) winbits[(size_t)e * wpr + lane] = word;
    }
  }
}

// outputs of one segment from its winners (one wave): aggregate, arg table, activation derivative, LayerNorm (+ packed copy)
template <int NV, bool HAS_LN, bool GELU2>
__device__ __forceinline__ void segmax_finish(const float* __restrict__ x, int ldx, int seg, int D, int act, float (&best)[NV],
                                              const int (&barg)[NV], const float (&raw)[NV], const float (&werf)[NV], float* __restrict__ out,
                                              int* __restrict__ arg, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                              float eps, float* __restrict__ ln_out, float* __restrict__ mean_out,
                                              float* __restrict__ rstd_out, float* __restrict__ dact,
                                              uint32_t* __restrict__ ln_out_packed, float dscale = 1.f) {
  // dscale: d aggregate / d (item sum) of the "mean" aggregation (1 / items; segment_sum_kernel), folded into dact
  const int lane = threadIdx.x & 63;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int d = lane + 64 * j;
    if (barg[j] < 0) best[j] = 0.f;  // empty segment (torch_scatter leaves 0) or padding lane
    float dv_agg = 0.f;
    if (act == BL_ACT_GELU_AGG && barg[j] >= 0) {
      // activation on the AGGREGATE (ptgnn's order): the scan compared raw messages, best[j] is the largest one; one erf per
      // (segment, channel) gives both gelu(best) and gelu'(best)
      const float r = best[j], er = erff(r * 0.70710678118654752440f);
      dv_agg = 0.5f * (1.0f + er) + r * (0.39894228040143267794f * expf(-0.5f * r * r));
      best[j] = 0.5f * r * (1.0f + er);
    }
    if (d < D) {
      if (out) out[(size_t)seg * D + d] = best[j];  // (forward-only layer calls keep nothing but the LayerNorm output)
      if (arg) arg[(size_t)seg * D + d] = barg[j];
      s += best[j];
      if (dact) {  // d act / d pre at the winner, so that backward never needs the [E, D] messages again
        float dv = 1.f;
        // (the winner's raw message is in a register: re-reading it would be 64 scattered 4-byte loads per wave,
        // several times the address-coalescing work of the whole message sweep above)
        if (act == BL_ACT_GELU_AGG) {
          dv = dv_agg;
        } else if (act == BL_ACT_GELU) {
          if (GELU2)  // bl_gelu_grad(raw) with the erf that segmax_pick already evaluated
            dv = barg[j] >= 0 ? 0.5f * (1.0f + werf[j]) + raw[j] * (0.39894228040143267794f * expf(-0.5f * raw[j] * raw[j])) : 0.f;
          else
            dv = barg[j] >= 0 ? bl_gelu_grad(x[(size_t)barg[j] * ldx + d]) : 0.f;
        }
        dact[(size_t)seg * D + d] = dv * dscale;
      }
    }
  }
  if (HAS_LN) {
    const float mean = bl_wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int d = lane + 64 * j;
      if (d < D) { const float c = best[j] - mean; q += c * c; }
    }
    const float rstd = rsqrtf(bl_wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int d = lane + 64 * j;
      const float y = d < D ? (best[j] - mean) * rstd * ln_g[d] + ln_b[d] : 0.f;
      if (d < D && ln_out) ln_out[(size_t)seg * D + d] = y;
      if (ln_out_packed) {
        // bf16x3-packed copy (the operand form of the bf16x6 dense GEMMs; D % 8 == 0): even lanes store channel pairs
        const float y1 = __shfl_down(y, 1, 64);
        if (d < D && !(lane & 1)) {
          uint16_t h0, m0, l0, h1, m1, l1;
          split3(y, h0, m0, l0);
          split3(y1, h1, m1, l1);
          const int halfD = D >> 1;
          uint32_t* o = ln_out_packed + (size_t)seg * 3 * halfD + (d >> 1);
          o[0] = (uint32_t)h0 | ((uint32_t)h1 << 16);
          o[halfD] = (uint32_t)m0 | ((uint32_t)m1 << 16);
          o[2 * halfD] = (uint32_t)l0 | ((uint32_t)l1 << 16);
        }
      }
    }
    if (lane == 0 && mean_out) { mean_out[seg] = mean; rstd_out[seg] = rstd; }
  }
}

#define SEGMAX_PARAMS                                                                                                             \
  const float *__restrict__ x, int ldx, const int *__restrict__ seg_ptr, const int *__restrict__ seg_items, int nseg, int D,     \
      int act, float *__restrict__ out, int *__restrict__ arg, const float *__restrict__ ln_g, const float *__restrict__ ln_b,   \
      float eps, float *__restrict__ ln_out, float *__restrict__ mean_out, float *__restrict__ rstd_out,                         \
      float *__restrict__ dact, uint32_t *__restrict__ winbits, const int *__restrict__ seg_order,                               \
      uint32_t *__restrict__ ln_out_packed, int hub_slots

// One launch for every segment.  The first hub_slots workgroups look at one entry of seg_order each (the collator sorts the
// high-degree nodes to the front): a segment longer than SEGMAX_HUB items is taken by all four waves -- wave w scans the w-th
// contiguous share, the partial extremes of waves 1..3 go through LDS and wave 0 merges them in share order with the same
// strict comparisons (so the first of equal values still wins, and a NaN sticks); the final winners come back through LDS for
// the routing bitmask, which every wave writes for its own share.  A shorter segment in a hub slot is wave 0's alone.  The
// remaining workgroups take four ordinary segments each, one per wave.  Until round 6 the hubs had a launch of their own in
// front of this one: the kernel boundary made the ordinary segments wait for the longest hub's last round trip (c4: 0.9 ms
// per step at ~2 TB/s); inside one launch the dispatcher backfills ordinary workgroups as hub workgroups retire.
// HUBS = false: the instantiation for launches without hub candidates (hub_slots == 0: every minibatch of uniform-degree graphs) --
// without the hub path's registers the 128-channel form keeps 62 registers = eight waves per SIMD; with it, 66 = seven, and the
// kernel lives on waves in flight (hidden-128 layers measured 6 % slower).
template <int NV, bool HAS_LN, bool GELU2, bool HUBS>
__global__ __launch_bounds__(256) void segment_max_kernel(SEGMAX_PARAMS) {
  constexpr int W = HUBS ? 64 * NV : 1;
  __shared__ float hub_f[3][W];
  __shared__ int hub_i[3][W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool hub_wg = HUBS && (int)blockIdx.x < hub_slots;  // (uniform over the workgroup)
  const int slot = hub_wg ? (int)blockIdx.x : hub_slots + ((int)blockIdx.x - hub_slots) * 4 + wave;
  if (slot >= nseg) return;
  // seg_order (optional): processing order, long segments first
  const int seg = seg_order ? seg_order[slot] : slot;
  const int beg = seg_ptr[seg], end = seg_ptr[seg + 1];
  const bool split = HUBS && hub_wg && end - beg > SEGMAX_HUB;  // (uniform over the workgroup: every wave reaches the barriers below)
  if (hub_wg && !split && wave > 0) return;
  int wb = beg, we = end;
  if (split) {
    const int share = (end - beg + 3) >> 2;
    wb = min(end, beg + wave * share);
    we = min(end, wb + share);
  }
  float best[NV], low[NV], raw[NV], werf[NV];
  int barg[NV], larg[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) { best[j] = NEG_INF; barg[j] = -1; low[j] = -NEG_INF; larg[j] = -1; raw[j] = 0.f; werf[j] = 0.f; }
  segmax_scan<NV, GELU2>(x, ldx, seg_items, wb, we, D, act, best, barg, low, larg);
  if (HUBS && split) {
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) { hub_f[wave - 1][lane + 64 * j] = best[j]; hub_i[wave - 1][lane + 64 * j] = barg[j]; }
    }
    __syncthreads();
    if (wave == 0) {
      for (int w = 0; w < 3; ++w) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const float tb = hub_f[w][lane + 64 * j];
          const int ab = hub_i[w][lane + 64 * j];
          if (ab >= 0 && BL_MAX_WINS(tb, best[j])) { best[j] = tb; barg[j] = ab; }
        }
      }
    }
    if (GELU2) {  // the smallest raw items the same way (the buffers are re-used)
      __syncthreads();
      if (wave > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) { hub_f[wave - 1][lane + 64 * j] = low[j]; hub_i[wave - 1][lane + 64 * j] = larg[j]; }
      }
      __syncthreads();
      if (wave == 0) {
        for (int w = 0; w < 3; ++w) {
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            const float tl = hub_f[w][lane + 64 * j];
            const int al = hub_i[w][lane + 64 * j];
            if (al >= 0 && tl < low[j]) { low[j] = tl; larg[j] = al; }
          }
        }
      }
    }
    if (wave == 0) {
      segmax_pick<NV, GELU2>(best, barg, low, larg, raw, werf);
#pragma unroll
      for (int j = 0; j < NV; ++j) hub_i[0][lane + 64 * j] = barg[j];  // the final winners (wave 0 was the only reader of hub_i)
    }
    __syncthreads();
    if (winbits) {
      int fin[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) fin[j] = hub_i[0][lane + 64 * j];
      segmax_winbits<NV>(seg_items, wb, we, D, fin, winbits);
    }
    if (wave > 0) return;
  } else {
    segmax_pick<NV, GELU2>(best, barg, low, larg, raw, werf);
    if (winbits) segmax_winbits<NV>(seg_items, beg, end, D, barg, winbits);
  }
  segmax_finish<NV, HAS_LN, GELU2>(x, ldx, seg, D, act, best, barg, raw, werf, out, arg, ln_g, ln_b, eps, ln_out, mean_out, rstd_out, dact,
                                   ln_out_packed);
}

// ------------------------------------------------------------------------------------------------
// "sum" / "mean" aggregation (ptgnn's other message_aggregation_function values; the reference's recipe passes "max",
// gnnlayerdefs.py:11,21): a_v = sum_{e -> v} m_e (/ number of items for "mean"; 0 for an empty segment, like torch_scatter),
// then the same epilogue as the max: activation ON THE AGGREGATE (BL_ACT_GELU_AGG) or none, its derivative, LayerNorm (+ packed
// copy).  One wave per segment, items summed in CSR order (a fixed order: bit-reproducible).  No winner table and no routing
// bits: every message receives its target's gradient (x dact, which carries the 1 / items of "mean").
template <int NV, bool HAS_LN>
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ seg_ptr,
                                                          const int* __restrict__ seg_items, int nseg, int D, int act, int mean_agg,
                                                          float* __restrict__ out, const float* __restrict__ ln_g,
                                                          const float* __restrict__ ln_b, float eps, float* __restrict__ ln_out,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                          float* __restrict__ dact, const int* __restrict__ seg_order,
                                                          uint32_t* __restrict__ ln_out_packed) {
  constexpr int U = NV <= 4 ? 8 : 4;
  const int lane = threadIdx.x & 63;
  const int slot = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= nseg) return;
  const int seg = seg_order ? seg_order[slot] : slot;
  const int beg = seg_ptr[seg], end = seg_ptr[seg + 1];
  float best[NV], raw[NV], werf[NV];
  int barg[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) { best[j] = 0.f; raw[j] = 0.f; werf[j] = 0.f; barg[j] = (end > beg && lane + 64 * j < D) ? 0 : -1; }
  for (int base = beg; base < end; base += 64) {
    const int cnt = min(64, end - base);
    const int mine = lane < cnt ? (seg_items ? seg_items[base + lane] : base + lane) : 0;
    for (int i = 0; i < cnt; i += U) {
      float v[U][NV];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = __shfl(mine, min(i + u, cnt - 1), 64);
        const float* __restrict__ row = x + (size_t)e * ldx;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const int d = lane + 64 * j;
          v[u][j] = d < D ? row[d] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i + u >= cnt) break;  // wave-uniform
#pragma unroll
        for (int j = 0; j < NV; ++j) best[j] += v[u][j];
      }
    }
  }
  float dscale = 1.f;
  if (mean_agg && end > beg) {
    dscale = 1.0f / (float)(end - beg);
#pragma unroll
    for (int j = 0; j < NV; ++j) best[j] *= dscale;
  }
  segmax_finish<NV, HAS_LN, false>(x, ldx, seg, D, act, best, barg, raw, werf, out, nullptr, ln_g, ln_b, eps, ln_out, mean_out, rstd_out, dact,
                                   ln_out_packed, dscale);
}

// backward of the segmented max in gather form: each item row looks up its segment's argmax
__global__ __launch_bounds__(256) void segment_max_bwd_kernel(const float* __restrict__ g_out,
                                                              const int* __restrict__ arg, const float* x, int ldx,
                                                              const int* __restrict__ seg_of, long long nitems, int D,
                                                              int act, float* g_x) {
  const int d4n = D >> 2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nitems * d4n) return;
  const int i = (int)(t / d4n), d = (int)(t % d4n) * 4;
  const int seg = seg_of[i];
  const int4 a = *reinterpret_cast<const int4*>(arg + (size_t)seg * D + d);
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.x == i || a.y == i || a.z == i || a.w == i) {
    const float4 go = *reinterpret_cast<const float4*>(g_out + (size_t)seg * D + d);
    float4 dv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (act == BL_ACT_GELU || act == BL_ACT_GELU_AGG) {  // the winner's raw item IS the aggregate's pre-activation: same derivative
      const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)i * ldx + d);
      dv.x = bl_gelu_grad(xv.x); dv.y = bl_gelu_grad(xv.y); dv.z = bl_gelu_grad(xv.z); dv.w = bl_gelu_grad(xv.w);
    }
    g.x = a.x == i ? go.x * dv.x : 0.f;
    g.y = a.y == i ? go.y * dv.y : 0.f;
    g.z = a.z == i ? go.z * dv.z : 0.f;
    g.w = a.w == i ? go.w * dv.w : 0.f;
  }
  *reinterpret_cast<float4*>(g_x + (size_t)i * ldx + d) = g;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row, LN_RIF rows in flight per wave (grid-stride); a lane owns the
// channel PAIRS 2*lane + 128*j (+0, +1): float2 loads/stores, and the optional bf16x3-packed copy of
// the result (the operand form of the bf16x6 GEMMs) is written as 32-bit stores.  Column sums are
// reduced per block, then atomics -- from few blocks: same-address atomics serialise (~40 ns each).
// PD (the relational transformer block, csrc/bl_great_layer.hip): the PACKED copy is the gradient of the branch
// y_branch = dropout(Linear(.)) that was added to the residual before the LayerNorm -- g_x through that dropout's mask (element
// index row * D + d) -- and g_bias += its column sums (the Linear's bias gradient); the fp32 g_x stays unmasked (the residual).
#define LN_RIF 4
template <int NP, bool PD = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ g_y, const float* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, int nrows, int D,
                                                            float* __restrict__ g_x, float* __restrict__ g_gamma,
                                                            float* __restrict__ g_beta,
                                                            const float* __restrict__ post_scale,
                                                            uint32_t* __restrict__ g_x_packed,
                                                            unsigned* __restrict__ order_ctr, bl_drop_dev pdrop,
                                                            float* __restrict__ g_bias) {
  __shared__ float red[PD ? 3 : 2][4][128 * NP];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nw = gridDim.x * 4;
  float2 dg[NP], db[NP], gam[NP], dm[PD ? NP : 1];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    dg[j] = make_float2(0.f, 0.f);
    db[j] = make_float2(0.f, 0.f);
    if (PD) dm[j] = make_float2(0.f, 0.f);
    const int d = 2 * lane + 128 * j;
    gam[j] = d < D ? *reinterpret_cast<const float2*>(gamma + d) : make_float2(0.f, 0.f);
  }
  const float invD = 1.0f / (float)D;
  const int halfD = D >> 1;
  for (int r0 = (blockIdx.x * 4 + w) * LN_RIF; r0 < nrows; r0 += nw * LN_RIF) {
    float2 gy[LN_RIF][NP], xh[LN_RIF][NP], ps[LN_RIF][NP];
    float mu[LN_RIF], rs[LN_RIF];
#pragma unroll
    for (int q = 0; q < LN_RIF; ++q) {
      const int r = min(r0 + q, nrows - 1);
      mu[q] = mean[r];
      rs[q] = rstd[r];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int d = 2 * lane + 128 * j;
        const bool ok = d < D;
        gy[q][j] = ok ? *reinterpret_cast<const float2*>(g_y + (size_t)r * D + d) : make_float2(0.f, 0.f);
        xh[q][j] = ok ? *reinterpret_cast<const float2*>(x + (size_t)r * D + d) : make_float2(0.f, 0.f);
        ps[q][j] = (ok && post_scale) ? *reinterpret_cast<const float2*>(post_scale + (size_t)r * D + d) : make_float2(1.f, 1.f);
      }
    }
#pragma unroll
    for (int q = 0; q < LN_RIF; ++q) {
      const int r = r0 + q;
      if (r >= nrows) break;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const bool ok = 2 * lane + 128 * j < D;
        xh[q][j].x = ok ? (xh[q][j].x - mu[q]) * rs[q] : 0.f;
        xh[q][j].y = ok ? (xh[q][j].y - mu[q]) * rs[q] : 0.f;
        const float g0 = gy[q][j].x * gam[j].x, g1 = gy[q][j].y * gam[j].y;
        a += g0 + g1;
        b += g0 * xh[q][j].x + g1 * xh[q][j].y;
        dg[j].x += gy[q][j].x * xh[q][j].x; dg[j].y += gy[q][j].y * xh[q][j].y;
        db[j].x += gy[q][j].x; db[j].y += gy[q][j].y;
      }
      a = bl_wave_sum(a) * invD;
      b = bl_wave_sum(b) * invD;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int d = 2 * lane + 128 * j;
        if (d < D) {
          float2 gx;
          gx.x = rs[q] * (gy[q][j].x * gam[j].x - a - xh[q][j].x * b) * ps[q][j].x;
          gx.y = rs[q] * (gy[q][j].y * gam[j].y - a - xh[q][j].y * b) * ps[q][j].y;
          if (g_x) *reinterpret_cast<float2*>(g_x + (size_t)r * D + d) = gx;
          if (PD) {
            const uint32_t e0 = (uint32_t)r * (uint32_t)D + (uint32_t)d;
            gx.x = (pdrop.thresh == 0u || bl_keep(pdrop, e0)) ? gx.x * pdrop.scale : 0.f;
            gx.y = (pdrop.thresh == 0u || bl_keep(pdrop, e0 + 1)) ? gx.y * pdrop.scale : 0.f;
            dm[j].x += gx.x; dm[j].y += gx.y;
          }
          if (g_x_packed) {
            uint16_t h0, m0, l0, h1, m1, l1;
            split3(gx.x, h0, m0, l0);
            split3(gx.y, h1, m1, l1);
            uint32_t* o = g_x_packed + (size_t)r * 3 * halfD + (d >> 1);
            o[0] = (uint32_t)h0 | ((uint32_t)h1 << 16);
            o[halfD] = (uint32_t)m0 | ((uint32_t)m1 << 16);
            o[2 * halfD] = (uint32_t)l0 | ((uint32_t)l1 << 16);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    red[0][w][2 * lane + 128 * j] = dg[j].x; red[0][w][2 * lane + 128 * j + 1] = dg[j].y;
    red[1][w][2 * lane + 128 * j] = db[j].x; red[1][w][2 * lane + 128 * j + 1] = db[j].y;
    if (PD) { red[2][w][2 * lane + 128 * j] = dm[j].x; red[2][w][2 * lane + 128 * j + 1] = dm[j].y; }
  }
  __syncthreads();
  bl_ordered_enter(order_ctr, blockIdx.x);  // deterministic mode: blocks flush in block order
  for (int d = threadIdx.x; d < D; d += 256) {
    unsafeAtomicAdd(&g_gamma[d], red[0][0][d] + red[0][1][d] + red[0][2][d] + red[0][3][d]);
    unsafeAtomicAdd(&g_beta[d], red[1][0][d] + red[1][1][d] + red[1][2][d] + red[1][3][d]);
    if (PD && g_bias) unsafeAtomicAdd(&g_bias[d], red[2][0][d] + red[2][1][d] + red[2][2][d] + red[2][3][d]);
  }
  bl_ordered_leave(order_ctr, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// backward through y = drop(act(z + bias)) from y; column sums -> g_bias
// 256 threads = (256 / tpr) rows x tpr float4-columns (tpr = power of two >= N/4); a block walks row
// chunks of ACT_BWD_ROWS grid-stride with four independent rows of loads in flight per thread and
// flushes its column sums ONCE: same-address fp32 atomics serialise at the L2 (~40 ns each), so the
// number of blocks, not the row count, sets the cost of the bias gradient (<= 512 blocks).
#define ACT_BWD_ROWS 64
#define ACT_BWD_MAX_BLOCKS 512
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* g_y, const float* __restrict__ y, int nrows, int N,
                                                      int ld, int act, bl_drop_dev drop, float* g_z,
                                                      float* __restrict__ g_bias, int tpr_log2,
                                                      unsigned* __restrict__ order_ctr, uint2* __restrict__ g_z_packed) {
  __shared__ float4 red[256];
  const int tpr = 1 << tpr_log2;
  const int tx = threadIdx.x & (tpr - 1), ty = threadIdx.x >> tpr_log2;
  const int rows_per_iter = 256 >> tpr_log2;
  const int c = (blockIdx.y * tpr + tx) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < N) {
    for (int chunk = blockIdx.x * ACT_BWD_ROWS; chunk < nrows; chunk += gridDim.x * ACT_BWD_ROWS) {
    const int r_end = min(nrows, chunk + ACT_BWD_ROWS);
    for (int r0 = chunk + ty; r0 < r_end; r0 += 4 * rows_per_iter) {
      float4 gin[4], yin[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r0 + q * rows_per_iter;
        const size_t o = (size_t)(r < r_end ? r : r0) * ld + c;
        gin[q] = *reinterpret_cast<const float4*>(g_y + o);
        yin[q] = y ? *reinterpret_cast<const float4*>(y + o) : make_float4(0.f, 0.f, 0.f, 0.f);  // (no activation: y is not read)
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r0 + q * rows_per_iter;
        if (r >= r_end) break;
        float yy[4] = {yin[q].x, yin[q].y, yin[q].z, yin[q].w};
        float gg[4] = {gin[q].x, gin[q].y, gin[q].z, gin[q].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float yu = yy[u];
          if (drop.thresh) {
            const bool keep = bl_keep(drop, (uint32_t)r * (uint32_t)N + (uint32_t)(c + u));
            gg[u] = keep ? gg[u] * drop.scale : 0.f;
            yu = yu * (1.0f / drop.scale);  // undo the dropout scale to recover act(z)
          }
          gg[u] *= bl_act_grad_from_out(act, yu);
        }
        const float4 g = make_float4(gg[0], gg[1], gg[2], gg[3]);
        if (g_z) *reinterpret_cast<float4*>(g_z + (size_t)r * ld + c) = g;
        if (g_z_packed) {  // bf16x3-packed copy, rows N wide (the operand form of the bf16x6 dense GEMMs)
          uint16_t h[4], m[4], l[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) split3(gg[u], h[u], m[u], l[u]);
          const int q = N >> 2;  // uint2 (4 bf16) per plane of a row
          uint2* o = g_z_packed + (size_t)r * 3 * q + (c >> 2);
          o[0] = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
          o[q] = make_uint2((uint32_t)m[0] | ((uint32_t)m[1] << 16), (uint32_t)m[2] | ((uint32_t)m[3] << 16));
          o[2 * q] = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
        }
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
      }
    }
    }
  }
  if (g_bias) {
    red[threadIdx.x] = acc;
    __syncthreads();
    unsigned* ctr = order_ctr ? order_ctr + blockIdx.y : nullptr;  // deterministic mode: row blocks of a column block in order
    bl_ordered_enter(ctr, blockIdx.x);
    if (ty == 0 && c < N) {
      float4 t = red[tx];
      for (int k = 1; k < rows_per_iter; ++k) {
        const float4 o = red[k * tpr + tx];
        t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
      }
      unsafeAtomicAdd(&g_bias[c + 0], t.x);
      unsafeAtomicAdd(&g_bias[c + 1], t.y);
      unsafeAtomicAdd(&g_bias[c + 2], t.z);
      unsafeAtomicAdd(&g_bias[c + 3], t.w);
    }
    bl_ordered_leave(ctr, blockIdx.x);
  }
}

// ------------------------------------------------------------------------------------------------
// node gradient = segmented sums of per-message input gradients over the source and target CSRs
template <int NV>
__global__ __launch_bounds__(256) void mp_scatter_grad_kernel(const float* __restrict__ g_a, int ld_ga,
                                                              const int* __restrict__ src_ptr,
                                                              const int* __restrict__ src_msgs,
                                                              const int* __restrict__ tgt_ptr,
                                                              const int* __restrict__ tgt_msgs, int N, int Din,
                                                              int accumulate, float* __restrict__ g_h, int ld_gh,
                                                              const int* __restrict__ node_order, int split,
                                                              float* __restrict__ g_h2, int ld_gh2, int hub_slots) {
  // The first hub_slots entries of node_order are the nodes with many incident messages (the collators put them in front).
  // One wave walking a 512-row segment eight rows at a time is 64 dependent round trips -- at the power-law configuration
  // that one wave WAS the kernel's duration (207 vs 152 us at hidden 256).  Workgroups 0 .. hub_slots - 1 therefore take ONE
  // such node each, its segments cut into four contiguous shares (one per wave), partial sums merged through LDS in wave
  // order; they are dispatched first and finish inside the time the ordinary workgroups (four nodes each) need anyway.
  // Same launch, same registers per wave (a separate hub launch, and more rows in flight, were both measured slower).
  __shared__ float hub_part[3][64 * NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool hub_wg = (int)blockIdx.x < hub_slots;  // (uniform over the workgroup)
  const int slot = hub_wg ? (int)blockIdx.x : hub_slots + ((int)blockIdx.x - hub_slots) * 4 + wave;
  if (slot >= N) return;
  const int n = node_order ? node_order[slot] : slot;  // hubs first (see segment_max_kernel)
  // the four-way split pays from a few round trips per wave on: a node with fewer than MPS_SPLIT_MIN incident messages in a hub
  // slot (the collator counts a node as a hub candidate above 32) is wave 0's alone -- no LDS merge, no barrier, and the same
  // summation order as on the ordinary path (uniform over the workgroup: it depends on n only)
  constexpr int MPS_SPLIT_MIN = 64;
  const bool split4 = hub_wg && (src_ptr[n + 1] - src_ptr[n]) + (tgt_ptr ? tgt_ptr[n + 1] - tgt_ptr[n] : 0) >= MPS_SPLIT_MIN;
  if (hub_wg && !split4 && wave > 0) return;
  float acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int d = lane + 64 * j;
    acc[j] = (accumulate && d < Din && !(split4 && wave > 0)) ? (d < split ? g_h[(size_t)n * ld_gh + d] : g_h2[(size_t)n * ld_gh2 + d - split]) : 0.f;
  }
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    if (part == 1 && tgt_ptr == nullptr) continue;  // source-only messages (GGNN)
    const int* __restrict__ ptr = part == 0 ? src_ptr : tgt_ptr;
    const int* __restrict__ items = part == 0 ? src_msgs : tgt_msgs;
    const int coff = part == 0 ? 0 : Din;
    int beg = ptr[n], end = ptr[n + 1];
    if (split4) {  // this wave's contiguous share of the segment
      const int share = (end - beg + 3) >> 2;
      beg = min(end, beg + wave * share);
      end = min(end, beg + share);
    }
    for (int base = beg; base < end; base += 64) {
      const int cnt = min(64, end - base);
      const int mine = lane < cnt ? items[base + lane] : 0;
      // U independent row loads in flight per step, summed in item order; a short last step re-reads the last row and
      // drops it, so that up to U items are ONE round trip
      constexpr int U = NV <= 4 ? 8 : 4;
      for (int i = 0; i < cnt; i += U) {
        float v[U][NV];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = __shfl(mine, min(i + u, cnt - 1), 64);
          const float* __restrict__ row = g_a + (size_t)e * ld_ga + coff;
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            const int d = lane + 64 * j;
            v[u][j] = d < Din ? row[d] : 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (i + u >= cnt) break;  // wave-uniform
#pragma unroll
          for (int j = 0; j < NV; ++j) acc[j] += v[u][j];
        }
      }
    }
  }
  if (split4) {  // partial sums of waves 1..3 -> wave 0, added in wave order (a fixed order: deterministic)
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) hub_part[wave - 1][lane + 64 * j] = acc[j];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int j = 0; j < NV; ++j) acc[j] += hub_part[w][lane + 64 * j];
  }
  // columns >= split go to the second output (the two inputs of a folded ConcatResidual: [stash ; current])
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int d = lane + 64 * j;
    if (d < Din) {
      if (d < split) g_h[(size_t)n * ld_gh + d] = acc[j];
      else g_h2[(size_t)n * ld_gh2 + d - split] = acc[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// GRU cell of the gated (GGNN) node update, torch.nn.GRUCell gate order [r | z | n]:
//   r = sig(gi_r + gh_r), z = sig(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = drop((1 - z) * n + z * h)
// gi = x W_i + b_i and gh = h W_h + b_h come from the MFMA GEMM; this kernel is the elementwise part.
__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                           const float* __restrict__ h, int ld_h, long long N, int D,
                                                           bl_drop_dev drop, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * D) return;
  const long long n = t / D;
  const int d = (int)(t % D);
  const float* a = gi + n * 3 * D;
  const float* b = gh + n * 3 * D;
  const float r = 1.f / (1.f + expf(-(a[d] + b[d])));
  const float z = 1.f / (1.f + expf(-(a[D + d] + b[D + d])));
  const float nn = tanhf(a[2 * D + d] + r * b[2 * D + d]);
  float v = (1.f - z) * nn + z * h[n * ld_h + d];
  if (drop.thresh) v = bl_keep(drop, (uint32_t)t) ? v * drop.scale : 0.f;
  out[t] = v;
}

__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ gi,
                                                           const float* __restrict__ gh, const float* __restrict__ h,
                                                           int ld_h, long long N, int D, bl_drop_dev drop,
                                                           float* __restrict__ g_gi, float* __restrict__ g_gh,
                                                           float* __restrict__ g_h) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * D) return;
  const long long n = t / D;
  const int d = (int)(t % D);
  const float* a = gi + n * 3 * D;
  const float* b = gh + n * 3 * D;
  const float r = 1.f / (1.f + expf(-(a[d] + b[d])));
  const float z = 1.f / (1.f + expf(-(a[D + d] + b[D + d])));
  const float bn = b[2 * D + d];
  const float nn = tanhf(a[2 * D + d] + r * bn);
  const float hv = h[n * ld_h + d];
  float g = g_out[t];
  if (drop.thresh) g = bl_keep(drop, (uint32_t)t) ? g * drop.scale : 0.f;
  const float g_n = g * (1.f - z);
  const float g_z = g * (hv - nn);
  const float g_npre = g_n * (1.f - nn * nn);
  const float g_r = g_npre * bn;
  const float g_rpre = g_r * r * (1.f - r);
  const float g_zpre = g_z * z * (1.f - z);
  float* ga = g_gi + n * 3 * D;
  float* gb = g_gh + n * 3 * D;
  ga[d] = g_rpre;
  gb[d] = g_rpre;
  ga[D + d] = g_zpre;
  gb[D + d] = g_zpre;
  ga[2 * D + d] = g_npre;
  gb[2 * D + d] = g_npre * r;
  g_h[t] = g * z;
}

// ================================================================================================
// C ABI
#define DISPATCH_NV(D, ...)                                        \
  if ((D) <= 64) { constexpr int NV = 1; __VA_ARGS__; }            \
  else if ((D) <= 128) { constexpr int NV = 2; __VA_ARGS__; }      \
  else if ((D) <= 256) { constexpr int NV = 4; __VA_ARGS__; }      \
  else { constexpr int NV = 8; __VA_ARGS__; }

// subtoken_combination of the node model (reference modelregistry.py:65-66: "max" unless node_representations says otherwise):
// BL_POOL_MAX 0, BL_POOL_SUM 1, BL_POOL_MEAN 2 (sum / len).  argsub: the winners' slots, max only (NULL otherwise).
extern "C" int bl_embed_subtoken_pool_fwd(const float* table, int32_t V, int32_t H, const int32_t* ids, const int32_t* lens, int32_t N,
                                          int32_t S, int32_t combination, bl_dropout_t drop, int32_t drop_before_pool, float* out,
                                          int32_t ld_out, int8_t* argsub, void* stream) {
  const char* who = "bl_embed_subtoken_pool_fwd";
  if (N == 0) return BL_OK;
  BL_CHECK_ARG(combination >= 0 && combination <= 2, "%s: combination must be 0 (max), 1 (sum) or 2 (mean)", who);
  BL_CHECK_ARG(table && ids && lens && out && (argsub || combination != 0), "%s: null pointer", who);
  BL_CHECK_ARG(H > 0 && H % 4 == 0 && ld_out % 4 == 0 && S >= 1 && S <= 127 && V > 0, "%s: H %% 4, 1 <= S <= 127", who);
  BL_CHECK_ARG(bl_aligned16(table) && bl_aligned16(out), "%s: misaligned", who);
  BL_CHECK_ARG(!drop_before_pool || (uint64_t)N * (uint64_t)S * (uint64_t)H < (1ull << 32), "%s: dropout index space is 32 bit", who);
  const long long total = (long long)N * (H / 4);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, H,
                     ids, lens, N, S, bl_make_drop(drop), out, ld_out, combination == 0 ? argsub : nullptr, drop_before_pool, combination);
  BL_LAUNCH_CHECK(who);
  return BL_OK;
}

extern "C" int bl_embed_subtoken_pool_bwd(const float* g_out, int32_t ld_g, const int32_t* ids, const int32_t* lens, const int8_t* argsub,
                                          int32_t N, int32_t S, int32_t H, int32_t V, int32_t combination, bl_dropout_t drop,
                                          int32_t drop_before_pool, float* g_table, void* stream) {
  const char* who = "bl_embed_subtoken_pool_bwd";
  if (N == 0) return BL_OK;
  BL_CHECK_ARG(combination >= 0 && combination <= 2, "%s: combination must be 0 (max), 1 (sum) or 2 (mean)", who);
  BL_CHECK_ARG(g_out && ids && g_table && V > 0 && (combination == 0 ? argsub != nullptr : lens != nullptr), "%s: null pointer", who);
  const long long total = (long long)N * H;
  if (bl_get_deterministic())
    hipLaunchKernelGGL(embed_bwd_serial_kernel, dim3((H + 255) / 256), dim3(256), 0, (hipStream_t)stream, g_out, ld_g, ids, argsub, N,
                       S, H, bl_make_drop(drop), g_table, drop_before_pool, combination, lens);
  else
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_out,
                       ld_g, ids, argsub, N, S, H, bl_make_drop(drop), g_table, drop_before_pool, combination, lens);
  BL_LAUNCH_CHECK(who);
  return BL_OK;
}

extern "C" int bl_embed_subtoken_pool_bwd_sorted(const float* g_out, int32_t ld_g, const int32_t* occ, const int32_t* chunk_ptr,
                                                 const int32_t* chunk_tok, int32_t nchunks, const int32_t* lens, const int8_t* argsub,
                                                 int32_t S, int32_t H, int32_t combination, bl_dropout_t drop, int32_t drop_before_pool,
                                                 float* g_table, void* stream) {
  const char* who = "bl_embed_subtoken_pool_bwd_sorted";
  if (nchunks == 0) return BL_OK;
  BL_CHECK_ARG(combination >= 0 && combination <= 2, "%s: combination must be 0 (max), 1 (sum) or 2 (mean)", who);
  BL_CHECK_ARG(g_out && occ && chunk_ptr && chunk_tok && g_table && (combination == 0 ? argsub != nullptr : true) &&
                   (combination == 2 ? lens != nullptr : true), "%s: null pointer", who);
  BL_CHECK_ARG(H > 0 && H <= 512 && S >= 1 && S <= 127, "%s: H in 1..512, 1 <= S <= 127", who);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_NV(H, hipLaunchKernelGGL((embed_bwd_sorted_kernel<NV>), dim3((nchunks + 3) / 4), dim3(256), 0, st, g_out, ld_g, occ,
                                     chunk_ptr, chunk_tok, nchunks, argsub, S, H, bl_make_drop(drop), g_table, drop_before_pool,
                                     combination, lens))
  BL_LAUNCH_CHECK(who);
  return BL_OK;
}

// the "max" forms SURVEY section 8b names (= combination 0)
extern "C" int bl_embed_subtoken_max_fwd(const float* table, int32_t V, int32_t H, const int32_t* ids,
                                         const int32_t* lens, int32_t N, int32_t S, bl_dropout_t drop, int32_t drop_before_pool,
                                         float* out, int32_t ld_out, int8_t* argsub, void* stream) {
  return bl_embed_subtoken_pool_fwd(table, V, H, ids, lens, N, S, 0, drop, drop_before_pool, out, ld_out, argsub, stream);
}

extern "C" int bl_embed_subtoken_max_bwd(const float* g_out, int32_t ld_g, const int32_t* ids, const int8_t* argsub,
                                         int32_t N, int32_t S, int32_t H, int32_t V, bl_dropout_t drop, int32_t drop_before_pool,
                                         float* g_table, void* stream) {
  return bl_embed_subtoken_pool_bwd(g_out, ld_g, ids, nullptr, argsub, N, S, H, V, 0, drop, drop_before_pool, g_table, stream);
}

extern "C" int bl_embed_subtoken_max_bwd_sorted(const float* g_out, int32_t ld_g, const int32_t* occ,
                                               const int32_t* chunk_ptr, const int32_t* chunk_tok, int32_t nchunks,
                                               const int8_t* argsub, int32_t S, int32_t H, bl_dropout_t drop,
                                               int32_t drop_before_pool, float* g_table, void* stream) {
  return bl_embed_subtoken_pool_bwd_sorted(g_out, ld_g, occ, chunk_ptr, chunk_tok, nchunks, nullptr, argsub, S, H, 0, drop, drop_before_pool,
                                           g_table, stream);
}

extern "C" int bl_segment_max_fwd(const float* x, int32_t ldx, const int32_t* seg_ptr, const int32_t* seg_items,
                                  int32_t nseg, int32_t D, int32_t act, float* out, int32_t* arg, const float* ln_g,
                                  const float* ln_b, float eps, float* ln_out, float* mean, float* rstd, float* dact,
                                  uint32_t* winbits, const int32_t* seg_order, void* stream) {
  return bl_segment_max_fwd_impl(x, ldx, seg_ptr, seg_items, nseg, D, act, out, arg, ln_g, ln_b, eps, ln_out, mean, rstd, dact, winbits,
                                 seg_order, nullptr, -1, stream);
}

// + ln_out_packed: the LayerNorm output also (or only: ln_out may then be NULL) in bl_pack_bf16x3's packed form
int bl_segment_max_fwd_impl(const float* x, int32_t ldx, const int32_t* seg_ptr, const int32_t* seg_items, int32_t nseg, int32_t D,
                            int32_t act, float* out, int32_t* arg, const float* ln_g, const float* ln_b, float eps, float* ln_out,
                            float* mean, float* rstd, float* dact, uint32_t* winbits, const int32_t* seg_order,
                            uint16_t* ln_out_packed, int32_t num_hub_slots, void* stream) {
  if (nseg == 0) return BL_OK;
  // arg (the winner table) is optional; so are out / mean / rstd when only the LayerNorm output is wanted (forward-only calls)
  BL_CHECK_ARG(seg_ptr && (out || ln_g), "bl_segment_max_fwd: null pointer");
  BL_CHECK_ARG(D > 0 && D <= 512, "bl_segment_max_fwd: D must be in 1..512 (got %d)", D);
  BL_CHECK_ARG(act == BL_ACT_NONE || act == BL_ACT_GELU || act == BL_ACT_GELU_AGG, "bl_segment_max_fwd: act must be NONE, GELU or GELU_AGG");
  const bool has_ln = ln_g != nullptr;
  BL_CHECK_ARG(!has_ln || (ln_b && (ln_out || ln_out_packed) && (mean == nullptr) == (rstd == nullptr)), "bl_segment_max_fwd: LayerNorm outputs missing");
  BL_CHECK_ARG(ln_out_packed == nullptr || (has_ln && D % 8 == 0), "bl_segment_max_fwd: the packed LayerNorm output needs D %% 8 == 0");
  hipStream_t st = (hipStream_t)stream;
  uint32_t* lnp = reinterpret_cast<uint32_t*>(ln_out_packed);
  // hub candidates: the first hub_slots entries of seg_order (the collator sorts high-degree nodes to the front) get a
  // workgroup each at the front of the grid, so that the longest segments start at t = 0
  const int hub_slots = seg_order ? (num_hub_slots < 0 ? min(nseg, SEGMAX_HUB_SLOTS) : min(nseg, num_hub_slots)) : 0;
  dim3 grid(hub_slots + (nseg - hub_slots + 3) / 4), block(256);
#define SEGMAX_GO(LN_, G2_)                                                                                                          \
  DISPATCH_NV(D, {                                                                                                                   \
    if (hub_slots > 0)                                                                                                               \
      hipLaunchKernelGGL((segment_max_kernel<NV, LN_, G2_, true>), grid, block, 0, st, x, ldx, seg_ptr, seg_items, nseg, D, act, out, \
                         arg, ln_g, ln_b, eps, ln_out, mean, rstd, dact, winbits, seg_order, lnp, hub_slots);                        \
    else                                                                                                                             \
      hipLaunchKernelGGL((segment_max_kernel<NV, LN_, G2_, false>), grid, block, 0, st, x, ldx, seg_ptr, seg_items, nseg, D, act,     \
                         out, arg, ln_g, ln_b, eps, ln_out, mean, rstd, dact, winbits, seg_order, lnp, 0);                           \
  })
  if (has_ln) {
    if (act == BL_ACT_GELU) SEGMAX_GO(true, true) else SEGMAX_GO(true, false)
  } else {
    if (act == BL_ACT_GELU) SEGMAX_GO(false, true) else SEGMAX_GO(false, false)
  }
  BL_LAUNCH_CHECK("bl_segment_max_fwd");
  return BL_OK;
}

// segmented sum / mean + activation on the aggregate + LayerNorm (the fused layer's "sum" / "mean" aggregation): segment_sum_kernel
int bl_segment_sum_fwd_impl(const float* x, int32_t ldx, const int32_t* seg_ptr, const int32_t* seg_items, int32_t nseg, int32_t D,
                            int32_t act, int32_t mean_agg, float* out, const float* ln_g, const float* ln_b, float eps, float* ln_out,
                            float* mean, float* rstd, float* dact, const int32_t* seg_order, uint16_t* ln_out_packed, void* stream) {
  if (nseg == 0) return BL_OK;
  BL_CHECK_ARG(seg_ptr && (out || ln_g), "bl_segment_sum_fwd: null pointer");
  BL_CHECK_ARG(D > 0 && D <= 512, "bl_segment_sum_fwd: D must be in 1..512 (got %d)", D);
  BL_CHECK_ARG(act == BL_ACT_NONE || act == BL_ACT_GELU_AGG, "bl_segment_sum_fwd: the activation sits on the aggregate (GELU_AGG) or is absent");
  const bool has_ln = ln_g != nullptr;
  BL_CHECK_ARG(!has_ln || (ln_b && (ln_out || ln_out_packed) && (mean == nullptr) == (rstd == nullptr)), "bl_segment_sum_fwd: LayerNorm outputs missing");
  BL_CHECK_ARG(ln_out_packed == nullptr || (has_ln && D % 8 == 0), "bl_segment_sum_fwd: the packed LayerNorm output needs D %% 8 == 0");
  hipStream_t st = (hipStream_t)stream;
  uint32_t* lnp = reinterpret_cast<uint32_t*>(ln_out_packed);
  dim3 grid((nseg + 3) / 4), block(256);
  DISPATCH_NV(D, {
    if (has_ln)
      hipLaunchKernelGGL((segment_sum_kernel<NV, true>), grid, block, 0, st, x, ldx, seg_ptr, seg_items, nseg, D, act, mean_agg, out, ln_g, ln_b,
                         eps, ln_out, mean, rstd, dact, seg_order, lnp);
    else
      hipLaunchKernelGGL((segment_sum_kernel<NV, false>), grid, block, 0, st, x, ldx, seg_ptr, seg_items, nseg, D, act, mean_agg, out, ln_g, ln_b,
                         eps, ln_out, mean, rstd, dact, seg_order, lnp);
  })
  BL_LAUNCH_CHECK("bl_segment_sum_fwd");
  return BL_OK;
}

extern "C" int bl_segment_max_bwd(const float* g_out, const int32_t* arg, const float* x, int32_t ldx,
                                  const int32_t* seg_of, int32_t nitems, int32_t D, int32_t act, float* g_x,
                                  void* stream) {
  if (nitems == 0) return BL_OK;
  BL_CHECK_ARG(g_out && arg && seg_of && g_x, "bl_segment_max_bwd: null pointer");
  BL_CHECK_ARG(D > 0 && D % 4 == 0 && ldx % 4 == 0, "bl_segment_max_bwd: D and ldx must be multiples of 4");
  BL_CHECK_ARG(act == BL_ACT_NONE || ((act == BL_ACT_GELU || act == BL_ACT_GELU_AGG) && x), "bl_segment_max_bwd: act must be NONE or GELU / GELU_AGG (+x)");
  const long long total = (long long)nitems * (D / 4);
  hipLaunchKernelGGL(segment_max_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     g_out, arg, x, ldx, seg_of, (long long)nitems, D, act, g_x);
  BL_LAUNCH_CHECK("bl_segment_max_bwd");
  return BL_OK;
}

extern "C" int bl_layernorm_bwd(const float* g_y, const float* x, const float* mean, const float* rstd,
                                const float* gamma, int32_t nrows, int32_t D, float* g_x, float* g_gamma, float* g_beta,
                                const float* post_scale, uint16_t* g_x_packed, void* stream) {
  if (nrows == 0) return BL_OK;
  BL_CHECK_ARG(g_y && x && mean && rstd && gamma && (g_x || g_x_packed) && g_gamma && g_beta, "bl_layernorm_bwd: null pointer");
  BL_CHECK_ARG(D > 0 && D <= 512 && D % 2 == 0, "bl_layernorm_bwd: D must be even and in 2..512 (got %d)", D);
  BL_CHECK_ARG(g_x_packed == nullptr || D % 8 == 0, "bl_layernorm_bwd: the packed output needs D %% 8 == 0");
  const int blocks = min((nrows + 4 * LN_RIF - 1) / (4 * LN_RIF), 512);  // also bounds the same-address atomics on g_gamma / g_beta
  hipStream_t st = (hipStream_t)stream;
  uint32_t* gp = reinterpret_cast<uint32_t*>(g_x_packed);
  const bl_drop_dev nodrop = {0u, 0u, 1.f};
#define LN_BWD_GO(NP_) hipLaunchKernelGGL((layernorm_bwd_kernel<NP_>), dim3(blocks), dim3(256), 0, st, g_y, x, mean, rstd, gamma, nrows, D, g_x, g_gamma, g_beta, post_scale, gp, bl_order_counters(1, stream), nodrop, (float*)nullptr)
  if (D <= 128) LN_BWD_GO(1);
  else if (D <= 256) LN_BWD_GO(2);
  else LN_BWD_GO(4);
  BL_LAUNCH_CHECK("bl_layernorm_bwd");
  return BL_OK;
}

extern "C" int bl_layernorm_bwd_branch(const float* g_y, const float* x, const float* mean, const float* rstd, const float* gamma,
                                       int32_t nrows, int32_t D, float* g_x, float* g_gamma, float* g_beta, bl_dropout_t branch_drop,
                                       float* g_bias, uint16_t* g_branch_packed, void* stream) {
  if (nrows == 0) return BL_OK;
  BL_CHECK_ARG(g_y && x && mean && rstd && gamma && g_x && g_branch_packed && g_gamma && g_beta, "bl_layernorm_bwd_branch: null pointer");
  BL_CHECK_ARG(D > 0 && D <= 512 && D % 8 == 0, "bl_layernorm_bwd_branch: D must be a multiple of 8 up to 512 (got %d)", D);
  BL_CHECK_ARG((uint64_t)nrows * (uint64_t)D < (1ull << 32) || branch_drop.p <= 0.f, "bl_layernorm_bwd_branch: dropout index space is 32 bit");
  const int blocks = min((nrows + 4 * LN_RIF - 1) / (4 * LN_RIF), 512);
  hipStream_t st = (hipStream_t)stream;
  uint32_t* gp = reinterpret_cast<uint32_t*>(g_branch_packed);
  const bl_drop_dev pd = bl_make_drop(branch_drop);
#define LN_BWD_BR(NP_) hipLaunchKernelGGL((layernorm_bwd_kernel<NP_, true>), dim3(blocks), dim3(256), 0, st, g_y, x, mean, rstd, gamma, nrows, D, g_x, g_gamma, g_beta, (const float*)nullptr, gp, bl_order_counters(1, stream), pd, g_bias)
  if (D <= 128) LN_BWD_BR(1);
  else if (D <= 256) LN_BWD_BR(2);
  else LN_BWD_BR(4);
  BL_LAUNCH_CHECK("bl_layernorm_bwd_branch");
  return BL_OK;
}

extern "C" int bl_act_bwd(const float* g_y, const float* y, int32_t nrows, int32_t N, int32_t ld, int32_t act,
                          bl_dropout_t drop, float* g_z, float* g_bias, void* stream) {
  BL_CHECK_ARG(nrows == 0 || g_z, "bl_act_bwd: null pointer");
  return bl_act_bwd_impl(g_y, y, nrows, N, ld, act, drop, g_z, g_bias, nullptr, stream);
}

extern "C" int bl_act_bwd_packed(const float* g_y, const float* y, int32_t nrows, int32_t N, int32_t ld, int32_t act,
                                 bl_dropout_t drop, float* g_z, float* g_bias, uint16_t* g_z_packed, void* stream) {
  return bl_act_bwd_impl(g_y, y, nrows, N, ld, act, drop, g_z, g_bias, g_z_packed, stream);
}

// + g_z_packed: the result also (or only: g_z may then be NULL) in bl_pack_bf16x3's packed form, rows N wide
int bl_act_bwd_impl(const float* g_y, const float* y, int32_t nrows, int32_t N, int32_t ld, int32_t act, bl_dropout_t drop,
                    float* g_z, float* g_bias, uint16_t* g_z_packed, void* stream) {
  if (nrows == 0) return BL_OK;
  BL_CHECK_ARG(g_y && y && (g_z || g_z_packed), "bl_act_bwd: null pointer");
  BL_CHECK_ARG(g_z_packed == nullptr || (N % 8 == 0 && bl_aligned16(g_z_packed)), "bl_act_bwd: the packed output needs N %% 8 == 0");
  BL_CHECK_ARG(N > 0 && N % 4 == 0 && ld % 4 == 0, "bl_act_bwd: N/ld multiples of 4");
  BL_CHECK_ARG(act != BL_ACT_GELU, "bl_act_bwd: GELU needs the pre-activation (use bl_segment_max_bwd)");
  int tpr_log2 = 0;
  while ((1 << tpr_log2) < N / 4 && tpr_log2 < 8) ++tpr_log2;
  const int gy = (N / 4 + (1 << tpr_log2) - 1) >> tpr_log2;
  dim3 grid(min((nrows + ACT_BWD_ROWS - 1) / ACT_BWD_ROWS, max(1, ACT_BWD_MAX_BLOCKS / gy)), gy);
  // without an activation the derivative does not depend on y (dropout's mask comes from the counter hash): 4 of the 14 bytes
  // per element of the packed form are not moved (the sequence models' QKV / output / second feed-forward projections)
  if (act == BL_ACT_NONE) y = nullptr;
  hipLaunchKernelGGL(act_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, g_y, y, nrows, N, ld, act,
                     bl_make_drop(drop), g_z, g_bias, tpr_log2, g_bias ? bl_order_counters(gy, stream) : nullptr,
                     reinterpret_cast<uint2*>(g_z_packed));
  BL_LAUNCH_CHECK("bl_act_bwd");
  return BL_OK;
}

// one launch of the segmented sums: hub_slots leading entries of node_order get a workgroup each (see the kernel), the
// other nodes a wave each
static int mp_scatter_launch(const char* who, const float* g_a, int32_t ld_ga, const int32_t* src_ptr, const int32_t* src_msgs,
                             const int32_t* tgt_ptr, const int32_t* tgt_msgs, int32_t N, int32_t Din, int32_t accumulate, float* g_h,
                             int32_t ld_gh, const int32_t* node_order, int32_t split, float* g_h2, int32_t ld_gh2, int32_t hub_slots,
                             void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int hubs = (node_order && hub_slots > 0) ? min(hub_slots, N) : 0;
  const dim3 grid(hubs + (N - hubs + 3) / 4);
  DISPATCH_NV(Din, hipLaunchKernelGGL((mp_scatter_grad_kernel<NV>), grid, dim3(256), 0, st, g_a, ld_ga, src_ptr, src_msgs, tgt_ptr,
                                       tgt_msgs, N, Din, accumulate, g_h, ld_gh, node_order, split, g_h2, ld_gh2, hubs))
  BL_LAUNCH_CHECK(who);
  return BL_OK;
}

extern "C" int bl_mp_scatter_grad(const float* g_a, int32_t ld_ga, const int32_t* src_ptr, const int32_t* src_msgs,
                                  const int32_t* tgt_ptr, const int32_t* tgt_msgs, int32_t N, int32_t Din,
                                  int32_t accumulate, float* g_h, int32_t ld_gh, const int32_t* node_order, void* stream) {
  if (N == 0) return BL_OK;
  BL_CHECK_ARG(g_a && src_ptr && src_msgs && g_h && (tgt_ptr == nullptr) == (tgt_msgs == nullptr), "bl_mp_scatter_grad: null pointer");
  BL_CHECK_ARG(Din > 0 && Din <= 512 && ld_ga >= (tgt_ptr ? 2 : 1) * Din, "bl_mp_scatter_grad: Din in 1..512, ld_ga >= (1 or 2)*Din");
  return mp_scatter_launch("bl_mp_scatter_grad", g_a, ld_ga, src_ptr, src_msgs, tgt_ptr, tgt_msgs, N, Din, accumulate, g_h, ld_gh, node_order,
                           Din, nullptr, 0, 0, stream);
}

// The layer calls' form of bl_mp_scatter_grad / _split with the number of hub entries at the front of node_order
// (bl_mp_layer_t.num_hub_slots): g_h_hi == NULL -> one output of Din columns (split ignored)
int bl_mp_scatter_grad_hubs_impl(const float* g_a, int32_t ld_ga, const int32_t* src_ptr, const int32_t* src_msgs, const int32_t* tgt_ptr,
                                 const int32_t* tgt_msgs, int32_t N, int32_t Din, int32_t split, float* g_h_lo, int32_t ld_lo, float* g_h_hi,
                                 int32_t ld_hi, const int32_t* node_order, int32_t hub_slots, void* stream) {
  if (N == 0) return BL_OK;
  BL_CHECK_ARG(g_a && src_ptr && src_msgs && tgt_ptr && tgt_msgs && g_h_lo && Din > 0 && Din <= 512 && ld_ga >= 2 * Din,
               "bl_mp_scatter_grad (layer call): bad argument");
  BL_CHECK_ARG(g_h_hi == nullptr || (split > 0 && split < Din && ld_lo >= split && ld_hi >= Din - split), "bl_mp_scatter_grad (layer call): split");
  return mp_scatter_launch("bl_mp_scatter_grad", g_a, ld_ga, src_ptr, src_msgs, tgt_ptr, tgt_msgs, N, Din, 0, g_h_lo, ld_lo, node_order,
                           g_h_hi ? split : Din, g_h_hi, ld_hi, hub_slots, stream);
}

// g_h (+)= sums of the SOURCE-half rows g_src [E, Din] over the source CSR (the second step of the half-atomic input gradient,
// bl_routed_dgrad_nodes_rows); split / g_h_hi as in bl_mp_scatter_grad_split, accumulate on top of what the atomics left there
int bl_mp_scatter_src_accum_impl(const float* g_src, int32_t ld_src, const int32_t* src_ptr, const int32_t* src_msgs, int32_t N,
                                 int32_t Din, int32_t split, float* g_h_lo, int32_t ld_lo, float* g_h_hi, int32_t ld_hi,
                                 const int32_t* node_order, int32_t hub_slots, void* stream) {
  if (N == 0) return BL_OK;
  BL_CHECK_ARG(g_src && src_ptr && src_msgs && g_h_lo && Din > 0 && Din <= 512 && ld_src >= Din, "bl_mp_scatter_src_accum: bad argument");
  BL_CHECK_ARG((split == Din && g_h_hi == nullptr) || (split > 0 && split < Din && g_h_hi), "bl_mp_scatter_src_accum: split");
  return mp_scatter_launch("bl_mp_scatter_src_accum", g_src, ld_src, src_ptr, src_msgs, nullptr, nullptr, N, Din, 1, g_h_lo, ld_lo, node_order,
                           split, g_h_hi, ld_hi, hub_slots, stream);
}

extern "C" int bl_mp_scatter_grad_split(const float* g_a, int32_t ld_ga, const int32_t* src_ptr, const int32_t* src_msgs,
                                        const int32_t* tgt_ptr, const int32_t* tgt_msgs, int32_t N, int32_t Din,
                                        int32_t split, float* g_h_lo, int32_t ld_lo, float* g_h_hi, int32_t ld_hi,
                                        const int32_t* node_order, void* stream) {
  if (N == 0) return BL_OK;
  BL_CHECK_ARG(g_a && src_ptr && src_msgs && tgt_ptr && tgt_msgs && g_h_lo && g_h_hi, "bl_mp_scatter_grad_split: null pointer");
  BL_CHECK_ARG(Din > 0 && Din <= 512 && ld_ga >= 2 * Din && split > 0 && split < Din && ld_lo >= split && ld_hi >= Din - split,
               "bl_mp_scatter_grad_split: Din in 1..512, 0 < split < Din, ld_ga >= 2*Din");
  return mp_scatter_launch("bl_mp_scatter_grad_split", g_a, ld_ga, src_ptr, src_msgs, tgt_ptr, tgt_msgs, N, Din, 0, g_h_lo, ld_lo, node_order,
                           split, g_h_hi, ld_hi, 0, stream);
}

extern "C" int bl_gru_cell_fwd(const float* gi, const float* gh, const float* h, int32_t ld_h, int32_t N, int32_t D,
                               bl_dropout_t drop, float* out, void* stream) {
  if (N == 0) return BL_OK;
  BL_CHECK_ARG(gi && gh && h && out && D > 0, "bl_gru_cell_fwd: null pointer");
  const long long total = (long long)N * D;
  hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gi, gh, h,
                     ld_h, (long long)N, D, bl_make_drop(drop), out);
  BL_LAUNCH_CHECK("bl_gru_cell_fwd");
  return BL_OK;
}

extern "C" int bl_gru_cell_bwd(const float* g_out, const float* gi, const float* gh, const float* h, int32_t ld_h,
                               int32_t N, int32_t D, bl_dropout_t drop, float* g_gi, float* g_gh, float* g_h,
                               void* stream) {
  if (N == 0) return BL_OK;
  BL_CHECK_ARG(g_out && gi && gh && h && g_gi && g_gh && g_h && D > 0, "bl_gru_cell_bwd: null pointer");
  const long long total = (long long)N * D;
  hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_out, gi,
                     gh, h, ld_h, (long long)N, D, bl_make_drop(drop), g_gi, g_gh, g_h);
  BL_LAUNCH_CHECK("bl_gru_cell_bwd");
  return BL_OK;
}
