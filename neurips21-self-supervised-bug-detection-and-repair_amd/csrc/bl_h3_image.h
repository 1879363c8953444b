// Weight image of the f16x3 row GEMM (bl_gemm_h3.hip); shared with the one-launch weight packer of bl_gemm_x6.hip.
#pragma once
#include "bl_common.h"

#define BL_PKH(a_, b_) ((uint32_t)(a_) | ((uint32_t)(b_) << 16))
// Weights TILED like bl_pack_weights_x6, two planes: per group, 128-column tile and 32-k stage one contiguous 16 KB block
// [i (2)][plane (2)][row_lo (64)][k-group (4)] x 8 halves, column n = 64 i + row_lo of the tile.
__device__ __forceinline__ void pack_weights_h_thread(const float* __restrict__ w, int G, int K, int N, int w_is_kn, uint4* __restrict__ out,
                                      long long t, float scale, unsigned* __restrict__ sat_counter) {
  const int nst = K >> 5, ntn = (N + 127) >> 7;
  if (t >= (long long)G * ntn * nst * 512) return;
  int r = (int)(t & 511);
  const long long blk = t >> 9;
  const int st = (int)(blk % nst), tile = (int)((blk / nst) % ntn), g = (int)(blk / ((long long)nst * ntn));
  int i, row_lo, kg;
  if (w_is_kn) { row_lo = r & 63; i = (r >> 6) & 1; kg = r >> 7; }
  else { kg = r & 3; row_lo = (r >> 2) & 63; i = r >> 8; }
  const int n = tile * 128 + 64 * i + row_lo, k0 = st * 32 + 8 * kg;
  uint16_t h[8], l[8];
  bool sat = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = 0.f;
    if (n < N) v = w_is_kn ? w[((size_t)g * K + k0 + j) * N + n] : w[((size_t)g * N + n) * K + k0 + j];
    split2h(v * scale, h[j], l[j], sat);
  }
  if (sat && sat_counter) atomicAdd(sat_counter, 1u);
  uint4* o = out + (size_t)blk * 1024 + i * 512 + row_lo * 4 + kg;
  o[0] = make_uint4(BL_PKH(h[0], h[1]), BL_PKH(h[2], h[3]), BL_PKH(h[4], h[5]), BL_PKH(h[6], h[7]));
  o[256] = make_uint4(BL_PKH(l[0], l[1]), BL_PKH(l[2], l[3]), BL_PKH(l[4], l[5]), BL_PKH(l[6], l[7]));
}

