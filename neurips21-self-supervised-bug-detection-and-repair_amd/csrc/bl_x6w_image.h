// Weight image of the wide bf16x6 row GEMM (bl_gemm_x6w.hip); shared with the one-launch weight packer of bl_gemm_x6.hip.
#pragma once
#include "bl_common.h"

#define WBM 128
#define WBN 256
#define W_STAGE_UINT4 ((WBM + WBN) * 12)    // uint4 per stage buffer: (128 + 256) rows x 3 planes x 4 k-groups
#define W_STAGE_BYTES (W_STAGE_UINT4 * 16)  // 73 728
#define W_A_PLANE_BYTES (WBM * 64)          // 8 192
#define W_B_OFF_BYTES (WBM * 12 * 16)       // 24 576: B image behind the A image
#define W_B_PLANE_BYTES (WBN * 64)          // 16 384
#define W_BLK (WBN * 12)                    // uint4 per (group, column tile, stage) weight block

// ---- weight image -----------------------------------------------------------------------------------
// slot (plane p, column n, k-group kg) at uint4 index (p * 256 + n) * 4 + (kg ^ ((n >> 2) & 3)) of the 48 KB block of
// (group, 256-column tile, 32-k stage); columns past N are zero.  w is [G][K][N] (w_is_kn = 1) or [G][N][K].
__device__ __forceinline__ void pack_weights_wide_thread(const float* __restrict__ w, int G, int K, int N, int w_is_kn,
                                                         uint4* __restrict__ out, long long t) {
  const int nst = K >> 5, ntn = (N + WBN - 1) / WBN;
  if (t >= (long long)G * ntn * nst * (WBN * 4)) return;
  const int r = (int)(t % (WBN * 4));
  const long long blk = t / (WBN * 4);
  const int st = (int)(blk % nst), tile = (int)((blk / nst) % ntn), g = (int)(blk / ((long long)nst * ntn));
  int n_lo, kg;  // consecutive threads -> consecutive n when the source is [K][N] (coalesced), consecutive k-groups otherwise
  if (w_is_kn) { n_lo = r % WBN; kg = r / WBN; }
  else { kg = r & 3; n_lo = r >> 2; }
  const int n = tile * WBN + n_lo, k0 = st * 32 + 8 * kg;
  uint16_t h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = 0.f;
    if (n < N) v = w_is_kn ? w[((size_t)g * K + k0 + j) * N + n] : w[((size_t)g * N + n) * K + k0 + j];
    split3(v, h[j], m[j], l[j]);
  }
#define PKW(a_, b_) ((uint32_t)(a_) | ((uint32_t)(b_) << 16))
  uint4* o = out + (size_t)blk * W_BLK + n_lo * 4 + (kg ^ ((n_lo >> 2) & 3));
  o[0] = make_uint4(PKW(h[0], h[1]), PKW(h[2], h[3]), PKW(h[4], h[5]), PKW(h[6], h[7]));
  o[WBN * 4] = make_uint4(PKW(m[0], m[1]), PKW(m[2], m[3]), PKW(m[4], m[5]), PKW(m[6], m[7]));
  o[WBN * 8] = make_uint4(PKW(l[0], l[1]), PKW(l[2], l[3]), PKW(l[4], l[5]), PKW(l[6], l[7]));
#undef PKW
}

