// Work-item lookup and routing-mask helpers shared by the bf16x6 row GEMMs (bl_gemm_x6.hip, bl_gemm_x6w.hip).
#pragma once
#include "bl_common.h"

__device__ __forceinline__ bool x6_find_piece(const int* __restrict__ group_ptr, int G, int M, int piece, int t,
                                              int& g, int& row0, int& nrows) {
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

// routing byte (8 channels, bit c = keep channel c) -> AND-masks for the 8 packed bf16 of a plane
// Work-item of a workgroup.  Workgroups are dealt round-robin to the 8 XCDs in linear-id order; with
// xcd_remap every XCD gets one CONTIGUOUS range of (tile, y) work items, so its private 4 MB L2 sees
// consecutive tiles: they share the edge type's weights, the target-sorted node rows, and -- when the
// output has several column tiles -- the whole row tile.  Measured at c2 shapes: weight-gradient GEMM
// 0.314 -> 0.287 ms (H=128), 1.34 -> 1.00 ms (concat layer), input-gradient GEMM 1.36 -> 1.16 ms.
__device__ __forceinline__ bool x6_locate(const int* __restrict__ group_ptr, int G, int M, int piece, int xcd_remap,
                                          int& tile_y, int& g, int& row0, int& nrows) {
  int tx = blockIdx.x;
  tile_y = blockIdx.y;
  if (xcd_remap) {
    // XCD c owns the work items [c q + min(c, r), ...): a bijection of [0, total) for any total
    const int lin = blockIdx.x + blockIdx.y * gridDim.x, total = gridDim.x * gridDim.y;
    const int q = total >> 3, r = total & 7, c = lin & 7;
    const int v = c * q + min(c, r) + (lin >> 3);
    tx = v / gridDim.y;
    tile_y = v - tx * gridDim.y;
  }
  return x6_find_piece(group_ptr, G, M, piece, tx, g, row0, nrows);
}

__device__ __forceinline__ uint4 keep_from_bits(uint32_t b) {
  uint4 k;
  k.x = (__builtin_amdgcn_sbfe(b, 0, 1) & 0x0000FFFFu) | (__builtin_amdgcn_sbfe(b, 1, 1) & 0xFFFF0000u);
  k.y = (__builtin_amdgcn_sbfe(b, 2, 1) & 0x0000FFFFu) | (__builtin_amdgcn_sbfe(b, 3, 1) & 0xFFFF0000u);
  k.z = (__builtin_amdgcn_sbfe(b, 4, 1) & 0x0000FFFFu) | (__builtin_amdgcn_sbfe(b, 5, 1) & 0xFFFF0000u);
  k.w = (__builtin_amdgcn_sbfe(b, 6, 1) & 0x0000FFFFu) | (__builtin_amdgcn_sbfe(b, 7, 1) & 0xFFFF0000u);
  return k;
}

