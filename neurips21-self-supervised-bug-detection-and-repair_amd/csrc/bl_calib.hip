// Box calibration for bench.py (measurement support, not on the training path): two fixed kernels whose rates say what THIS
// chip sustains today -- boxes of the pool differ by several per cent in the clock they hold at the 1 400 W package limit.
//   bl_calib_mfma_bf16   dense v_mfma_f32_32x32x16_bf16 from registers, nothing else: the matrix pipes' rate at the clock the
//                        package settles on under a pure matrix load (paper peak 2 500 TF/s at 2.4 GHz)
//   bl_calib_stream_copy a 16 B / lane grid-stride copy: HBM read + write bandwidth (paper peak 8 TB/s)
// The caller times them with HIP events on `stream` (hip_ops.box_calibration).
#include "bl_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Every wave: `iters` rounds of CHAINS independent accumulator chains (a chain's MFMA depends on its previous one; four chains
// keep the pipe full), operands in registers.  FLOP per wave = iters * CHAINS * 2 * 32 * 32 * 16.
#define CALIB_CHAINS 4
__global__ __launch_bounds__(256) void calib_mfma_kernel(int iters, float* __restrict__ sink) {
  bf16x8 a, b;
  // operands with busy mantissas (what a real GEMM's data looks like to the multipliers: a loop over constants draws less power
  // and runs at a clock the training step never sees), magnitudes ~2^-6 so that the accumulators stay finite
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t ha = bl_lowbias32((uint32_t)(threadIdx.x * 8 + i) + 0x9E3779B9u * (blockIdx.x + 1));
    const uint32_t hb = bl_lowbias32(ha ^ 0x85EBCA6Bu);
    a[i] = (__bf16)(((float)(ha >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.03125f);
    b[i] = (__bf16)(((float)(hb >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.03125f);
  }
  f32x16 acc[CALIB_CHAINS];
#pragma unroll
  for (int c = 0; c < CALIB_CHAINS; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CALIB_CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CALIB_CHAINS; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 12345.678f) sink[0] = s;  // never true: keeps the chains alive without a store on the timed path
}

__global__ __launch_bounds__(256) void calib_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n16) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

// -> *flop = the launch's floating-point operations (workgroups x 4 waves x iters x CHAINS x 32 768)
extern "C" int bl_calib_mfma_bf16(int32_t iters, int32_t workgroups, float* sink, double* flop, void* stream) {
  BL_CHECK_ARG(iters > 0 && workgroups > 0 && sink && flop, "bl_calib_mfma_bf16: bad argument");
  hipLaunchKernelGGL(calib_mfma_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, sink);
  BL_LAUNCH_CHECK("bl_calib_mfma_bf16");
  *flop = (double)workgroups * 4.0 * (double)iters * CALIB_CHAINS * 2.0 * 32.0 * 32.0 * 16.0;
  return BL_OK;
}

// copies nbytes (a multiple of 16, both pointers 16-byte aligned) from src to dst: 2 x nbytes of HBM traffic
extern "C" int bl_calib_stream_copy(const void* src, void* dst, int64_t nbytes, void* stream) {
  BL_CHECK_ARG(src && dst && nbytes > 0 && nbytes % 16 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0),
               "bl_calib_stream_copy: pointers and size must be 16-byte aligned");
  const int grid = bl_num_cus() * 16;
  hipLaunchKernelGGL(calib_copy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst,
                     (long long)(nbytes / 16));
  BL_LAUNCH_CHECK("bl_calib_stream_copy");
  return BL_OK;
}
