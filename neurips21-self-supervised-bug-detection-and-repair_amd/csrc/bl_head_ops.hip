// Scoring-head kernels and the flat-buffer optimiser (T1).
// All of these work on 10^2..10^4 elements per minibatch: they are launch-latency bound, so each
// is a single small launch with no host round trip; segment work is one wave per segment.
#include <math.h>

#include "bl_common.h"

#define NEG_INF (-__builtin_huge_valf())

// H7  scatter_log_softmax over CSR segments (reference buglab/models/utils.py:15-28)
__global__ __launch_bounds__(256) void seg_logsoftmax_fwd_kernel(const float* __restrict__ x,
                                                                 const int* __restrict__ seg_ptr,
                                                                 const int* __restrict__ seg_items, int nseg, float eps,
                                                                 float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int seg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (seg >= nseg) return;
  const int beg = seg_ptr[seg], end = seg_ptr[seg + 1];
  if (beg == end) return;
  float m = NEG_INF;
  for (int i = beg + lane; i < end; i += 64) m = fmaxf(m, x[seg_items ? seg_items[i] : i]);
  m = bl_wave_max(m);
  float s = 0.f;
  for (int i = beg + lane; i < end; i += 64) s += expf(x[seg_items ? seg_items[i] : i] - m);
  s = bl_wave_sum(s);
  const float lz = logf(s + eps);
  for (int i = beg + lane; i < end; i += 64) {
    const int it = seg_items ? seg_items[i] : i;
    y[it] = (x[it] - m) - lz;
  }
}

__global__ __launch_bounds__(256) void seg_logsoftmax_bwd_kernel(const float* __restrict__ g_y,
                                                                 const float* __restrict__ y,
                                                                 const int* __restrict__ seg_ptr,
                                                                 const int* __restrict__ seg_items, int nseg,
                                                                 float* __restrict__ g_x) {
  const int lane = threadIdx.x & 63;
  const int seg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (seg >= nseg) return;
  const int beg = seg_ptr[seg], end = seg_ptr[seg + 1];
  float s = 0.f;
  for (int i = beg + lane; i < end; i += 64) s += g_y[seg_items ? seg_items[i] : i];
  s = bl_wave_sum(s);
  for (int i = beg + lane; i < end; i += 64) {
    const int it = seg_items ? seg_items[i] : i;
    g_x[it] = g_y[it] - expf(y[it]) * s;
  }
}

// Linear(H -> 1): one wave per row
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ w, const float* __restrict__ b,
                                                         int R, int H, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;
  float s = 0.f;
  for (int h = lane; h < H; h += 64) s += x[(size_t)r * ldx + h] * w[h];
  s = bl_wave_sum(s);
  if (lane == 0) y[r] = s + (b ? b[0] : 0.f);
}

__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ g_y, const float* __restrict__ x,
                                                         int ldx, const float* __restrict__ w, int R, int H,
                                                         float* __restrict__ g_x, int ld_gx, float* __restrict__ g_w,
                                                         float* __restrict__ g_b, unsigned* __restrict__ order_ctr) {
  // thread t owns column h = t (+ blockDim multiples); rows are split over blocks
  const int rows_per_block = (R + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
  bl_ordered_enter(order_ctr, blockIdx.x);  // deterministic mode: blocks add their column sums in block order
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    const float wh = w[h];
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) {
      const float g = g_y[r];
      g_x[(size_t)r * ld_gx + h] = g * wh;
      acc += g * x[(size_t)r * ldx + h];
    }
    if (r1 > r0) unsafeAtomicAdd(&g_w[h], acc);
  }
  if (g_b && threadIdx.x == 0 && r1 > r0) {
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) acc += g_y[r];
    unsafeAtomicAdd(g_b, acc);
  }
  bl_ordered_leave(order_ctr, blockIdx.x);
}

// out[idx[r], :] += src[r, col_off : col_off + width]
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, int ld_src, int col_off,
                                                               int width, const int* __restrict__ idx, long long R,
                                                               float* __restrict__ out, int ld_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * width) return;
  const int r = (int)(t / width), c = (int)(t % width);
  unsafeAtomicAdd(&out[(size_t)idx[r] * ld_out + c], src[(size_t)r * ld_src + col_off + c]);
}

// deterministic mode: thread c owns column c and walks the rows in order
__global__ __launch_bounds__(256) void scatter_add_rows_serial_kernel(const float* __restrict__ src, int ld_src, int col_off, int width,
                                                                      const int* __restrict__ idx, long long R, float* __restrict__ out,
                                                                      int ld_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= width) return;
  for (long long r = 0; r < R; ++r) out[(size_t)idx[r] * ld_out + c] += src[(size_t)r * ld_src + col_off + c];
}

// out[r, :] = x[idx[r], 0:width]  (float4 per thread)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, int ld_x, const int* __restrict__ idx,
                                                          long long R, int width4, float* __restrict__ out, int ld_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * width4) return;
  const long long r = t / width4;
  const int c = (int)(t % width4) * 4;
  *reinterpret_cast<float4*>(out + (size_t)r * ld_out + c) = *reinterpret_cast<const float4*>(x + (size_t)idx[r] * ld_x + c);
}

// ------------------------------------------------------------------------------------------------
// T1  flat-buffer optimiser
// Two passes so that the sum has ONE order of additions whatever the block scheduling: data-parallel replicas clip by
// this number and must get it bit-identical from bit-identical gradients (atomics would let them drift by an ulp a step).
constexpr int SQNORM_MAX_BLOCKS = 2048;
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long long n, float* __restrict__ partials) {
  __shared__ float red[4];
  float s = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = g4[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += g[i] * g[i];
  s = bl_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ partials, int count, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < count; i += 256) s += partials[i];
  s = bl_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long long n,
                                                        const float* __restrict__ sqnorm, float prescale, float clip,
                                                        float lr_over_bc1, float beta1, float beta2, float eps,
                                                        float inv_sqrt_bc2, const float* __restrict__ batch_total) {
  float scale = prescale;
  if (batch_total) {  // data parallel: the summed gradient is weighted by graphs per rank; divide by the global count
    const float bt = batch_total[0];
    if (!(bt > 0.f)) return;  // no rank had a minibatch (the step after the last one of an epoch): leave everything untouched
    scale /= bt;
  }
  if (clip > 0.f) {
    const float total = sqrtf(sqnorm[0]) * scale;  // norm of the gradient the update uses (after the 1/count of data parallel)
    scale *= fminf(1.0f, clip / (total + 1e-6f));  // torch.nn.utils.clip_grad_norm_
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * scale;
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_over_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}

// ================================================================================================
extern "C" int bl_segment_log_softmax_fwd(const float* x, const int32_t* seg_ptr, const int32_t* seg_items,
                                          int32_t nseg, float eps, float* y, void* stream) {
  if (nseg == 0) return BL_OK;
  BL_CHECK_ARG(x && seg_ptr && y, "bl_segment_log_softmax_fwd: null pointer");
  hipLaunchKernelGGL(seg_logsoftmax_fwd_kernel, dim3((nseg + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, seg_ptr,
                     seg_items, nseg, eps, y);
  BL_LAUNCH_CHECK("bl_segment_log_softmax_fwd");
  return BL_OK;
}

extern "C" int bl_segment_log_softmax_bwd(const float* g_y, const float* y, const int32_t* seg_ptr,
                                          const int32_t* seg_items, int32_t nseg, float* g_x, void* stream) {
  if (nseg == 0) return BL_OK;
  BL_CHECK_ARG(g_y && y && seg_ptr && g_x, "bl_segment_log_softmax_bwd: null pointer");
  hipLaunchKernelGGL(seg_logsoftmax_bwd_kernel, dim3((nseg + 3) / 4), dim3(256), 0, (hipStream_t)stream, g_y, y,
                     seg_ptr, seg_items, nseg, g_x);
  BL_LAUNCH_CHECK("bl_segment_log_softmax_bwd");
  return BL_OK;
}

extern "C" int bl_rowdot_fwd(const float* x, int32_t ldx, const float* w, const float* b, int32_t R, int32_t H,
                             float* y, void* stream) {
  if (R == 0) return BL_OK;
  BL_CHECK_ARG(x && w && y && H > 0, "bl_rowdot_fwd: null pointer");
  hipLaunchKernelGGL(rowdot_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, w, b, R, H, y);
  BL_LAUNCH_CHECK("bl_rowdot_fwd");
  return BL_OK;
}

extern "C" int bl_rowdot_bwd(const float* g_y, const float* x, int32_t ldx, const float* w, int32_t R, int32_t H,
                             float* g_x, int32_t ld_gx, float* g_w, float* g_b, void* stream) {
  if (R == 0) return BL_OK;
  BL_CHECK_ARG(g_y && x && w && g_x && g_w && H > 0, "bl_rowdot_bwd: null pointer");
  const int blocks = min(256, (R + 15) / 16);
  hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g_y, x, ldx, w, R, H, g_x,
                     ld_gx, g_w, g_b, bl_order_counters(1, stream));
  BL_LAUNCH_CHECK("bl_rowdot_bwd");
  return BL_OK;
}

extern "C" int bl_scatter_add_rows(const float* src, int32_t ld_src, int32_t col_off, int32_t width,
                                   const int32_t* idx, int32_t R, float* out, int32_t ld_out, void* stream) {
  if (R == 0) return BL_OK;
  BL_CHECK_ARG(src && idx && out && width > 0, "bl_scatter_add_rows: null pointer");
  const long long total = (long long)R * width;
  if (bl_get_deterministic())
    hipLaunchKernelGGL(scatter_add_rows_serial_kernel, dim3((width + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                       col_off, width, idx, (long long)R, out, ld_out);
  else
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       src, ld_src, col_off, width, idx, (long long)R, out, ld_out);
  BL_LAUNCH_CHECK("bl_scatter_add_rows");
  return BL_OK;
}

extern "C" int bl_gather_rows(const float* x, int32_t ld_x, const int32_t* idx, int32_t R, int32_t width, float* out,
                              int32_t ld_out, void* stream) {
  if (R == 0) return BL_OK;
  BL_CHECK_ARG(x && idx && out && bl_aligned16(x) && bl_aligned16(out), "bl_gather_rows: null or misaligned pointer");
  BL_CHECK_ARG(width > 0 && width % 4 == 0 && ld_x % 4 == 0 && ld_out % 4 == 0, "bl_gather_rows: width / ld must be multiples of 4");
  const long long total = (long long)R * (width / 4);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld_x, idx,
                     (long long)R, width / 4, out, ld_out);
  BL_LAUNCH_CHECK("bl_gather_rows");
  return BL_OK;
}

extern "C" int64_t bl_sqnorm_scratch_bytes(void) { return (int64_t)SQNORM_MAX_BLOCKS * sizeof(float); }

extern "C" int bl_sqnorm(const float* g, int64_t n, float* out, float* scratch, void* stream) {
  BL_CHECK_ARG(g && out && scratch && bl_aligned16(g), "bl_sqnorm: null or misaligned pointer");
  const int blocks = n == 0 ? 0 : (int)fmin((double)SQNORM_MAX_BLOCKS, (double)((n / 4 + 255) / 256 + 1));
  if (blocks) {
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, (long long)n, scratch);
    BL_LAUNCH_CHECK("bl_sqnorm");
  }
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, blocks, out);
  BL_LAUNCH_CHECK("bl_sqnorm");
  return BL_OK;
}

extern "C" int bl_adam_clip_step(float* param, const float* grad, float* m, float* v, int64_t n,
                                 const float* grad_sqnorm, float grad_prescale, float clip_norm, float lr, float beta1,
                                 float beta2, float eps, int32_t step, void* stream) {
  if (n == 0) return BL_OK;
  BL_CHECK_ARG(param && grad && m && v, "bl_adam_clip_step: null pointer");
  BL_CHECK_ARG(clip_norm <= 0.f || grad_sqnorm, "bl_adam_clip_step: clipping needs grad_sqnorm");
  BL_CHECK_ARG(step >= 1, "bl_adam_clip_step: step is 1-based");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int blocks = (int)fmin(2048.0, (double)((n + 255) / 256));
  hipLaunchKernelGGL(adam_clip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, (long long)n,
                     grad_sqnorm, grad_prescale, clip_norm, (float)((double)lr / bc1), beta1, beta2, eps,
                     (float)(1.0 / sqrt(bc2)), (const float*)nullptr);
  BL_LAUNCH_CHECK("bl_adam_clip_step");
  return BL_OK;
}

extern "C" int bl_adam_clip_step_dp(float* param, const float* grad, float* m, float* v, int64_t n, const float* grad_sqnorm,
                                    const float* batch_total, float clip_norm, float lr, float beta1, float beta2, float eps,
                                    int32_t step, void* stream) {
  if (n == 0) return BL_OK;
  BL_CHECK_ARG(param && grad && m && v && batch_total && (clip_norm <= 0.f || grad_sqnorm) && step >= 1, "bl_adam_clip_step_dp: null pointer or step < 1");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int blocks = (int)fmin(2048.0, (double)((n + 255) / 256));
  hipLaunchKernelGGL(adam_clip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, (long long)n,
                     grad_sqnorm, 1.0f, clip_norm, (float)((double)lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)),
                     batch_total);
  BL_LAUNCH_CHECK("bl_adam_clip_step_dp");
  return BL_OK;
}
