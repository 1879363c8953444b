// Loss assembly of the detector / repair model in ONE kernel per direction (SURVEY.md rows H2 + H6 + H7 + H8).
//
// Replaces, on the training path, the chain of small framework ops between the scorers' logits and the scalar loss:
//   reference buglab/models/layers/localizationmodule.py:63-124  NO_BUG logit (constant 1.0) appended per graph, segmented
//             log-softmax, pick of the correct location (torch.where), clamp at log 0.995, abstain term, (weighted) mean,
//             arg-max accuracy counters;
//   reference buglab/models/gnn.py:295-311  one log-softmax of the three scorers' logits over the location groups
//             (utils.py:15-28: x - max - log(sum exp(x - max) + 1e-12)) and the "is this candidate its group's arg-max" flags;
//   reference buglab/models/layers/fixermodules.py:41-53, 86-98, 134-147  -logprob[target] and the fixer accuracy counters;
//   reference buglab/models/gnn.py:221-251  loss = localization loss + w_buggy * (sum of the repair losses) / B.
// Sizes are tiny (C + R ~ 10^3-10^4 values per minibatch): one 1024-thread workgroup walks everything, one wave per
// graph / location group, sums in a fixed order (per-wave partials, then wave order): the loss is bit-reproducible.
#include "bl_common.h"

namespace {
constexpr int LOSS_THREADS = 1024;
constexpr int LOSS_WAVES = LOSS_THREADS / 64;
constexpr float LOG_0995 = -0.0050125418235442820f;  // log(0.995), localizationmodule.py:93
constexpr float LSM_EPS = 1e-12f;                     // utils.py:27
#define NEG_INF_F (-__builtin_huge_valf())

enum { ST_B = 0, ST_LOC_OK, ST_NOBUG, ST_NOBUG_OK, ST_LOC_NLL, ST_TEXT_OK, ST_TEXT_N, ST_VAR_OK, ST_VAR_N, ST_SWAP_OK, ST_SWAP_N,
       ST_LOSS, ST_REPAIR_LOSS, ST_HAS_BUG, ST_BATCHES, ST_UNUSED, ST_COUNT };

__device__ __forceinline__ float loc_value(const bl_bug_loss_t& d, int item) { return item < d.C ? d.loc_scores[item] : 1.0f; }

// fixed-order sum of per-wave partials (slot `k` of every wave), by thread 0
__device__ __forceinline__ float sum_partials(float (*part)[16], int k) {
  float s = 0.f;
  for (int w = 0; w < LOSS_WAVES; ++w) s += part[w][k];
  return s;
}

// The index arrays of the descriptor arrive with the minibatch's H2D copy: nothing has touched them on the device, and the kernels below
// walk them as a chain (ptr -> items -> values -> ...), i.e. one cold miss after the other, from one workgroup.  Every thread therefore
// first requests its share of ALL of them at once -- independent loads, one round trip -- so that the chain runs on L2 hits
// (bug_loss_fwd 75 -> 58 us at 64 graphs, nothing at 15 graphs where both kernels take 22 us, nothing for the backward kernel, whose time
// is its four per-graph iterations per wave: profiles/r06zzq_loss_touch.log).
__device__ __forceinline__ void warm_index_arrays(const bl_bug_loss_t& d, int tid) {
  int sink = 0;
#define BL_TOUCH(p_, n_) \
  if (p_) for (int i_ = tid; i_ < (n_); i_ += LOSS_THREADS) sink ^= (int)(p_)[i_];
  BL_TOUCH(d.loc_group_ptr, d.B + 1)
  BL_TOUCH(d.loc_group_items, d.C + d.B)
  BL_TOUCH(d.candidate_ptr, d.B + 1)
  BL_TOUCH(d.has_bug, d.B)
  BL_TOUCH(d.correct_candidate_idxs, d.B)
  BL_TOUCH(d.repair_group_ptr, d.G + 1)
  BL_TOUCH(d.repair_group_items, d.Rt + d.Rv + d.Rs)
  BL_TOUCH(d.logit_group[0], d.Rt)
  BL_TOUCH(d.logit_group[1], d.Rv)
  BL_TOUCH(d.logit_group[2], d.Rs)
  BL_TOUCH(d.target[0], d.ntarget[0])
  BL_TOUCH(d.target[1], d.ntarget[1])
  BL_TOUCH(d.target[2], d.ntarget[2])
#undef BL_TOUCH
  asm volatile("" ::"v"(sink));  // (the loads have to happen; their values do not matter)
}

__global__ __launch_bounds__(LOSS_THREADS) void bug_loss_fwd_kernel(bl_bug_loss_t d, float* __restrict__ loc_lp, float* __restrict__ rep_lp,
                                                                    float* __restrict__ gmax, float* __restrict__ loss,
                                                                    float* __restrict__ stats) {
  __shared__ float part[LOSS_WAVES][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float a_lp = 0.f, a_lpw = 0.f, a_w = 0.f, a_ok = 0.f, a_nb = 0.f, a_nbok = 0.f, a_hb = 0.f;  // lane 0 of each wave
  warm_index_arrays(d, tid);
  // ---- localization: one wave per graph ------------------------------------------------------------------------------
  for (int b = wave; b < d.B; b += LOSS_WAVES) {
    const int beg = d.loc_group_ptr[b], end = d.loc_group_ptr[b + 1];
    float m = NEG_INF_F, lz;
    if (end - beg <= 64) {
      // the usual case (tens of candidate locations per graph): every lane keeps its one item and value -- one dependent
      // round of loads instead of three (same arithmetic in the same order: a lane of the loops below has one term too)
      const int it = beg + lane < end ? d.loc_group_items[beg + lane] : -1;
      const float v = it >= 0 ? loc_value(d, it) : NEG_INF_F;
      m = bl_wave_max(v);
      const float s = bl_wave_sum(it >= 0 ? expf(v - m) : 0.f);
      lz = logf(s + LSM_EPS);
      if (it >= 0) loc_lp[it] = (v - m) - lz;
    } else {
      for (int i = beg + lane; i < end; i += 64) m = fmaxf(m, loc_value(d, d.loc_group_items[i]));
      m = bl_wave_max(m);
      float s = 0.f;
      for (int i = beg + lane; i < end; i += 64) s += expf(loc_value(d, d.loc_group_items[i]) - m);
      s = bl_wave_sum(s);
      lz = logf(s + LSM_EPS);
      for (int i = beg + lane; i < end; i += 64) {
        const int it = d.loc_group_items[i];
        loc_lp[it] = (loc_value(d, it) - m) - lz;
      }
    }
    // arg-max over the graph's candidate rows (contiguous rows candidate_ptr[b] .. candidate_ptr[b+1]); ties -> first row
    const int c0 = d.candidate_ptr[b], c1 = d.candidate_ptr[b + 1];
    float best = NEG_INF_F;
    int barg = 0x7fffffff;
    for (int r = c0 + lane; r < c1; r += 64) {
      const float v = (d.loc_scores[r] - m) - lz;
      if (v > best) { best = v; barg = r; }
    }
    const float wbest = bl_wave_max(best);
    int cand = (best == wbest && barg != 0x7fffffff) ? barg : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    if (lane == 0) {
      const bool hb = d.has_bug[b] != 0;
      const float no_bug_lp = (1.0f - m) - lz;
      const int correct = hb ? d.correct_candidate_idxs[b] : d.C + b;
      const float lp_c = correct < d.C ? (d.loc_scores[correct] - m) - lz : no_bug_lp;
      float lp = fminf(lp_c, LOG_0995);
      if (d.abstain_weight > 0.f && hb) lp += d.abstain_weight * no_bug_lp;
      const int pred = (c1 > c0 && wbest >= no_bug_lp) ? cand : d.C + b;
      const bool ok = pred == correct;
      const float w = hb ? d.w_buggy : 1.0f;
      a_lp += lp; a_lpw += lp * w; a_w += w;
      a_ok += ok ? 1.f : 0.f; a_nb += hb ? 0.f : 1.f; a_nbok += (!hb && ok) ? 1.f : 0.f; a_hb += hb ? 1.f : 0.f;
    }
  }
  // ---- repair: one wave per location group ---------------------------------------------------------------------------
  for (int g = wave; g < d.G; g += LOSS_WAVES) {
    const int beg = d.repair_group_ptr[g], end = d.repair_group_ptr[g + 1];
    if (beg == end) { if (lane == 0) gmax[g] = NEG_INF_F; continue; }
    float m = NEG_INF_F;
    if (end - beg <= 64) {  // (one item per lane, as above)
      const int it = beg + lane < end ? d.repair_group_items[beg + lane] : -1;
      const float v = it >= 0 ? d.repair_logits[it] : NEG_INF_F;
      m = bl_wave_max(v);
      const float s = bl_wave_sum(it >= 0 ? expf(v - m) : 0.f);
      const float lz = logf(s + LSM_EPS);
      if (it >= 0) rep_lp[it] = (v - m) - lz;
    } else {
      for (int i = beg + lane; i < end; i += 64) m = fmaxf(m, d.repair_logits[d.repair_group_items[i]]);
      m = bl_wave_max(m);
      float s = 0.f;
      for (int i = beg + lane; i < end; i += 64) s += expf(d.repair_logits[d.repair_group_items[i]] - m);
      s = bl_wave_sum(s);
      const float lz = logf(s + LSM_EPS);
      for (int i = beg + lane; i < end; i += 64) {
        const int it = d.repair_group_items[i];
        rep_lp[it] = (d.repair_logits[it] - m) - lz;
      }
    }
    if (lane == 0) gmax[g] = m;
  }
  __syncthreads();  // rep_lp / gmax written by other waves of this workgroup are read below
  // ---- -logprob of the correct rewrites + "was it the arg-max of its group" counters ------------------------------------
  float r_nll[3] = {0.f, 0.f, 0.f}, r_ok[3] = {0.f, 0.f, 0.f};
  int off = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    for (int i = tid; i < d.ntarget[k]; i += LOSS_THREADS) {
      const int idx = d.target[k][i];
      const int it = off + idx;
      r_nll[k] -= rep_lp[it];
      r_ok[k] += d.repair_logits[it] == gmax[d.logit_group[k][idx]] ? 1.f : 0.f;
    }
    off += k == 0 ? d.Rt : (k == 1 ? d.Rv : d.Rs);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { r_nll[k] = bl_wave_sum(r_nll[k]); r_ok[k] = bl_wave_sum(r_ok[k]); }
  if (lane == 0) {
    float* p = part[wave];
    p[0] = a_lp; p[1] = a_lpw; p[2] = a_w; p[3] = a_ok; p[4] = a_nb; p[5] = a_nbok; p[6] = a_hb;
    p[7] = r_nll[0]; p[8] = r_nll[1]; p[9] = r_nll[2]; p[10] = r_ok[0]; p[11] = r_ok[1]; p[12] = r_ok[2];
  }
  __syncthreads();
  if (tid == 0) {
    const float s_lp = sum_partials(part, 0), s_lpw = sum_partials(part, 1), s_w = sum_partials(part, 2);
    const float loc_loss = d.w_buggy == 1.0f ? -s_lp / (float)d.B : -s_lpw / s_w;  // localizationmodule.py:116-124
    const float text = sum_partials(part, 7), var = sum_partials(part, 8), swap = sum_partials(part, 9);
    const float repair = ((text + var) + swap) * d.w_buggy;                        // gnn.py:240-242
    const float total = loc_loss + repair / (float)d.B;                            // gnn.py:251
    loss[0] = total;
    stats[ST_B] = (float)d.B; stats[ST_LOC_OK] = sum_partials(part, 3); stats[ST_NOBUG] = sum_partials(part, 4);
    stats[ST_NOBUG_OK] = sum_partials(part, 5); stats[ST_LOC_NLL] = -s_lp;
    stats[ST_TEXT_OK] = sum_partials(part, 10); stats[ST_TEXT_N] = (float)d.ntarget[0];
    stats[ST_VAR_OK] = sum_partials(part, 11); stats[ST_VAR_N] = (float)d.ntarget[1];
    stats[ST_SWAP_OK] = sum_partials(part, 12); stats[ST_SWAP_N] = (float)d.ntarget[2];
    stats[ST_LOSS] = total; stats[ST_REPAIR_LOSS] = repair; stats[ST_HAS_BUG] = sum_partials(part, 6); stats[ST_BATCHES] = 1.f;
    stats[ST_UNUSED] = 0.f;
  }
}

__global__ __launch_bounds__(LOSS_THREADS) void bug_loss_bwd_kernel(bl_bug_loss_t d, const float* __restrict__ loc_lp,
                                                                    const float* __restrict__ rep_lp, const float* __restrict__ g_loss,
                                                                    float* __restrict__ gy_loc, float* __restrict__ gy_rep,
                                                                    float* __restrict__ g_scores, float* __restrict__ g_logits) {
  __shared__ float part[LOSS_WAVES][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = d.Rt + d.Rv + d.Rs;
  warm_index_arrays(d, tid);
  const float gl = g_loss[0];
  // sum of the per-sample weights (the denominator of the weighted mean) in the forward's order
  float a_w = 0.f;
  for (int b = wave; b < d.B; b += LOSS_WAVES)
    if (lane == 0) a_w += d.has_bug[b] ? d.w_buggy : 1.0f;
  if (lane == 0) part[wave][0] = a_w;
  for (int i = tid; i < d.C + d.B; i += LOSS_THREADS) gy_loc[i] = 0.f;
  for (int i = tid; i < R; i += LOSS_THREADS) gy_rep[i] = 0.f;
  __syncthreads();
  const float s_w = d.w_buggy == 1.0f ? (float)d.B : sum_partials(part, 0);
  // d loss / d logprob: one entry per graph (+ the NO_BUG entry under an abstain weight), one per correct rewrite
  for (int b = tid; b < d.B; b += LOSS_THREADS) {
    const bool hb = d.has_bug[b] != 0;
    const int correct = hb ? d.correct_candidate_idxs[b] : d.C + b;
    const float coef = -gl * (hb ? d.w_buggy : 1.0f) / s_w;
    if (loc_lp[correct] <= LOG_0995) gy_loc[correct] = coef;  // clamp(max=...) passes the gradient where x <= max
    if (d.abstain_weight > 0.f && hb) gy_loc[d.C + b] = coef * d.abstain_weight;
  }
  int off = 0;
  const float rc = -gl * d.w_buggy / (float)d.B;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    for (int i = tid; i < d.ntarget[k]; i += LOSS_THREADS) atomicAdd(&gy_rep[off + d.target[k][i]], rc);
    off += k == 0 ? d.Rt : (k == 1 ? d.Rv : d.Rs);
  }
  __syncthreads();
  // log-softmax backward per segment: g_x = g_y - exp(y) * sum_seg g_y   (the NO_BUG logit is a constant: no output)
  for (int b = wave; b < d.B; b += LOSS_WAVES) {
    const int beg = d.loc_group_ptr[b], end = d.loc_group_ptr[b + 1];
    if (end - beg <= 64) {  // one item per lane: its index, gradient and log-probability are loaded once
      const int it = beg + lane < end ? d.loc_group_items[beg + lane] : -1;
      const float gy = it >= 0 ? gy_loc[it] : 0.f;
      const float lp = (it >= 0 && it < d.C) ? loc_lp[it] : 0.f;
      const float s = bl_wave_sum(gy);
      if (it >= 0 && it < d.C) g_scores[it] = gy - expf(lp) * s;
      continue;
    }
    float s = 0.f;
    for (int i = beg + lane; i < end; i += 64) s += gy_loc[d.loc_group_items[i]];
    s = bl_wave_sum(s);
    for (int i = beg + lane; i < end; i += 64) {
      const int it = d.loc_group_items[i];
      if (it < d.C) g_scores[it] = gy_loc[it] - expf(loc_lp[it]) * s;
    }
  }
  for (int g = wave; g < d.G; g += LOSS_WAVES) {
    const int beg = d.repair_group_ptr[g], end = d.repair_group_ptr[g + 1];
    if (end - beg <= 64) {
      const int it = beg + lane < end ? d.repair_group_items[beg + lane] : -1;
      const float gy = it >= 0 ? gy_rep[it] : 0.f;
      const float lp = it >= 0 ? rep_lp[it] : 0.f;
      const float s = bl_wave_sum(gy);
      if (it >= 0) g_logits[it] = gy - expf(lp) * s;
      continue;
    }
    float s = 0.f;
    for (int i = beg + lane; i < end; i += 64) s += gy_rep[d.repair_group_items[i]];
    s = bl_wave_sum(s);
    for (int i = beg + lane; i < end; i += 64) {
      const int it = d.repair_group_items[i];
      g_logits[it] = gy_rep[it] - expf(rep_lp[it]) * s;
    }
  }
}

int check_desc(const bl_bug_loss_t* d, const char* who) {
  BL_CHECK_ARG(d && d->B > 0 && d->C >= 0 && d->Rt >= 0 && d->Rv >= 0 && d->Rs >= 0 && d->G >= 0, "%s: bad sizes", who);
  BL_CHECK_ARG((d->C == 0 || d->loc_scores) && d->loc_group_ptr && d->loc_group_items && d->candidate_ptr && d->has_bug &&
                   d->correct_candidate_idxs,
               "%s: null localization input", who);
  const int R = d->Rt + d->Rv + d->Rs;
  BL_CHECK_ARG(R == 0 || (d->repair_logits && d->repair_group_ptr && d->repair_group_items), "%s: null repair input", who);
  for (int k = 0; k < 3; ++k)
    BL_CHECK_ARG(d->ntarget[k] >= 0 && (d->ntarget[k] == 0 || (d->target[k] && d->logit_group[k])), "%s: null target list %d", who, k);
  return BL_OK;
}
}  // namespace

extern "C" int bl_bug_loss_fwd(const bl_bug_loss_t* d, float* loc_logprobs, float* repair_logprobs, float* group_max, float* loss,
                               float* stats, void* stream) {
  int rc = check_desc(d, "bl_bug_loss_fwd");
  if (rc != BL_OK) return rc;
  BL_CHECK_ARG(loc_logprobs && loss && stats && (d->Rt + d->Rv + d->Rs == 0 || repair_logprobs) && (d->G == 0 || group_max),
               "bl_bug_loss_fwd: null output");
  hipLaunchKernelGGL(bug_loss_fwd_kernel, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, *d, loc_logprobs, repair_logprobs, group_max,
                     loss, stats);
  BL_LAUNCH_CHECK("bl_bug_loss_fwd");
  return BL_OK;
}

extern "C" int bl_bug_loss_bwd(const bl_bug_loss_t* d, const float* loc_logprobs, const float* repair_logprobs, const float* g_loss,
                               float* scratch, float* g_loc_scores, float* g_repair_logits, void* stream) {
  int rc = check_desc(d, "bl_bug_loss_bwd");
  if (rc != BL_OK) return rc;
  const int R = d->Rt + d->Rv + d->Rs;
  BL_CHECK_ARG(loc_logprobs && g_loss && scratch && (d->C == 0 || g_loc_scores) && (R == 0 || (repair_logprobs && g_repair_logits)),
               "bl_bug_loss_bwd: null buffer");
  hipLaunchKernelGGL(bug_loss_bwd_kernel, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, *d, loc_logprobs, repair_logprobs, g_loss,
                     scratch, scratch + d->C + d->B, g_loc_scores, g_repair_logits);
  BL_LAUNCH_CHECK("bl_bug_loss_bwd");
  return BL_OK;
}
