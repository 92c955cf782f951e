// BatchNorm2d (+ LeakyReLU) for NHWC fp32 activations, train-mode and eval-mode, forward and backward.
// Reference semantics: nn.BatchNorm2d(eps 1e-5, momentum 0.1, affine, unbiased running_var) followed by
// nn.LeakyReLU(0.01) -- models/unet.py:189-193, models/aux_path_memory.py:25-26.
//
// A launch covers `groups` independent batches laid back to back along the pixel axis (the weak and the
// strong view of the siamese step): statistics are taken per group, exactly as two separate module calls
// would, and the running statistics are updated once per group in order.
//
// All kernels are HBM-bound streaming kernels: 16-B per lane accesses along the channel axis,
// per-thread partial sums, LDS + fixed-order double accumulation for the per-channel reductions
// (deterministic: no float atomics).
#include "pp_common.h"
#include <stdlib.h>

PP_NS_BEGIN
#define NORM_THREADS 256

struct ColPlan { int c4, rows, nblk, chunk; };

// rows of pixels handled per block iteration, number of blocks per group and pixels per block
static ColPlan col_plan(int C, int Ppg, int groups) {
  ColPlan p;
  p.c4 = C / 4;
  p.rows = NORM_THREADS / p.c4;
  if (p.rows < 1) p.rows = 1;
  constexpr int blocks = 512;
  int target = blocks / groups;                     // 2 blocks per CU; swept again in round 6 (BatchNorm family alone on the chip: 512 -> 5.60-5.72 ms,
                                                    // 1024 -> 5.69-5.84, 2048 -> 6.09-6.21, 4096 -> 6.40-6.43; profiles/r06_experiments/bn_blocks_sweep.log)
  if (target < 1) target = 1;
  p.chunk = pp_cdiv(Ppg, target);
  const int min_chunk = p.rows * 8;
  if (p.chunk < min_chunk) p.chunk = min_chunk;
  p.chunk = pp_cdiv(p.chunk, p.rows) * p.rows;
  p.nblk = pp_cdiv(Ppg, p.chunk);
  return p;
}

// ---- per-channel sum / sum of squares: partial[g][blk][2][C] (double) ----
__global__ __launch_bounds__(NORM_THREADS) void bn_stats_partial_kernel(const act_t* __restrict__ z, int ld, int C,
                                                                        int Ppg, int chunk, int rows,
                                                                        double* __restrict__ partial) {
  __shared__ float sh[2 * NORM_THREADS * 4];
  const int c4n = C >> 2;
  const int tid = threadIdx.x;
  const int cq = tid % c4n, row = tid / c4n;
  const bool active = row < rows;
  const int g = blockIdx.y, blk = blockIdx.x;
  const int p_lo = blk * chunk;
  int p_hi = p_lo + chunk;
  if (p_hi > Ppg) p_hi = Ppg;
  float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
  if (active) {
    const act_t* base = z + (size_t)g * Ppg * ld + cq * 4;
    int p = p_lo + row;
#define PP_ACC(v)                                                        \
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;                      \
    q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
    for (; p + 3 * rows < p_hi; p += 4 * rows) {        // four independent 16-B loads in flight per lane
      const float4 v0 = act_ld4f(base + (size_t)p * ld);
      const float4 v1 = act_ld4f(base + (size_t)(p + rows) * ld);
      const float4 v2 = act_ld4f(base + (size_t)(p + 2 * rows) * ld);
      const float4 v3 = act_ld4f(base + (size_t)(p + 3 * rows) * ld);
      PP_ACC(v0) PP_ACC(v1) PP_ACC(v2) PP_ACC(v3)
    }
    for (; p < p_hi; p += rows) {
      const float4 v = act_ld4f(base + (size_t)p * ld);
      PP_ACC(v)
    }
#undef PP_ACC
    float* d = sh + (row * c4n + cq) * 8;
    d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w; d[4] = q.x; d[5] = q.y; d[6] = q.z; d[7] = q.w;
  }
  __syncthreads();
  for (int c = tid; c < C; c += NORM_THREADS) {
    double ds = 0.0, dq = 0.0;
    const int cq2 = c >> 2, e = c & 3;
    for (int r = 0; r < rows; ++r) {
      ds += (double)sh[(r * c4n + cq2) * 8 + e];
      dq += (double)sh[(r * c4n + cq2) * 8 + 4 + e];
    }
    double* o = partial + ((size_t)(g * gridDim.x + blk) * 2) * C;
    o[c] = ds;
    o[C + c] = dq;
  }
}

// Reduce the per-block partials of 16 channels per block: thread (slice, channel) walks the partial rows slice,
// slice + 64, ... so every load instruction reads 16 consecutive doubles (128 B) -- the first version (one wave per
// channel, lanes over rows) touched a separate cache line per lane and took 18 us per call.  Fixed slice order and a
// fixed LDS summation order keep it deterministic.
#define FIN_CH 16
#define FIN_SL 64
__device__ __forceinline__ void fin_reduce2(const double* __restrict__ partial, int nblk, int C, int g, int c, int slice,
                                            double (*red)[FIN_CH][2], double& s, double& q) {
  double a = 0.0, b = 0.0;
  if (c < C) {
    // four rows (eight loads) in flight per thread: as a dependent load -> add chain the 256 rows of a layer were four round
    // trips to memory per group, and a finalize launch 10 - 12 us on the critical chain of every layer (45 per step)
    int blk = slice;
    const size_t step = (size_t)FIN_SL * 2 * C;
    for (; blk + 3 * FIN_SL < nblk; blk += 4 * FIN_SL) {
      const double* o = partial + ((size_t)(g * nblk + blk) * 2) * C;
      const double a0 = o[c], b0 = o[C + c], a1 = o[step + c], b1 = o[step + C + c];
      const double a2 = o[2 * step + c], b2 = o[2 * step + C + c], a3 = o[3 * step + c], b3 = o[3 * step + C + c];
      a += (a0 + a1) + (a2 + a3);
      b += (b0 + b1) + (b2 + b3);
    }
    for (; blk < nblk; blk += FIN_SL) {
      const double* o = partial + ((size_t)(g * nblk + blk) * 2) * C;
      a += o[c];
      b += o[C + c];
    }
  }
  __syncthreads();                                  // red[] may still be read from the previous group
  red[slice][threadIdx.x & (FIN_CH - 1)][0] = a;
  red[slice][threadIdx.x & (FIN_CH - 1)][1] = b;
  __syncthreads();
  s = 0.0; q = 0.0;
  if (slice == 0)
#pragma unroll
    for (int i = 0; i < FIN_SL; ++i) { s += red[i][threadIdx.x & (FIN_CH - 1)][0]; q += red[i][threadIdx.x & (FIN_CH - 1)][1]; }
}

__global__ __launch_bounds__(FIN_CH * FIN_SL) void bn_stats_finalize_kernel(const double* __restrict__ partial, int nblk, int C,
                                                               int Ppg, int groups, float eps, float momentum,
                                                               const float* gamma, const float* beta,
                                                               float* running_mean, float* running_var, long long* nbt,
                                                               float* save_mean, float* save_invstd, float* scale,
                                                               float* shift, float* lazy_coef, int lazy_ld, float lazy_slope) {
  __shared__ double red[FIN_SL][FIN_CH][2];
  const int cl = threadIdx.x & (FIN_CH - 1), slice = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  const bool owner = slice == 0 && c < C;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += groups;
  float rm = (owner && running_mean) ? running_mean[c] : 0.f, rv = (owner && running_var) ? running_var[c] : 1.f;
  const double n = (double)Ppg;
  for (int g = 0; g < groups; ++g) {
    double s, q;
    fin_reduce2(partial, nblk, C, g, c, slice, red, s, q);
    if (!owner) continue;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, varf = (float)var;
    const float invstd = 1.0f / sqrtf(varf + eps);
    const float sc = invstd * gamma[c];
    save_mean[g * C + c] = meanf;
    save_invstd[g * C + c] = invstd;
    scale[g * C + c] = sc;
    shift[g * C + c] = beta[c] - meanf * sc;
    if (lazy_coef) {                       // coefficient rows of the lazy output tensor (pp_common.h: PpLazy)
      lazy_coef[(size_t)(g * 3 + 0) * lazy_ld + c] = sc;
      lazy_coef[(size_t)(g * 3 + 1) * lazy_ld + c] = beta[c] - meanf * sc;
      lazy_coef[(size_t)(g * 3 + 2) * lazy_ld + c] = lazy_slope;
    }
    const float unbiased = (float)(var * (n / (n > 1.0 ? n - 1.0 : 1.0)));
    rm = (1.f - momentum) * rm + momentum * meanf;        // weak view first, then strong view
    rv = (1.f - momentum) * rv + momentum * unbiased;
  }
  if (owner) {
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
  }
}

__global__ void bn_eval_coeffs_kernel(int C, int groups, float eps, const float* gamma, const float* beta,
                                      const float* running_mean, const float* running_var, float* save_mean,
                                      float* save_invstd, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(running_var[c] + eps);
  const float sc = invstd * gamma[c];
  for (int g = 0; g < groups; ++g) {
    save_mean[g * C + c] = running_mean[c];
    save_invstd[g * C + c] = invstd;
    scale[g * C + c] = sc;
    shift[g * C + c] = beta[c] - running_mean[c] * sc;
  }
}

// fall-backs used by the fused-epilogue convolution entry points when the selected kernel has no fused form
int pp_bn_partial_rows(int C, int P_per_group, int groups) { return col_plan(C, P_per_group, groups).nblk; }
int pp_bn_stats_partial_launch(const pp_act* z, int ld, int C, int P_per_group, int groups, double* partial, hipStream_t s) {
  ColPlan p = col_plan(C, P_per_group, groups);
  pp_prof_begin(PP_K_BN, 0.0, 4.0 * (double)groups * P_per_group * C, s);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, z, ld, C, P_per_group,
                     p.chunk, p.rows, partial);
  pp_prof_end(s);
  return pp_launch_status("bn_stats_partial");
}

extern "C" size_t PP_FN(pp_bn_workspace)(int C, int P_per_group, int groups) {
  ColPlan p = col_plan(C, P_per_group, groups);
  return (size_t)groups * p.nblk * 2 * C * sizeof(double) + 256;
}

static int bn_check(const void* z, int ld, int C, int Ppg, int groups) {
  PP_CHECK_ARG(z != nullptr, "bn: null pointer");
  PP_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 1024 && ld % 4 == 0 && ld >= C, "bn: C=%d ld=%d (C%%4==0, C<=1024)", C, ld);
  PP_CHECK_ARG(Ppg > 0 && groups > 0, "bn: bad pixel/group count");
  PP_CHECK_ARG(((uintptr_t)z & PP_ACT_ALIGN) == 0, "bn: tensor must be 16-byte aligned");
  return 0;
}

extern "C" int PP_FN(pp_bn_train_stats)(const pp_act* z, int ld, int C, int P_per_group, int groups, float eps,
                                 float momentum, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, int64_t* num_batches_tracked, float* save_mean,
                                 float* save_invstd, float* scale, float* shift, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = bn_check(z, ld, C, P_per_group, groups)) return rc;
  PP_CHECK_ARG(gamma && beta && save_mean && save_invstd && scale && shift && workspace, "bn_train_stats: null pointer");
  if (workspace_bytes < pp_bn_workspace(C, P_per_group, groups)) {
    pp_set_error("bn_train_stats: workspace too small");
    return PP_ERR_WORKSPACE;
  }
  ColPlan p = col_plan(C, P_per_group, groups);
  double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  pp_prof_begin(PP_K_BN, 0.0, 4.0 * (double)groups * P_per_group * C, s);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, z, ld, C, P_per_group,
                     p.chunk, p.rows, partial);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(pp_cdiv(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, partial, p.nblk, C, P_per_group,
                     groups, eps, momentum, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked,
                     save_mean, save_invstd, scale, shift, (float*)nullptr, 0, 0.f);
  pp_prof_end(s);
  return pp_launch_status("bn_train_stats");
}

extern "C" int PP_FN(pp_bn_eval_coeffs)(int C, int groups, float eps, const float* gamma, const float* beta,
                                 const float* running_mean, const float* running_var, float* save_mean,
                                 float* save_invstd, float* scale, float* shift, void* stream) {
  PP_CHECK_ARG(gamma && beta && running_mean && running_var && save_mean && save_invstd && scale && shift,
               "bn_eval_coeffs: null pointer");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(pp_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, C, groups, eps,
                     gamma, beta, running_mean, running_var, save_mean, save_invstd, scale, shift);
  return pp_launch_status("bn_eval_coeffs");
}

#ifndef PP_ACT_16      // no activation operand: one copy, in the fp32 build
// The same for every BatchNorm layer of a network in ONE launch (round 6): in eval mode -- the reference's state from epoch 1 on,
// train_chaos.py:370 -- the coefficients depend on parameters and running statistics only, i.e. they are all known when the step
// starts, and 22 launches of a few hundred threads each sat between the convolutions of the critical chain (~8 us apiece).
#define COEF_BATCH_MAX 24
struct CoefItem { const float *gamma, *beta, *rm, *rv; float *mean, *invstd, *scale, *shift; int C, groups, blk0; };
struct CoefBatch { CoefItem it[COEF_BATCH_MAX]; int n; float eps; };
__global__ void bn_eval_coeffs_batch_kernel(CoefBatch b) {
  int k = 0;
  for (int i = 1; i < b.n; ++i)
    if ((int)blockIdx.x >= b.it[i].blk0) k = i;
  const CoefItem it = b.it[k];
  const int c = ((int)blockIdx.x - it.blk0) * blockDim.x + threadIdx.x;
  if (c >= it.C) return;
  const float invstd = 1.0f / sqrtf(it.rv[c] + b.eps);          // the arithmetic of bn_eval_coeffs_kernel: bit-identical rows
  const float sc = invstd * it.gamma[c];
  for (int g = 0; g < it.groups; ++g) {
    it.mean[g * it.C + c] = it.rm[c];
    it.invstd[g * it.C + c] = invstd;
    it.scale[g * it.C + c] = sc;
    it.shift[g * it.C + c] = it.beta[c] - it.rm[c] * sc;
  }
}

extern "C" int pp_bn_eval_coeffs_batch(const pp_bn_coef_item* items, int n, float eps, void* stream) {
  PP_CHECK_ARG(items && n >= 1, "bn_eval_coeffs_batch: no items");
  for (int i0 = 0; i0 < n; i0 += COEF_BATCH_MAX) {
    CoefBatch b;
    b.n = n - i0 < COEF_BATCH_MAX ? n - i0 : COEF_BATCH_MAX;
    b.eps = eps;
    int blk = 0;
    for (int i = 0; i < b.n; ++i) {
      const pp_bn_coef_item& q = items[i0 + i];
      PP_CHECK_ARG(q.C >= 1 && q.groups >= 1 && q.gamma && q.beta && q.running_mean && q.running_var && q.save_mean && q.save_invstd &&
                       q.scale && q.shift, "bn_eval_coeffs_batch: bad item %d", i0 + i);
      b.it[i] = CoefItem{q.gamma, q.beta, q.running_mean, q.running_var, q.save_mean, q.save_invstd, q.scale, q.shift, q.C, q.groups, blk};
      blk += pp_cdiv(q.C, 64);
    }
    for (int i = b.n; i < COEF_BATCH_MAX; ++i) b.it[i] = CoefItem{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, 0x7fffffff};
    hipLaunchKernelGGL(bn_eval_coeffs_batch_kernel, dim3(blk), dim3(64), 0, (hipStream_t)stream, b);
    if (int rc = pp_launch_status("bn_eval_coeffs_batch")) return rc;
  }
  return 0;
}
#endif  // !PP_ACT_16

// ---- y = lrelu(z * scale[g][c] + shift[g][c]) ----
// Each thread owns one 16-B channel column and walks pixels (no per-element index arithmetic).
__global__ __launch_bounds__(NORM_THREADS) void bn_lrelu_fwd_kernel(const act_t* __restrict__ z, int ld_z,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ shift,
                                                                    act_t* __restrict__ y, int ld_y, int C, int Ppg,
                                                                    int chunk, int rows, float slope, int coef_stride) {
  const int c4n = C >> 2;
  const int tid = threadIdx.x;
  const int cq = tid % c4n, row = tid / c4n;
  if (row >= rows) return;
  const int g = blockIdx.y;
  const int p_lo = blockIdx.x * chunk;
  int p_hi = p_lo + chunk;
  if (p_hi > Ppg) p_hi = Ppg;
  const float4 sc = *reinterpret_cast<const float4*>(scale + g * coef_stride + cq * 4);
  const float4 sh = *reinterpret_cast<const float4*>(shift + g * coef_stride + cq * 4);
  const act_t* zb = z + (size_t)g * Ppg * ld_z + cq * 4;
  act_t* yb = y + (size_t)g * Ppg * ld_y + cq * 4;
#define PP_APPLY(v, o)                      \
  o.x = pp_lrelu(pp_bn_pre(v.x, sc.x, sh.x), slope); \
  o.y = pp_lrelu(pp_bn_pre(v.y, sc.y, sh.y), slope); \
  o.z = pp_lrelu(pp_bn_pre(v.z, sc.z, sh.z), slope); \
  o.w = pp_lrelu(pp_bn_pre(v.w, sc.w, sh.w), slope);
  int p = p_lo + row;
  for (; p + 3 * rows < p_hi; p += 4 * rows) {
    const float4 v0 = act_ld4f(zb + (size_t)p * ld_z);
    const float4 v1 = act_ld4f(zb + (size_t)(p + rows) * ld_z);
    const float4 v2 = act_ld4f(zb + (size_t)(p + 2 * rows) * ld_z);
    const float4 v3 = act_ld4f(zb + (size_t)(p + 3 * rows) * ld_z);
    float4 o0, o1, o2, o3;
    PP_APPLY(v0, o0) PP_APPLY(v1, o1) PP_APPLY(v2, o2) PP_APPLY(v3, o3)
    act_st4f(yb + (size_t)p * ld_y, o0);
    act_st4f(yb + (size_t)(p + rows) * ld_y, o1);
    act_st4f(yb + (size_t)(p + 2 * rows) * ld_y, o2);
    act_st4f(yb + (size_t)(p + 3 * rows) * ld_y, o3);
  }
  for (; p < p_hi; p += rows) {
    const float4 v = act_ld4f(zb + (size_t)p * ld_z);
    float4 o;
    PP_APPLY(v, o)
    act_st4f(yb + (size_t)p * ld_y, o);
  }
#undef PP_APPLY
}

extern "C" int PP_FN(pp_bn_lrelu_fwd)(const pp_act* z, int ld_z, const float* scale, const float* shift, pp_act* y, int ld_y,
                               int C, int P_per_group, int groups, float slope, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = bn_check(z, ld_z, C, P_per_group, groups)) return rc;
  PP_CHECK_ARG(scale && shift && y && ld_y % 4 == 0 && ld_y >= C && ((uintptr_t)y & PP_ACT_ALIGN) == 0, "bn_lrelu_fwd: bad output");
  ColPlan p = col_plan(C, P_per_group, groups);
  pp_prof_begin(PP_K_BN, 0.0, 8.0 * (double)groups * P_per_group * C, s);
  hipLaunchKernelGGL(bn_lrelu_fwd_kernel, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, z, ld_z, scale, shift, y, ld_y,
                     C, P_per_group, p.chunk, p.rows, slope, C);
  pp_prof_end(s);
  return pp_launch_status("bn_lrelu_fwd");
}

int pp_bn_apply_launch(const pp_act* z, int ld_z, const float* scale, const float* shift, int coef_groups, pp_act* y, int ld_y,
                       int C, int P_per_group, int groups, float slope, hipStream_t s) {
  ColPlan p = col_plan(C, P_per_group, groups);
  pp_prof_begin(PP_K_BN, 0.0, 8.0 * (double)groups * P_per_group * C, s);
  hipLaunchKernelGGL(bn_lrelu_fwd_kernel, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, z, ld_z, scale, shift, y, ld_y,
                     C, P_per_group, p.chunk, p.rows, slope, coef_groups > 1 ? C : 0);
  pp_prof_end(s);
  return pp_launch_status("bn_apply");
}

// ---- backward ----
// g = dy * lrelu'(z*scale+shift);  s1 = sum g;  s2 = sum g * (z-mean)*invstd    (per group, per channel)
__global__ __launch_bounds__(NORM_THREADS) void bn_bwd_partial_kernel(
    const act_t* __restrict__ dy, int ld_dy, const act_t* __restrict__ z, int ld_z, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd, int C, int Ppg,
    int chunk, int rows, float slope, double* __restrict__ partial) {
  __shared__ float sh[2 * NORM_THREADS * 4];
  const int c4n = C >> 2;
  const int tid = threadIdx.x;
  const int cq = tid % c4n, row = tid / c4n;
  const bool active = row < rows;
  const int g = blockIdx.y, blk = blockIdx.x;
  const int p_lo = blk * chunk;
  int p_hi = p_lo + chunk;
  if (p_hi > Ppg) p_hi = Ppg;
  if (active) {
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const float4 sc = *reinterpret_cast<const float4*>(scale + g * C + cq * 4);
    const float4 sf = *reinterpret_cast<const float4*>(shift + g * C + cq * 4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + g * C + cq * 4);
    const float4 is = *reinterpret_cast<const float4*>(invstd + g * C + cq * 4);
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, sfv[4] = {sf.x, sf.y, sf.z, sf.w};
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
    const size_t gbase = (size_t)g * Ppg;
#define PP_BWD_ACC(d4, z4)                                                                   \
    {                                                                                        \
      const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w};          \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                        \
        const float pre = pp_bn_pre(zv[e], scv[e], sfv[e]);                                  \
        const float gg = pre > 0.f ? dv[e] : dv[e] * slope;                                  \
        s1[e] += gg;                                                                         \
        s2[e] += gg * ((zv[e] - muv[e]) * isv[e]);                                           \
      }                                                                                      \
    }
    int p = p_lo + row;
    for (; p + 3 * rows < p_hi; p += 4 * rows) {
      const float4 d0 = act_ld4f(dy + (gbase + p) * ld_dy + cq * 4);
      const float4 d1 = act_ld4f(dy + (gbase + p + rows) * ld_dy + cq * 4);
      const float4 d2 = act_ld4f(dy + (gbase + p + 2 * rows) * ld_dy + cq * 4);
      const float4 d3 = act_ld4f(dy + (gbase + p + 3 * rows) * ld_dy + cq * 4);
      const float4 z0 = act_ld4f(z + (gbase + p) * ld_z + cq * 4);
      const float4 z1 = act_ld4f(z + (gbase + p + rows) * ld_z + cq * 4);
      const float4 z2 = act_ld4f(z + (gbase + p + 2 * rows) * ld_z + cq * 4);
      const float4 z3 = act_ld4f(z + (gbase + p + 3 * rows) * ld_z + cq * 4);
      PP_BWD_ACC(d0, z0) PP_BWD_ACC(d1, z1) PP_BWD_ACC(d2, z2) PP_BWD_ACC(d3, z3)
    }
    for (; p < p_hi; p += rows) {
      const float4 d4 = act_ld4f(dy + (gbase + p) * ld_dy + cq * 4);
      const float4 z4 = act_ld4f(z + (gbase + p) * ld_z + cq * 4);
      PP_BWD_ACC(d4, z4)
    }
#undef PP_BWD_ACC
    float* d = sh + (row * c4n + cq) * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[e] = s1[e]; d[4 + e] = s2[e]; }
  }
  __syncthreads();
  for (int c = tid; c < C; c += NORM_THREADS) {
    double a = 0.0, b = 0.0;
    const int cq2 = c >> 2, e = c & 3;
    for (int r = 0; r < rows; ++r) {
      a += (double)sh[(r * c4n + cq2) * 8 + e];
      b += (double)sh[(r * c4n + cq2) * 8 + 4 + e];
    }
    double* o = partial + ((size_t)(g * gridDim.x + blk) * 2) * C;
    o[c] = a;
    o[C + c] = b;
  }
}

// coefficients of dz = kA*g + kB*z + kC, parameter gradients (same 16-channel x 64-slice reduction as above)
__global__ __launch_bounds__(FIN_CH * FIN_SL) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nblk, int C,
                                                             int Ppg, int groups, int training, const float* gamma,
                                                             const float* mean, const float* invstd, float* kA,
                                                             float* kB, float* kC, float* dgamma, float* dbeta,
                                                             float* dbias, int accumulate, float* amax) {
  __shared__ double red[FIN_SL][FIN_CH][2];
  if (amax && blockIdx.x == 0 && threadIdx.x == 0) *amax = 0.f;      // bn_bwd_apply_kernel accumulates max |dz| into it
  const int cl = threadIdx.x & (FIN_CH - 1), slice = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  const bool owner = slice == 0 && c < C;
  double dg = 0.0, db = 0.0, dbc = 0.0;
  const double n = (double)Ppg;
  for (int g = 0; g < groups; ++g) {
    double s1, s2;
    fin_reduce2(partial, nblk, C, g, c, slice, red, s1, s2);
    if (!owner) continue;
    dg += s2;
    db += s1;
    const double A = (double)gamma[c] * (double)invstd[g * C + c];
    if (training) {
      const double B = -A * (double)invstd[g * C + c] * s2 / n;
      kA[g * C + c] = (float)A;
      kB[g * C + c] = (float)B;
      kC[g * C + c] = (float)(-A * s1 / n - B * (double)mean[g * C + c]);
      // sum_p dz == 0 exactly in train mode (the batch mean is removed): conv bias gets no gradient
    } else {
      kA[g * C + c] = (float)A;
      kB[g * C + c] = 0.f;
      kC[g * C + c] = 0.f;
      dbc += A * s1;
    }
  }
  if (owner) {
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)dg;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)db;
    if (dbias) dbias[c] = (accumulate ? dbias[c] : 0.f) + (float)dbc;
  }
}

__global__ __launch_bounds__(NORM_THREADS) void bn_bwd_apply_kernel(
    const act_t* __restrict__ dy, int ld_dy, const act_t* __restrict__ z, int ld_z, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ kA, const float* __restrict__ kB,
    const float* __restrict__ kC, act_t* __restrict__ dz, int ld_dz, int C, int Ppg, int chunk, int rows, float slope,
    float* __restrict__ amax /* nullable: max |dz| of the launch, zeroed by bn_bwd_finalize_kernel */) {
  const int c4n = C >> 2;
  const int tid = threadIdx.x;
  const int cq = tid % c4n, row = tid / c4n;
  if (row >= rows) return;
  float mx = 0.f;
  const int g = blockIdx.y;
  const int p_lo = blockIdx.x * chunk;
  int p_hi = p_lo + chunk;
  if (p_hi > Ppg) p_hi = Ppg;
  const int co = g * C + cq * 4;
  const float4 sc = *reinterpret_cast<const float4*>(scale + co);
  const float4 sf = *reinterpret_cast<const float4*>(shift + co);
  const float4 a4 = *reinterpret_cast<const float4*>(kA + co);
  const float4 b4 = *reinterpret_cast<const float4*>(kB + co);
  const float4 c4 = *reinterpret_cast<const float4*>(kC + co);
  const size_t gb = (size_t)g * Ppg;
  const act_t* dyb = dy + gb * ld_dy + cq * 4;
  const act_t* zb = z + gb * ld_z + cq * 4;
  act_t* dzb = dz + gb * ld_dz + cq * 4;
#define PP_DZ(d4, z4, o)                                                                 \
  o.x = a4.x * (pp_bn_pre(z4.x, sc.x, sf.x) > 0.f ? d4.x : d4.x * slope) + b4.x * z4.x + c4.x;  \
  o.y = a4.y * (pp_bn_pre(z4.y, sc.y, sf.y) > 0.f ? d4.y : d4.y * slope) + b4.y * z4.y + c4.y;  \
  o.z = a4.z * (pp_bn_pre(z4.z, sc.z, sf.z) > 0.f ? d4.z : d4.z * slope) + b4.z * z4.z + c4.z;  \
  o.w = a4.w * (pp_bn_pre(z4.w, sc.w, sf.w) > 0.f ? d4.w : d4.w * slope) + b4.w * z4.w + c4.w;
#define PP_MX(o) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
  int p = p_lo + row;
  for (; p + rows < p_hi; p += 2 * rows) {
    const float4 d0 = act_ld4f(dyb + (size_t)p * ld_dy);
    const float4 d1 = act_ld4f(dyb + (size_t)(p + rows) * ld_dy);
    const float4 z0 = act_ld4f(zb + (size_t)p * ld_z);
    const float4 z1 = act_ld4f(zb + (size_t)(p + rows) * ld_z);
    float4 o0, o1;
    PP_DZ(d0, z0, o0) PP_DZ(d1, z1, o1)
    act_st4f(dzb + (size_t)p * ld_dz, o0);
    act_st4f(dzb + (size_t)(p + rows) * ld_dz, o1);
    PP_MX(o0) PP_MX(o1)
  }
  for (; p < p_hi; p += rows) {
    const float4 d0 = act_ld4f(dyb + (size_t)p * ld_dy);
    const float4 z0 = act_ld4f(zb + (size_t)p * ld_z);
    float4 o0;
    PP_DZ(d0, z0, o0)
    act_st4f(dzb + (size_t)p * ld_dz, o0);
    PP_MX(o0)
  }
#undef PP_DZ
#undef PP_MX
  if (amax) {                                   // max is order independent: the atomic keeps the result deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0 && mx > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(mx));
  }
}

static int bn_lrelu_bwd_impl(const pp_act* dy, int ld_dy, const pp_act* z, int ld_z, const float* scale,
                             const float* shift, const float* save_mean, const float* save_invstd,
                             const float* gamma, int training, pp_act* dz, int ld_dz, float* dgamma, float* dbeta,
                             float* dbias_conv, int accumulate_param_grads, int C, int P_per_group, int groups,
                             float slope, void* workspace, size_t workspace_bytes, float* dz_amax, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = bn_check(z, ld_z, C, P_per_group, groups)) return rc;
  PP_CHECK_ARG(dy && dz && scale && shift && save_mean && save_invstd && gamma && workspace, "bn_lrelu_bwd: null pointer");
  PP_CHECK_ARG(ld_dy % 4 == 0 && ld_dz % 4 == 0 && ld_dy >= C && ld_dz >= C, "bn_lrelu_bwd: bad ld");
  PP_CHECK_ARG(((uintptr_t)dy & PP_ACT_ALIGN) == 0 && ((uintptr_t)dz & PP_ACT_ALIGN) == 0, "bn_lrelu_bwd: tensors must be 16-byte aligned");
  const size_t need = pp_bn_workspace(C, P_per_group, groups) + (size_t)3 * groups * C * sizeof(float);
  if (workspace_bytes < need) {
    pp_set_error("bn_lrelu_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    return PP_ERR_WORKSPACE;
  }
  ColPlan p = col_plan(C, P_per_group, groups);
  double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  float* kA = reinterpret_cast<float*>(partial + (size_t)groups * p.nblk * 2 * C);
  float* kB = kA + (size_t)groups * C;
  float* kC = kB + (size_t)groups * C;
  pp_prof_begin(PP_K_BN, 0.0, 20.0 * (double)groups * P_per_group * C, s);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, dy, ld_dy, z, ld_z, scale,
                     shift, save_mean, save_invstd, C, P_per_group, p.chunk, p.rows, slope, partial);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(pp_cdiv(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, partial, p.nblk, C, P_per_group,
                     groups, training, gamma, save_mean, save_invstd, kA, kB, kC, dgamma, dbeta, dbias_conv,
                     accumulate_param_grads, dz_amax);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, dy, ld_dy, z, ld_z, scale,
                     shift, kA, kB, kC, dz, ld_dz, C, P_per_group, p.chunk, p.rows, slope, dz_amax);
  pp_prof_end(s);
  return pp_launch_status("bn_lrelu_bwd");
}

extern "C" int PP_FN(pp_bn_lrelu_bwd)(const pp_act* dy, int ld_dy, const pp_act* z, int ld_z, const float* scale,
                               const float* shift, const float* save_mean, const float* save_invstd,
                               const float* gamma, int training, pp_act* dz, int ld_dz, float* dgamma, float* dbeta,
                               float* dbias_conv, int accumulate_param_grads, int C, int P_per_group, int groups,
                               float slope, void* workspace, size_t workspace_bytes, void* stream) {
  return bn_lrelu_bwd_impl(dy, ld_dy, z, ld_z, scale, shift, save_mean, save_invstd, gamma, training, dz, ld_dz, dgamma, dbeta,
                           dbias_conv, accumulate_param_grads, C, P_per_group, groups, slope, workspace, workspace_bytes,
                           nullptr, stream);
}

// same, and additionally *dz_amax = max |dz| (device float): the power-of-two operand scale of the split-fp16
// convolution kernels that consume dz (pp_conv3x3_bwd_data_f16x3)
extern "C" int PP_FN(pp_bn_lrelu_bwd_amax)(const pp_act* dy, int ld_dy, const pp_act* z, int ld_z, const float* scale,
                                    const float* shift, const float* save_mean, const float* save_invstd,
                                    const float* gamma, int training, pp_act* dz, int ld_dz, float* dgamma, float* dbeta,
                                    float* dbias_conv, int accumulate_param_grads, int C, int P_per_group, int groups,
                                    float slope, void* workspace, size_t workspace_bytes, float* dz_amax, void* stream) {
  PP_CHECK_ARG(dz_amax, "bn_lrelu_bwd_amax: null dz_amax");
  return bn_lrelu_bwd_impl(dy, ld_dy, z, ld_z, scale, shift, save_mean, save_invstd, gamma, training, dz, ld_dz, dgamma, dbeta,
                           dbias_conv, accumulate_param_grads, C, P_per_group, groups, slope, workspace, workspace_bytes,
                           dz_amax, stream);
}

// ---- split forms for synchronised BatchNorm (data-parallel epoch 0: statistics over the GLOBAL batch) ----
// The reference normalises over the whole batch (models/unet.py:189); with the batch sharded over ranks the caller
// all-reduces the per-channel sums between the two halves of each call:
//   forward : pp_bn_stats_sums -> all_reduce(sums) -> pp_bn_train_finalize(n = global pixels per group)
//   backward: pp_bn_lrelu_bwd_sums -> all_reduce(copy) -> pp_bn_lrelu_bwd_apply(local sums, global sums, global n)
// sums layout: double [groups][2][C] (= a `partial` array with one row per group, so the finalize kernels run on it).
__global__ __launch_bounds__(FIN_CH * FIN_SL) void bn_reduce_partials_kernel(const double* __restrict__ partial, int nblk, int C,
                                                                             int groups, double* __restrict__ sums) {
  __shared__ double red[FIN_SL][FIN_CH][2];
  const int cl = threadIdx.x & (FIN_CH - 1), slice = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  for (int g = 0; g < groups; ++g) {
    double s, q;
    fin_reduce2(partial, nblk, C, g, c, slice, red, s, q);
    if (slice == 0 && c < C) {
      sums[((size_t)g * 2) * C + c] = s;
      sums[((size_t)g * 2 + 1) * C + c] = q;
    }
  }
}

extern "C" int PP_FN(pp_bn_stats_sums)(const pp_act* z, int ld, int C, int P_per_group, int groups, double* sums,
                                void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = bn_check(z, ld, C, P_per_group, groups)) return rc;
  PP_CHECK_ARG(sums && workspace && ((uintptr_t)sums & 7) == 0, "bn_stats_sums: null / misaligned pointer");
  if (workspace_bytes < pp_bn_workspace(C, P_per_group, groups)) {
    pp_set_error("bn_stats_sums: workspace too small");
    return PP_ERR_WORKSPACE;
  }
  ColPlan p = col_plan(C, P_per_group, groups);
  double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  pp_prof_begin(PP_K_BN, 0.0, 4.0 * (double)groups * P_per_group * C, s);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, z, ld, C, P_per_group,
                     p.chunk, p.rows, partial);
  hipLaunchKernelGGL(bn_reduce_partials_kernel, dim3(pp_cdiv(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, partial, p.nblk, C,
                     groups, sums);
  pp_prof_end(s);
  return pp_launch_status("bn_stats_sums");
}

static int bn_train_finalize_impl(const double* sums, int rows, int C, int n_per_group, int groups, float eps, float momentum,
                                  const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  int64_t* num_batches_tracked, float* save_mean, float* save_invstd, float* scale,
                                  float* shift, float* lazy_coef, int lazy_ld, float lazy_slope, void* stream) {
  PP_CHECK_ARG(sums && gamma && beta && save_mean && save_invstd && scale && shift, "bn_train_finalize: null pointer");
  PP_CHECK_ARG(C > 0 && n_per_group > 0 && groups > 0 && rows > 0, "bn_train_finalize: bad shape");
  PP_CHECK_ARG(!lazy_coef || lazy_ld >= C, "bn_train_finalize: lazy coefficient rows shorter than C");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(pp_cdiv(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, (hipStream_t)stream, sums, rows, C,
                     n_per_group, groups, eps, momentum, gamma, beta, running_mean, running_var,
                     (long long*)num_batches_tracked, save_mean, save_invstd, scale, shift, lazy_coef, lazy_ld, lazy_slope);
  return pp_launch_status("bn_train_finalize");
}

extern "C" int PP_FN(pp_bn_train_finalize)(const double* sums, int rows, int C, int n_per_group, int groups, float eps, float momentum,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    int64_t* num_batches_tracked, float* save_mean, float* save_invstd, float* scale,
                                    float* shift, void* stream) {
  return bn_train_finalize_impl(sums, rows, C, n_per_group, groups, eps, momentum, gamma, beta, running_mean, running_var,
                                num_batches_tracked, save_mean, save_invstd, scale, shift, nullptr, 0, 0.f, stream);
}

// ... and additionally the coefficient rows (scale, shift, slope) of the layer's LAZY output tensor: lazy_coef points at
// channel 0 of the layer inside rows of lazy_ld floats (pp_lazy_in, include/pacingpseudo_hip.h)
extern "C" int PP_FN(pp_bn_train_finalize_lazy)(const double* sums, int rows, int C, int n_per_group, int groups, float eps,
                                         float momentum, const float* gamma, const float* beta, float* running_mean,
                                         float* running_var, int64_t* num_batches_tracked, float* save_mean,
                                         float* save_invstd, float* scale, float* shift, float* lazy_coef, int lazy_ld,
                                         float slope, void* stream) {
  PP_CHECK_ARG(lazy_coef, "bn_train_finalize_lazy: null coefficient pointer");
  return bn_train_finalize_impl(sums, rows, C, n_per_group, groups, eps, momentum, gamma, beta, running_mean, running_var,
                                num_batches_tracked, save_mean, save_invstd, scale, shift, lazy_coef, lazy_ld, slope, stream);
}

// ---- lazy tensor -> ordinary tensor: y = lrelu(z * scale + shift) with the rows of a pp_lazy_in ----
// For consumers without a *_lazy form and for the end points handed back to the caller.  src and dst may not alias when
// the producing layer's backward still needs z (the training engine never materialises in place).
__global__ __launch_bounds__(NORM_THREADS) void lazy_materialize_kernel(const act_t* __restrict__ z, int ld_z, PpLazy lz,
                                                                        act_t* __restrict__ y, int ld_y, int C, int HW,
                                                                        long long P) {
  const int c4n = C >> 2;
  const long long total = P * c4n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4n);
    const long long p = i / c4n;
    pp_f32x4 sc, sh, sl;
    pp_lazy_rows4(lz, (int)(p / HW), cq * 4, sc, sh, sl);
    const pp_f32x4 v = act_ld4(z + (size_t)p * ld_z + cq * 4);
    act_st4(y + (size_t)p * ld_y + cq * 4, pp_lazy_apply4(v, sc, sh, sl));
  }
}

extern "C" int PP_FN(pp_lazy_materialize)(const pp_act* src, int ld_src, const pp_lazy_in* lazy, pp_act* dst, int ld_dst, int C, int B,
                                   int HW, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(src && dst && lazy && lazy->coef, "lazy_materialize: null pointer");
  PP_CHECK_ARG(C > 0 && C % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_src >= C && ld_dst >= C && B > 0 && HW > 0,
               "lazy_materialize: bad shape");
  PP_CHECK_ARG(lazy->groups >= 1 && B % lazy->groups == 0 && lazy->ld % 4 == 0 && lazy->ld >= C, "lazy_materialize: bad descriptor");
  PP_CHECK_ARG(((((uintptr_t)src) | ((uintptr_t)dst)) & PP_ACT_ALIGN) == 0 && ((uintptr_t)lazy->coef & 15) == 0, "lazy_materialize: alignment");
  const long long P = (long long)B * HW;
  long long blocks = (P * (C / 4) + NORM_THREADS * 4 - 1) / (NORM_THREADS * 4);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  pp_prof_begin(PP_K_BN, 0.0, 8.0 * (double)P * C, s);
  hipLaunchKernelGGL(lazy_materialize_kernel, dim3((unsigned)blocks), dim3(NORM_THREADS), 0, s, src, ld_src,
                     PpLazy{lazy->coef, lazy->ld, B / lazy->groups}, dst, ld_dst, C, HW, P);
  pp_prof_end(s);
  return pp_launch_status("lazy_materialize");
}

extern "C" int PP_FN(pp_bn_lrelu_bwd_sums)(const pp_act* dy, int ld_dy, const pp_act* z, int ld_z, const float* scale,
                                    const float* shift, const float* save_mean, const float* save_invstd, int C,
                                    int P_per_group, int groups, float slope, double* sums, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = bn_check(z, ld_z, C, P_per_group, groups)) return rc;
  PP_CHECK_ARG(dy && scale && shift && save_mean && save_invstd && sums && workspace, "bn_lrelu_bwd_sums: null pointer");
  PP_CHECK_ARG(ld_dy % 4 == 0 && ld_dy >= C && ((uintptr_t)dy & PP_ACT_ALIGN) == 0, "bn_lrelu_bwd_sums: bad dy");
  if (workspace_bytes < pp_bn_workspace(C, P_per_group, groups)) {
    pp_set_error("bn_lrelu_bwd_sums: workspace too small");
    return PP_ERR_WORKSPACE;
  }
  ColPlan p = col_plan(C, P_per_group, groups);
  double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  pp_prof_begin(PP_K_BN, 0.0, 8.0 * (double)groups * P_per_group * C, s);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, dy, ld_dy, z, ld_z, scale,
                     shift, save_mean, save_invstd, C, P_per_group, p.chunk, p.rows, slope, partial);
  hipLaunchKernelGGL(bn_reduce_partials_kernel, dim3(pp_cdiv(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, partial, p.nblk, C,
                     groups, sums);
  pp_prof_end(s);
  return pp_launch_status("bn_lrelu_bwd_sums");
}

// dz from the GLOBAL sums (n_global pixels per group), parameter gradients from the LOCAL sums (the gradient
// all-reduce adds the ranks' contributions afterwards).  workspace >= 6*groups*C floats.
extern "C" int PP_FN(pp_bn_lrelu_bwd_apply)(const pp_act* dy, int ld_dy, const pp_act* z, int ld_z, const float* scale,
                                     const float* shift, const float* save_mean, const float* save_invstd,
                                     const float* gamma, int training, const double* local_sums, const double* global_sums,
                                     int n_global_per_group, pp_act* dz, int ld_dz, float* dgamma, float* dbeta,
                                     float* dbias_conv, int accumulate_param_grads, int C, int P_per_group, int groups,
                                     float slope, void* workspace, size_t workspace_bytes, float* dz_amax, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = bn_check(z, ld_z, C, P_per_group, groups)) return rc;
  PP_CHECK_ARG(dy && dz && scale && shift && save_mean && save_invstd && gamma && local_sums && global_sums && workspace,
               "bn_lrelu_bwd_apply: null pointer");
  PP_CHECK_ARG(ld_dy % 4 == 0 && ld_dz % 4 == 0 && ld_dy >= C && ld_dz >= C && n_global_per_group > 0, "bn_lrelu_bwd_apply: bad ld / n");
  PP_CHECK_ARG(((uintptr_t)dy & PP_ACT_ALIGN) == 0 && ((uintptr_t)dz & PP_ACT_ALIGN) == 0, "bn_lrelu_bwd_apply: tensors must be 16-byte aligned");
  const size_t need = (size_t)6 * groups * C * sizeof(float) + 16;
  if (workspace_bytes < need) {
    pp_set_error("bn_lrelu_bwd_apply: workspace too small (%zu < %zu)", workspace_bytes, need);
    return PP_ERR_WORKSPACE;
  }
  ColPlan p = col_plan(C, P_per_group, groups);
  float* kA = reinterpret_cast<float*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  float* kB = kA + (size_t)groups * C;
  float* kC = kB + (size_t)groups * C;
  float* scratch = kC + (size_t)groups * C;          // coefficients of the local-sums pass: not used
  pp_prof_begin(PP_K_BN, 0.0, 12.0 * (double)groups * P_per_group * C, s);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(pp_cdiv(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, local_sums, 1, C,
                     n_global_per_group, groups, training, gamma, save_mean, save_invstd, scratch, scratch + (size_t)groups * C,
                     scratch + (size_t)2 * groups * C, dgamma, dbeta, dbias_conv, accumulate_param_grads, (float*)nullptr);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(pp_cdiv(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, global_sums, 1, C,
                     n_global_per_group, groups, training, gamma, save_mean, save_invstd, kA, kB, kC, (float*)nullptr,
                     (float*)nullptr, (float*)nullptr, 0, dz_amax);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, dy, ld_dy, z, ld_z, scale,
                     shift, kA, kB, kC, dz, ld_dz, C, P_per_group, p.chunk, p.rows, slope, dz_amax);
  pp_prof_end(s);
  return pp_launch_status("bn_lrelu_bwd_apply");
}

// ---- eval-mode BatchNorm + LeakyReLU backward in ONE pass (the reference's state from epoch 1 on, train_chaos.py:370) ----
// With running statistics dz = gamma * invstd * g needs no batch reduction, and the forward epilogue wrote y only
// (pp_conv3x3_fwd_bn mode 2), so everything comes from dy and y:  g = dy * lrelu'(pre), the branch read from the
// sign of y;  pre = y (y > 0) or y / slope;  xhat = (pre - beta) / gamma.
//   dz = scale * g,   dbeta = sum g,   dgamma = sum g * xhat = (sum g * pre - beta * sum g) / gamma,
//   dbias_conv = scale * sum g.     One read of dy and y, one write of dz: 12 B per element instead of 20.
__global__ __launch_bounds__(NORM_THREADS) void bn_bwd_eval_kernel(
    const act_t* __restrict__ dy, int ld_dy, const act_t* __restrict__ y, int ld_y, const float* __restrict__ scale,
    act_t* __restrict__ dz, int ld_dz, int C, int P, int chunk, int rows, float slope, float inv_slope,
    double* __restrict__ partial, float* __restrict__ amax) {
  __shared__ float sh[2 * NORM_THREADS * 4];
  const int c4n = C >> 2;
  const int tid = threadIdx.x;
  const int cq = tid % c4n, row = tid / c4n;
  const bool active = row < rows;
  const int blk = blockIdx.x;
  const int p_lo = blk * chunk;
  int p_hi = p_lo + chunk;
  if (p_hi > P) p_hi = P;
  float mx = 0.f;
  if (active) {
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const float4 sc = *reinterpret_cast<const float4*>(scale + cq * 4);
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
#define PP_EV(d4, y4, o)                                                                      \
    {                                                                                         \
      const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};           \
      float ov[4];                                                                            \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                         \
        const bool pos = yv[e] > 0.f;                                                         \
        const float gg = pos ? dv[e] : dv[e] * slope;                                         \
        const float pre = pos ? yv[e] : yv[e] * inv_slope;                                    \
        s1[e] += gg;                                                                          \
        s2[e] += gg * pre;                                                                    \
        ov[e] = scv[e] * gg;                                                                  \
        mx = fmaxf(mx, fabsf(ov[e]));                                                         \
      }                                                                                       \
      o = make_float4(ov[0], ov[1], ov[2], ov[3]);                                            \
    }
    int p = p_lo + row;
    for (; p + rows < p_hi; p += 2 * rows) {
      const float4 d0 = act_ld4f(dy + (size_t)p * ld_dy + cq * 4);
      const float4 d1 = act_ld4f(dy + (size_t)(p + rows) * ld_dy + cq * 4);
      const float4 y0 = act_ld4f(y + (size_t)p * ld_y + cq * 4);
      const float4 y1 = act_ld4f(y + (size_t)(p + rows) * ld_y + cq * 4);
      float4 o0, o1;
      PP_EV(d0, y0, o0) PP_EV(d1, y1, o1)
      act_st4f(dz + (size_t)p * ld_dz + cq * 4, o0);
      act_st4f(dz + (size_t)(p + rows) * ld_dz + cq * 4, o1);
    }
    for (; p < p_hi; p += rows) {
      const float4 d0 = act_ld4f(dy + (size_t)p * ld_dy + cq * 4);
      const float4 y0 = act_ld4f(y + (size_t)p * ld_y + cq * 4);
      float4 o0;
      PP_EV(d0, y0, o0)
      act_st4f(dz + (size_t)p * ld_dz + cq * 4, o0);
    }
#undef PP_EV
    float* d = sh + (row * c4n + cq) * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[e] = s1[e]; d[4 + e] = s2[e]; }
  }
  __syncthreads();
  for (int c = tid; c < C; c += NORM_THREADS) {
    double a = 0.0, b = 0.0;
    const int cq2 = c >> 2, e = c & 3;
    for (int r = 0; r < rows; ++r) {
      a += (double)sh[(r * c4n + cq2) * 8 + e];
      b += (double)sh[(r * c4n + cq2) * 8 + 4 + e];
    }
    double* o = partial + ((size_t)blk * 2) * C;
    o[c] = a;
    o[C + c] = b;
  }
  if (amax) {                                   // max is order independent: the atomic keeps the result deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0 && mx > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(mx));
  }
}

__global__ __launch_bounds__(FIN_CH * FIN_SL) void bn_bwd_eval_finalize_kernel(const double* __restrict__ partial, int nblk, int C,
                                                                  const float* gamma, const float* beta, const float* scale,
                                                                  float* dgamma, float* dbeta, float* dbias, int accumulate) {
  __shared__ double red[FIN_SL][FIN_CH][2];
  const int cl = threadIdx.x & (FIN_CH - 1), slice = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  double s1, s2;
  fin_reduce2(partial, nblk, C, 0, c, slice, red, s1, s2);
  if (slice != 0 || c >= C) return;
  // xhat is recovered from y = gamma * xhat + beta, so the quotient needs gamma != 0.  A channel whose gamma is EXACTLY 0
  // has lost xhat (y == beta everywhere): its dgamma is reported as 0 instead of Inf / NaN -- which Adam would keep in m and
  // v for ever -- and that channel stays where weight decay leaves it; for tiny |gamma| the quotient is finite and carries
  // the fp32 rounding of y amplified by |beta / (gamma * xhat)| (tests/test_gpu_round2.py: gamma = 1e-6 within 5 %).
  const double gm = (double)gamma[c];
  const double dg = gm != 0.0 ? (s2 - (double)beta[c] * s1) / gm : 0.0;
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)dg;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
  if (dbias) dbias[c] = (accumulate ? dbias[c] : 0.f) + (float)((double)scale[c] * s1);
}

extern "C" int PP_FN(pp_bn_lrelu_bwd_eval)(const pp_act* dy, int ld_dy, const pp_act* y, int ld_y, const float* scale,
                                    const float* gamma, const float* beta, pp_act* dz, int ld_dz, float* dgamma,
                                    float* dbeta, float* dbias_conv, int accumulate_param_grads, int C, int P_total,
                                    float slope, void* workspace, size_t workspace_bytes, float* dz_amax, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = bn_check(y, ld_y, C, P_total, 1)) return rc;
  PP_CHECK_ARG(dy && dz && scale && gamma && beta && workspace, "bn_lrelu_bwd_eval: null pointer");
  PP_CHECK_ARG(ld_dy % 4 == 0 && ld_dz % 4 == 0 && ld_dy >= C && ld_dz >= C && slope > 0.f, "bn_lrelu_bwd_eval: bad ld / slope");
  PP_CHECK_ARG(((uintptr_t)dy & PP_ACT_ALIGN) == 0 && ((uintptr_t)dz & PP_ACT_ALIGN) == 0 && ((uintptr_t)scale & 15) == 0,
               "bn_lrelu_bwd_eval: tensors must be 16-byte aligned");
  if (workspace_bytes < pp_bn_workspace(C, P_total, 1)) {
    pp_set_error("bn_lrelu_bwd_eval: workspace too small");
    return PP_ERR_WORKSPACE;
  }
  ColPlan p = col_plan(C, P_total, 1);
  double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  if (dz_amax && hipMemsetAsync(dz_amax, 0, sizeof(float), s) != hipSuccess) return pp_launch_status("bn_bwd_eval_memset");
  pp_prof_begin(PP_K_BN, 0.0, 12.0 * (double)P_total * C, s);
  hipLaunchKernelGGL(bn_bwd_eval_kernel, dim3(p.nblk), dim3(NORM_THREADS), 0, s, dy, ld_dy, y, ld_y, scale, dz, ld_dz, C,
                     P_total, p.chunk, p.rows, slope, 1.0f / slope, partial, dz_amax);
  hipLaunchKernelGGL(bn_bwd_eval_finalize_kernel, dim3(pp_cdiv(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, partial, p.nblk, C,
                     gamma, beta, scale, dgamma, dbeta, dbias_conv, accumulate_param_grads);
  pp_prof_end(s);
  return pp_launch_status("bn_lrelu_bwd_eval");
}

// ---- BatchNorm + LeakyReLU backward with the gradient of the FOLLOWING 2x2 max-pooling folded in (round 4) ----
// The output y of an encoder stage feeds the skip connection and nn.MaxPool2d(2, 2) (models/unet.py:109,123-127), so its
// gradient is dskip + unpool(dpooled).  maxpool2_bwd_kernel used to add the second term into the skip-gradient buffer
// (read y, read dpooled, read + write the buffer: 1.0 GB per launch at the benchmark shape, 0.54 ms per step) just before the
// BatchNorm backward read that buffer again.  Here the BatchNorm backward walks 2x2 WINDOWS instead of pixels, finds each
// window's winner itself -- first maximum of y in (0,0),(0,1),(1,0),(1,1) order, PyTorch's rule, with y recomputed from z by
// the one expression every kernel uses (pp_bn_pre) -- and adds dpooled to that element on the fly: one extra quarter-size
// read per pass instead of a whole pass.  MODE 0: partial sums (s1 = sum g, s2 = sum g xhat); MODE 1: dz = kA g + kB z + kC
// (+ max |dz|); MODE 2: eval-mode one-pass form on y (see bn_bwd_eval_kernel): sums of (g, g pre) and dz = scale g.
template <int MODE>
__global__ __launch_bounds__(NORM_THREADS) void bn_bwd_pool_kernel(
    const act_t* __restrict__ dy, int ld_dy, const act_t* __restrict__ dp, int ld_dp, const act_t* __restrict__ zy, int ld_z,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ kA, const float* __restrict__ kB, const float* __restrict__ kC,
    act_t* __restrict__ dz, int ld_dz, int C, int H, int W, int Wpg /* windows per group */, int chunk, int rows, float slope,
    float inv_slope, double* __restrict__ partial, float* __restrict__ amax) {
  __shared__ float sh[2 * NORM_THREADS * 4];
  const int c4n = C >> 2;
  const int tid = threadIdx.x;
  const int cq = tid % c4n, row = tid / c4n;
  const bool active = row < rows;
  const int g = blockIdx.y, blk = blockIdx.x;
  const int w_lo = blk * chunk;
  int w_hi = w_lo + chunk;
  if (w_hi > Wpg) w_hi = Wpg;
  const int Ho = H >> 1, Wo = W >> 1;
  float mx = 0.f;
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (active) {
    const int co = g * C + cq * 4;
    float scv[4], sfv[4], muv[4], isv[4], av[4], bv[4], cv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      scv[e] = scale[(MODE == 2 ? 0 : g * C) + cq * 4 + e];
      sfv[e] = MODE == 2 ? 0.f : shift[co + e];
      muv[e] = MODE == 0 ? mean[co + e] : 0.f;
      isv[e] = MODE == 0 ? invstd[co + e] : 0.f;
      av[e] = MODE == 1 ? kA[co + e] : 0.f;
      bv[e] = MODE == 1 ? kB[co + e] : 0.f;
      cv[e] = MODE == 1 ? kC[co + e] : 0.f;
    }
    const size_t img0 = (size_t)g * (Wpg / (Ho * Wo));                 // first image of the group
    for (int w = w_lo + row; w < w_hi; w += rows) {
      const int n = w / (Ho * Wo), r = w - n * (Ho * Wo), yo = r / Wo, xo = r - yo * Wo;
      const size_t p0 = ((img0 + n) * H + 2 * yo) * W + 2 * xo;
      const size_t pix[4] = {p0, p0 + 1, p0 + W, p0 + W + 1};
      float4 z4[4], d4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        z4[i] = act_ld4f(zy + pix[i] * ld_z + cq * 4);
        d4[i] = act_ld4f(dy + pix[i] * ld_dy + cq * 4);
      }
      const float4 gp4 = act_ld4f(dp + ((size_t)g * Wpg + w) * ld_dp + cq * 4);
      const float gpv[4] = {gp4.x, gp4.y, gp4.z, gp4.w};
      float zv[4][4], dv[4][4], ov[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        zv[i][0] = z4[i].x; zv[i][1] = z4[i].y; zv[i][2] = z4[i].z; zv[i][3] = z4[i].w;
        dv[i][0] = d4[i].x; dv[i][1] = d4[i].y; dv[i][2] = d4[i].z; dv[i][3] = d4[i].w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float yv[4], pre[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (MODE == 2) { yv[i] = zv[i][e]; pre[i] = yv[i] > 0.f ? yv[i] : yv[i] * inv_slope; }
          else { pre[i] = pp_bn_pre(zv[i][e], scv[e], sfv[e]); yv[i] = pre[i] > 0.f ? pre[i] : pre[i] * slope; }
        }
        int k = 0;
        float m = yv[0];
        if (yv[1] > m) { m = yv[1]; k = 1; }
        if (yv[2] > m) { m = yv[2]; k = 2; }
        if (yv[3] > m) { m = yv[3]; k = 3; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d = dv[i][e] + (i == k ? gpv[e] : 0.f);
          const bool pos = MODE == 2 ? yv[i] > 0.f : pre[i] > 0.f;
          const float gg = pos ? d : d * slope;
          if (MODE == 0) { s1[e] += gg; s2[e] += gg * ((zv[i][e] - muv[e]) * isv[e]); }
          if (MODE == 1) { ov[i][e] = av[e] * gg + bv[e] * zv[i][e] + cv[e]; mx = fmaxf(mx, fabsf(ov[i][e])); }
          if (MODE == 2) { s1[e] += gg; s2[e] += gg * pre[i]; ov[i][e] = scv[e] * gg; mx = fmaxf(mx, fabsf(ov[i][e])); }
        }
      }
      if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          act_st4f(dz + pix[i] * ld_dz + cq * 4, make_float4(ov[i][0], ov[i][1], ov[i][2], ov[i][3]));
      }
    }
    if (MODE != 1) {
      float* d = sh + (row * c4n + cq) * 8;
#pragma unroll
      for (int e = 0; e < 4; ++e) { d[e] = s1[e]; d[4 + e] = s2[e]; }
    }
  }
  if (MODE != 1) {
    __syncthreads();
    for (int c = tid; c < C; c += NORM_THREADS) {
      double a = 0.0, b = 0.0;
      const int cq2 = c >> 2, e = c & 3;
      for (int r = 0; r < rows; ++r) {
        a += (double)sh[(r * c4n + cq2) * 8 + e];
        b += (double)sh[(r * c4n + cq2) * 8 + 4 + e];
      }
      double* o = partial + ((size_t)(g * gridDim.x + blk) * 2) * C;
      o[c] = a;
      o[C + c] = b;
    }
  }
  if (MODE != 0 && amax) {                      // max is order independent: the atomic keeps the result deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0 && mx > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(mx));
  }
}

static int bn_pool_check(const pp_act* dy, int ld_dy, const pp_act* dp, int ld_dp, const pp_act* dz, int ld_dz, int C, int B, int H,
                         int W, int groups) {
  PP_CHECK_ARG(dy && dp && dz, "bn_lrelu_bwd_pool: null pointer");
  PP_CHECK_ARG(H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && B > 0 && groups > 0 && B % groups == 0,
               "bn_lrelu_bwd_pool: H, W must be even and groups must divide the batch (B=%d H=%d W=%d groups=%d)", B, H, W, groups);
  PP_CHECK_ARG(ld_dy % 4 == 0 && ld_dp % 4 == 0 && ld_dz % 4 == 0 && ld_dy >= C && ld_dp >= C && ld_dz >= C, "bn_lrelu_bwd_pool: bad ld");
  PP_CHECK_ARG(((((uintptr_t)dy) | ((uintptr_t)dp) | ((uintptr_t)dz)) & PP_ACT_ALIGN) == 0, "bn_lrelu_bwd_pool: tensors must be 16-byte aligned");
  return 0;
}

// train-mode (or eval-statistics, training = 0) BatchNorm + LeakyReLU backward of a layer whose output also feeds a 2x2
// max-pooling: dy = gradient through the skip connection (B, H, W), dpool = gradient of the pooled tensor (B, H/2, W/2).
// Same outputs and workspace as pp_bn_lrelu_bwd_amax (dz_amax nullable).
extern "C" int PP_FN(pp_bn_lrelu_bwd_pool)(const pp_act* dy, int ld_dy, const pp_act* dpool, int ld_dpool, const pp_act* z, int ld_z,
                                    const float* scale, const float* shift, const float* save_mean, const float* save_invstd,
                                    const float* gamma, int training, pp_act* dz, int ld_dz, float* dgamma, float* dbeta,
                                    float* dbias_conv, int accumulate_param_grads, int C, int B, int H, int W, int groups,
                                    float slope, void* workspace, size_t workspace_bytes, float* dz_amax, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int Ppg = (B / (groups > 0 ? groups : 1)) * H * W;
  if (int rc = bn_check(z, ld_z, C, Ppg, groups)) return rc;
  if (int rc = bn_pool_check(dy, ld_dy, dpool, ld_dpool, dz, ld_dz, C, B, H, W, groups)) return rc;
  PP_CHECK_ARG(scale && shift && save_mean && save_invstd && gamma && workspace, "bn_lrelu_bwd_pool: null pointer");
  const int Wpg = Ppg / 4;
  const size_t need = pp_bn_workspace(C, Ppg, groups) + (size_t)3 * groups * C * sizeof(float);
  if (workspace_bytes < need) {
    pp_set_error("bn_lrelu_bwd_pool: workspace too small (%zu < %zu)", workspace_bytes, need);
    return PP_ERR_WORKSPACE;
  }
  ColPlan p = col_plan(C, Wpg, groups);               // windows play the role of pixels (never more blocks than col_plan(Ppg))
  double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  float* kA = reinterpret_cast<float*>(partial + (size_t)groups * col_plan(C, Ppg, groups).nblk * 2 * C);
  float* kB = kA + (size_t)groups * C;
  float* kC = kB + (size_t)groups * C;
  pp_prof_begin(PP_K_BN, 0.0, 22.0 * (double)groups * Ppg * C, s);
  hipLaunchKernelGGL(bn_bwd_pool_kernel<0>, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, dy, ld_dy, dpool, ld_dpool, z, ld_z,
                     scale, shift, save_mean, save_invstd, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     (act_t*)nullptr, 0, C, H, W, Wpg, p.chunk, p.rows, slope, 1.0f / slope, partial, (float*)nullptr);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(pp_cdiv(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, partial, p.nblk, C, Ppg,
                     groups, training, gamma, save_mean, save_invstd, kA, kB, kC, dgamma, dbeta, dbias_conv,
                     accumulate_param_grads, dz_amax);
  hipLaunchKernelGGL(bn_bwd_pool_kernel<1>, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, dy, ld_dy, dpool, ld_dpool, z, ld_z,
                     scale, shift, (const float*)nullptr, (const float*)nullptr, kA, kB, kC, dz, ld_dz, C, H, W, Wpg, p.chunk,
                     p.rows, slope, 1.0f / slope, (double*)nullptr, dz_amax);
  pp_prof_end(s);
  return pp_launch_status("bn_lrelu_bwd_pool");
}

// the eval-mode one-pass form (pp_bn_lrelu_bwd_eval) with the pooling gradient folded in: y is the layer's stored output
extern "C" int PP_FN(pp_bn_lrelu_bwd_eval_pool)(const pp_act* dy, int ld_dy, const pp_act* dpool, int ld_dpool, const pp_act* y, int ld_y,
                                         const float* scale, const float* gamma, const float* beta, pp_act* dz, int ld_dz,
                                         float* dgamma, float* dbeta, float* dbias_conv, int accumulate_param_grads, int C, int B,
                                         int H, int W, float slope, void* workspace, size_t workspace_bytes, float* dz_amax,
                                         void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int P = B * H * W;
  if (int rc = bn_check(y, ld_y, C, P, 1)) return rc;
  if (int rc = bn_pool_check(dy, ld_dy, dpool, ld_dpool, dz, ld_dz, C, B, H, W, 1)) return rc;
  PP_CHECK_ARG(scale && gamma && beta && workspace && slope > 0.f && ((uintptr_t)scale & 15) == 0, "bn_lrelu_bwd_eval_pool: bad arguments");
  if (workspace_bytes < pp_bn_workspace(C, P, 1)) {
    pp_set_error("bn_lrelu_bwd_eval_pool: workspace too small");
    return PP_ERR_WORKSPACE;
  }
  ColPlan p = col_plan(C, P / 4, 1);
  double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  if (dz_amax && hipMemsetAsync(dz_amax, 0, sizeof(float), s) != hipSuccess) return pp_launch_status("bn_bwd_eval_pool_memset");
  pp_prof_begin(PP_K_BN, 0.0, 13.0 * (double)P * C, s);
  hipLaunchKernelGGL(bn_bwd_pool_kernel<2>, dim3(p.nblk, 1), dim3(NORM_THREADS), 0, s, dy, ld_dy, dpool, ld_dpool, y, ld_y, scale,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, dz, ld_dz, C, H, W, P / 4, p.chunk, p.rows, slope, 1.0f / slope,
                     partial, dz_amax);
  hipLaunchKernelGGL(bn_bwd_eval_finalize_kernel, dim3(pp_cdiv(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, partial, p.nblk, C,
                     gamma, beta, scale, dgamma, dbeta, dbias_conv, accumulate_param_grads);
  pp_prof_end(s);
  return pp_launch_status("bn_lrelu_bwd_eval_pool");
}

// ---- y = lrelu(z * scale + shift) AND its 2x2 max-pooled copy in one pass (train-mode forward of an encoder stage's last layer) ----
// The stage output feeds the skip connection (y, written into its concatenation slot) and nn.MaxPool2d(2, 2) (models/unet.py:
// 109,123-127): maxpool2_fwd_kernel used to read y back right after bn_lrelu_fwd_kernel had written it.  Window-major like
// bn_bwd_pool_kernel: one thread normalises the four pixels of a window, stores them and their maximum.
__global__ __launch_bounds__(NORM_THREADS) void bn_lrelu_fwd_pool_kernel(const act_t* __restrict__ z, int ld_z, const float* __restrict__ scale,
                                                                         const float* __restrict__ shift, act_t* __restrict__ y, int ld_y,
                                                                         act_t* __restrict__ pooled, int ld_p, int C, int H, int W, int Wpg,
                                                                         int chunk, int rows, float slope) {
  const int c4n = C >> 2;
  const int tid = threadIdx.x;
  const int cq = tid % c4n, row = tid / c4n;
  if (row >= rows) return;
  const int g = blockIdx.y;
  const int w_lo = blockIdx.x * chunk;
  int w_hi = w_lo + chunk;
  if (w_hi > Wpg) w_hi = Wpg;
  const int Ho = H >> 1, Wo = W >> 1;
  const float4 sc = *reinterpret_cast<const float4*>(scale + g * C + cq * 4);
  const float4 sh = *reinterpret_cast<const float4*>(shift + g * C + cq * 4);
  const size_t img0 = (size_t)g * (Wpg / (Ho * Wo));
  for (int w = w_lo + row; w < w_hi; w += rows) {
    const int n = w / (Ho * Wo), r = w - n * (Ho * Wo), yo = r / Wo, xo = r - yo * Wo;
    const size_t p0 = ((img0 + n) * H + 2 * yo) * W + 2 * xo;
    const size_t pix[4] = {p0, p0 + 1, p0 + W, p0 + W + 1};
    float4 v[4], o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = act_ld4f(z + pix[i] * ld_z + cq * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[i].x = pp_lrelu(pp_bn_pre(v[i].x, sc.x, sh.x), slope);
      o[i].y = pp_lrelu(pp_bn_pre(v[i].y, sc.y, sh.y), slope);
      o[i].z = pp_lrelu(pp_bn_pre(v[i].z, sc.z, sh.z), slope);
      o[i].w = pp_lrelu(pp_bn_pre(v[i].w, sc.w, sh.w), slope);
      act_st4f(y + pix[i] * ld_y + cq * 4, o[i]);
    }
    float4 m;
    m.x = fmaxf(fmaxf(o[0].x, o[1].x), fmaxf(o[2].x, o[3].x));
    m.y = fmaxf(fmaxf(o[0].y, o[1].y), fmaxf(o[2].y, o[3].y));
    m.z = fmaxf(fmaxf(o[0].z, o[1].z), fmaxf(o[2].z, o[3].z));
    m.w = fmaxf(fmaxf(o[0].w, o[1].w), fmaxf(o[2].w, o[3].w));
    act_st4f(pooled + ((size_t)g * Wpg + w) * ld_p + cq * 4, m);
  }
}

extern "C" int PP_FN(pp_bn_lrelu_fwd_pool)(const pp_act* z, int ld_z, const float* scale, const float* shift, pp_act* y, int ld_y,
                                    pp_act* pooled, int ld_pooled, int C, int B, int H, int W, int groups, float slope, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(groups > 0 && B > 0 && B % groups == 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0,
               "bn_lrelu_fwd_pool: H, W must be even and groups must divide the batch");
  const int Ppg = (B / groups) * H * W;
  if (int rc = bn_check(z, ld_z, C, Ppg, groups)) return rc;
  PP_CHECK_ARG(scale && shift && y && pooled && ld_y % 4 == 0 && ld_y >= C && ld_pooled % 4 == 0 && ld_pooled >= C &&
                   ((((uintptr_t)y) | ((uintptr_t)pooled)) & PP_ACT_ALIGN) == 0, "bn_lrelu_fwd_pool: bad output");
  ColPlan p = col_plan(C, Ppg / 4, groups);
  pp_prof_begin(PP_K_BN, 0.0, 9.0 * (double)groups * Ppg * C, s);
  hipLaunchKernelGGL(bn_lrelu_fwd_pool_kernel, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, z, ld_z, scale, shift, y, ld_y, pooled,
                     ld_pooled, C, H, W, Ppg / 4, p.chunk, p.rows, slope);
  pp_prof_end(s);
  return pp_launch_status("bn_lrelu_fwd_pool");
}
// ---- BatchNorm + LeakyReLU backward of the FIRST layer with its weight gradient folded in (round 5) ----
// The network's first convolution has one input channel (grey-scale slices: every dataset of the reference) and nobody asks for
// its data gradient, so dz of that layer had exactly one reader: the weight gradient, the last kernel of every backward pass,
// alone on the chip behind the BatchNorm backward that had just written those 0.54 GB (benchmark shape: 32 channels at 256^2,
// 64 images; bn_bwd_apply 290 us -> conv3x3_c4_wgrad 242 us -> finalize -> optimizer).  Here the pass that forms dz keeps it in
// registers: a thread owns a channel quad and walks pixels as in bn_bwd_apply_kernel, loads the 3x3 neighbourhood of the
// one-channel input around each pixel (the eight threads of a pixel read the same nine words) and adds dz[o] * x[tap] into
// 4 x 9 accumulators; a block leaves one [C][9] partial, summed over blocks in a fixed order: dW[o][0][ky][kx] =
// sum_p dz[p][o] * x[p + (ky - 1, kx - 1)] (zero padding).  dz is never written.  EVAL: the one-pass eval-mode form on (dy, y).
struct Wg1Args { const act_t* x; int ld_x; int H, W; float* part; };

template <bool EVAL>
__global__ __launch_bounds__(NORM_THREADS) void bn_bwd_wg1_kernel(
    const act_t* __restrict__ dy, int ld_dy, const act_t* __restrict__ zy, int ld_z, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ kA, const float* __restrict__ kB, const float* __restrict__ kC,
    int C, int Ppg, int chunk, int rows, float slope, float inv_slope, double* __restrict__ partial, Wg1Args w) {
  __shared__ float shw[NORM_THREADS * 36];
  const int c4n = C >> 2;
  const int tid = threadIdx.x;
  const int cq = tid % c4n, row = tid / c4n;
  const bool active = row < rows;
  const int g = blockIdx.y, blk = blockIdx.x;
  const int p_lo = blk * chunk;
  int p_hi = p_lo + chunk;
  if (p_hi > Ppg) p_hi = Ppg;
  float acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.f;
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (active) {
    const int co = g * C + cq * 4;
    const float4 sc = *reinterpret_cast<const float4*>(scale + co);
    float4 sf = sc, a4 = sc, b4 = sc, c4 = sc;
    if (!EVAL) {
      sf = *reinterpret_cast<const float4*>(shift + co);
      a4 = *reinterpret_cast<const float4*>(kA + co);
      b4 = *reinterpret_cast<const float4*>(kB + co);
      c4 = *reinterpret_cast<const float4*>(kC + co);
    }
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, sfv[4] = {sf.x, sf.y, sf.z, sf.w};
    const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w}, cv[4] = {c4.x, c4.y, c4.z, c4.w};
    const size_t gb = (size_t)g * Ppg;
    const act_t* dyb = dy + gb * ld_dy + cq * 4;
    const act_t* zb = zy + gb * ld_z + cq * 4;
    const act_t* xb = w.x + gb * w.ld_x;
    int p = p_lo + row;
    int xx = p % w.W, yy = (p / w.W) % w.H;          // a group is whole images: the position inside the image follows from p
    const int dxx = rows % w.W, dyy = rows / w.W;    // advance by `rows` pixels
    auto step = [&]() __attribute__((always_inline)) {
      p += rows; xx += dxx; yy += dyy;
      if (xx >= w.W) { xx -= w.W; ++yy; }
      if (yy >= w.H) yy -= w.H;
      if (yy >= w.H) yy %= w.H;                      // (rows > H * W: tiny images)
    };
    auto load = [&](float4& d4, float4& z4, float (&xs)[9]) __attribute__((always_inline)) {
      d4 = act_ld4f(dyb + (size_t)p * ld_dy);
      z4 = act_ld4f(zb + (size_t)p * ld_z);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int oy = t / 3 - 1, ox = t % 3 - 1;
        const bool ok = (unsigned)(yy + oy) < (unsigned)w.H && (unsigned)(xx + ox) < (unsigned)w.W;
        const float v = act_ld1(xb + (size_t)(ok ? p + oy * w.W + ox : p) * w.ld_x);
        xs[t] = ok ? v : 0.f;
      }
    };
    auto consume = [&](const float4& d4, const float4& z4, const float (&xs)[9]) __attribute__((always_inline)) {
      const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float o;
        if (EVAL) {
          const bool pos = zv[e] > 0.f;
          const float gg = pos ? dv[e] : dv[e] * slope;
          const float pre = pos ? zv[e] : zv[e] * inv_slope;
          s1[e] += gg;
          s2[e] += gg * pre;
          o = scv[e] * gg;
        } else {
          o = av[e] * (pp_bn_pre(zv[e], scv[e], sfv[e]) > 0.f ? dv[e] : dv[e] * slope) + bv[e] * zv[e] + cv[e];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[e * 9 + t] = __builtin_fmaf(o, xs[t], acc[e * 9 + t]);
      }
    };
    float4 d0, z0, d1, z1;
    float x0[9], x1[9];
    while (p + rows < p_hi) {                        // two pixels per round: four 16-byte loads in flight per lane
      load(d0, z0, x0); step();
      load(d1, z1, x1); step();
      consume(d0, z0, x0);
      consume(d1, z1, x1);
    }
    if (p < p_hi) {
      load(d0, z0, x0);
      consume(d0, z0, x0);
    }
#pragma unroll
    for (int k = 0; k < 36; ++k) shw[(row * c4n + cq) * 36 + k] = acc[k];
  }
  __syncthreads();
  const int n_out = C * 9;
  for (int idx = tid; idx < n_out; idx += NORM_THREADS) {
    const int o = idx / 9, t = idx - o * 9, cq2 = o >> 2, e = o & 3;
    double a = 0.0;
    for (int r = 0; r < rows; ++r) a += (double)shw[(r * c4n + cq2) * 36 + e * 9 + t];
    w.part[(size_t)(g * gridDim.x + blk) * n_out + idx] = (float)a;
  }
  if (EVAL) {                                        // s1 / s2 rows as bn_bwd_eval_kernel writes them
    __syncthreads();
    if (active) {
      float* d = shw + (row * c4n + cq) * 8;
#pragma unroll
      for (int e = 0; e < 4; ++e) { d[e] = s1[e]; d[4 + e] = s2[e]; }
    }
    __syncthreads();
    for (int c = tid; c < C; c += NORM_THREADS) {
      double a = 0.0, b = 0.0;
      const int cq2 = c >> 2, e = c & 3;
      for (int r = 0; r < rows; ++r) {
        a += (double)shw[(r * c4n + cq2) * 8 + e];
        b += (double)shw[(r * c4n + cq2) * 8 + 4 + e];
      }
      double* o = partial + ((size_t)blk * 2) * C;
      o[c] = a;
      o[C + c] = b;
    }
  }
}

// dw[idx] (+)= sum over the per-block partials, 16 outputs x 64 slices per block, fixed order
__global__ __launch_bounds__(FIN_CH * FIN_SL) void wg1_finalize_kernel(const float* __restrict__ part, int nslab, int n_out,
                                                                      float* __restrict__ dw, int accumulate) {
  __shared__ double red[FIN_SL][FIN_CH];
  const int cl = threadIdx.x & (FIN_CH - 1), slice = threadIdx.x / FIN_CH;
  const int idx = blockIdx.x * FIN_CH + cl;
  double a = 0.0;
  if (idx < n_out)
    for (int sl = slice; sl < nslab; sl += FIN_SL) a += (double)part[(size_t)sl * n_out + idx];
  red[slice][cl] = a;
  __syncthreads();
  if (slice == 0 && idx < n_out) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < FIN_SL; ++i) t += red[i][cl];
    dw[idx] = (accumulate ? dw[idx] : 0.f) + (float)t;
  }
}

// grid of the folded pass: more, smaller blocks than the streaming BatchNorm kernels use (col_plan: 2 per CU) -- it carries 36
// accumulators and nine neighbour loads per pixel, so it needs the occupancy to hide their latency
static ColPlan wg1_plan(int C, int Ppg, int groups) {
  ColPlan p = col_plan(C, Ppg, groups);
  constexpr int blocks = 1024;                      // r05 microbenchmark: 512 -> 489 us, 1024 -> 407, 2048 -> 410
  int target = blocks / groups;
  if (target < 1) target = 1;
  p.chunk = pp_cdiv(Ppg, target);
  if (p.chunk < p.rows * 8) p.chunk = p.rows * 8;
  p.chunk = pp_cdiv(p.chunk, p.rows) * p.rows;
  p.nblk = pp_cdiv(Ppg, p.chunk);
  return p;
}

extern "C" size_t PP_FN(pp_bn_lrelu_bwd_wgrad_c1_workspace)(int C, int P_per_group, int groups) {
  ColPlan p = col_plan(C, P_per_group, groups), q = wg1_plan(C, P_per_group, groups);
  const int nb = p.nblk > q.nblk ? p.nblk : q.nblk;
  return (size_t)groups * nb * 2 * C * sizeof(double) + 256 + (size_t)3 * groups * C * sizeof(float) +
         (size_t)groups * q.nblk * C * 9 * sizeof(float) + 64;
}

static int wg1_check(const void* x, int ld_x, int H, int W, int P_per_group, const float* dw, int C, const void* workspace,
                     size_t workspace_bytes, size_t need) {
  PP_CHECK_ARG(x && dw && workspace && ld_x >= 1 && H > 0 && W > 0 && P_per_group % (H * W) == 0,
               "bn_lrelu_bwd_wgrad_c1: x / dw / workspace null, or a group is not a whole number of H x W images");
  PP_CHECK_ARG(C % 4 == 0 && C <= 4 * NORM_THREADS, "bn_lrelu_bwd_wgrad_c1: C=%d", C);
  if (workspace_bytes < need) {
    pp_set_error("bn_lrelu_bwd_wgrad_c1: workspace too small (%zu < %zu)", workspace_bytes, need);
    return PP_ERR_WORKSPACE;
  }
  return 0;
}

extern "C" int PP_FN(pp_bn_lrelu_bwd_wgrad_c1)(const pp_act* dy, int ld_dy, const pp_act* z, int ld_z, const float* scale,
                                        const float* shift, const float* save_mean, const float* save_invstd,
                                        const float* gamma, int training, const pp_act* x, int ld_x, int H, int W,
                                        float* dw_o1hw, int accumulate_dw, float* dgamma, float* dbeta, float* dbias_conv,
                                        int accumulate_param_grads, int C, int P_per_group, int groups, float slope,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = bn_check(z, ld_z, C, P_per_group, groups)) return rc;
  PP_CHECK_ARG(dy && scale && shift && save_mean && save_invstd && gamma, "bn_lrelu_bwd_wgrad_c1: null pointer");
  PP_CHECK_ARG(ld_dy % 4 == 0 && ld_dy >= C && ((uintptr_t)dy & PP_ACT_ALIGN) == 0, "bn_lrelu_bwd_wgrad_c1: bad dy");
  if (int rc = wg1_check(x, ld_x, H, W, P_per_group, dw_o1hw, C, workspace, workspace_bytes,
                         PP_FN(pp_bn_lrelu_bwd_wgrad_c1_workspace)(C, P_per_group, groups))) return rc;
  ColPlan p = col_plan(C, P_per_group, groups), q = wg1_plan(C, P_per_group, groups);
  double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  float* kA = reinterpret_cast<float*>(partial + (size_t)groups * (p.nblk > q.nblk ? p.nblk : q.nblk) * 2 * C);
  float* kB = kA + (size_t)groups * C;
  float* kC = kB + (size_t)groups * C;
  float* part = kC + (size_t)groups * C;
  pp_prof_begin(PP_K_BN, 0.0, 16.0 * (double)groups * P_per_group * C, s);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(p.nblk, groups), dim3(NORM_THREADS), 0, s, dy, ld_dy, z, ld_z, scale,
                     shift, save_mean, save_invstd, C, P_per_group, p.chunk, p.rows, slope, partial);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(pp_cdiv(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, partial, p.nblk, C, P_per_group,
                     groups, training, gamma, save_mean, save_invstd, kA, kB, kC, dgamma, dbeta, dbias_conv,
                     accumulate_param_grads, (float*)nullptr);
  hipLaunchKernelGGL(bn_bwd_wg1_kernel<false>, dim3(q.nblk, groups), dim3(NORM_THREADS), 0, s, dy, ld_dy, z, ld_z, scale, shift,
                     kA, kB, kC, C, P_per_group, q.chunk, q.rows, slope, 1.0f / slope, (double*)nullptr, Wg1Args{x, ld_x, H, W, part});
  hipLaunchKernelGGL(wg1_finalize_kernel, dim3(pp_cdiv(C * 9, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, part, groups * q.nblk, C * 9,
                     dw_o1hw, accumulate_dw);
  pp_prof_end(s);
  return pp_launch_status("bn_lrelu_bwd_wgrad_c1");
}

// the eval-mode one-pass form (pp_bn_lrelu_bwd_eval) with the same fold: y is the layer's stored output
extern "C" int PP_FN(pp_bn_lrelu_bwd_eval_wgrad_c1)(const pp_act* dy, int ld_dy, const pp_act* y, int ld_y, const float* scale,
                                             const float* gamma, const float* beta, const pp_act* x, int ld_x, int H, int W,
                                             float* dw_o1hw, int accumulate_dw, float* dgamma, float* dbeta,
                                             float* dbias_conv, int accumulate_param_grads, int C, int P_total, float slope,
                                             void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = bn_check(y, ld_y, C, P_total, 1)) return rc;
  PP_CHECK_ARG(dy && scale && gamma && beta && slope > 0.f, "bn_lrelu_bwd_eval_wgrad_c1: null pointer / slope");
  PP_CHECK_ARG(ld_dy % 4 == 0 && ld_dy >= C && ((uintptr_t)dy & PP_ACT_ALIGN) == 0 && ((uintptr_t)scale & 15) == 0,
               "bn_lrelu_bwd_eval_wgrad_c1: bad dy / scale");
  if (int rc = wg1_check(x, ld_x, H, W, P_total, dw_o1hw, C, workspace, workspace_bytes,
                         PP_FN(pp_bn_lrelu_bwd_wgrad_c1_workspace)(C, P_total, 1))) return rc;
  ColPlan p0 = col_plan(C, P_total, 1), p = wg1_plan(C, P_total, 1);
  double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  float* part = reinterpret_cast<float*>(partial + (size_t)(p.nblk > p0.nblk ? p.nblk : p0.nblk) * 2 * C) + (size_t)3 * C;
  pp_prof_begin(PP_K_BN, 0.0, 8.0 * (double)P_total * C, s);
  hipLaunchKernelGGL(bn_bwd_wg1_kernel<true>, dim3(p.nblk, 1), dim3(NORM_THREADS), 0, s, dy, ld_dy, y, ld_y, scale, scale, scale,
                     scale, scale, C, P_total, p.chunk, p.rows, slope, 1.0f / slope, partial, Wg1Args{x, ld_x, H, W, part});
  hipLaunchKernelGGL(bn_bwd_eval_finalize_kernel, dim3(pp_cdiv(C, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, partial, p.nblk, C,
                     gamma, beta, scale, dgamma, dbeta, dbias_conv, accumulate_param_grads);
  hipLaunchKernelGGL(wg1_finalize_kernel, dim3(pp_cdiv(C * 9, FIN_CH)), dim3(FIN_CH * FIN_SL), 0, s, part, p.nblk, C * 9, dw_o1hw,
                     accumulate_dw);
  pp_prof_end(s);
  return pp_launch_status("bn_lrelu_bwd_eval_wgrad_c1");
}
PP_NS_END
