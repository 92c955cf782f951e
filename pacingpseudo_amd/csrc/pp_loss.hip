// Loss-side kernels of the PacingPseudo step (all HBM/latency-bound, NCHW logits with K <= 8 classes):
//   * channel arg-max (scribble one-hot -> int64 target, prediction masks)  consistency_reglur_memory.py:31
//   * fused partial-CE + entropy-minimisation + decoder-consistency sums     losses/losses.py:9-116
//   * their gradient wrt the weak and strong logits
//   * auxiliary head: bilinear x(H/h) up-sampling of the low-res logits fused with partial-CE, and the
//     gather-form gradient back to the low-res logits                        aux_path_memory.py:52, losses.py:43
//   * class-prototype memory bank update (batch sample 0 only) and the bank classification CE
//                                                                            aux_path_memory.py:61,68-116
//   * validation Dice counts                                                 utils/metrics.py:7-34
// Reductions are two-stage (per-block partials, fixed-order double finalize): deterministic, no float atomics.
#include "pp_common.h"

#define LS_THREADS 256
#define LS_MAXK 8

// ---------------------------------------------------------------- arg-max over channels (first maximum wins)
__global__ void argmax_channels_kernel(const float* __restrict__ x, int N, int C, int HW, long long* __restrict__ out) {
  const long long P = (long long)N * HW;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
    // (32-bit division where the pixel index fits: the 64-bit form is ~150 VALU instructions of these kernels' ~870 per pixel)
    const int n = p < 0x7fffffffLL ? (int)((unsigned)p / (unsigned)HW) : (int)(p / HW), hw = (int)(p - (long long)n * HW);
    const float* b = x + (size_t)n * C * HW + hw;
    float m = b[0];
    int k = 0;
    for (int c = 1; c < C; ++c) {
      const float v = b[(size_t)c * HW];
      if (v > m) { m = v; k = c; }
    }
    out[p] = k;
  }
}

static inline int ls_blocks(long long total, int cap = 4096) {
  int b = pp_cdiv(total, LS_THREADS);
  return b > cap ? cap : (b < 1 ? 1 : b);
}

extern "C" int pp_argmax_channels(const float* x, int N, int C, int HW, int64_t* out, void* stream) {
  PP_CHECK_ARG(x && out && N > 0 && C > 0 && HW > 0, "argmax_channels: bad arguments");
  hipLaunchKernelGGL(argmax_channels_kernel, dim3(ls_blocks((long long)N * HW)), dim3(LS_THREADS), 0,
                     (hipStream_t)stream, x, N, C, HW, (long long*)out);
  return pp_launch_status("argmax_channels");
}

// ---------------------------------------------------------------- softmax helpers
struct SM { float p[LS_MAXK]; float l[LS_MAXK]; };   // softmax and log-softmax of one pixel
// softmax / log-softmax of one pixel from its K logits (the one expression every loss kernel uses: the scalar and the
// four-pixels-per-thread forms of a kernel give the same bits)
__device__ __forceinline__ void softmax_vals(const float (&v)[LS_MAXK], int K, SM& o) {
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k)
    if (k < K) m = fmaxf(m, v[k]);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k)
    if (k < K) { o.p[k] = expf(v[k] - m); s += o.p[k]; }
  const float ls = logf(s), inv = 1.f / s;
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k)
    if (k < K) { o.l[k] = v[k] - m - ls; o.p[k] *= inv; }
}
__device__ __forceinline__ void pixel_softmax(const float* __restrict__ z, size_t stride, int K, SM& o) {
  float v[LS_MAXK];
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k)
    if (k < K) v[k] = z[(size_t)k * stride];
  softmax_vals(v, K, o);
}

// consistency variants (train_chaos.py:138): 0 none, 1 ce_loss, 2 l1_loss, 3 l2_loss, 4 kl_loss
__device__ __forceinline__ float cr_value(const SM& w, const SM& s, int K, int variant) {
  float L = 0.f;
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k)
    if (k < K) {
      if (variant == 1) L -= w.p[k] * s.l[k];
      else if (variant == 2) L += fabsf(s.p[k] - w.p[k]);
      else if (variant == 3) { const float d = s.p[k] - w.p[k]; L += d * d; }
      else L += w.p[k] * (w.l[k] - s.l[k]);
    }
  return L;
}

// sums[0]=pce_sum [1]=n_labelled [2]=ent_sum [3]=ent_den [4]=cr_sum [5]=cr_den   (double)
struct SegAcc { float pce, n, ent, cr, m; };
__device__ __forceinline__ void seg_fwd_pixel(const float (&vw)[LS_MAXK], const float (&vs)[LS_MAXK], long long t, float m, int K,
                                              int ignore_index, int do_ent, int variant, SegAcc& a) {
  SM w;
  softmax_vals(vw, K, w);
  if (t != ignore_index && t >= 0 && t < K) {
    float lt = 0.f;
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k) if (k == (int)t) lt = w.l[k];
    a.pce -= lt;
    a.n += 1.f;
  }
  a.m += m;
  if (do_ent) {
    float h = 0.f;
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k) if (k < K) h -= w.p[k] * w.l[k];
    a.ent += h * m;
  }
  if (variant) {
    SM s;
    softmax_vals(vs, K, s);
    a.cr += cr_value(w, s, K, variant) * m;
  }
}

__device__ __forceinline__ void seg_acc_store(const SegAcc& a, float* sh, double* __restrict__ partial) {
  const float r0 = pp_block_sum(a.pce, sh), r1 = pp_block_sum(a.n, sh), r2 = pp_block_sum(a.ent, sh);
  const float r3 = pp_block_sum(a.cr, sh), r4 = pp_block_sum(a.m, sh);
  if (threadIdx.x == 0) {
    double* o = partial + (size_t)blockIdx.x * 5;
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; o[4] = r4;
  }
}

__global__ __launch_bounds__(LS_THREADS) void seg_losses_partial_kernel(
    const float* __restrict__ zw, const float* __restrict__ zs, const long long* __restrict__ target,
    const float* __restrict__ mask, int N, int K, int HW, int ignore_index, int do_ent, int variant,
    double* __restrict__ partial /*[blocks][5]*/) {
  __shared__ float sh[16];
  const long long P = (long long)N * HW;
  SegAcc a{0.f, 0.f, 0.f, 0.f, 0.f};
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
    // (32-bit division where the pixel index fits: the 64-bit form is ~150 VALU instructions of these kernels' ~870 per pixel)
    const int n = p < 0x7fffffffLL ? (int)((unsigned)p / (unsigned)HW) : (int)(p / HW), hw = (int)(p - (long long)n * HW);
    const size_t off = (size_t)n * K * HW + hw;
    float vw[LS_MAXK], vs[LS_MAXK];
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k) {
      vw[k] = k < K ? zw[off + (size_t)k * HW] : 0.f;
      vs[k] = (k < K && variant) ? zs[off + (size_t)k * HW] : 0.f;
    }
    seg_fwd_pixel(vw, vs, target[p], mask ? mask[p] : 1.f, K, ignore_index, do_ent, variant, a);
  }
  seg_acc_store(a, sh, partial);
}

// Four consecutive pixels per thread, K a compile-time constant (round 6): 16-byte loads of every logit plane, the class map
// and the mask, all issued before the first use -- the one-pixel form above ran at 1.2 - 1.7 TB/s of its bytes (r05 profile:
// two dependent rounds of ten 4-byte loads per thread).  The per-pixel arithmetic is the same function in the same order; the
// per-thread and per-block partial sums group the pixels differently, so the SUMS agree to rounding, not bit for bit.
// Host side: HW % 4 == 0, pointers 16-byte aligned, K in {2, 4, 5} (the three data sets), P < 2^31.
template <int KC>
__global__ __launch_bounds__(LS_THREADS) void seg_losses_partial_v4_kernel(
    const float* __restrict__ zw, const float* __restrict__ zs, const long long* __restrict__ target,
    const float* __restrict__ mask, int N, int HW, int ignore_index, int do_ent, int variant,
    double* __restrict__ partial /*[blocks][5]*/) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef long long i64x2 __attribute__((ext_vector_type(2)));
  __shared__ float sh[16];
  constexpr int K = KC;
  const int Q = (int)(((long long)N * HW) >> 2), HW4 = HW >> 2;
  SegAcc a{0.f, 0.f, 0.f, 0.f, 0.f};
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += gridDim.x * blockDim.x) {
    const int n = (int)((unsigned)q / (unsigned)HW4), hw = (q - n * HW4) << 2;
    const size_t off = (size_t)n * K * HW + hw;
    f32x4 w4[K], s4[K];
#pragma unroll
    for (int k = 0; k < K; ++k) w4[k] = *reinterpret_cast<const f32x4*>(zw + off + (size_t)k * HW);
    if (variant) {
#pragma unroll
      for (int k = 0; k < K; ++k) s4[k] = *reinterpret_cast<const f32x4*>(zs + off + (size_t)k * HW);
    }
    const i64x2 t01 = *reinterpret_cast<const i64x2*>(target + (size_t)q * 4), t23 = *reinterpret_cast<const i64x2*>(target + (size_t)q * 4 + 2);
    const f32x4 m4 = mask ? *reinterpret_cast<const f32x4*>(mask + (size_t)q * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float vw[LS_MAXK], vs[LS_MAXK];
#pragma unroll
      for (int k = 0; k < LS_MAXK; ++k) { vw[k] = k < K ? w4[k < K ? k : 0][j] : 0.f; vs[k] = (k < K && variant) ? s4[k < K ? k : 0][j] : 0.f; }
      seg_fwd_pixel(vw, vs, j < 2 ? t01[j] : t23[j - 2], m4[j], K, ignore_index, do_ent, variant, a);
    }
  }
  seg_acc_store(a, sh, partial);
}

__global__ __launch_bounds__(64) void seg_losses_reduce_kernel(const double* __restrict__ partial, int nblocks,
                                                               int has_mask, int variant, double unmasked_ent_den,
                                                               double unmasked_cr_den, double* __restrict__ sums) {
  double a[5] = {0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < nblocks; b += 64)
    for (int j = 0; j < 5; ++j) a[j] += partial[(size_t)b * 5 + j];
  for (int j = 0; j < 5; ++j) a[j] = pp_wave_sum_d(a[j]);
  if (threadIdx.x != 0) return;
  sums[0] = a[0];
  sums[1] = a[1];
  sums[2] = a[2];
  sums[3] = has_mask ? a[4] : unmasked_ent_den;
  sums[4] = a[3];
  sums[5] = has_mask ? a[4] : unmasked_cr_den;
  (void)variant;
}

// loss = sum / den;   masked denominators are clamped as max(mask.sum(), 1e-8)  (losses/losses.py:21,59)
__global__ void losses_finalize_kernel(const double* __restrict__ sums, int has_mask, float* loss_pce, float* loss_ent,
                                       float* loss_cr) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (loss_pce) *loss_pce = (float)(sums[0] / sums[1]);              // 0/0 -> NaN like F.cross_entropy
  const double de = has_mask ? fmax(sums[3], 1e-8) : sums[3];
  const double dc = has_mask ? fmax(sums[5], 1e-8) : sums[5];
  if (loss_ent) *loss_ent = (float)(sums[2] / de);
  if (loss_cr) *loss_cr = (float)(sums[4] / dc);
}

// the four-pixels-per-thread forms: 16-byte accesses of every tensor, 32-bit pixel-quad indices
static inline bool seg_v4_ok(const void* a, const void* b, const void* c, const void* d, const void* e, const void* f, int N, int K,
                             int HW) {
  const uintptr_t bits = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e | (uintptr_t)f;
  return HW % 4 == 0 && (bits & 15) == 0 && (long long)N * HW * K < 0x7fffffffLL && (K == 2 || K == 4 || K == 5);
}

extern "C" size_t pp_seg_losses_workspace(int N, int HW) {
  return (size_t)ls_blocks((long long)N * HW, 1024) * 5 * sizeof(double);
}

extern "C" int pp_seg_losses_fwd(const float* logits_w, const float* logits_s, const int64_t* target,
                                 const float* valid_mask, int N, int K, int HW, int ignore_index, int do_ent,
                                 int cr_variant, double* sums, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(logits_w && target && sums && workspace, "seg_losses_fwd: null pointer");
  PP_CHECK_ARG(K >= 1 && K <= LS_MAXK && cr_variant >= 0 && cr_variant <= 4, "seg_losses_fwd: K=%d variant=%d", K, cr_variant);
  PP_CHECK_ARG(cr_variant == 0 || logits_s, "seg_losses_fwd: consistency loss needs the strong logits");
  if (workspace_bytes < pp_seg_losses_workspace(N, HW)) {
    pp_set_error("seg_losses_fwd: workspace too small");
    return PP_ERR_WORKSPACE;
  }
  const int blocks = ls_blocks((long long)N * HW, 1024);
  const double P = (double)N * HW;
  const double den_ent = P * K;                                       // loss.mean() over (N,K,H,W)
  const double den_cr = (cr_variant == 2 || cr_variant == 3) ? P : P * K;   // l1/l2 reduce channels first
  pp_prof_begin(PP_K_LOSS, 0.0, P * (8.0 * K + 12.0), s);
  const bool v4 = seg_v4_ok(logits_w, logits_s, target, valid_mask, nullptr, nullptr, N, K, HW);
#define SEG_FWD_V4(KC) hipLaunchKernelGGL(seg_losses_partial_v4_kernel<KC>, dim3(blocks), dim3(LS_THREADS), 0, s, logits_w, logits_s, \
                     (const long long*)target, valid_mask, N, HW, ignore_index, do_ent, cr_variant, (double*)workspace)
  if (v4 && K == 5) SEG_FWD_V4(5);
  else if (v4 && K == 4) SEG_FWD_V4(4);
  else if (v4 && K == 2) SEG_FWD_V4(2);
  else
    hipLaunchKernelGGL(seg_losses_partial_kernel, dim3(blocks), dim3(LS_THREADS), 0, s, logits_w, logits_s,
                       (const long long*)target, valid_mask, N, K, HW, ignore_index, do_ent, cr_variant, (double*)workspace);
#undef SEG_FWD_V4
  hipLaunchKernelGGL(seg_losses_reduce_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, blocks,
                     valid_mask ? 1 : 0, cr_variant, den_ent, den_cr, sums);
  pp_prof_end(s);
  return pp_launch_status("seg_losses_fwd");
}

extern "C" int pp_losses_finalize(const double* sums, int has_mask, float* loss_pce, float* loss_ent, float* loss_cr,
                                  void* stream) {
  PP_CHECK_ARG(sums != nullptr, "losses_finalize: null pointer");
  hipLaunchKernelGGL(losses_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, has_mask, loss_pce,
                     loss_ent, loss_cr);
  return pp_launch_status("losses_finalize");
}

// d(sum_i g_i * loss_i)/d logits.  For a per-pixel loss L(q, s) of the weak / strong probabilities with
// partials u = dL/dq, v = dL/ds:  dL/dz_w[k] = q_k (u_k - sum_c q_c u_c),  dL/dz_s[k] = s_k (v_k - sum_c s_c v_c).
struct SegG { float gp, ge, gc; };
__device__ __forceinline__ SegG seg_bwd_scales(const double* __restrict__ sums, int has_mask, int do_ent, int variant,
                                               const float* g_pce, const float* g_ent, const float* g_cr, float grad_scale) {
  SegG g;
  g.gp = (g_pce ? *g_pce : 0.f) * grad_scale / (float)sums[1];
  const double de = has_mask ? fmax(sums[3], 1e-8) : sums[3];
  const double dc = has_mask ? fmax(sums[5], 1e-8) : sums[5];
  g.ge = (do_ent && g_ent) ? (float)((double)(*g_ent * grad_scale) / de) : 0.f;
  g.gc = (variant && g_cr) ? (float)((double)(*g_cr * grad_scale) / dc) : 0.f;
  return g;
}
__device__ __forceinline__ void seg_bwd_pixel(const float (&vw)[LS_MAXK], const float (&vs)[LS_MAXK], long long t, float m, int K,
                                              int ignore_index, int do_ent, int variant, int detach_weak, const SegG& g,
                                              float (&dw)[LS_MAXK], float (&ds)[LS_MAXK]) {
  SM w;
  softmax_vals(vw, K, w);
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k) { dw[k] = 0.f; ds[k] = 0.f; }
  if (t != ignore_index && t >= 0 && t < K) {
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k)
      if (k < K) dw[k] = g.gp * (w.p[k] - (k == (int)t ? 1.f : 0.f));
  }
  if (do_ent) {
    float h = 0.f;
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k) if (k < K) h -= w.p[k] * w.l[k];
    const float f = g.ge * m;
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k) if (k < K) dw[k] -= f * w.p[k] * (w.l[k] + h);
  }
  if (variant) {
    SM s;
    softmax_vals(vs, K, s);
    float u[LS_MAXK], v[LS_MAXK];
    float qu = 0.f, sv = 0.f;
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k)
      if (k < K) {
        if (variant == 1) { u[k] = -s.l[k]; v[k] = 0.f; }
        else if (variant == 2) {
          const float d = s.p[k] - w.p[k];
          const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
          u[k] = -sg; v[k] = sg;
        } else if (variant == 3) { const float d = s.p[k] - w.p[k]; u[k] = -2.f * d; v[k] = 2.f * d; }
        else { u[k] = w.l[k] - s.l[k] + 1.f; v[k] = 0.f; }
        qu += w.p[k] * u[k];
        sv += s.p[k] * v[k];
      }
    const float f = g.gc * m;
    const bool weak_grad = !(detach_weak && variant != 4);    // kl_loss reads the logits, never detached
    // ce_loss / kl_loss: v = -q/s, so s_k (v_k - sum_c s_c v_c) = s_k - q_k.  The closed form is used because the
    // quotient form is 0 * inf = NaN once a strong-view probability underflows (logit gap > 87): found by the r02
    // multi-seed Dice runs, which died with NaN gradients after ~550 steps; torch's log_softmax backward
    // (the reference, losses/losses.py:54-59) is the closed form too.
    const bool closed = variant == 1 || variant == 4;
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k)
      if (k < K) {
        if (weak_grad) dw[k] += f * w.p[k] * (u[k] - qu);
        ds[k] = closed ? f * (s.p[k] - w.p[k]) : f * s.p[k] * (v[k] - sv);
      }
  }
}

__global__ __launch_bounds__(LS_THREADS) void seg_losses_bwd_kernel(
    const float* __restrict__ zw, const float* __restrict__ zs, const long long* __restrict__ target,
    const float* __restrict__ mask, int N, int K, int HW, int ignore_index, int do_ent, int variant, int detach_weak,
    const double* __restrict__ sums, int has_mask, const float* __restrict__ g_pce, const float* __restrict__ g_ent,
    const float* __restrict__ g_cr, float grad_scale, float* __restrict__ dzw, float* __restrict__ dzs) {
  const long long P = (long long)N * HW;
  const SegG g = seg_bwd_scales(sums, has_mask, do_ent, variant, g_pce, g_ent, g_cr, grad_scale);
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
    // (32-bit division where the pixel index fits: the 64-bit form is ~150 VALU instructions of these kernels' ~870 per pixel)
    const int n = p < 0x7fffffffLL ? (int)((unsigned)p / (unsigned)HW) : (int)(p / HW), hw = (int)(p - (long long)n * HW);
    const size_t off = (size_t)n * K * HW + hw;
    float vw[LS_MAXK], vs[LS_MAXK], dw[LS_MAXK], ds[LS_MAXK];
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k) {
      vw[k] = k < K ? zw[off + (size_t)k * HW] : 0.f;
      vs[k] = (k < K && variant) ? zs[off + (size_t)k * HW] : 0.f;
    }
    seg_bwd_pixel(vw, vs, target[p], mask ? mask[p] : 1.f, K, ignore_index, do_ent, variant, detach_weak, g, dw, ds);
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k)
      if (k < K) {
        if (variant) dzs[off + (size_t)k * HW] = ds[k];
        dzw[off + (size_t)k * HW] = dw[k];
      }
  }
}

// four consecutive pixels per thread, compile-time K (see seg_losses_partial_v4_kernel): bit-identical gradients
template <int KC>
__global__ __launch_bounds__(LS_THREADS) void seg_losses_bwd_v4_kernel(
    const float* __restrict__ zw, const float* __restrict__ zs, const long long* __restrict__ target,
    const float* __restrict__ mask, int N, int HW, int ignore_index, int do_ent, int variant, int detach_weak,
    const double* __restrict__ sums, int has_mask, const float* __restrict__ g_pce, const float* __restrict__ g_ent,
    const float* __restrict__ g_cr, float grad_scale, float* __restrict__ dzw, float* __restrict__ dzs) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef long long i64x2 __attribute__((ext_vector_type(2)));
  constexpr int K = KC;
  const int Q = (int)(((long long)N * HW) >> 2), HW4 = HW >> 2;
  const SegG g = seg_bwd_scales(sums, has_mask, do_ent, variant, g_pce, g_ent, g_cr, grad_scale);
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += gridDim.x * blockDim.x) {
    const int n = (int)((unsigned)q / (unsigned)HW4), hw = (q - n * HW4) << 2;
    const size_t off = (size_t)n * K * HW + hw;
    f32x4 w4[K], s4[K], dw4[K], ds4[K];
#pragma unroll
    for (int k = 0; k < K; ++k) w4[k] = *reinterpret_cast<const f32x4*>(zw + off + (size_t)k * HW);
    if (variant) {
#pragma unroll
      for (int k = 0; k < K; ++k) s4[k] = *reinterpret_cast<const f32x4*>(zs + off + (size_t)k * HW);
    }
    const i64x2 t01 = *reinterpret_cast<const i64x2*>(target + (size_t)q * 4), t23 = *reinterpret_cast<const i64x2*>(target + (size_t)q * 4 + 2);
    const f32x4 m4 = mask ? *reinterpret_cast<const f32x4*>(mask + (size_t)q * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float vw[LS_MAXK], vs[LS_MAXK], dw[LS_MAXK], ds[LS_MAXK];
#pragma unroll
      for (int k = 0; k < LS_MAXK; ++k) { vw[k] = k < K ? w4[k < K ? k : 0][j] : 0.f; vs[k] = (k < K && variant) ? s4[k < K ? k : 0][j] : 0.f; }
      seg_bwd_pixel(vw, vs, j < 2 ? t01[j] : t23[j - 2], m4[j], K, ignore_index, do_ent, variant, detach_weak, g, dw, ds);
#pragma unroll
      for (int k = 0; k < K; ++k) { dw4[k][j] = dw[k]; ds4[k][j] = ds[k]; }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (variant) *reinterpret_cast<f32x4*>(dzs + off + (size_t)k * HW) = ds4[k];
      *reinterpret_cast<f32x4*>(dzw + off + (size_t)k * HW) = dw4[k];
    }
  }
}

extern "C" int pp_seg_losses_bwd(const float* logits_w, const float* logits_s, const int64_t* target,
                                 const float* valid_mask, int N, int K, int HW, int ignore_index, int do_ent,
                                 int cr_variant, int detach_weak, const double* sums, const float* g_pce,
                                 const float* g_ent, const float* g_cr, float grad_scale, float* dlogits_w,
                                 float* dlogits_s, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(logits_w && target && sums && dlogits_w, "seg_losses_bwd: null pointer");
  PP_CHECK_ARG(K >= 1 && K <= LS_MAXK && cr_variant >= 0 && cr_variant <= 4, "seg_losses_bwd: K=%d variant=%d", K, cr_variant);
  PP_CHECK_ARG(cr_variant == 0 || (logits_s && dlogits_s), "seg_losses_bwd: consistency loss needs the strong logits");
  const double P = (double)N * HW;
  pp_prof_begin(PP_K_LOSS, 0.0, P * (16.0 * K + 12.0), s);
  const bool v4 = seg_v4_ok(logits_w, logits_s, target, valid_mask, dlogits_w, dlogits_s, N, K, HW);
#define SEG_BWD_V4(KC) hipLaunchKernelGGL(seg_losses_bwd_v4_kernel<KC>, dim3(ls_blocks((long long)N * HW / 4, 8192)), dim3(LS_THREADS), 0, s, \
                     logits_w, logits_s, (const long long*)target, valid_mask, N, HW, ignore_index, do_ent, cr_variant,        \
                     detach_weak, sums, valid_mask ? 1 : 0, g_pce, g_ent, g_cr, grad_scale, dlogits_w, dlogits_s)
  if (v4 && K == 5) SEG_BWD_V4(5);
  else if (v4 && K == 4) SEG_BWD_V4(4);
  else if (v4 && K == 2) SEG_BWD_V4(2);
  else
    hipLaunchKernelGGL(seg_losses_bwd_kernel, dim3(ls_blocks((long long)N * HW)), dim3(LS_THREADS), 0, s, logits_w,
                       logits_s, (const long long*)target, valid_mask, N, K, HW, ignore_index, do_ent, cr_variant,
                       detach_weak, sums, valid_mask ? 1 : 0, g_pce, g_ent, g_cr, grad_scale, dlogits_w, dlogits_s);
#undef SEG_BWD_V4
  pp_prof_end(s);
  return pp_launch_status("seg_losses_bwd");
}

// ---------------------------------------------------------------- auxiliary head: up-sample + partial CE
__device__ __forceinline__ void lin_coeff_l(int o, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
  const float src = scale * (float)o;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}
static inline float lin_scale_l(int in_size, int out_size) {
  return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}

__global__ __launch_bounds__(LS_THREADS) void aux_pce_fwd_kernel(const float* __restrict__ lo, int N, int K, int h,
                                                                 int w, int H, int W, float sy, float sx,
                                                                 const long long* __restrict__ target,
                                                                 int ignore_index, float* __restrict__ up,
                                                                 double* __restrict__ partial /*[blocks][2]*/) {
  __shared__ float sh[16];
  const long long P = (long long)N * H * W;
  const int HW = H * W, hw_lo = h * w;
  float a_pce = 0.f, a_n = 0.f;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(p / HW), pix = (int)(p % HW);
    const int y = pix / W, x = pix % W;
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    lin_coeff_l(y, sy, h, y0, y1, wy0, wy1);
    lin_coeff_l(x, sx, w, x0, x1, wx0, wx1);
    float v[LS_MAXK];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k)
      if (k < K) {
        const float* b = lo + ((size_t)n * K + k) * hw_lo;
        v[k] = wy0 * (wx0 * b[y0 * w + x0] + wx1 * b[y0 * w + x1]) + wy1 * (wx0 * b[y1 * w + x0] + wx1 * b[y1 * w + x1]);
        up[((size_t)n * K + k) * HW + pix] = v[k];
        m = fmaxf(m, v[k]);
      }
    const long long t = target[p];
    if (t != ignore_index && t >= 0 && t < K) {
      float s = 0.f, vt = 0.f;
#pragma unroll
      for (int k = 0; k < LS_MAXK; ++k)
        if (k < K) { s += expf(v[k] - m); if (k == (int)t) vt = v[k]; }
      a_pce -= vt - m - logf(s);
      a_n += 1.f;
    }
  }
  const float r0 = pp_block_sum(a_pce, sh), r1 = pp_block_sum(a_n, sh);
  if (threadIdx.x == 0) { partial[(size_t)blockIdx.x * 2] = r0; partial[(size_t)blockIdx.x * 2 + 1] = r1; }
}

__global__ __launch_bounds__(64) void pair_reduce_kernel(const double* __restrict__ partial, int nblocks,
                                                         double* __restrict__ sums) {
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) { a += partial[(size_t)i * 2]; b += partial[(size_t)i * 2 + 1]; }
  a = pp_wave_sum_d(a);
  b = pp_wave_sum_d(b);
  if (threadIdx.x == 0) { sums[0] = a; sums[1] = b; }
}

extern "C" int pp_aux_pce_fwd(const float* lo, int N, int K, int h, int w, int H, int W, const int64_t* target,
                              int ignore_index, float* logits_up, double* sums, void* workspace,
                              size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(lo && target && logits_up && sums && workspace, "aux_pce_fwd: null pointer");
  PP_CHECK_ARG(K >= 1 && K <= LS_MAXK, "aux_pce_fwd: K=%d", K);
  const int blocks = ls_blocks((long long)N * H * W, 1024);
  if (workspace_bytes < (size_t)blocks * 2 * sizeof(double)) {
    pp_set_error("aux_pce_fwd: workspace too small");
    return PP_ERR_WORKSPACE;
  }
  pp_prof_begin(PP_K_LOSS, 0.0, (double)N * H * W * (4.0 * K + 8.0), s);
  hipLaunchKernelGGL(aux_pce_fwd_kernel, dim3(blocks), dim3(LS_THREADS), 0, s, lo, N, K, h, w, H, W, lin_scale_l(h, H),
                     lin_scale_l(w, W), (const long long*)target, ignore_index, logits_up, (double*)workspace);
  hipLaunchKernelGGL(pair_reduce_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, blocks, sums);
  pp_prof_end(s);
  return pp_launch_status("aux_pce_fwd");
}

__device__ __forceinline__ void touch_range_l(int i, float scale, int out_size, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out_size - 1; return; }
  lo = (int)floorf((float)(i - 1) / scale) - 1;
  hi = (int)ceilf((float)(i + 1) / scale) + 1;
  if (lo < 0) lo = 0;
  if (hi > out_size - 1) hi = out_size - 1;
}

// One WAVE per low-res pixel gathers the gradient of every labelled high-res pixel that taps it (round 6; round 2 used 16 lanes
// per pixel, each walking its rows of the ~20 x 20 support window one class-map load at a time: 149 us for 99 MB, 0.67 TB/s).
// Phase 1: the lanes take the window positions lane, lane + 64, ... (AUXB_SLOTS per lane and round) and load their class-map
// entries together -- independent loads, one latency.  The labelled ones (a few per cent of a scribble map) are COMPACTED into
// a per-wave list in LDS, ranked by ballot + population count (position order: deterministic).  Phase 2: lane r evaluates the
// softmax of the up-sampled logits and the bilinear weight of list entry r -- ONE pass for up to 64 labelled positions, where
// a direct walk over the slots executed the softmax body once per slot with most lanes idle.  The 64 per-lane sums are folded by
// the xor butterfly: a fixed order.
#define AUXB_SLOTS 8
#define AUXB_LIST (64 * AUXB_SLOTS)        // labelled positions of one phase-1 round: at most every slot of every lane
__global__ __launch_bounds__(256) void aux_pce_bwd_kernel(const float* __restrict__ up, const long long* __restrict__ target,
                                                          int ignore_index, const float* __restrict__ g_aux,
                                                          float grad_scale, const double* __restrict__ sums,
                                                          float* __restrict__ dlo, int N, int K, int h, int w, int H,
                                                          int W, float sy, float sx) {
  __shared__ int l_pos[4][AUXB_LIST];
  __shared__ unsigned char l_t[4][AUXB_LIST];
  const int total = N * h * w;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);        // wave-uniform
  if (i >= total) return;                                                    // (no block-wide barrier below: whole waves may leave)
  const int xl = i % w, yl = (i / w) % h, n = i / (w * h);
  const int HW = H * W;
  const float gs = (g_aux ? *g_aux : 0.f) * grad_scale / (float)sums[1];
  int ylo, yhi, xlo, xhi;
  touch_range_l(yl, sy, H, ylo, yhi);
  touch_range_l(xl, sx, W, xlo, xhi);
  const int wx = xhi - xlo + 1, cnt = (yhi - ylo + 1) * wx;
  const long long* tgt = target + (size_t)n * HW;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  float acc[LS_MAXK];
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k) acc[k] = 0.f;
  for (int base = 0; base < cnt; base += 64 * AUXB_SLOTS) {
    long long t[AUXB_SLOTS];
    int pos[AUXB_SLOTS];
#pragma unroll
    for (int j = 0; j < AUXB_SLOTS; ++j) {
      const int idx = base + j * 64 + lane;
      const int dy = (int)((unsigned)idx / (unsigned)wx), dx = idx - dy * wx;
      pos[j] = idx < cnt ? (ylo + dy) * W + xlo + dx : -1;
      t[j] = pos[j] >= 0 ? tgt[pos[j]] : (long long)ignore_index;
    }
    int n_list = 0;                                                         // wave-uniform
#pragma unroll
    for (int j = 0; j < AUXB_SLOTS; ++j) {
      const bool lab = pos[j] >= 0 && t[j] != ignore_index && t[j] >= 0 && t[j] < K;
      const unsigned long long m = __ballot(lab);
      if (lab) {
        const int r = n_list + __popcll(m & lt_mask);
        l_pos[wv][r] = pos[j];
        l_t[wv][r] = (unsigned char)t[j];
      }
      n_list += __popcll(m);
    }
    // the list is written and read by the same wave: its LDS operations complete in issue order, the fence keeps the compiler from
    // moving the reads of other lanes' entries above the writes
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int r = lane; r < n_list; r += 64) {
      const int p = l_pos[wv][r], tt = (int)l_t[wv][r];
      const int y = p / W, x = p - y * W;
      int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
      lin_coeff_l(y, sy, h, y0, y1, wy0, wy1);
      lin_coeff_l(x, sx, w, x0, x1, wx0, wx1);
      const float wgt = ((y0 == yl ? wy0 : 0.f) + (y1 == yl ? wy1 : 0.f)) * ((x0 == xl ? wx0 : 0.f) + (x1 == xl ? wx1 : 0.f));
      if (wgt == 0.f) continue;
      SM sm;
      pixel_softmax(up + (size_t)n * K * HW + p, HW, K, sm);
#pragma unroll
      for (int k = 0; k < LS_MAXK; ++k)
        if (k < K) acc[k] += wgt * (sm.p[k] - (k == tt ? 1.f : 0.f));
    }
  }
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k)
    if (k < K) {
      const float v = pp_wave_sum(acc[k]);
      if (lane == 0) dlo[((size_t)n * K + k) * h * w + yl * w + xl] = v * gs;
    }
}

extern "C" int pp_aux_pce_bwd(const float* logits_up, const int64_t* target, int ignore_index, const float* g_aux,
                              float grad_scale, const double* sums, float* dlo, int N, int K, int h, int w, int H,
                              int W, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(logits_up && target && sums && dlo, "aux_pce_bwd: null pointer");
  PP_CHECK_ARG(K >= 1 && K <= LS_MAXK, "aux_pce_bwd: K=%d", K);
  pp_prof_begin(PP_K_LOSS, 0.0, (double)N * H * W * 8.0, s);
  hipLaunchKernelGGL(aux_pce_bwd_kernel, dim3(pp_cdiv((long long)N * h * w * 64, 256)), dim3(256), 0, s, logits_up,
                     (const long long*)target, ignore_index, g_aux, grad_scale, sums, dlo, N, K, h, w, H, W,
                     lin_scale_l(h, H), lin_scale_l(w, W));
  pp_prof_end(s);
  return pp_launch_status("aux_pce_bwd");
}

// ---------------------------------------------------------------- memory bank (aux_path_memory.py:68-116)
// Round 6: two launches.  memory_partial_kernel, grid (classes, MEMU_SLICES): a block of four waves scans one slice of sample 0's
// scribble plane -- every wave requests its MEM_SCAN 64-pixel strips together, then visits the selected pixels MEM_VB at a time
// with the 64 lanes spread over the hid channels of the bilinearly up-sampled feature -- and leaves per-wave partial sums
// (sum of [weighted] embeddings, sum of weights, count).  memory_finalize_kernel, one block per class, adds them in (slice, wave)
// order and applies the first-visit / EMA rule.  (Round 1 ran ONE block of 16 waves per class, a dependent load per 64 pixels:
// 92 us for 2 MB on five of 256 CUs.)
#define MEMU_SLICES 64
#define MEMU_WAVES 4
#define MEM_CPL 4                 // channels per lane: hid <= 256
#define MEM_SCAN 4                // plane loads in flight per wave and round
#define MEM_VB 4                  // selected pixels whose corner rows are requested together

extern "C" size_t pp_memory_update_workspace(int K, int hid) {
  return (size_t)(K > 0 ? K : 0) * MEMU_SLICES * MEMU_WAVES * ((hid > 0 ? hid : 0) + 2) * sizeof(float);
}

// bank row: all-zero test and L2-normalised copy (wave 0 of the block; the caller synchronises)
__device__ __forceinline__ void memory_row_state(const float* __restrict__ row, int hid, int lane, float* row_hat, int* first_visit) {
  float nz = 0.f, sq = 0.f;
  for (int c = lane; c < hid; c += 64) { const float v = row[c]; nz += (v != 0.f) ? 1.f : 0.f; sq += v * v; }
  nz = pp_wave_sum(nz);
  sq = pp_wave_sum(sq);
  const float inv = 1.f / (sqrtf(sq) + 1e-8f);
  for (int c = lane; c < hid; c += 64) row_hat[c] = row[c] * inv;
  if (lane == 0) *first_visit = (nz == 0.f);
}

template <class FT>               // element type of the feature map: float, or _Float16 / __bf16 in the 16-bit storage modes
__global__ __launch_bounds__(MEMU_WAVES * 64) void memory_partial_kernel(
    const FT* __restrict__ feat, int ld, int hid, int h, int w, const float* __restrict__ scb0, int H, int W,
    float sy, float sx, const float* __restrict__ bank, int cosine_mode, float* __restrict__ partial) {
  __shared__ float row_hat[MEM_CPL * 64];
  __shared__ int first_visit;
  const int cls = blockIdx.x, slice = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* plane = scb0 + (size_t)cls * H * W;
  if (wv == 0) memory_row_state(bank + (size_t)cls * hid, hid, lane, row_hat, &first_visit);
  __syncthreads();
  const bool plain_mean = first_visit || !cosine_mode;
  float U[MEM_CPL] = {0.f, 0.f, 0.f, 0.f};
  float S = 0.f, cnt = 0.f;
  float rh[MEM_CPL];
#pragma unroll
  for (int j = 0; j < MEM_CPL; ++j) rh[j] = (lane + 64 * j < hid) ? row_hat[lane + 64 * j] : 0.f;
  const int HW = H * W;
  const int per = (HW + MEMU_SLICES - 1) / MEMU_SLICES;
  const int p_lo = slice * per, p_hi = (p_lo + per < HW) ? p_lo + per : HW;
  for (int base = p_lo + wv * 64; base < p_hi; base += MEMU_WAVES * 64 * MEM_SCAN) {
    unsigned long long bits_u[MEM_SCAN];
#pragma unroll
    for (int u = 0; u < MEM_SCAN; ++u) {
      const int pix = base + u * MEMU_WAVES * 64 + lane;
      bits_u[u] = __ballot(pix < p_hi && plane[pix] == 1.f);
    }
#pragma unroll
    for (int u = 0; u < MEM_SCAN; ++u) {
      unsigned long long bits = bits_u[u];
      const int base_u = base + u * MEMU_WAVES * 64;
      while (bits) {
        int q[MEM_VB];
#pragma unroll
        for (int b = 0; b < MEM_VB; ++b) {
          q[b] = -1;
          if (bits) { q[b] = base_u + __ffsll((long long)bits) - 1; bits &= bits - 1; }
        }
        float fa[MEM_VB][MEM_CPL], fb[MEM_VB][MEM_CPL], fc[MEM_VB][MEM_CPL], fd[MEM_VB][MEM_CPL];
        float wy0[MEM_VB], wy1[MEM_VB], wx0[MEM_VB], wx1[MEM_VB];
#pragma unroll
        for (int b = 0; b < MEM_VB; ++b) {
          if (q[b] < 0) continue;
          const int y = q[b] / W, x = q[b] % W;
          int y0, y1, x0, x1;
          lin_coeff_l(y, sy, h, y0, y1, wy0[b], wy1[b]);
          lin_coeff_l(x, sx, w, x0, x1, wx0[b], wx1[b]);
#pragma unroll
          for (int j = 0; j < MEM_CPL; ++j) {
            const int c = lane + 64 * j;
            if (c < hid) {
              fa[b][j] = feat[(size_t)(y0 * w + x0) * ld + c]; fb[b][j] = feat[(size_t)(y0 * w + x1) * ld + c];
              fc[b][j] = feat[(size_t)(y1 * w + x0) * ld + c]; fd[b][j] = feat[(size_t)(y1 * w + x1) * ld + c];
            }
          }
        }
#pragma unroll
        for (int b = 0; b < MEM_VB; ++b) {
          if (q[b] < 0) continue;
          float e[MEM_CPL];
          float sq = 0.f;
#pragma unroll
          for (int j = 0; j < MEM_CPL; ++j) {
            const int c = lane + 64 * j;
            e[j] = 0.f;
            if (c < hid) {
              const float a = fa[b][j], bb = fb[b][j], cc = fc[b][j], d = fd[b][j];
              e[j] = wy0[b] * (wx0[b] * a + wx1[b] * bb) + wy1[b] * (wx0[b] * cc + wx1[b] * d);
              sq += e[j] * e[j];
            }
          }
          cnt += 1.f;
          if (plain_mean) {
#pragma unroll
            for (int j = 0; j < MEM_CPL; ++j) U[j] += e[j];
          } else {
            sq = pp_wave_sum(sq);
            const float inv = 1.f / (sqrtf(sq) + 1e-8f);
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < MEM_CPL; ++j) { e[j] *= inv; dot += e[j] * rh[j]; }
            dot = pp_wave_sum(dot);
            const float om = 1.f - dot;
            S += om;
#pragma unroll
            for (int j = 0; j < MEM_CPL; ++j) U[j] += e[j] * om;
          }
        }
      }
    }
  }
  float* o = partial + ((size_t)(cls * MEMU_SLICES + slice) * MEMU_WAVES + wv) * (hid + 2);
#pragma unroll
  for (int j = 0; j < MEM_CPL; ++j)
    if (lane + 64 * j < hid) o[lane + 64 * j] = U[j];
  if (lane == 0) { o[hid] = S; o[hid + 1] = cnt; }
}

#define MEMU_FIN_WAVES 4
__global__ __launch_bounds__(MEMU_FIN_WAVES * 64) void memory_finalize_kernel(const float* __restrict__ partial, int hid,
                                                                              float* __restrict__ bank, float mom, int cosine_mode) {
  __shared__ float row_hat[MEM_CPL * 64];
  __shared__ int first_visit;
  __shared__ float part[MEMU_FIN_WAVES][MEM_CPL * 64 + 2];
  const int cls = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* row = bank + (size_t)cls * hid;
  if (wv == 0) memory_row_state(row, hid, lane, row_hat, &first_visit);
  // wave wv adds its quarter of the (slice, wave) partial rows in order, eight loads in flight; the four quarter sums are then
  // added in wave order: a fixed order over all MEMU_SLICES * MEMU_WAVES rows
  constexpr int ROWS = MEMU_SLICES * MEMU_WAVES, PER = ROWS / MEMU_FIN_WAVES;
  const float* p0 = partial + ((size_t)cls * ROWS + (size_t)wv * PER) * (hid + 2);
  for (int c = lane; c < hid + 2; c += 64) {
    float u = 0.f;
#pragma unroll 8
    for (int k = 0; k < PER; ++k) u += p0[(size_t)k * (hid + 2) + c];
    part[wv][c] = u;
  }
  __syncthreads();
  if (wv != 0) return;
  float St = 0.f, ct = 0.f;
#pragma unroll
  for (int q = 0; q < MEMU_FIN_WAVES; ++q) { St += part[q][hid]; ct += part[q][hid + 1]; }
  if (ct == 0.f) return;                                       // no scribble of this class in sample 0 (uniform over the block)
  for (int c = lane; c < hid; c += 64) {
    float u = 0.f;
#pragma unroll
    for (int q = 0; q < MEMU_FIN_WAVES; ++q) u += part[q][c];
    float nv;
    if (first_visit) {
      nv = u / ct;                                             // first visit: plain mean, no EMA
    } else {
      const float upd = cosine_mode ? u / (St + 1e-8f) : u / ct;
      const float old = cosine_mode ? row_hat[c] : row[c];     // cosine mode normalises the stored row in place
      nv = (1.f - mom) * old + mom * upd;
    }
    row[c] = nv;
  }
}

template <class FT>
static int memory_update_impl(const FT* feat0, int ld, int hid, int h, int w, const float* scribble0, int K,
                              int H, int W, float* bank, float momentum_now, int cosine_mode, void* workspace,
                              size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(feat0 && scribble0 && bank && workspace, "memory_update: null pointer");
  PP_CHECK_ARG(hid >= 1 && hid <= MEM_CPL * 64 && K >= 1 && ld >= hid, "memory_update: hid=%d (<=256) K=%d", hid, K);
  if (workspace_bytes < pp_memory_update_workspace(K, hid)) {
    pp_set_error("memory_update: workspace too small (%zu < %zu)", workspace_bytes, pp_memory_update_workspace(K, hid));
    return PP_ERR_WORKSPACE;
  }
  pp_prof_begin(PP_K_LOSS, 0.0, 4.0 * K * H * W, s);
  hipLaunchKernelGGL(memory_partial_kernel<FT>, dim3(K, MEMU_SLICES), dim3(MEMU_WAVES * 64), 0, s, feat0, ld, hid, h, w, scribble0, H,
                     W, lin_scale_l(h, H), lin_scale_l(w, W), (const float*)bank, cosine_mode, (float*)workspace);
  hipLaunchKernelGGL(memory_finalize_kernel, dim3(K), dim3(MEMU_FIN_WAVES * 64), 0, s, (const float*)workspace, hid, bank, momentum_now, cosine_mode);
  pp_prof_end(s);
  return pp_launch_status("memory_update");
}

extern "C" int pp_memory_update(const float* feat0, int ld, int hid, int h, int w, const float* scribble0, int K,
                                int H, int W, float* bank, float momentum_now, int cosine_mode, void* workspace,
                                size_t workspace_bytes, void* stream) {
  return memory_update_impl(feat0, ld, hid, h, w, scribble0, K, H, W, bank, momentum_now, cosine_mode, workspace, workspace_bytes, stream);
}
// the same with the features stored as IEEE fp16 (16-bit storage mode, include/pacingpseudo_hip_h16.h)
extern "C" int pp_memory_update_h16(const void* feat0, int ld, int hid, int h, int w, const float* scribble0, int K,
                                    int H, int W, float* bank, float momentum_now, int cosine_mode, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return memory_update_impl(reinterpret_cast<const _Float16*>(feat0), ld, hid, h, w, scribble0, K, H, W, bank, momentum_now,
                            cosine_mode, workspace, workspace_bytes, stream);
}
// ... and as bfloat16 (include/pacingpseudo_hip_bf16.h)
extern "C" int pp_memory_update_bf16(const void* feat0, int ld, int hid, int h, int w, const float* scribble0, int K,
                                     int H, int W, float* bank, float momentum_now, int cosine_mode, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return memory_update_impl(reinterpret_cast<const __bf16*>(feat0), ld, hid, h, w, scribble0, K, H, W, bank, momentum_now,
                            cosine_mode, workspace, workspace_bytes, stream);
}

// bank classification: logits[r][k] = <bank[r], wfc[k]>, loss = mean_r CE(logits[r], r)   (aux_path_memory.py:61,
// consistency_reglur_memory.py:94-97).  mode 0: write loss;  mode 1: dwfc (+)= g * dloss/dwfc.
__global__ __launch_bounds__(64) void memory_ce_kernel(const float* __restrict__ bank, const float* __restrict__ wfc,
                                                       int K, int hid, float* loss, const float* g, float grad_scale,
                                                       float* dwfc, int accumulate, int mode) {
  __shared__ float lg[LS_MAXK][LS_MAXK];
  __shared__ float dl[LS_MAXK][LS_MAXK];
  const int lane = threadIdx.x;
  for (int r = 0; r < K; ++r)
    for (int k = 0; k < K; ++k) {
      float a = 0.f;
      for (int c = lane; c < hid; c += 64) a += bank[r * hid + c] * wfc[k * hid + c];
      a = pp_wave_sum(a);
      if (lane == 0) lg[r][k] = a;
    }
  __syncthreads();
  if (lane == 0) {
    float tot = 0.f;
    for (int r = 0; r < K; ++r) {
      float m = -INFINITY, s = 0.f;
      for (int k = 0; k < K; ++k) m = fmaxf(m, lg[r][k]);
      for (int k = 0; k < K; ++k) s += expf(lg[r][k] - m);
      const float ls = logf(s);
      tot -= lg[r][r] - m - ls;
      for (int k = 0; k < K; ++k) dl[r][k] = (expf(lg[r][k] - m - ls) - (k == r ? 1.f : 0.f)) / (float)K;
    }
    if (mode == 0 && loss) *loss = tot / (float)K;
  }
  __syncthreads();
  if (mode == 1) {
    const float gs = (g ? *g : 0.f) * grad_scale;
    for (int k = 0; k < K; ++k)
      for (int c = lane; c < hid; c += 64) {
        float a = 0.f;
        for (int r = 0; r < K; ++r) a += dl[r][k] * bank[r * hid + c];
        dwfc[k * hid + c] = (accumulate ? dwfc[k * hid + c] : 0.f) + gs * a;
      }
  }
}

extern "C" int pp_memory_ce_fwd(const float* bank, const float* wfc, int K, int hid, float* loss, void* stream) {
  PP_CHECK_ARG(bank && wfc && loss && K >= 1 && K <= LS_MAXK, "memory_ce_fwd: bad arguments");
  hipLaunchKernelGGL(memory_ce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, bank, wfc, K, hid, loss, nullptr, 0.f,
                     nullptr, 0, 0);
  return pp_launch_status("memory_ce_fwd");
}

extern "C" int pp_memory_ce_bwd(const float* bank, const float* wfc, int K, int hid, const float* g, float grad_scale,
                                float* dwfc, int accumulate, void* stream) {
  PP_CHECK_ARG(bank && wfc && dwfc && K >= 1 && K <= LS_MAXK, "memory_ce_bwd: bad arguments");
  hipLaunchKernelGGL(memory_ce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, bank, wfc, K, hid, nullptr, g,
                     grad_scale, dwfc, accumulate, 1);
  return pp_launch_status("memory_ce_bwd");
}

// ---------------------------------------------------------------- validation Dice counts (utils/metrics.py:7-34)
// counts[n][k] = { sum pred_k * target_k, sum pred_k, sum target_k } with pred = one-hot(argmax_k logits)
__global__ __launch_bounds__(LS_THREADS) void dice_counts_kernel(const float* __restrict__ logits,
                                                                 const float* __restrict__ label, int K, int HW,
                                                                 float* __restrict__ counts) {
  __shared__ float sh[16];
  const int n = blockIdx.x;
  float inter[LS_MAXK], ps[LS_MAXK], ts[LS_MAXK];
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k) inter[k] = ps[k] = ts[k] = 0.f;
  const float* lz = logits + (size_t)n * K * HW;
  const float* lb = label + (size_t)n * K * HW;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    float m = lz[p];
    int a = 0;
    for (int k = 1; k < K; ++k) { const float v = lz[(size_t)k * HW + p]; if (v > m) { m = v; a = k; } }
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k)
      if (k < K) {
        const float t = lb[(size_t)k * HW + p];
        const float pr = (k == a) ? 1.f : 0.f;
        inter[k] += pr * t; ps[k] += pr; ts[k] += t;
      }
  }
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k)
    if (k < K) {
      const float a = pp_block_sum(inter[k], sh), b = pp_block_sum(ps[k], sh), c = pp_block_sum(ts[k], sh);
      if (threadIdx.x == 0) {
        float* o = counts + ((size_t)n * K + k) * 3;
        o[0] = a; o[1] = b; o[2] = c;
      }
    }
}

extern "C" int pp_dice_counts(const float* logits, const float* label_onehot, int N, int K, int HW, float* counts,
                              void* stream) {
  PP_CHECK_ARG(logits && label_onehot && counts && K >= 1 && K <= LS_MAXK, "dice_counts: bad arguments");
  hipLaunchKernelGGL(dice_counts_kernel, dim3(N), dim3(LS_THREADS), 0, (hipStream_t)stream, logits, label_onehot, K, HW,
                     counts);
  return pp_launch_status("dice_counts");
}

// ---------------------------------------------------------------- soft Dice loss (losses/losses.py:147-162, upper_bound_chaos.py)
// dice[n][k] = 2 sum_i p_k t_k / (sum_i p_k + sum_i t_k + 1e-5) on the soft-max probabilities; loss = -mean_{n,k} dice.
// Forward: blocks (x, n) reduce their pixel range of image n into partial[n][blk][K][3] (double); a finalize folds
// the partials into sums[n][K][3] = {sum p t, sum p, sum t} and writes the loss.
// Backward: dL/dp_k = -g/(N K) * (2 t_k down - up) / down^2 per pixel, then through the soft-max.
#define DICE_BLOCKS 64
__global__ __launch_bounds__(LS_THREADS) void dice_loss_partial_kernel(const float* __restrict__ logits,
                                                                       const float* __restrict__ label, int K, int HW,
                                                                       double* __restrict__ partial) {
  __shared__ float sh[16];
  const int n = blockIdx.y;
  float inter[LS_MAXK], ps[LS_MAXK], ts[LS_MAXK];
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k) inter[k] = ps[k] = ts[k] = 0.f;
  const float* lz = logits + (size_t)n * K * HW;
  const float* lb = label + (size_t)n * K * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    SM w;
    pixel_softmax(lz + p, HW, K, w);
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k)
      if (k < K) {
        const float t = lb[(size_t)k * HW + p];
        inter[k] += w.p[k] * t; ps[k] += w.p[k]; ts[k] += t;
      }
  }
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k)
    if (k < K) {
      const float a = pp_block_sum(inter[k], sh), b = pp_block_sum(ps[k], sh), c = pp_block_sum(ts[k], sh);
      if (threadIdx.x == 0) {
        double* o = partial + (((size_t)n * gridDim.x + blockIdx.x) * K + k) * 3;
        o[0] = a; o[1] = b; o[2] = c;
      }
    }
}

__global__ __launch_bounds__(64) void dice_loss_finalize_kernel(const double* __restrict__ partial, int nblk, int N, int K,
                                                                double* __restrict__ sums, float* __restrict__ loss) {
  __shared__ double dice[64];
  const int i = threadIdx.x;                     // (n, k) pairs handled in strides of 64
  double tot = 0.0;
  for (int e = i; e < N * K; e += 64) {
    const int n = e / K, k = e % K;
    double a = 0.0, b = 0.0, c = 0.0;
    for (int blk = 0; blk < nblk; ++blk) {
      const double* o = partial + (((size_t)n * nblk + blk) * K + k) * 3;
      a += o[0]; b += o[1]; c += o[2];
    }
    double* s = sums + (size_t)e * 3;
    s[0] = a; s[1] = b; s[2] = c;
    tot += 2.0 * a / (b + c + 1e-5);
  }
  dice[i] = tot;
  __syncthreads();
  if (i == 0) {
    double t = 0.0;
    for (int j = 0; j < 64; ++j) t += dice[j];
    *loss = (float)(-t / (double)(N * K));
  }
}

__global__ __launch_bounds__(LS_THREADS) void dice_loss_bwd_kernel(const float* __restrict__ logits,
                                                                   const float* __restrict__ label, int N, int K, int HW,
                                                                   const double* __restrict__ sums, const float* __restrict__ g,
                                                                   float grad_scale, float* __restrict__ dlogits,
                                                                   int accumulate) {
  const int n = blockIdx.y;
  const float gs = (g ? *g : 1.f) * grad_scale / (float)(N * K);
  float A[LS_MAXK], B[LS_MAXK];                  // dL/dp_k = A_k t_k + B_k
#pragma unroll
  for (int k = 0; k < LS_MAXK; ++k) {
    A[k] = 0.f; B[k] = 0.f;
    if (k < K) {
      const double* s = sums + ((size_t)n * K + k) * 3;
      const double up = 2.0 * s[0], down = s[1] + s[2] + 1e-5;
      A[k] = (float)(-2.0 * gs / down);
      B[k] = (float)(gs * up / (down * down));
    }
  }
  const float* lz = logits + (size_t)n * K * HW;
  const float* lb = label + (size_t)n * K * HW;
  float* dz = dlogits + (size_t)n * K * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    SM w;
    pixel_softmax(lz + p, HW, K, w);
    float u[LS_MAXK], pu = 0.f;
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k)
      if (k < K) { u[k] = A[k] * lb[(size_t)k * HW + p] + B[k]; pu += w.p[k] * u[k]; }
#pragma unroll
    for (int k = 0; k < LS_MAXK; ++k)
      if (k < K) {
        const float v = w.p[k] * (u[k] - pu);
        float* o = dz + (size_t)k * HW + p;
        *o = accumulate ? *o + v : v;
      }
  }
}

extern "C" size_t pp_dice_loss_workspace(int N, int K) { return (size_t)N * DICE_BLOCKS * K * 3 * sizeof(double) + 64; }

extern "C" int pp_dice_loss_fwd(const float* logits, const float* label_onehot, int N, int K, int HW, double* sums,
                                float* loss, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(logits && label_onehot && sums && loss && workspace && N >= 1 && K >= 1 && K <= LS_MAXK && HW >= 1,
               "dice_loss_fwd: bad arguments");
  if (workspace_bytes < pp_dice_loss_workspace(N, K)) {
    pp_set_error("dice_loss_fwd: workspace too small");
    return PP_ERR_WORKSPACE;
  }
  double* partial = reinterpret_cast<double*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  int bx = pp_cdiv(HW, LS_THREADS);
  if (bx > DICE_BLOCKS) bx = DICE_BLOCKS;
  pp_prof_begin(PP_K_LOSS, 0.0, 8.0 * (double)N * K * HW, s);
  hipLaunchKernelGGL(dice_loss_partial_kernel, dim3(bx, N), dim3(LS_THREADS), 0, s, logits, label_onehot, K, HW, partial);
  hipLaunchKernelGGL(dice_loss_finalize_kernel, dim3(1), dim3(64), 0, s, partial, bx, N, K, sums, loss);
  pp_prof_end(s);
  return pp_launch_status("dice_loss_fwd");
}

extern "C" int pp_dice_loss_bwd(const float* logits, const float* label_onehot, int N, int K, int HW, const double* sums,
                                const float* g, float grad_scale, float* dlogits, int accumulate, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(logits && label_onehot && sums && dlogits && N >= 1 && K >= 1 && K <= LS_MAXK, "dice_loss_bwd: bad arguments");
  int bx = pp_cdiv(HW, LS_THREADS);
  if (bx > 256) bx = 256;
  pp_prof_begin(PP_K_LOSS, 0.0, 12.0 * (double)N * K * HW, s);
  hipLaunchKernelGGL(dice_loss_bwd_kernel, dim3(bx, N), dim3(LS_THREADS), 0, s, logits, label_onehot, N, K, HW, sums, g,
                     grad_scale, dlogits, accumulate);
  pp_prof_end(s);
  return pp_launch_status("dice_loss_bwd");
}

// ---------------------------------------------------------------- 95 % Hausdorff distance (inference.py:217-237)
// The reference calls medpy.metric.binary.hd95(pred == k, label == k, spacing, connectivity 1): the surface of a mask is
// mask XOR erode(mask, 4-neighbourhood cross, outside = background); each directed set is the Euclidean distance (in
// units of the pixel spacing) from every surface pixel of one mask to the nearest surface pixel of the other, and the
// metric is the 95th percentile of both sets together.  Per (image, class): one block compacts the two surfaces into
// coordinate lists (order irrelevant: only minima and a percentile are taken), then every surface pixel scans the
// other list staged through LDS.  dist: [N][K][2][cap] floats, counts: [N][K][4] = {surface a, surface b, |a|, |b|}.
#define HD_THREADS 256
__global__ __launch_bounds__(HD_THREADS) void hd_surface_kernel(const long long* __restrict__ pred, const long long* __restrict__ label,
                                                                int K, int H, int W, int cap, int* __restrict__ coords,
                                                                int* __restrict__ counts) {
  const int n = blockIdx.x / K, k = blockIdx.x % K;
  const long long* A = pred + (size_t)n * H * W;
  const long long* B = label + (size_t)n * H * W;
  int* ca = coords + ((size_t)blockIdx.x * 2 + 0) * cap;
  int* cb = coords + ((size_t)blockIdx.x * 2 + 1) * cap;
  int* cnt = counts + (size_t)blockIdx.x * 4;
  __shared__ int na, nb, ta, tb;
  if (threadIdx.x == 0) { na = 0; nb = 0; ta = 0; tb = 0; }
  __syncthreads();
  int my_ta = 0, my_tb = 0;
  for (int p = threadIdx.x; p < H * W; p += HD_THREADS) {
    const int y = p / W, x = p % W;
    const bool a = A[p] == k, b = B[p] == k;
    my_ta += a; my_tb += b;
    if (a) {
      const bool inner = y > 0 && y < H - 1 && x > 0 && x < W - 1 && A[p - W] == k && A[p + W] == k && A[p - 1] == k && A[p + 1] == k;
      if (!inner) { const int i = atomicAdd(&na, 1); if (i < cap) ca[i] = p; }
    }
    if (b) {
      const bool inner = y > 0 && y < H - 1 && x > 0 && x < W - 1 && B[p - W] == k && B[p + W] == k && B[p - 1] == k && B[p + 1] == k;
      if (!inner) { const int i = atomicAdd(&nb, 1); if (i < cap) cb[i] = p; }
    }
  }
  atomicAdd(&ta, my_ta);
  atomicAdd(&tb, my_tb);
  __syncthreads();
  if (threadIdx.x == 0) { cnt[0] = na; cnt[1] = nb; cnt[2] = ta; cnt[3] = tb; }
}

__global__ __launch_bounds__(HD_THREADS) void hd_distance_kernel(const int* __restrict__ coords, const int* __restrict__ counts,
                                                                 int W, int cap, float sy, float sx, float* __restrict__ dist) {
  __shared__ int tile[HD_THREADS];
  const int item = blockIdx.x, dir = blockIdx.y;             // dir 0: a -> b, 1: b -> a
  const int* cnt = counts + (size_t)item * 4;
  const int n_from = min(cnt[dir], cap), n_to = min(cnt[1 - dir], cap);
  const int* from = coords + ((size_t)item * 2 + dir) * cap;
  const int* to = coords + ((size_t)item * 2 + (1 - dir)) * cap;
  float* out = dist + ((size_t)item * 2 + dir) * cap;
  for (int base = 0; base < n_from; base += HD_THREADS) {
    const int i = base + threadIdx.x;
    const int p = i < n_from ? from[i] : 0;
    const int py = p / W, px = p % W;            // integer offsets first: coincident pixels give exactly 0
    float best = 3.0e38f;
    for (int t0 = 0; t0 < n_to; t0 += HD_THREADS) {
      __syncthreads();
      if (t0 + threadIdx.x < n_to) tile[threadIdx.x] = to[t0 + threadIdx.x];
      __syncthreads();
      const int m = min(HD_THREADS, n_to - t0);
      for (int j = 0; j < m; ++j) {
        const int q = tile[j];
        const float dy = (float)(q / W - py) * sy, dx = (float)(q % W - px) * sx;
        best = fminf(best, dy * dy + dx * dx);
      }
    }
    if (i < n_from) out[i] = sqrtf(best);
  }
}

// ---- loss assembly (train_chaos.py:273-310): total = t0 * w0 + t1 * w1 + ... of 0-dim device losses, one launch ----
// The Python form is eight element-wise launches forward and five backward on 0-dim tensors, ~6 us each behind one another on
// the stream between the forward and the backward pass.  Same arithmetic as that chain: every product and every sum rounded to
// fp32, left to right (no fused multiply-add).
struct WSumArgs { const float* t[8]; float w[8]; int n; };
__global__ void weighted_sum_fwd_kernel(WSumArgs a, float* out) {
#pragma clang fp contract(off)                      // (hipcc contracts a * b + c by default: __fmul_rn / __fadd_rn are plain operators)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float acc = a.t[0][0] * a.w[0];
  for (int i = 1; i < a.n; ++i) {
    const float prod = a.t[i][0] * a.w[i];
    acc = acc + prod;
  }
  out[0] = acc;
}
__global__ void weighted_sum_bwd_kernel(const float* g, WSumArgs a, float* gout) {
  const int i = threadIdx.x;
  if (i < a.n) gout[i] = __fmul_rn(g[0], a.w[i]);
}
extern "C" int pp_weighted_sum_fwd(const float* const* terms, const float* weights, int n, float* out, void* stream) {
  PP_CHECK_ARG(terms && weights && out && n >= 1 && n <= 8, "weighted_sum_fwd: 1..8 terms");
  WSumArgs a;
  for (int i = 0; i < 8; ++i) { a.t[i] = i < n ? terms[i] : nullptr; a.w[i] = i < n ? weights[i] : 0.f; }
  for (int i = 0; i < n; ++i) PP_CHECK_ARG(terms[i], "weighted_sum_fwd: null term");
  a.n = n;
  hipLaunchKernelGGL(weighted_sum_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, out);
  return pp_launch_status("weighted_sum_fwd");
}
extern "C" int pp_weighted_sum_bwd(const float* g, const float* weights, int n, float* gout, void* stream) {
  PP_CHECK_ARG(g && weights && gout && n >= 1 && n <= 8, "weighted_sum_bwd: 1..8 terms");
  WSumArgs a;
  for (int i = 0; i < 8; ++i) { a.t[i] = nullptr; a.w[i] = i < n ? weights[i] : 0.f; }
  a.n = n;
  hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g, a, gout);
  return pp_launch_status("weighted_sum_bwd");
}

extern "C" size_t pp_hd95_workspace(int N, int K, int H, int W) {
  return (size_t)N * K * 2 * H * W * sizeof(int) + 64;
}

// pred / label: int64 class maps [N][H][W] (device).  dist: [N*K][2][H*W] floats, counts: [N*K][4] ints (device); the
// caller takes the 95th percentile of dist[item][0][:counts[0]] ++ dist[item][1][:counts[1]] (numpy.percentile, linear).
extern "C" int pp_hd95_surface_distances(const int64_t* pred, const int64_t* label, int N, int K, int H, int W,
                                         float spacing_y, float spacing_x, float* dist, int* counts, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(pred && label && dist && counts && workspace && N >= 1 && K >= 1 && H >= 1 && W >= 1, "hd95: bad arguments");
  if (workspace_bytes < pp_hd95_workspace(N, K, H, W)) {
    pp_set_error("hd95: workspace too small");
    return PP_ERR_WORKSPACE;
  }
  int* coords = reinterpret_cast<int*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
  const int cap = H * W;
  pp_prof_begin(PP_K_LOSS, 0.0, 16.0 * N * K * H * W, s);
  hipLaunchKernelGGL(hd_surface_kernel, dim3(N * K), dim3(HD_THREADS), 0, s, (const long long*)pred, (const long long*)label, K, H,
                     W, cap, coords, counts);
  hipLaunchKernelGGL(hd_distance_kernel, dim3(N * K, 2), dim3(HD_THREADS), 0, s, coords, counts, W, cap, spacing_y, spacing_x, dist);
  pp_prof_end(s);
  return pp_launch_status("hd95_surface_distances");
}
