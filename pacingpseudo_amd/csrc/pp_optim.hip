// Fused Adam over one flat fp32 parameter slab (torch.optim.Adam(lr, betas=(.9,.999), eps=1e-8,
// weight_decay=wd) semantics: L2-coupled decay, bias correction as PyTorch) -- train_chaos.py:219,315.
// HBM-bound: 16 B per lane, reads p,g,m,v and writes p,m,v once (28 B / parameter).
#include "pp_common.h"

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   float lr, float b1, float b2, float eps, float wd, float bc1,
                                                   float sqrt_bc2, int* __restrict__ skip, const int* __restrict__ step_dev,
                                                   const float* __restrict__ lr_dev) {
  // skip (nullable): skip[0] != 0 -> the gradients of this step are not finite (16-bit storage: loss-scale overflow, found by
  // pp_scale_guard): leave p, m, v untouched.  The skipped update is counted in skip[1] by the caller's form: the *_guard entry
  // points count here (once per launch), the *_dev entry points in their commit kernel (once per optimizer step).
  if (skip && skip[0]) {
    if (!step_dev && blockIdx.x == 0 && threadIdx.x == 0) skip[1] += 1;
    return;
  }
  // step_dev (nullable): number of updates applied to this segment so far, kept ON THE DEVICE -- the bias corrections follow the
  // updates that really happened (a skipped step does not advance them) and a captured hipGraph replays with the live count;
  // lr_dev (nullable): the learning rate as a device scalar, for the same reason
  if (step_dev) {
    const double t = (double)(step_dev[0] + 1);
    bc1 = (float)(1.0 - pow((double)b1, t));
    sqrt_bc2 = (float)sqrt(1.0 - pow((double)b2, t));
  }
  if (lr_dev) lr = lr_dev[0];
  const long long n4 = n >> 2;
  const float step = lr / bc1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define PP_ADAM1(f)                                              \
    {                                                            \
      const float gr = gg.f + wd * pp.f;                         \
      mm.f = b1 * mm.f + (1.f - b1) * gr;                        \
      vv.f = b2 * vv.f + (1.f - b2) * gr * gr;                   \
      pp.f -= step * (mm.f / (sqrtf(vv.f) / sqrt_bc2 + eps));    \
    }
    PP_ADAM1(x) PP_ADAM1(y) PP_ADAM1(z) PP_ADAM1(w)
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n not a multiple of 4)
  const long long t = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    const float gr = g[t] + wd * p[t];
    const float m1 = b1 * m[t] + (1.f - b1) * gr;
    const float v1 = b2 * v[t] + (1.f - b2) * gr * gr;
    m[t] = m1; v[t] = v1;
    p[t] -= step * (m1 / (sqrtf(v1) / sqrt_bc2 + eps));
  }
}

// after the update kernel(s) of one optimizer step: advance the device-side step counter of a segment, or -- skipped step --
// count it once (count_skip: only the first segment's commit counts, so one skipped step is one count)
__global__ void optim_commit_kernel(int* __restrict__ step_dev, int* __restrict__ skip, int count_skip) {
  if (skip && skip[0]) {
    if (count_skip) skip[1] += 1;
    return;
  }
  step_dev[0] += 1;
}

static int adam_step_impl(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int step, int* skip, void* stream,
                          int* step_dev = nullptr, const float* lr_dev = nullptr, int count_skip = 0) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(p && g && m && v && n > 0 && (step >= 1 || step_dev), "adam_step: bad arguments");
  PP_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: slabs must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  int blocks = pp_cdiv(n / 4 + 1, 256);
  if (blocks > 4096) blocks = 4096;
  pp_prof_begin(PP_K_OPTIM, 0.0, 28.0 * (double)n, s);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                     (float)bc1, (float)sqrt(bc2), skip, (const int*)step_dev, lr_dev);
  if (step_dev) hipLaunchKernelGGL(optim_commit_kernel, dim3(1), dim3(1), 0, s, step_dev, skip, count_skip);
  pp_prof_end(s);
  return pp_launch_status("adam_step");
}

extern "C" int pp_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int step, void* stream) {
  return adam_step_impl(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, nullptr, stream);
}
// the same, skipped (and counted in skip[1]) when skip[0] != 0: see pp_scale_guard
extern "C" int pp_adam_step_guard(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, int step, int* skip, void* stream) {
  return adam_step_impl(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, skip, stream);
}

// The form the fused optimizers use (round 5): the step count of the segment lives on the device (step_dev[0] =
// updates applied so far; the kernel uses step_dev[0] + 1 for the bias corrections and a one-thread commit kernel advances it
// afterwards unless the step was skipped), lr_dev (nullable) overrides lr with a device scalar, skip as in pp_adam_step_guard
// except that a skipped STEP is counted once: only the call with count_skip != 0 adds to skip[1].
extern "C" int pp_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float lr, const float* lr_dev, float beta1,
                                float beta2, float eps, float weight_decay, int* step_dev, int* skip, int count_skip, void* stream) {
  PP_CHECK_ARG(step_dev, "adam_step_dev: step_dev is null");
  return adam_step_impl(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, 1, skip, stream, step_dev, lr_dev, count_skip);
}

// torch.optim.SGD(lr, momentum, weight_decay) (train_chaos.py:220-221, --optimizer momentum): g += wd*p;
// buf = g on the first step, momentum*buf + g afterwards (dampening 0, no Nesterov); p -= lr*buf.  20 B / parameter.
__global__ __launch_bounds__(256) void sgd_momentum_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ buf, long long n, float lr, float mom,
                                                           float wd, int first, int* __restrict__ skip,
                                                           const int* __restrict__ step_dev, const float* __restrict__ lr_dev) {
  if (skip && skip[0]) {                      // non-finite gradients this step (see adam_kernel)
    if (!step_dev && blockIdx.x == 0 && threadIdx.x == 0) skip[1] += 1;
    return;
  }
  if (step_dev) first = step_dev[0] == 0;
  if (lr_dev) lr = lr_dev[0];
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 bb = first ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(buf)[i];
#define PP_SGD1(f)                                   \
    {                                                \
      const float gr = gg.f + wd * pp.f;             \
      bb.f = first ? gr : mom * bb.f + gr;           \
      pp.f -= lr * bb.f;                             \
    }
    PP_SGD1(x) PP_SGD1(y) PP_SGD1(z) PP_SGD1(w)
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(buf)[i] = bb;
  }
  const long long t = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    const float gr = g[t] + wd * p[t];
    const float b1 = first ? gr : mom * buf[t] + gr;
    buf[t] = b1;
    p[t] -= lr * b1;
  }
}

static int sgd_step_impl(float* p, const float* g, float* buf, long long n, float lr, float momentum,
                         float weight_decay, int step, int* skip, void* stream, int* step_dev = nullptr,
                         const float* lr_dev = nullptr, int count_skip = 0) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(p && g && buf && n > 0 && (step >= 1 || step_dev), "sgd_momentum_step: bad arguments");
  PP_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) == 0, "sgd_momentum_step: slabs must be 16-byte aligned");
  int blocks = pp_cdiv(n / 4 + 1, 256);
  if (blocks > 4096) blocks = 4096;
  pp_prof_begin(PP_K_OPTIM, 0.0, 20.0 * (double)n, s);
  hipLaunchKernelGGL(sgd_momentum_kernel, dim3(blocks), dim3(256), 0, s, p, g, buf, n, lr, momentum, weight_decay,
                     step == 1 ? 1 : 0, skip, (const int*)step_dev, lr_dev);
  if (step_dev) hipLaunchKernelGGL(optim_commit_kernel, dim3(1), dim3(1), 0, s, step_dev, skip, count_skip);
  pp_prof_end(s);
  return pp_launch_status("sgd_momentum_step");
}

extern "C" int pp_sgd_momentum_step(float* p, const float* g, float* buf, long long n, float lr, float momentum,
                                    float weight_decay, int step, void* stream) {
  return sgd_step_impl(p, g, buf, n, lr, momentum, weight_decay, step, nullptr, stream);
}
extern "C" int pp_sgd_momentum_step_guard(float* p, const float* g, float* buf, long long n, float lr, float momentum,
                                          float weight_decay, int step, int* skip, void* stream) {
  return sgd_step_impl(p, g, buf, n, lr, momentum, weight_decay, step, skip, stream);
}

extern "C" int pp_sgd_momentum_step_dev(float* p, const float* g, float* buf, long long n, float lr, const float* lr_dev,
                                        float momentum, float weight_decay, int* step_dev, int* skip, int count_skip, void* stream) {
  PP_CHECK_ARG(step_dev, "sgd_momentum_step_dev: step_dev is null");
  return sgd_step_impl(p, g, buf, n, lr, momentum, weight_decay, 2, skip, stream, step_dev, lr_dev, count_skip);
}

// dst (+)= src over a flat slab (gradient accumulation across bucket copies, test helper)
__global__ void fill_kernel(float* __restrict__ p, long long n, float value) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = value;
}

extern "C" int pp_fill(float* p, long long n, float value, void* stream) {
  PP_CHECK_ARG(p && n >= 0, "fill: bad arguments");
  if (n == 0) return 0;
  int blocks = pp_cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n, value);
  return pp_launch_status("fill");
}

// p *= value over a flat slab: removes the static loss scale from the gradient slab of a 16-bit-storage step (the scale is a
// power of two: exact) before the optimizer / after the all-reduce
__global__ void scale_kernel(float* __restrict__ p, long long n, float value, int* __restrict__ bad) {
  bool nf = false;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = p[i] * value;
    p[i] = v;
    nf |= !(fabsf(v) <= 3.4028234e38f);       // inf or NaN
  }
  if (bad && nf) atomicOr(bad, 1);            // rare; order independent
}

extern "C" int pp_scale(float* p, long long n, float value, void* stream) {
  PP_CHECK_ARG(p && n >= 0, "scale: bad arguments");
  if (n == 0) return 0;
  int blocks = pp_cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n, value, (int*)nullptr);
  return pp_launch_status("scale");
}

// the same, and bad[0] |= 1 when a scaled value is not finite: the overflow check of the 16-bit storage mode (a static loss
// scale can overflow fp16 in a bad step; pp_adam_step_guard / pp_sgd_momentum_step_guard then leave the weights alone)
extern "C" int pp_scale_guard(float* p, long long n, float value, int* bad, void* stream) {
  PP_CHECK_ARG(p && bad && n >= 0, "scale_guard: bad arguments");
  if (n == 0) return 0;
  int blocks = pp_cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n, value, bad);
  return pp_launch_status("scale_guard");
}
