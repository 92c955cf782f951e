// Input pipeline on the GPU: the two-stream (weak / strong) augmentation of the reference's data loader
// (datasets/chaos/chaos_dataset.py:58-90, datasets/augmentations.py:11-461, datasets/chaos/chaos_aug_configs.py:16-86).
//
// The reference augments one slice at a time on the CPU with scipy / skimage / cv2 (cubic-spline resize, elastic
// map_coordinates, warpAffine); at >= 800 images/s per GPU that pipeline is the bottleneck.  Here a
// whole batch is augmented in HBM by a handful of launches:
//   * the random DECISIONS (which transform fires, its parameters) are drawn on the host by numpy in the reference's
//     own call order (pacingpseudo_amd/augment.py), a few dozen scalars per sample;
//   * the geometric transforms -- Scaling, RandomRotation, Mirroring x2, RandomCrop / padding -- are composed on the
//     host into ONE inverse affine map per sample and applied by ONE resampling kernel (bicubic for the image with the
//     Keys a = -0.75 kernel cv2.INTER_CUBIC uses, nearest for label / scribble, constant padding, valid mask), instead
//     of three successive CPU resamplings; ElasticTransform adds a displacement field to the same map;
//   * MeanStdNorm / Brightness / Contrast / Gamma / GaussianNoise are per-sample scalar maps x -> clip(a x + b), a power
//     law, and a counter-based (Philox-4x32-10) normal generator; their statistics come from one reduction kernel.
// All kernels are HBM-bound streaming kernels over (B, H, W) fp32 images; one thread per output pixel.
#include "pp_common.h"

#define AUG_THREADS 256

// ---------------------------------------------------------------- per-sample statistics: mean, std (population), min, max
// over the rectangle rect[n] = {top, left, h, w} of sample n's Hp x Wp plane (rect == nullptr: the whole plane)
__global__ __launch_bounds__(AUG_THREADS) void aug_stats_kernel(const float* __restrict__ x, int Hp, int Wp,
                                                                const int* __restrict__ rect, double* __restrict__ out) {
  __shared__ double sh_s[AUG_THREADS / 64], sh_q[AUG_THREADS / 64];
  __shared__ float sh_lo[AUG_THREADS / 64], sh_hi[AUG_THREADS / 64];
  const int n = blockIdx.x;
  const float* p = x + (size_t)n * Hp * Wp;
  const int top = rect ? rect[n * 4] : 0, left = rect ? rect[n * 4 + 1] : 0, h = rect ? rect[n * 4 + 2] : Hp,
            w = rect ? rect[n * 4 + 3] : Wp;
  double s = 0.0, q = 0.0;
  float lo = 3.0e38f, hi = -3.0e38f;
  for (int i = threadIdx.x; i < h * w; i += AUG_THREADS) {
    const float v = p[(top + i / w) * Wp + left + i % w];
    s += v; q += (double)v * v;
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  s = pp_wave_sum_d(s); q = pp_wave_sum_d(q);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh_s[wv] = s; sh_q[wv] = q; sh_lo[wv] = lo; sh_hi[wv] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double S = 0.0, Q = 0.0;
    for (int i = 0; i < AUG_THREADS / 64; ++i) { S += sh_s[i]; Q += sh_q[i]; lo = fminf(lo, sh_lo[i]); hi = fmaxf(hi, sh_hi[i]); }
    const double cnt = (double)h * w;
    const double mean = cnt > 0 ? S / cnt : 0.0;
    double var = cnt > 0 ? Q / cnt - mean * mean : 0.0;
    if (var < 0.0) var = 0.0;
    double* o = out + (size_t)n * 4;
    o[0] = mean; o[1] = sqrt(var); o[2] = cnt > 0 ? lo : 0.0; o[3] = cnt > 0 ? hi : 0.0;
  }
}

extern "C" int pp_aug_stats(const float* x, int B, int Hp, int Wp, const int* rect, double* stats, void* stream) {
  PP_CHECK_ARG(x && stats && B >= 1 && Hp >= 1 && Wp >= 1, "aug_stats: bad arguments");
  hipLaunchKernelGGL(aug_stats_kernel, dim3(B), dim3(AUG_THREADS), 0, (hipStream_t)stream, x, Hp, Wp, rect, stats);
  return pp_launch_status("aug_stats");
}

// ---------------------------------------------------------------- coefficients of the per-sample scalar maps, built on the device
// from the statistics (no device -> host round trip).  param[n] <= PP_AUG_SKIP (-1e30) means "transform not drawn".
//   mode 0  MeanStdNorm        (augmentations.py:11-21):    a = 1 / (std + eps), b = -mean a
//   mode 1  Contrast           (augmentations.py:112-129):  a = s, b = mean (1 - s), clip to [min, max];  s = param
//   mode 2  Gamma, power step  (augmentations.py:131-166):  {min, max - min + eps, gamma};                gamma = param
//   mode 3  Gamma, retain_stats: a = std0 / (std1 + eps), b = mean0 - mean1 a  (stats = after the power, stats0 = before)
//   mode 4  Brightness         (augmentations.py:97-110):   a = 1, b = s
#define PP_AUG_SKIP (-1.0e30f)
__global__ void aug_coef_kernel(const double* __restrict__ stats, const double* __restrict__ stats0, const float* __restrict__ param,
                                int mode, int B, float* __restrict__ coef) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= B) return;
  const float eps = 1e-8f, inf = 3.0e38f;
  float* c = coef + n * 4;
  const bool skip = param && param[n] <= PP_AUG_SKIP;
  c[0] = 1.f; c[1] = 0.f; c[2] = -inf; c[3] = inf;
  if (mode == 2) { c[0] = 0.f; c[1] = 1.f; c[2] = -1.f; c[3] = 0.f; }
  if (skip) return;
  const double* s = stats ? stats + n * 4 : nullptr;
  if (mode == 0) {
    const float a = 1.f / ((float)s[1] + eps);
    c[0] = a; c[1] = -(float)s[0] * a;
  } else if (mode == 1) {
    c[0] = param[n]; c[1] = (float)s[0] * (1.f - param[n]); c[2] = (float)s[2]; c[3] = (float)s[3];
  } else if (mode == 2) {
    c[0] = (float)s[2]; c[1] = (float)s[3] - (float)s[2] + eps; c[2] = param[n];
  } else if (mode == 3) {
    const double* s0 = stats0 + n * 4;
    const float a = (float)s0[1] / ((float)s[1] + eps);
    c[0] = a; c[1] = (float)s0[0] - (float)s[0] * a;
  } else {
    c[1] = param[n];
  }
}

extern "C" int pp_aug_coef(const double* stats, const double* stats0, const float* param, int mode, int B, float* coef,
                           void* stream) {
  PP_CHECK_ARG(coef && B >= 1 && mode >= 0 && mode <= 4, "aug_coef: bad arguments");
  PP_CHECK_ARG((mode == 4 || stats) && (mode != 3 || stats0) && (mode == 0 || param), "aug_coef: missing statistics / parameters");
  hipLaunchKernelGGL(aug_coef_kernel, dim3(pp_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, stats, stats0, param, mode, B, coef);
  return pp_launch_status("aug_coef");
}

// rectangle test shared by the elementwise kernels
__device__ __forceinline__ bool in_rect(const int* rect, int n, int y, int x) {
  if (!rect) return true;
  const int* r = rect + n * 4;
  return y >= r[0] && y < r[0] + r[2] && x >= r[1] && x < r[1] + r[3];
}

// ---------------------------------------------------------------- x <- clip(a[n] * x + b[n], lo[n], hi[n]) inside rect[n]
__global__ void aug_scalar_map_kernel(float* __restrict__ x, int Hp, int Wp, long long total, const float* __restrict__ coef,
                                      const int* __restrict__ rect) {
  const int HW = Hp * Wp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / HW), p = (int)(i % HW);
    if (!in_rect(rect, n, p / Wp, p % Wp)) continue;
    const float* c = coef + n * 4;
    x[i] = fminf(fmaxf(c[0] * x[i] + c[1], c[2]), c[3]);
  }
}

extern "C" int pp_aug_scalar_map(float* x, int B, int Hp, int Wp, const float* coef, const int* rect, void* stream) {
  PP_CHECK_ARG(x && coef && B >= 1 && Hp >= 1 && Wp >= 1, "aug_scalar_map: bad arguments");
  const long long total = (long long)B * Hp * Wp;
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aug_scalar_map_kernel, dim3(blocks), dim3(AUG_THREADS), 0, (hipStream_t)stream, x, Hp, Wp, total, coef, rect);
  return pp_launch_status("aug_scalar_map");
}

// ---------------------------------------------------------------- x <- ((x - lo[n]) / range[n]) ^ gamma[n] inside rect[n]   (gamma <= 0: untouched)
__global__ void aug_gamma_kernel(float* __restrict__ x, int Hp, int Wp, long long total, const float* __restrict__ coef,
                                 const int* __restrict__ rect) {
  const int HW = Hp * Wp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / HW), p = (int)(i % HW);
    const float* c = coef + n * 4;               // {min, max - min + eps, gamma, unused}
    if (c[2] > 0.f && in_rect(rect, n, p / Wp, p % Wp)) x[i] = powf(fmaxf((x[i] - c[0]) / c[1], 0.f), c[2]);
  }
}

extern "C" int pp_aug_gamma(float* x, int B, int Hp, int Wp, const float* coef, const int* rect, void* stream) {
  PP_CHECK_ARG(x && coef && B >= 1 && Hp >= 1 && Wp >= 1, "aug_gamma: bad arguments");
  const long long total = (long long)B * Hp * Wp;
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aug_gamma_kernel, dim3(blocks), dim3(AUG_THREADS), 0, (hipStream_t)stream, x, Hp, Wp, total, coef, rect);
  return pp_launch_status("aug_gamma");
}

// ---------------------------------------------------------------- additive Gaussian noise: x += sigma[n] * N(0,1) inside rect[n]
// Philox-4x32-10 keyed by the seed, counter = (pixel quad, sample); Box-Muller on the four outputs.  Reproducible for
// a given seed and independent of the launch geometry.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void aug_noise_kernel(float* __restrict__ x, int Hp, int Wp, int B, const float* __restrict__ sigma,
                                 const int* __restrict__ rect, unsigned long long seed) {
  const int HW = Hp * Wp, quads = (HW + 3) / 4;
  const long long total = (long long)B * quads;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / quads), qd = (int)(i % quads);
    const float sg = sigma[n];
    if (sg <= 0.f) continue;
    unsigned r[4];
    philox4x32_10((unsigned)qd, 0u, (unsigned)n, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float u1 = ((float)(r[2 * h] >> 8) + 0.5f) * 5.9604644775390625e-8f;      // (0, 1), 24 bits
      const float u2 = ((float)(r[2 * h + 1] >> 8) + 0.5f) * 5.9604644775390625e-8f;
      const float rad = sqrtf(-2.f * logf(u1));
      z[2 * h] = rad * cosf(6.283185307179586f * u2);
      z[2 * h + 1] = rad * sinf(6.283185307179586f * u2);
    }
    float* p = x + (size_t)n * HW;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int px = qd * 4 + e;
      if (px < HW && in_rect(rect, n, px / Wp, px % Wp)) p[px] += sg * z[e];
    }
  }
}

extern "C" int pp_aug_add_noise(float* x, int B, int Hp, int Wp, const float* sigma, const int* rect, unsigned long long seed,
                                void* stream) {
  PP_CHECK_ARG(x && sigma && B >= 1 && Hp >= 1 && Wp >= 1, "aug_add_noise: bad arguments");
  const long long total = (long long)B * ((Hp * Wp + 3) / 4);
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aug_noise_kernel, dim3(blocks), dim3(AUG_THREADS), 0, (hipStream_t)stream, x, Hp, Wp, B, sigma, rect, seed);
  return pp_launch_status("aug_add_noise");
}

// ---------------------------------------------------------------- the one resampling kernel
// For every output pixel (yo, xo) of sample n:  (ys, xs) = A[n] (yo, xo, 1), the source coordinates in the sample's
// hs x ws slice (stored in an Hp x Wp plane); with a displacement field d (elastic transform) the point moves to
// (ys, xs) + d[n](yo, xo), clamped to the slice like scipy's mode='nearest' if (ys, xs) itself was inside.
// Image: bicubic (cubic = 1: Keys, a = -0.75, as cv2.INTER_CUBIC), bilinear (0) or nearest (2), taps outside the slice read `img_pad`, the result
// is clipped to the sample's [min, max] (clip_stats, as skimage / the elastic transform do with clip=True).
// Label / scribble: nearest neighbour, `lab_pad` outside.  Outside the canvas rectangle {top, left, ph, pw} (the patch
// RandomCrop copies, datasets/augmentations.py:383-418) everything is padding and valid = 0.
// maps[n] = {a00, a01, a02, a10, a11, a12, top, left, ph, pw, hs, ws}: ys = a00 yo + a01 xo + a02, xs = a10 yo + a11 xo + a12.
#define PP_AUG_MAP_FLOATS 12
__device__ __forceinline__ void keys_weights(float t, float w[4]) {
  const float a = -0.75f;
  w[0] = ((a * (t + 1.f) - 5.f * a) * (t + 1.f) + 8.f * a) * (t + 1.f) - 4.f * a;
  w[1] = ((a + 2.f) * t - (a + 3.f)) * t * t + 1.f;
  w[2] = ((a + 2.f) * (1.f - t) - (a + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
  w[3] = 1.f - w[0] - w[1] - w[2];
}

// ---------------------------------------------------------------- cubic B-spline interpolation as scipy does it
// ElasticTransform (augmentations.py:270) resamples the image with scipy.ndimage.map_coordinates(order = 3, mode = 'nearest'):
// the image is padded by 12 pixels with its edge values, prefiltered into B-spline coefficients (float64; one pole
// z = sqrt(3) - 2, gain (1 - z)(1 - 1/z), causal + anticausal recursion with the 'reflect' initialisation -- scipy's
// ni_splines.c), and evaluated with the four cubic B-spline weights.  oracle/augment_oracle.py restates the algorithm and
// checks it against scipy; these kernels follow it line by line.
#define PP_SPLINE_PAD 12
__device__ __forceinline__ void spline_weights(double t, double* w) {
  const double z = 1.0 - t;
  w[1] = (t * t * (t - 2.0) * 3.0 + 4.0) / 6.0;
  w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
  w[0] = z * z * z / 6.0;
  w[3] = 1.0 - w[0] - w[1] - w[2];
}
// in-place prefilter of one line c[0], c[stride], ... of n elements
__device__ void spline_prefilter_line(double* c, int n, size_t stride) {
  const double z = -0.26794919243112270647;       // sqrt(3) - 2
  const double gain = (1.0 - z) * (1.0 - 1.0 / z);
  for (int i = 0; i < n; ++i) c[i * stride] *= gain;
  const double z_n = pow(z, (double)n);
  const double c0 = c[0];
  double z_i = z, acc = c[0] + z_n * c[(size_t)(n - 1) * stride];
  for (int i = 1; i < n; ++i) {
    acc += z_i * (c[i * stride] + z_n * c[(size_t)(n - 1 - i) * stride]);
    z_i *= z;
  }
  acc *= z / (1.0 - z_n * z_n);
  c[0] = acc + c0;
  for (int i = 1; i < n; ++i) c[i * stride] += z * c[(i - 1) * stride];
  c[(size_t)(n - 1) * stride] *= z / (z - 1.0);
  for (int i = n - 2; i >= 0; --i) c[i * stride] = z * (c[(i + 1) * stride] - c[i * stride]);
}
// pass 0: edge-padded copy + prefilter along axis 0 (one thread per padded column: coalesced); pass 1: along axis 1
__global__ void aug_spline_prefilter_kernel(const float* __restrict__ img, int B, int Hp, int Wp, const float* __restrict__ maps,
                                            const int* __restrict__ use, double* __restrict__ coef, int pass) {
  const int pitch = Wp + 2 * PP_SPLINE_PAD, lines = pass == 0 ? pitch : Hp + 2 * PP_SPLINE_PAD;
  const long long total = (long long)B * lines;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / lines), l = (int)(i % lines);
    if (!use[n]) continue;
    const float* mm = maps + (size_t)n * PP_AUG_MAP_FLOATS;
    const int hs = (int)mm[10], ws = (int)mm[11];
    const int Hq = hs + 2 * PP_SPLINE_PAD, Wq = ws + 2 * PP_SPLINE_PAD;
    double* cf = coef + (size_t)n * (Hp + 2 * PP_SPLINE_PAD) * pitch;
    if (pass == 0) {
      if (l >= Wq) continue;
      const int xs = min(max(l - PP_SPLINE_PAD, 0), ws - 1);
      const float* si = img + (size_t)n * Hp * Wp;
      for (int y = 0; y < Hq; ++y) cf[(size_t)y * pitch + l] = (double)si[(size_t)min(max(y - PP_SPLINE_PAD, 0), hs - 1) * Wp + xs];
      spline_prefilter_line(cf + l, Hq, (size_t)pitch);
    } else {
      if (l >= Hq) continue;
      spline_prefilter_line(cf + (size_t)l * pitch, Wq, 1);
    }
  }
}

extern "C" int pp_aug_spline_prefilter(const float* img, int B, int Hp, int Wp, const float* maps, const int* use, double* coef,
                                       void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(img && maps && use && coef && B >= 1 && Hp >= 1 && Wp >= 1, "aug_spline_prefilter: bad arguments");
  hipLaunchKernelGGL(aug_spline_prefilter_kernel, dim3(pp_cdiv((long long)B * (Wp + 2 * PP_SPLINE_PAD), 64)), dim3(64), 0, s, img, B, Hp,
                     Wp, maps, use, coef, 0);
  hipLaunchKernelGGL(aug_spline_prefilter_kernel, dim3(pp_cdiv((long long)B * (Hp + 2 * PP_SPLINE_PAD), 64)), dim3(64), 0, s, img, B, Hp,
                     Wp, maps, use, coef, 1);
  return pp_launch_status("aug_spline_prefilter");
}

__global__ void aug_warp_kernel(const float* __restrict__ img, const int* __restrict__ lab, const int* __restrict__ scb,
                                int Hp, int Wp, float* __restrict__ oimg, int* __restrict__ olab, int* __restrict__ oscb,
                                float* __restrict__ ovalid, int Ho, int Wo, int B, const float* __restrict__ maps,
                                const float* __restrict__ disp /* nullable: [B][2][Ho][Wo] (dy, dx) in source pixels */,
                                const double* __restrict__ clip_stats /* nullable: [B][4], min / max at [2] / [3] */,
                                float img_pad, int lab_pad, int cubic,
                                const double* __restrict__ spl /* nullable: cubic B-spline coefficients [B][Hp + 24][Wp + 24] */,
                                const int* __restrict__ spl_use /* [B]: sample n interpolates its image with the spline */,
                                const double* __restrict__ disp64 /* nullable: disp in double (replayed reference fields) */) {
  const long long total = (long long)B * Ho * Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / ((long long)Ho * Wo)), p = (int)(i % ((long long)Ho * Wo)), yo = p / Wo, xo = p % Wo;
    const float* mm = maps + (size_t)n * PP_AUG_MAP_FLOATS;
    const int top = (int)mm[6], left = (int)mm[7], ph = (int)mm[8], pw = (int)mm[9], hs = (int)mm[10], ws = (int)mm[11];
    const bool valid = yo >= top && yo < top + ph && xo >= left && xo < left + pw;
    if (ovalid) ovalid[i] = valid ? 1.f : 0.f;
    if (!valid) {
      oimg[i] = img_pad;
      if (olab) olab[i] = lab_pad;
      if (oscb) oscb[i] = lab_pad;
      continue;
    }
    float ys = mm[0] * yo + mm[1] * xo + mm[2], xs = mm[3] * yo + mm[4] * xo + mm[5];
    // double-precision source coordinates of the spline path: the reference adds its float64 displacement to the integer
    // grid (augmentations.py:268); (yd, xd) before the clamp is what map_coordinates receives
    double yd = (double)mm[0] * yo + (double)mm[1] * xo + (double)mm[2], xd = (double)mm[3] * yo + (double)mm[4] * xo + (double)mm[5];
    double yu = yd, xu = xd;                        // unclamped (the spline clamps in its own, padded, frame)
    if (disp || disp64) {
      const bool inside = ys >= -0.5f && ys < hs - 0.5f && xs >= -0.5f && xs < ws - 0.5f;
      const size_t o0 = ((size_t)n * 2 + 0) * Ho * Wo + p, o1 = ((size_t)n * 2 + 1) * Ho * Wo + p;
      const double dyv = disp64 ? disp64[o0] : (double)disp[o0], dxv = disp64 ? disp64[o1] : (double)disp[o1];
      yd += dyv; xd += dxv;
      yu = yd; xu = xd;
      ys += (float)dyv;
      xs += (float)dxv;
      if (inside) {
        ys = fminf(fmaxf(ys, 0.f), hs - 1.f); xs = fminf(fmaxf(xs, 0.f), ws - 1.f);
        yd = fmin(fmax(yd, 0.0), hs - 1.0); xd = fmin(fmax(xd, 0.0), ws - 1.0);
      }
    }
    const float* si = img + (size_t)n * Hp * Wp;
    const bool spline = spl && spl_use && spl_use[n];
    // class maps: order 0, floor(c + 0.5) of the clamped coordinate (scipy NI_GeometricTransform); in double on the spline
    // path so that the rounding decision is the reference's
    const int yn = spline ? (int)floor(yd + 0.5) : (int)floorf(ys + 0.5f), xn = spline ? (int)floor(xd + 0.5) : (int)floorf(xs + 0.5f);
    const bool in_src = (unsigned)yn < (unsigned)hs && (unsigned)xn < (unsigned)ws;
    if (olab) olab[i] = in_src ? lab[(size_t)n * Hp * Wp + yn * Wp + xn] : lab_pad;
    if (oscb) oscb[i] = in_src ? scb[(size_t)n * Hp * Wp + yn * Wp + xn] : lab_pad;
    float v;
    const int y0 = (int)floorf(ys), x0 = (int)floorf(xs);
    if (spline) {
      // scipy.ndimage.map_coordinates(order = 3, mode = 'nearest') on the prefiltered, edge-padded coefficients: coordinate + 12
      // clamped to the padded array, taps floor(c) - 1 .. + 2 (indices clamped), weights of get_spline_interpolation_weights
      const int Hq = hs + 2 * PP_SPLINE_PAD, Wq = ws + 2 * PP_SPLINE_PAD, pitch = Wp + 2 * PP_SPLINE_PAD;
      const double* cf = spl + (size_t)n * (Hp + 2 * PP_SPLINE_PAD) * pitch;
      const double cy = fmin(fmax(yu + PP_SPLINE_PAD, 0.0), Hq - 1.0), cx = fmin(fmax(xu + PP_SPLINE_PAD, 0.0), Wq - 1.0);
      const double fy = floor(cy), fx = floor(cx);
      double wy[4], wx[4];
      spline_weights(cy - fy, wy);
      spline_weights(cx - fx, wx);
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int yy = min(max((int)fy - 1 + r, 0), Hq - 1);
        double row = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int xx = min(max((int)fx - 1 + c, 0), Wq - 1);
          row += wx[c] * cf[(size_t)yy * pitch + xx];
        }
        acc += wy[r] * row;
      }
      v = (float)acc;
    } else if (cubic == 2) {                      // nearest neighbour (the down-sampling half of SimulationLowRes, order 0)
      v = in_src ? si[yn * Wp + xn] : img_pad;
    } else if (cubic) {
      float wy[4], wx[4];
      keys_weights(ys - y0, wy);
      keys_weights(xs - x0, wx);
      v = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int yy = y0 - 1 + r;
        float row = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int xx = x0 - 1 + c;
          const float s = ((unsigned)yy < (unsigned)hs && (unsigned)xx < (unsigned)ws) ? si[yy * Wp + xx] : img_pad;
          row += wx[c] * s;
        }
        v += wy[r] * row;
      }
    } else {
      const float ty = ys - y0, tx = xs - x0;
      float s[4];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int yy = y0 + r, xx = x0 + c;
          s[r * 2 + c] = ((unsigned)yy < (unsigned)hs && (unsigned)xx < (unsigned)ws) ? si[yy * Wp + xx] : img_pad;
        }
      v = (1.f - ty) * ((1.f - tx) * s[0] + tx * s[1]) + ty * ((1.f - tx) * s[2] + tx * s[3]);
    }
    if (clip_stats) v = fminf(fmaxf(v, fminf((float)clip_stats[n * 4 + 2], img_pad)), fmaxf((float)clip_stats[n * 4 + 3], img_pad));
    if (!in_src) v = img_pad;
    oimg[i] = v;
  }
}

extern "C" int pp_aug_warp(const float* img, const int* lab, const int* scb, int Hp, int Wp, float* out_img, int* out_lab,
                           int* out_scb, float* out_valid, int Ho, int Wo, int B, const float* maps, const float* disp,
                           const double* clip_stats, float img_pad, int lab_pad, int cubic, void* stream) {
  PP_CHECK_ARG(img && out_img && maps && B >= 1 && Hp >= 1 && Wp >= 1 && Ho >= 1 && Wo >= 1, "aug_warp: bad arguments");
  PP_CHECK_ARG((!out_lab || lab) && (!out_scb || scb), "aug_warp: class-map output without input");
  const long long total = (long long)B * Ho * Wo;
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(aug_warp_kernel, dim3(blocks), dim3(AUG_THREADS), 0, (hipStream_t)stream, img, lab, scb, Hp, Wp, out_img,
                     out_lab, out_scb, out_valid, Ho, Wo, B, maps, disp, clip_stats, img_pad, lab_pad, cubic,
                     (const double*)nullptr, (const int*)nullptr, (const double*)nullptr);
  return pp_launch_status("aug_warp");
}

// the same resampling with the samples flagged in `use` interpolating their image with scipy's cubic B-spline from the
// coefficients of pp_aug_spline_prefilter (ElasticTransform's interpolant, augmentations.py:270); disp64 (nullable) replaces
// disp with a double-precision field (tests replay the reference's own float64 fields through it)
extern "C" int pp_aug_warp_spline(const float* img, const int* lab, const int* scb, int Hp, int Wp, float* out_img, int* out_lab,
                                  int* out_scb, float* out_valid, int Ho, int Wo, int B, const float* maps, const float* disp,
                                  const double* disp64, const double* clip_stats, float img_pad, int lab_pad, int cubic,
                                  const double* spline_coef, const int* use, void* stream) {
  PP_CHECK_ARG(img && out_img && maps && spline_coef && use && B >= 1 && Hp >= 1 && Wp >= 1 && Ho >= 1 && Wo >= 1, "aug_warp_spline: bad arguments");
  PP_CHECK_ARG((!out_lab || lab) && (!out_scb || scb), "aug_warp_spline: class-map output without input");
  const long long total = (long long)B * Ho * Wo;
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(aug_warp_kernel, dim3(blocks), dim3(AUG_THREADS), 0, (hipStream_t)stream, img, lab, scb, Hp, Wp, out_img,
                     out_lab, out_scb, out_valid, Ho, Wo, B, maps, disp, clip_stats, img_pad, lab_pad, cubic, spline_coef, use, disp64);
  return pp_launch_status("aug_warp_spline");
}

// ---------------------------------------------------------------- elastic displacement fields (augmentations.py:228-276)
// d = gaussian_filter(U(-1, 1), sigma) * alpha per sample and axis: uniform noise from Philox, then a separable
// Gaussian (truncated at 4 sigma like scipy.ndimage.gaussian_filter, 'reflect' boundary) -- two passes over HBM.
__global__ void aug_uniform_kernel(float* __restrict__ out, long long total, unsigned long long seed) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (total + 3) / 4; i += (long long)gridDim.x * blockDim.x) {
    unsigned r[4];
    philox4x32_10((unsigned)i, (unsigned)(i >> 32), 0x5eedu, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i * 4 + e < total) out[i * 4 + e] = ((float)(r[e] >> 8) + 0.5f) * 1.1920928955078125e-7f - 1.f;      // U(-1, 1), 24 bits
  }
}

__device__ __forceinline__ int reflect_index(int i, int n) {       // scipy 'reflect': d c b a | a b c d | d c b a
  if (n == 1) return 0;
  const int period = 2 * n;
  i %= period;
  if (i < 0) i += period;
  return i < n ? i : period - 1 - i;
}

// one pass along `axis` (0: rows / y, 1: columns / x); planes = B * 2 fields of H x W; sigma / alpha per SAMPLE
__global__ void aug_gauss_pass_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int H, int W, int axis,
                                      const float* __restrict__ sigma_alpha /* [B][2] */, int apply_alpha, int pps /* planes per sample */) {
  const long long total = (long long)planes * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int pl = (int)(i / ((long long)H * W)), p = (int)(i % ((long long)H * W)), y = p / W, x = p % W;
    const float sg = sigma_alpha[(pl / pps) * 2], al = sigma_alpha[(pl / pps) * 2 + 1];
    float v = in[i];
    if (sg > 0.f) {
      const int rad = (int)(4.f * sg + 0.5f);
      const float inv = -0.5f / (sg * sg);
      float acc = 0.f, wsum = 0.f;
      const float* base = in + (size_t)pl * H * W;
      for (int k = -rad; k <= rad; ++k) {
        const float w = __expf(inv * k * k);
        const int yy = axis == 0 ? reflect_index(y + k, H) : y, xx = axis == 1 ? reflect_index(x + k, W) : x;
        acc += w * base[yy * W + xx];
        wsum += w;
      }
      v = acc / wsum;
    } else if (apply_alpha) {
      v = 0.f;                                    // no elastic transform for this sample
    }                                             // (a blur with sigma <= 0 copies the plane)
    out[i] = apply_alpha ? v * al : v;
  }
}

extern "C" int pp_aug_elastic_field(float* disp, float* scratch, int B, int H, int W, const float* sigma_alpha,
                                    unsigned long long seed, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(disp && scratch && sigma_alpha && B >= 1 && H >= 1 && W >= 1, "aug_elastic_field: bad arguments");
  const long long total = (long long)B * 2 * H * W;
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(aug_uniform_kernel, dim3(blocks), dim3(AUG_THREADS), 0, s, disp, total, seed);
  hipLaunchKernelGGL(aug_gauss_pass_kernel, dim3(blocks), dim3(AUG_THREADS), 0, s, disp, scratch, B * 2, H, W, 0, sigma_alpha, 0, 2);
  hipLaunchKernelGGL(aug_gauss_pass_kernel, dim3(blocks), dim3(AUG_THREADS), 0, s, scratch, disp, B * 2, H, W, 1, sigma_alpha, 1, 2);
  return pp_launch_status("aug_elastic_field");
}

// ---------------------------------------------------------------- GaussianBlur (augmentations.py:82-95): scipy.ndimage.gaussian_filter
// per sample, sigma_pad[n] = {sigma, unused}; sigma <= 0 leaves the sample untouched.  In place through `scratch`.
extern "C" int pp_aug_gaussian_blur(float* x, float* scratch, int B, int H, int W, const float* sigma_pad, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(x && scratch && sigma_pad && B >= 1 && H >= 1 && W >= 1, "aug_gaussian_blur: bad arguments");
  const long long total = (long long)B * H * W;
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(aug_gauss_pass_kernel, dim3(blocks), dim3(AUG_THREADS), 0, s, x, scratch, B, H, W, 0, sigma_pad, 0, 1);
  hipLaunchKernelGGL(aug_gauss_pass_kernel, dim3(blocks), dim3(AUG_THREADS), 0, s, scratch, x, B, H, W, 1, sigma_pad, 0, 1);
  return pp_launch_status("aug_gaussian_blur");
}

// ---------------------------------------------------------------- Mixup (augmentations.py:51-80): x <- lam[n] x + (1 - lam[n]) y;  lam[n] < 0: untouched
__global__ void aug_mix_kernel(float* __restrict__ x, const float* __restrict__ y, int HW, long long total, const float* __restrict__ lam) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const float l = lam[i / HW];
    if (l >= 0.f) x[i] = x[i] * l + y[i] * (1.f - l);
  }
}

extern "C" int pp_aug_mix(float* x, const float* y, int B, int HW, const float* lam, void* stream) {
  PP_CHECK_ARG(x && y && lam && B >= 1 && HW >= 1, "aug_mix: bad arguments");
  const long long total = (long long)B * HW;
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aug_mix_kernel, dim3(blocks), dim3(AUG_THREADS), 0, (hipStream_t)stream, x, y, HW, total, lam);
  return pp_launch_status("aug_mix");
}

// ---------------------------------------------------------------- x <- x + f inside rect[n]: GaussianNoise (augmentations.py:365) with the normal field
// supplied by the caller instead of pp_aug_add_noise's Philox stream (replaying the reference's own draws)
__global__ void aug_add_field_kernel(float* __restrict__ x, const float* __restrict__ f, int Hp, int Wp, long long total,
                                     const int* __restrict__ rect) {
  const int HW = Hp * Wp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / HW), p = (int)(i % HW);
    if (in_rect(rect, n, p / Wp, p % Wp)) x[i] += f[i];
  }
}

extern "C" int pp_aug_add_field(float* x, const float* f, int B, int Hp, int Wp, const int* rect, void* stream) {
  PP_CHECK_ARG(x && f && B >= 1 && Hp >= 1 && Wp >= 1, "aug_add_field: bad arguments");
  const long long total = (long long)B * Hp * Wp;
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aug_add_field_kernel, dim3(blocks), dim3(AUG_THREADS), 0, (hipStream_t)stream, x, f, Hp, Wp, total, rect);
  return pp_launch_status("aug_add_field");
}

// ---------------------------------------------------------------- one-hot encoding (augmentations.py:421-461)
__global__ void aug_onehot_kernel(const int* __restrict__ lab, float* __restrict__ out, int K, int HW, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / ((long long)K * HW);
    const int k = (int)((i / HW) % K), p = (int)(i % HW);
    out[i] = lab[n * HW + p] == k ? 1.f : 0.f;
  }
}

extern "C" int pp_aug_onehot(const int* lab, float* out, int B, int K, int HW, void* stream) {
  PP_CHECK_ARG(lab && out && B >= 1 && K >= 1 && HW >= 1, "aug_onehot: bad arguments");
  const long long total = (long long)B * K * HW;
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(aug_onehot_kernel, dim3(blocks), dim3(AUG_THREADS), 0, (hipStream_t)stream, lab, out, K, HW, total);
  return pp_launch_status("aug_onehot");
}
