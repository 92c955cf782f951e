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
__global__ __launch_bounds__(AUG_THREADS) void aug_stats_kernel(const float* __restrict__ x, int HW, double* __restrict__ out) {
  __shared__ double sh_s[AUG_THREADS / 64], sh_q[AUG_THREADS / 64];
  __shared__ float sh_lo[AUG_THREADS / 64], sh_hi[AUG_THREADS / 64];
  const float* p = x + (size_t)blockIdx.x * HW;
  double s = 0.0, q = 0.0;
  float lo = 3.0e38f, hi = -3.0e38f;
  for (int i = threadIdx.x; i < HW; i += AUG_THREADS) {
    const float v = p[i];
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
    const double mean = S / HW;
    double var = Q / HW - mean * mean;
    if (var < 0.0) var = 0.0;
    double* o = out + (size_t)blockIdx.x * 4;
    o[0] = mean; o[1] = sqrt(var); o[2] = lo; o[3] = hi;
  }
}

extern "C" int pp_aug_stats(const float* x, int B, int HW, double* stats, void* stream) {
  PP_CHECK_ARG(x && stats && B >= 1 && HW >= 1, "aug_stats: bad arguments");
  hipLaunchKernelGGL(aug_stats_kernel, dim3(B), dim3(AUG_THREADS), 0, (hipStream_t)stream, x, HW, stats);
  return pp_launch_status("aug_stats");
}

// ---------------------------------------------------------------- x <- clip(a[n] * x + b[n], lo[n], hi[n])
// coef[n] = {a, b, lo, hi}: MeanStdNorm (a = 1/(std+eps), b = -mean a), Brightness (a = 1, b = s), Contrast
// (a = s, b = mean (1 - s), lo / hi = min / max of the sample), the re-normalisation of GammaAugmentation.
__global__ void aug_scalar_map_kernel(float* __restrict__ x, int HW, long long total, const float* __restrict__ coef) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const float* c = coef + (i / HW) * 4;
    x[i] = fminf(fmaxf(c[0] * x[i] + c[1], c[2]), c[3]);
  }
}

extern "C" int pp_aug_scalar_map(float* x, int B, int HW, const float* coef, void* stream) {
  PP_CHECK_ARG(x && coef && B >= 1 && HW >= 1, "aug_scalar_map: bad arguments");
  const long long total = (long long)B * HW;
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aug_scalar_map_kernel, dim3(blocks), dim3(AUG_THREADS), 0, (hipStream_t)stream, x, HW, total, coef);
  return pp_launch_status("aug_scalar_map");
}

// ---------------------------------------------------------------- x <- ((x - lo[n]) / (range[n] + eps)) ^ gamma[n]   (gamma <= 0: untouched)
__global__ void aug_gamma_kernel(float* __restrict__ x, int HW, long long total, const float* __restrict__ coef) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const float* c = coef + (i / HW) * 4;               // {min, max - min + eps, gamma, unused}
    if (c[2] > 0.f) x[i] = powf(fmaxf((x[i] - c[0]) / c[1], 0.f), c[2]);
  }
}

extern "C" int pp_aug_gamma(float* x, int B, int HW, const float* coef, void* stream) {
  PP_CHECK_ARG(x && coef && B >= 1 && HW >= 1, "aug_gamma: bad arguments");
  const long long total = (long long)B * HW;
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aug_gamma_kernel, dim3(blocks), dim3(AUG_THREADS), 0, (hipStream_t)stream, x, HW, total, coef);
  return pp_launch_status("aug_gamma");
}

// ---------------------------------------------------------------- additive Gaussian noise: x += sigma[n] * N(0,1)
// Philox-4x32-10 keyed by (seed, sample), counter = pixel quad; Box-Muller on the four outputs.  Reproducible for a
// given seed and independent of the launch geometry.
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

__global__ void aug_noise_kernel(float* __restrict__ x, int HW, int B, const float* __restrict__ sigma,
                                 unsigned long long seed) {
  const int quads = (HW + 3) / 4;
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
      const float u1 = ((float)r[2 * h] + 0.5f) * 2.3283064365386963e-10f;      // (0, 1)
      const float u2 = ((float)r[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
      const float rad = sqrtf(-2.f * logf(u1));
      z[2 * h] = rad * cosf(6.283185307179586f * u2);
      z[2 * h + 1] = rad * sinf(6.283185307179586f * u2);
    }
    float* p = x + (size_t)n * HW + qd * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (qd * 4 + e < HW) p[e] += sg * z[e];
  }
}

extern "C" int pp_aug_add_noise(float* x, int B, int HW, const float* sigma, unsigned long long seed, void* stream) {
  PP_CHECK_ARG(x && sigma && B >= 1 && HW >= 1, "aug_add_noise: bad arguments");
  const long long total = (long long)B * ((HW + 3) / 4);
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(aug_noise_kernel, dim3(blocks), dim3(AUG_THREADS), 0, (hipStream_t)stream, x, HW, B, sigma, seed);
  return pp_launch_status("aug_add_noise");
}

// ---------------------------------------------------------------- the one resampling kernel
// For every output pixel (yo, xo) of sample n:  (ys, xs) = M[n] * (yo, xo, 1) [+ displacement(yo, xo)], the source
// coordinates in the sample's Hs x Ws slice.  Image: bicubic (Keys, a = -0.75, as cv2.INTER_CUBIC) or bilinear, taps
// outside the slice read `img_pad`; an output pixel whose source falls outside the slice by more than one pixel is
// padding.  Label / scribble: nearest neighbour, `lab_pad` outside.  valid = 1 where the source lies inside the canvas
// rectangle [0, Hc) x [0, Wc) given per sample (the region RandomCrop copies, datasets/augmentations.py:383-418).
// m[n] = {a00, a01, a02, a10, a11, a12, Hc, Wc}: ys = a00 yo + a01 xo + a02, xs = a10 yo + a11 xo + a12.
__device__ __forceinline__ void keys_weights(float t, float w[4]) {
  const float a = -0.75f;
  w[0] = ((a * (t + 1.f) - 5.f * a) * (t + 1.f) + 8.f * a) * (t + 1.f) - 4.f * a;
  w[1] = ((a + 2.f) * t - (a + 3.f)) * t * t + 1.f;
  w[2] = ((a + 2.f) * (1.f - t) - (a + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
  w[3] = 1.f - w[0] - w[1] - w[2];
}

__global__ void aug_warp_kernel(const float* __restrict__ img, const int* __restrict__ lab, const int* __restrict__ scb,
                                int Hs, int Ws, float* __restrict__ oimg, int* __restrict__ olab, int* __restrict__ oscb,
                                float* __restrict__ ovalid, int Ho, int Wo, int B, const float* __restrict__ m,
                                const float* __restrict__ disp /* nullable: [B][2][Ho][Wo] (dy, dx) in source pixels */,
                                float img_pad, int lab_pad, int cubic) {
  const long long total = (long long)B * Ho * Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i / ((long long)Ho * Wo)), p = (int)(i % ((long long)Ho * Wo)), yo = p / Wo, xo = p % Wo;
    const float* mm = m + (size_t)n * 8;
    float ys = mm[0] * yo + mm[1] * xo + mm[2], xs = mm[3] * yo + mm[4] * xo + mm[5];
    if (disp) {
      ys += disp[((size_t)n * 2 + 0) * Ho * Wo + p];
      xs += disp[((size_t)n * 2 + 1) * Ho * Wo + p];
    }
    const float* si = img + (size_t)n * Hs * Ws;
    // nearest neighbour for the class maps (round half away from zero on the sampling grid)
    const int yn = (int)floorf(ys + 0.5f), xn = (int)floorf(xs + 0.5f);
    const bool in_src = (unsigned)yn < (unsigned)Hs && (unsigned)xn < (unsigned)Ws;
    if (olab) olab[i] = in_src ? lab[(size_t)n * Hs * Ws + yn * Ws + xn] : lab_pad;
    if (oscb) oscb[i] = in_src ? scb[(size_t)n * Hs * Ws + yn * Ws + xn] : lab_pad;
    if (ovalid) ovalid[i] = (ys > -0.5f && ys < mm[6] - 0.5f && xs > -0.5f && xs < mm[7] - 0.5f) ? 1.f : 0.f;
    float v;
    const int y0 = (int)floorf(ys), x0 = (int)floorf(xs);
    if (cubic) {
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
          const float s = ((unsigned)yy < (unsigned)Hs && (unsigned)xx < (unsigned)Ws) ? si[yy * Ws + xx] : img_pad;
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
          s[r * 2 + c] = ((unsigned)yy < (unsigned)Hs && (unsigned)xx < (unsigned)Ws) ? si[yy * Ws + xx] : img_pad;
        }
      v = (1.f - ty) * ((1.f - tx) * s[0] + tx * s[1]) + ty * ((1.f - tx) * s[2] + tx * s[3]);
    }
    if (ys <= -1.f || ys >= (float)Hs || xs <= -1.f || xs >= (float)Ws) v = img_pad;
    oimg[i] = v;
  }
}

extern "C" int pp_aug_warp(const float* img, const int* lab, const int* scb, int Hs, int Ws, float* out_img, int* out_lab,
                           int* out_scb, float* out_valid, int Ho, int Wo, int B, const float* maps, const float* disp,
                           float img_pad, int lab_pad, int cubic, void* stream) {
  PP_CHECK_ARG(img && out_img && maps && B >= 1 && Hs >= 1 && Ws >= 1 && Ho >= 1 && Wo >= 1, "aug_warp: bad arguments");
  PP_CHECK_ARG((!out_lab || lab) && (!out_scb || scb), "aug_warp: class-map output without input");
  const long long total = (long long)B * Ho * Wo;
  int blocks = pp_cdiv(total, AUG_THREADS);
  if (blocks > 8192) blocks = 8192;
  pp_prof_begin(PP_K_MISC, 0.0, 4.0 * total * 5.0, (hipStream_t)stream);
  hipLaunchKernelGGL(aug_warp_kernel, dim3(blocks), dim3(AUG_THREADS), 0, (hipStream_t)stream, img, lab, scb, Hs, Ws, out_img,
                     out_lab, out_scb, out_valid, Ho, Wo, B, maps, disp, img_pad, lab_pad, cubic);
  pp_prof_end((hipStream_t)stream);
  return pp_launch_status("aug_warp");
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
      if (i * 4 + e < total) out[i * 4 + e] = ((float)r[e] + 0.5f) * 4.656612873077393e-10f - 1.f;      // U(-1, 1)
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
                                      const float* __restrict__ sigma_alpha /* [B][2] */, int apply_alpha) {
  const long long total = (long long)planes * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int pl = (int)(i / ((long long)H * W)), p = (int)(i % ((long long)H * W)), y = p / W, x = p % W;
    const float sg = sigma_alpha[(pl / 2) * 2], al = sigma_alpha[(pl / 2) * 2 + 1];
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
    }
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
  hipLaunchKernelGGL(aug_gauss_pass_kernel, dim3(blocks), dim3(AUG_THREADS), 0, s, disp, scratch, B * 2, H, W, 0, sigma_alpha, 0);
  hipLaunchKernelGGL(aug_gauss_pass_kernel, dim3(blocks), dim3(AUG_THREADS), 0, s, scratch, disp, B * 2, H, W, 1, sigma_alpha, 1);
  return pp_launch_status("aug_elastic_field");
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
