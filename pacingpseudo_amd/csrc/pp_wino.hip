// Winograd F(2x2, 3x3) convolution for the wide layers (>= 128 channels) of the U-Net, fp32 end to end.
//
// A 3x3 / stride-1 convolution spends 18 flop per (pixel, cin, cout).  F(2x2,3x3) computes each 2x2 output tile from a
// 4x4 input tile with 16 element-wise products in the transform domain: 16/4 = 4 multiply-adds per (pixel, cin, cout)
// = 8 flop, 2.25x less matrix work, at the price of three streaming transform passes.  On MI355X the wide layers are
// bound by the fp32 MFMA rate (157 TFLOP/s) while the transforms run at the HBM rate, so the trade pays from about
// 128 channels up.  fp32 accuracy: single layer 5e-7 vs 2e-7 for the direct form, whole-network
// logits unchanged at 1.2e-5 vs fp64 (measured on the oracle).
//
//   forward / data gradient:   V_b[t][c] = (B^T d B)_b          input transform   (16 planes b, tiles t)
//                              M_b[t][n] = sum_c V_b[t][c] U_b[n][c]   16 batched GEMMs on v_mfma_f32_32x32x2_f32
//                              y = A^T M A + bias              output transform
//   weight gradient:           W_b[t][o] = (A dY A^T)_b,  dU_b[o][c] = sum_t W_b[t][o] V_b[t][c],  dg = G^T dU G
//   dilation d (encoder stages 5/6): the image splits into d*d interleaved sub-images, each an ordinary pad-1 conv.
//
// Reference op: nn.Conv2d 3x3 (models/unet.py:188) and its autograd.
#include "pp_common.h"
#include <stdlib.h>

PP_NS_BEGIN

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define WLD 36          // padded LDS row (floats), see pp_conv.hip
#define WBK 32

struct WinoGeom {
  int N, H, W, dil;     // images, image size, dilation
  int Hs, Ws;           // sub-image size  (H/dil, W/dil)
  int m;                // output tile edge: 2 = F(2x2,3x3) (16 planes), 4 = F(4x4,3x3) (36 planes)
  int th, tw;           // tiles per sub-image (Hs/m, Ws/m)
  int T;                // total tiles = N * dil*dil * th * tw
  int nb;               // planes = (m+2)^2
};
// F(4x4,3x3) whenever the (sub-)image is a multiple of 4: 4.5 flop per pixel*cin*cout instead of 8 (direct: 18) and
// 2.25x instead of 4x transform-domain data; its fp32 error (1e-5 per layer, 2.8e-5 on the logits of the whole network
// vs fp64, measured on the oracle) stays inside the 1e-4 budget.  Other sizes run F(2x2,3x3).
static inline int wino_tile(int H, int W, int dil) {
  return (H % (4 * dil) == 0 && W % (4 * dil) == 0) ? 4 : 2;
}
static inline WinoGeom wino_geom(int N, int H, int W, int dil) {
  const int m = wino_tile(H, W, dil);
  WinoGeom g{N, H, W, dil, H / dil, W / dil, m, H / dil / m, W / dil / m, 0, (m + 2) * (m + 2)};
  g.T = N * dil * dil * g.th * g.tw;
  return g;
}
// tile index -> image n, sub-image offset (sy,sx), tile coordinates (ty,tx)
__device__ __forceinline__ void tile_coords(const WinoGeom& g, int t, int& n, int& sy, int& sx, int& ty, int& tx) {
  tx = t % g.tw; t /= g.tw;
  ty = t % g.th; t /= g.th;
  sx = t % g.dil; t /= g.dil;
  sy = t % g.dil;
  n = t / g.dil;
}

#ifndef PP_ACT_16     // the F(2x2,3x3) kernels and the fp32-operand F(4x4,3x3) transforms exist for fp32 activations only
// ---------------------------------------------------------------- input transform  V = B^T d B
// one thread per (tile, channel quad); writes 16 planes [b][T][C]
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int ld, int C, WinoGeom g,
                                                         float* __restrict__ V) {
  const int c4n = C >> 2;
  const long long total = (long long)g.T * c4n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4n);
    const int t = (int)(i / c4n);
    int n, sy, sx, ty, tx;
    tile_coords(g, t, n, sy, sx, ty, tx);
    f32x4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ys = 2 * ty - 1 + r;                       // row inside the sub-image
      const int y = ys * g.dil + sy;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int xs = 2 * tx - 1 + s;
        const int xx = xs * g.dil + sx;
        const bool ok = (unsigned)ys < (unsigned)g.Hs && (unsigned)xs < (unsigned)g.Ws;
        d[r][s] = ok ? *reinterpret_cast<const f32x4*>(x + ((size_t)(n * g.H + y) * g.W + xx) * ld + cq * 4)
                     : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    f32x4 tt[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      tt[0][s] = d[0][s] - d[2][s];
      tt[1][s] = d[1][s] + d[2][s];
      tt[2][s] = d[2][s] - d[1][s];
      tt[3][s] = d[1][s] - d[3][s];
    }
    float* o = V + (size_t)t * C + cq * 4;
    const size_t plane = (size_t)g.T * C;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      *reinterpret_cast<f32x4*>(o + (r * 4 + 0) * plane) = tt[r][0] - tt[r][2];
      *reinterpret_cast<f32x4*>(o + (r * 4 + 1) * plane) = tt[r][1] + tt[r][2];
      *reinterpret_cast<f32x4*>(o + (r * 4 + 2) * plane) = tt[r][2] - tt[r][1];
      *reinterpret_cast<f32x4*>(o + (r * 4 + 3) * plane) = tt[r][1] - tt[r][3];
    }
  }
}

// ---------------------------------------------------------------- output transform  y = A^T M A (+ bias) (+ y)
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, int Nc, WinoGeom g,
                                                          const float* __restrict__ bias, float* __restrict__ y, int ld,
                                                          int accumulate) {
  const int c4n = Nc >> 2;
  const long long total = (long long)g.T * c4n;
  const size_t plane = (size_t)g.T * Nc;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4n);
    const int t = (int)(i / c4n);
    int n, sy, sx, ty, tx;
    tile_coords(g, t, n, sy, sx, ty, tx);
    const float* m = M + (size_t)t * Nc + cq * 4;
    f32x4 s[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(m + (0 * 4 + c) * plane);
      const f32x4 m1 = *reinterpret_cast<const f32x4*>(m + (1 * 4 + c) * plane);
      const f32x4 m2 = *reinterpret_cast<const f32x4*>(m + (2 * 4 + c) * plane);
      const f32x4 m3 = *reinterpret_cast<const f32x4*>(m + (3 * 4 + c) * plane);
      s[0][c] = m0 + m1 + m2;
      s[1][c] = m1 - m2 - m3;
    }
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias) b4 = *reinterpret_cast<const f32x4*>(bias + cq * 4);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const f32x4 o0 = s[r][0] + s[r][1] + s[r][2] + b4;
      const f32x4 o1 = s[r][1] - s[r][2] - s[r][3] + b4;
      const int yy = (2 * ty + r) * g.dil + sy;
      const int x0 = (2 * tx) * g.dil + sx, x1 = (2 * tx + 1) * g.dil + sx;
      f32x4* p0 = reinterpret_cast<f32x4*>(y + ((size_t)(n * g.H + yy) * g.W + x0) * ld + cq * 4);
      f32x4* p1 = reinterpret_cast<f32x4*>(y + ((size_t)(n * g.H + yy) * g.W + x1) * ld + cq * 4);
      *p0 = accumulate ? *p0 + o0 : o0;
      *p1 = accumulate ? *p1 + o1 : o1;
    }
  }
}

// ---------------------------------------------------------------- gradient-side transform  W = A dY A^T  (2x2 -> 4x4)
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, int ld, int O, WinoGeom g,
                                                      float* __restrict__ Wt) {
  const int c4n = O >> 2;
  const long long total = (long long)g.T * c4n;
  const size_t plane = (size_t)g.T * O;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4n);
    const int t = (int)(i / c4n);
    int n, sy, sx, ty, tx;
    tile_coords(g, t, n, sy, sx, ty, tx);
    f32x4 d[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int yy = (2 * ty + r) * g.dil + sy, xx = (2 * tx + s) * g.dil + sx;
        d[r][s] = *reinterpret_cast<const f32x4*>(dy + ((size_t)(n * g.H + yy) * g.W + xx) * ld + cq * 4);
      }
    // rows of A: [1,0] [1,1] [1,-1] [0,-1]
    f32x4 a[4][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      a[0][s] = d[0][s];
      a[1][s] = d[0][s] + d[1][s];
      a[2][s] = d[0][s] - d[1][s];
      a[3][s] = -d[1][s];
    }
    float* o = Wt + (size_t)t * O + cq * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      *reinterpret_cast<f32x4*>(o + (r * 4 + 0) * plane) = a[r][0];
      *reinterpret_cast<f32x4*>(o + (r * 4 + 1) * plane) = a[r][0] + a[r][1];
      *reinterpret_cast<f32x4*>(o + (r * 4 + 2) * plane) = a[r][0] - a[r][1];
      *reinterpret_cast<f32x4*>(o + (r * 4 + 3) * plane) = -a[r][1];
    }
  }
}

#endif  // !PP_ACT_16

// ================================================================ F(4x4,3x3) transforms
// One thread per (tile, channel): 36 scalar loads / stores, each wave-instruction covering 256 contiguous bytes along the
// channel axis (a float4-per-lane form would need 288 VGPRs for the 6x6 tile).
template <class T>
__device__ __forceinline__ void f4_bt(const T* d, T* t) {          // t = B^T d   (6 -> 6)
  t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  t[1] = -4.f * d[1] - 4.f * d[2] + d[3] + d[4];
  t[2] = 4.f * d[1] - 4.f * d[2] - d[3] + d[4];
  t[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
  t[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
  t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
template <class T>
__device__ __forceinline__ void f4_at(const T* m, T* y) {          // y = A^T m   (6 -> 4)
  y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
  y[1] = m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
  y[2] = m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
  y[3] = m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
}
template <class T>
__device__ __forceinline__ void f4_a(const T* d, T* w) {           // w = A d     (4 -> 6)
  w[0] = d[0];
  w[1] = d[0] + d[1] + d[2] + d[3];
  w[2] = d[0] - d[1] + d[2] - d[3];
  w[3] = d[0] + 2.f * d[1] + 4.f * d[2] + 8.f * d[3];
  w[4] = d[0] - 2.f * d[1] + 4.f * d[2] - 8.f * d[3];
  w[5] = d[3];
}
template <class T>
__device__ __forceinline__ void f4_g(const T* g, T* u) {           // u = G g     (3 -> 6)
  u[0] = 0.25f * g[0];
  u[1] = -(g[0] + g[1] + g[2]) * (1.f / 6.f);
  u[2] = -(g[0] - g[1] + g[2]) * (1.f / 6.f);
  u[3] = g[0] * (1.f / 24.f) + g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
  u[4] = g[0] * (1.f / 24.f) - g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
  u[5] = g[2];
}
template <class T>
__device__ __forceinline__ void f4_gt(const T* u, T* p) {          // p = G^T u   (6 -> 3)
  p[0] = 0.25f * u[0] - (u[1] + u[2]) * (1.f / 6.f) + (u[3] + u[4]) * (1.f / 24.f);
  p[1] = (u[2] - u[1]) * (1.f / 6.f) + (u[3] - u[4]) * (1.f / 12.f);
  p[2] = -(u[1] + u[2]) * (1.f / 6.f) + (u[3] + u[4]) * (1.f / 6.f) + u[5];
}

template <int VEC> struct WVec;
template <> struct WVec<1> { typedef float type; };
template <> struct WVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct WVec<4> { typedef float type __attribute__((ext_vector_type(4))); };

// The three F(4x4) data transforms move 1 + 2.25 floats per element and are HBM-bound: one thread per (tile, VEC
// channels) so that every load / store instruction of a wave moves VEC * 256 B (VEC = 4 unless alignment forbids).
// block maximum -> one (conditional) atomic: max is order independent, so the result stays deterministic
__device__ __forceinline__ void wino_block_amax(float mx, float* amax) {
  if (!amax) return;
  __shared__ float wmax[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    // one atomic per block, and only when it can still raise the value (tens of thousands of same-address atomics
    // cost the HBM-bound transform kernels 25 %, r01)
    if (mx > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(mx));
  }
}

// VEC activation elements at p (act_t = float: plain vector access; fp16: converted)
template <int VEC>
__device__ __forceinline__ typename WVec<VEC>::type wv_ld(const act_t* p) {
#ifdef PP_ACT_16
  typedef act_t HT __attribute__((ext_vector_type(VEC == 1 ? 2 : VEC)));
  if constexpr (VEC == 1) return (float)*p;
  else return __builtin_convertvector(*reinterpret_cast<const HT*>(p), typename WVec<VEC>::type);
#else
  return *reinterpret_cast<const typename WVec<VEC>::type*>(p);
#endif
}
template <int VEC>
__device__ __forceinline__ void wv_st(act_t* p, typename WVec<VEC>::type v) {
#ifdef PP_ACT_16
  typedef act_t HT __attribute__((ext_vector_type(VEC == 1 ? 2 : VEC)));
  if constexpr (VEC == 1) *p = (act_t)v;
  else *reinterpret_cast<HT*>(p) = __builtin_convertvector(v, HT);
#else
  *reinterpret_cast<typename WVec<VEC>::type*>(p) = v;
#endif
}

__device__ __forceinline__ float wvec_amax(float v) { return fabsf(v); }
__device__ __forceinline__ float wvec_amax(WVec<2>::type v) { return fmaxf(fabsf(v[0]), fabsf(v[1])); }
__device__ __forceinline__ float wvec_amax(WVec<4>::type v) {
  return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

#ifndef PP_ACT_16
// amax (nullable): device float, zeroed by the caller; receives max |V| (operand scale of the split-fp16 GEMM)
template <int VEC>
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ x, int ld, int C, WinoGeom g,
                                                          float* __restrict__ V, float* __restrict__ amax) {
  typedef typename WVec<VEC>::type T;
  float mx = 0.f;
  const int cv = C / VEC;
  const long long total = (long long)g.T * cv;
  const size_t plane = (size_t)g.T * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    const int t = (int)(i / cv);
    int n, sy, sx, ty, tx;
    tile_coords(g, t, n, sy, sx, ty, tx);
    T tt[6][6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {                     // column pass: rows of the patch through B^T
      T d[6];
      const int xs = 4 * tx - 1 + s;
      const int xx = xs * g.dil + sx;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const int ys = 4 * ty - 1 + r;
        const bool ok = (unsigned)ys < (unsigned)g.Hs && (unsigned)xs < (unsigned)g.Ws;
        d[r] = ok ? *reinterpret_cast<const T*>(x + ((size_t)(n * g.H + ys * g.dil + sy) * g.W + xx) * ld + c) : T(0.f);
      }
      T col[6];
      f4_bt(d, col);
#pragma unroll
      for (int r = 0; r < 6; ++r) tt[r][s] = col[r];
    }
    float* o = V + (size_t)t * C + c;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      T v[6];
      f4_bt(tt[r], v);
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        *reinterpret_cast<T*>(o + (r * 6 + s) * plane) = v[s];
        mx = fmaxf(mx, wvec_amax(v[s]));
      }
    }
  }
  wino_block_amax(mx, amax);
}
#endif  // !PP_ACT_16

template <int VEC>
__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ M, int Nc, WinoGeom g,
                                                           const float* __restrict__ bias, act_t* __restrict__ y, int ld,
                                                           int accumulate) {
  typedef typename WVec<VEC>::type T;
  const int cv = Nc / VEC;
  const long long total = (long long)g.T * cv;
  const size_t plane = (size_t)g.T * Nc;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    const int t = (int)(i / cv);
    int n, sy, sx, ty, tx;
    tile_coords(g, t, n, sy, sx, ty, tx);
    const float* m = M + (size_t)t * Nc + c;
    T s4[4][6];
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) {
      T col[6], yc[4];
#pragma unroll
      for (int r = 0; r < 6; ++r) col[r] = *reinterpret_cast<const T*>(m + (r * 6 + cc) * plane);
      f4_at(col, yc);
#pragma unroll
      for (int r = 0; r < 4; ++r) s4[r][cc] = yc[r];
    }
    const T bv = bias ? *reinterpret_cast<const T*>(bias + c) : T(0.f);
    T old[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {          // all 16 reads first: read-add-write per pixel serialises the round trips
        old[r][q] = T(0.f);
        if (accumulate)
          old[r][q] = wv_ld<VEC>(y + ((size_t)(n * g.H + (4 * ty + r) * g.dil + sy) * g.W + (4 * tx + q) * g.dil + sx) * ld + c);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T o[4];
      f4_at(s4[r], o);
      const int yy = (4 * ty + r) * g.dil + sy;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        wv_st<VEC>(y + ((size_t)(n * g.H + yy) * g.W + (4 * tx + q) * g.dil + sx) * ld + c, o[q] + bv + old[r][q]);
      }
    }
  }
}

// The same output transform with the BatchNorm that follows the convolution fused in (PpEpi, pp_common.h): mode 1 also
// emits per-block (sum z, sum z^2) partials per channel, mode 2 writes y = lrelu(z * scale + shift) directly.
// Requires gridDim.x * 256 to be a multiple of cv = Nc / VEC, so that a thread keeps ONE channel vector over its
// grid-stride loop (checked by the host: cv divides 256 or is a multiple of it).
template <int VEC>
__global__ __launch_bounds__(256) void wino4_output_bn_kernel(const float* __restrict__ M, int Nc, WinoGeom g,
                                                              const float* __restrict__ bias, act_t* __restrict__ y, int ld,
                                                              PpEpi e, int imgs_per_group) {
  typedef typename WVec<VEC>::type T;
  __shared__ float red[256 * 4 * VEC];
  const int cv = Nc / VEC;
  const long long total = (long long)g.T * cv;
  const size_t plane = (size_t)g.T * Nc;
  const int c = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) % cv) * VEC;      // fixed for this thread
  const T bv = bias ? *reinterpret_cast<const T*>(bias + c) : T(0.f);
  const T sc = e.mode == 2 ? *reinterpret_cast<const T*>(e.scale + c) : T(1.f);
  const T sh = e.mode == 2 ? *reinterpret_cast<const T*>(e.shift + c) : T(0.f);
  T s0 = T(0.f), q0 = T(0.f), s1 = T(0.f), q1 = T(0.f);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i / cv);
    int n, sy, sx, ty, tx;
    tile_coords(g, t, n, sy, sx, ty, tx);
    const float* m = M + (size_t)t * Nc + c;
    T s4[4][6];
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) {
      T col[6], yc[4];
#pragma unroll
      for (int r = 0; r < 6; ++r) col[r] = *reinterpret_cast<const T*>(m + (r * 6 + cc) * plane);
      f4_at(col, yc);
#pragma unroll
      for (int r = 0; r < 4; ++r) s4[r][cc] = yc[r];
    }
    T ts = T(0.f), tq = T(0.f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T o[4];
      f4_at(s4[r], o);
      const int yy = (4 * ty + r) * g.dil + sy;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        act_t* p = y + ((size_t)(n * g.H + yy) * g.W + (4 * tx + q) * g.dil + sx) * ld + c;
        T v = o[q] + bv;
        if (e.mode == 1) { ts += v; tq += v * v; }
        if (e.mode == 2) {
          v = v * sc + sh;
          const T vs = v * e.slope;
          v = __builtin_elementwise_max(v, vs);
        }
        wv_st<VEC>(p, v);
      }
    }
    if (e.mode == 1) {
      if (n >= imgs_per_group) { s1 += ts; q1 += tq; } else { s0 += ts; q0 += tq; }
    }
  }
  if (e.mode == 1) {
    float* mine = red + threadIdx.x * 4 * VEC;
    *reinterpret_cast<T*>(mine) = s0; *reinterpret_cast<T*>(mine + VEC) = q0;
    *reinterpret_cast<T*>(mine + 2 * VEC) = s1; *reinterpret_cast<T*>(mine + 3 * VEC) = q1;
    __syncthreads();
    const int per_block = cv < 256 ? cv : 256;               // distinct channel vectors in this block
    const int slots = 256 / per_block;
    const int row = cv > 256 ? blockIdx.x / (cv / 256) : blockIdx.x;
    for (int idx = threadIdx.x; idx < per_block * VEC * 4; idx += 256) {
      const int lane = idx / (4 * VEC), rest = idx % (4 * VEC), k4 = rest / VEC, v = rest % VEC;   // k4: s0 q0 s1 q1
      double acc = 0.0;
      for (int sl = 0; sl < slots; ++sl) acc += (double)red[(lane + sl * per_block) * 4 * VEC + k4 * VEC + v];
      const int gg = k4 >> 1, which = k4 & 1;
      if (gg < e.groups) {
        const int cc = (int)(((long long)blockIdx.x * 256 + lane) % cv) * VEC + v;
        pp_epi_row(e, gg, row, which, Nc)[cc] = acc;
      }
    }
  }
}

#ifndef PP_ACT_16
template <int VEC>
__global__ __launch_bounds__(256) void wino4_dy_kernel(const float* __restrict__ dy, int ld, int O, WinoGeom g,
                                                       float* __restrict__ Wt, float* __restrict__ amax) {
  typedef typename WVec<VEC>::type T;
  float mx = 0.f;
  const int cv = O / VEC;
  const long long total = (long long)g.T * cv;
  const size_t plane = (size_t)g.T * O;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    const int t = (int)(i / cv);
    int n, sy, sx, ty, tx;
    tile_coords(g, t, n, sy, sx, ty, tx);
    T a6[6][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      T d[4], col[6];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        d[r] = *reinterpret_cast<const T*>(
            dy + ((size_t)(n * g.H + (4 * ty + r) * g.dil + sy) * g.W + (4 * tx + s) * g.dil + sx) * ld + c);
      f4_a(d, col);
#pragma unroll
      for (int r = 0; r < 6; ++r) a6[r][s] = col[r];
    }
    float* o = Wt + (size_t)t * O + c;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      T w[6];
      f4_a(a6[r], w);
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        *reinterpret_cast<T*>(o + (r * 6 + s) * plane) = w[s];
        mx = fmaxf(mx, wvec_amax(w[s]));
      }
    }
  }
  wino_block_amax(mx, amax);
}
#endif  // !PP_ACT_16

// ================================================================ pre-split ("ps") transform-domain operands
// The split-fp16 GEMMs used to receive fp32 V / W planes and convert every staged tile to (hi, lo) fp16 -- once per
// N-block, in the one wave a SIMD has (r02: the largest non-MFMA term of a K-step).  The transforms are HBM-bound with
// idle VALU slots, so THEY now write the operands split, in the layout the GEMMs stage by plain (LDS-DMA) copies:
//   row (tile t or output channel n) of K elements = K / 8 OCTETS of 32 bytes  [hi 8 x fp16 | lo 8 x fp16],
//   value ~ (hi + lo / 2048) / s_in,   hi = rne16(v * s_in),  lo = rne16((v * s_in - hi) * 2048)
// -- 4 bytes per element as before, and one 16-byte piece is exactly one MFMA fragment (8 consecutive k of one part).
// s_in is a power of two chosen BEFORE the transform runs (the split needs it): from max |input| (known for gradients:
// the BatchNorm backward that writes dz collects it) times the transform's worst-case amplification, so that no
// transform-domain value can exceed the fp16 range; activations (no maximum at hand, O(1) after BatchNorm) use 2^-4.
#define PS_BOUND_INPUT 100.f     // ||B^T||_inf^2 of F(4x4,3x3): (4 + 5 + 1)^2
#define PS_BOUND_DY 225.f        // ||A||_inf^2: (1 + 2 + 4 + 8)^2
__device__ __forceinline__ void ps_scales(const float* amax, float bound, float& s_in, float& s_out) {
  s_in = 0.0625f; s_out = 16.f;
  if (amax) {
    const float m = *amax * bound;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      (void)frexpf(m, &e);                       // m = f * 2^e, f in [0.5, 1)  ->  m * 2^(15 - e) in [2^14, 2^15)
      e = e < -100 ? -100 : (e > 100 ? 100 : e);
      s_in = ldexpf(1.f, 15 - e);
      s_out = ldexpf(1.f, e - 15);
    }
  }
}
// One thread holds 4 consecutive channels c..c+3 of a transform-domain value; lanes 2m / 2m+1 hold the two halves of an
// octet.  The even lane collects both hi quads, the odd lane both lo quads: every lane stores ONE 16-byte piece and a
// wave writes 1 KB of contiguous octets per plane.
// CLAMP: saturate instead of overflowing to inf.  Only operands with the FIXED scale need it (activations: 2^-4, i.e.
// |V| < 1e6, which eval-mode BatchNorm with frozen statistics does not guarantee -- ADVICE r03); operands scaled by their own
// maximum cannot overflow.  The tiled input transform clamps its INPUT once per element instead (36 v_med3 per tile and
// channel quad measured +0.12 ms per step in the strided kernel: the F(4x4) transforms are not VALU-idle).
template <bool CLAMP = false>
__device__ __forceinline__ void ps_store(char* __restrict__ dst /* octet base + (odd ? 16 : 0) */, f32x4 v, float s_in, bool odd) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  f32x4 sv = v * s_in;
  if (CLAMP) {
#pragma unroll
    for (int e = 0; e < 4; ++e) sv[e] = __builtin_amdgcn_fmed3f(sv[e], -65504.f, 65504.f);
  }
  const f16x4 hi = __builtin_convertvector(sv, f16x4);
  const f16x4 lo = __builtin_convertvector((sv - __builtin_convertvector(hi, f32x4)) * F16_LO_SCALE, f16x4);
  const u32x2 H = __builtin_bit_cast(u32x2, hi), L = __builtin_bit_cast(u32x2, lo);
  const u32x2 send = odd ? H : L;
  u32x2 recv;
  recv[0] = (unsigned)__shfl_xor((int)send[0], 1, 64);
  recv[1] = (unsigned)__shfl_xor((int)send[1], 1, 64);
  const u32x4 out = odd ? u32x4{recv[0], recv[1], L[0], L[1]} : u32x4{H[0], H[1], recv[0], recv[1]};
  *reinterpret_cast<u32x4*>(dst) = out;
}

// input transform V = B^T d B straight into octets: one thread per (tile, channel quad); C % 8 == 0
// CLAMP: fixed-scale operands (activations, in_amax == null) saturate instead of overflowing, see ps_store
// (Round 4 built a form that applied the producing layer's BatchNorm + LeakyReLU while loading, and an LDS-tiled form that stages
// every pixel once: both measured slower on the benchmark step -- the F(4x4) transform is not VALU-idle (DESIGN.md section 3,
// profiles/r04_experiments/) -- and were removed in round 5.)
template <bool CLAMP>
__global__ __launch_bounds__(256) void wino4_input_ps_kernel(const act_t* __restrict__ x, int ld, int C, WinoGeom g,
                                                             char* __restrict__ V, const float* __restrict__ in_amax) {
  float s_in, s_out;
  ps_scales(in_amax, PS_BOUND_INPUT, s_in, s_out);
  const int cv = C >> 2;
  const long long total = (long long)g.T * cv;
  const size_t plane = (size_t)g.T * C * 4;            // bytes per plane
  const bool odd = threadIdx.x & 1;
  // neighbouring tiles read overlapping 6x6 patches: every XCD gets one contiguous band of blocks, so the shared rows /
  // columns are fetched into one L2 (blocks b and b + 8 share an XCD)
  const int G = (int)gridDim.x;
  const int bx = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  for (long long i = (long long)bx * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * 4;
    const int t = (int)(i / cv);
    int n, sy, sx, ty, tx;
    tile_coords(g, t, n, sy, sx, ty, tx);
    f32x4 tt[6][6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      f32x4 d[6];
      const int xs = 4 * tx - 1 + s;
      const int xx = xs * g.dil + sx;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const int ys = 4 * ty - 1 + r;
        const bool ok = (unsigned)ys < (unsigned)g.Hs && (unsigned)xs < (unsigned)g.Ws;
#ifdef PP_ACT_16
        // unconditional load from a clamped address, then a select: with a conditional load hipcc folds the fp16 -> fp32
        // conversion into the conditional block -- load, s_waitcnt vmcnt(0), convert, 36 times in a row (148 us against the
        // 102 us of the fp32 build on the benchmark's layers; 88 us in this form, r04 kernel trace)
        const size_t off = ok ? ((size_t)(n * g.H + ys * g.dil + sy) * g.W + xx) * ld + c : (size_t)c;
        const act_raw4 raw = act_ld4_raw(x + off);
        d[r] = act_cvt4(ok ? raw : act_raw4_zero());
#else
        d[r] = ok ? act_ld4(x + ((size_t)(n * g.H + ys * g.dil + sy) * g.W + xx) * ld + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#endif
      }
      f32x4 col[6];
      f4_bt(d, col);
#pragma unroll
      for (int r = 0; r < 6; ++r) tt[r][s] = col[r];
    }
    char* o = V + ((size_t)t * C + (c & ~7)) * 4 + (odd ? 16 : 0);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      f32x4 v[6];
      f4_bt(tt[r], v);
#pragma unroll
      for (int s = 0; s < 6; ++s) ps_store<CLAMP>(o + (r * 6 + s) * plane, v[s], s_in, odd);
    }
  }
}

static inline int wino_blocks(long long total);
static void launch_wino4_input_ps(const act_t* in, int ld_in, int C, const WinoGeom& g, char* V, const float* in_amax, hipStream_t s) {
  const dim3 grid(wino_blocks((long long)g.T * (C / 4)));
  if (!in_amax) hipLaunchKernelGGL(wino4_input_ps_kernel<true>, grid, dim3(256), 0, s, in, ld_in, C, g, V, in_amax);
  else hipLaunchKernelGGL(wino4_input_ps_kernel<false>, grid, dim3(256), 0, s, in, ld_in, C, g, V, in_amax);
}

// gradient-side transform W = A dY A^T straight into octets
__global__ __launch_bounds__(256) void wino4_dy_ps_kernel(const act_t* __restrict__ dy, int ld, int O, WinoGeom g,
                                                          char* __restrict__ Wt, const float* __restrict__ in_amax) {
  float s_in, s_out;
  ps_scales(in_amax, PS_BOUND_DY, s_in, s_out);
  const int cv = O >> 2;
  const long long total = (long long)g.T * cv;
  const size_t plane = (size_t)g.T * O * 4;
  const bool odd = threadIdx.x & 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * 4;
    const int t = (int)(i / cv);
    int n, sy, sx, ty, tx;
    tile_coords(g, t, n, sy, sx, ty, tx);
    f32x4 a6[6][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 d[4], col[6];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        d[r] = act_ld4(dy + ((size_t)(n * g.H + (4 * ty + r) * g.dil + sy) * g.W + (4 * tx + s) * g.dil + sx) * ld + c);
      f4_a(d, col);
#pragma unroll
      for (int r = 0; r < 6; ++r) a6[r][s] = col[r];
    }
    char* o = Wt + ((size_t)t * O + (c & ~7)) * 4 + (odd ? 16 : 0);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      f32x4 w[6];
      f4_a(a6[r], w);
#pragma unroll
      for (int s = 0; s < 6; ++s) ps_store(o + (r * 6 + s) * plane, w[s], s_in, odd);
    }
  }
}

// max |x| of a strided NHWC tensor into *amax (zeroed by the caller): only for callers that do not bring the maximum of
// a gradient tensor along (the training engine does: pp_bn_lrelu_bwd_amax / pp_bn_lrelu_bwd_eval)
__global__ __launch_bounds__(256) void wino_amax_kernel(const act_t* __restrict__ x, int ld, int C, long long P, float* __restrict__ amax) {
  const int cv = C >> 2;
  const long long total = P * cv;
  float mx = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = act_ld4(x + (size_t)(i / cv) * ld + (i % cv) * 4);
    mx = fmaxf(mx, wvec_amax(v));
  }
  wino_block_amax(mx, amax);
}

// vector width of the F(4x4) transform kernels for a tensor (pointer, leading dimension, channels)
static inline int wino4_vec(const void* p, int ld, int C) {
  // widest vector allowed; measured on the full step (r01): 1 -> 8.9 ms, 2 -> 7.1 ms, 4 -> 7.2 ms of transforms per step
  constexpr int forced = PP_ACT_BYTES == 2 ? 4 : 2;                           // 8 bytes per lane
  int v = (C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)p & (4 * PP_ACT_BYTES - 1)) == 0) ? 4
          : (C % 2 == 0 && ld % 2 == 0 && ((uintptr_t)p & (2 * PP_ACT_BYTES - 1)) == 0) ? 2 : 1;
  if (forced && forced < v) v = forced;
  return v;
}
#define WINO4_LAUNCH(kern, vec, total, s, ...)                                                                      \
  do {                                                                                                              \
    if ((vec) == 4) hipLaunchKernelGGL(kern<4>, dim3(wino_blocks((total) / 4)), dim3(256), 0, s, __VA_ARGS__);      \
    else if ((vec) == 2) hipLaunchKernelGGL(kern<2>, dim3(wino_blocks((total) / 2)), dim3(256), 0, s, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern<1>, dim3(wino_blocks((total))), dim3(256), 0, s, __VA_ARGS__);                     \
  } while (0)

__global__ void wino4_weight_kernel(const float* __restrict__ w, int O, int I, float* __restrict__ Uf,
                                    float* __restrict__ Ub) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= O * I) return;
  const int o = idx / I, c = idx % I;
  const size_t plane = (size_t)O * I;
  for (int pass = 0; pass < 2; ++pass) {
    float* U = pass == 0 ? Uf : Ub;
    if (!U) continue;
    float t6[6][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      float gcol[3], u[6];
#pragma unroll
      for (int r = 0; r < 3; ++r) gcol[r] = pass == 0 ? w[(size_t)idx * 9 + r * 3 + s] : w[(size_t)idx * 9 + (2 - r) * 3 + (2 - s)];
      f4_g(gcol, u);
#pragma unroll
      for (int r = 0; r < 6; ++r) t6[r][s] = u[r];
    }
    const size_t at = pass == 0 ? (size_t)o * I + c : (size_t)c * O + o;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float u[6];
      f4_g(t6[r], u);
#pragma unroll
      for (int s = 0; s < 6; ++s) U[(r * 6 + s) * plane + at] = u[s];
    }
  }
}

// F(4x4) weight transform straight into the pre-split octet layout of the GEMM's B operand (see "pre-split operands"
// above): one thread transforms the 3x3 kernels of FOUR consecutive K elements (input channels for Uf [b][o][c], output
// channels for Ub [b][c][o]) and stores the hi quad and the lo quad of its half octet per plane -- no fp32 U in HBM.
// pass 0: Uf (quad along c), pass 1: Ub from the flipped kernel (quad along o); grid.y selects the pass.  K % 8 == 0.
__device__ __forceinline__ void wino4_weight_ps_body(const float* __restrict__ w, int O, int I, char* __restrict__ Uf,
                                                     char* __restrict__ Ub, int pass, int idx) {
  char* U = pass == 0 ? Uf : Ub;
  if (!U) return;
  const int nq = pass == 0 ? O * (I / 4) : I * (O / 4);
  if (idx >= nq) return;
  // pass 0: row = o, quad over c;  pass 1: row = c, quad over o
  const int per_row = pass == 0 ? I / 4 : O / 4;
  const int row = idx / per_row, q4 = (idx % per_row) * 4;
  float u[4][36];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int o = pass == 0 ? row : q4 + e, c = pass == 0 ? q4 + e : row;
    const float* g = w + ((size_t)o * I + c) * 9;
    float t6[6][3];
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      float gcol[3], col[6];
#pragma unroll
      for (int r = 0; r < 3; ++r) gcol[r] = pass == 0 ? g[r * 3 + sx] : g[(2 - r) * 3 + (2 - sx)];
      f4_g(gcol, col);
#pragma unroll
      for (int r = 0; r < 6; ++r) t6[r][sx] = col[r];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float v[6];
      f4_g(t6[r], v);
#pragma unroll
      for (int sx = 0; sx < 6; ++sx) u[e][r * 6 + sx] = v[sx];
    }
  }
  const int K = pass == 0 ? I : O;
  const size_t plane = (size_t)O * I * 4;                     // bytes per plane
  const size_t at = ((size_t)row * K + (q4 & ~7)) * 4 + (q4 & 4) * 2;     // octet + position of this quad inside its hi part
#pragma unroll
  for (int b = 0; b < 36; ++b) {
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const _Float16 h = (_Float16)u[e][b];
      hi[e] = h;
      lo[e] = (_Float16)((u[e][b] - (float)h) * F16_LO_SCALE);
    }
    char* d = U + (size_t)b * plane + at;
    *reinterpret_cast<f16x4*>(d) = hi;
    *reinterpret_cast<f16x4*>(d + 16) = lo;
  }
}

__global__ __launch_bounds__(256) void wino4_weight_ps_kernel(const float* __restrict__ w, int O, int I,
                                                              char* __restrict__ Uf, char* __restrict__ Ub) {
  wino4_weight_ps_body(w, O, I, Uf, Ub, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x);
}

// every Winograd layer's weight transform in ONE launch (round 5; see pack_weights_f16x3_batch_kernel)
#define WPACK_BATCH_MAX 24
struct WPackItem { const float* w; char* Uf; char* Ub; int O, I, blk0; };
struct WPackBatch { WPackItem it[WPACK_BATCH_MAX]; int n; };
__global__ __launch_bounds__(256) void wino4_weight_ps_batch_kernel(WPackBatch b) {
  int k = 0;
  for (int i = 1; i < b.n; ++i)
    if ((int)blockIdx.x >= b.it[i].blk0) k = i;
  const WPackItem it = b.it[k];
  wino4_weight_ps_body(it.w, it.O, it.I, it.Uf, it.Ub, blockIdx.y, (int)(blockIdx.x - it.blk0) * blockDim.x + threadIdx.x);
}

// ---------------------------------------------------------------- weight transforms  U = G g G^T
// Uf[b][o][c] from w[o][c][3][3];  Ub[b][c][o] from the flipped kernel (data gradient)
__global__ void wino_weight_kernel(const float* __restrict__ w, int O, int I, float* __restrict__ Uf,
                                   float* __restrict__ Ub) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= O * I) return;
  const int o = idx / I, c = idx % I;
  float gk[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) gk[r][s] = w[(size_t)idx * 9 + r * 3 + s];
  for (int pass = 0; pass < 2; ++pass) {
    float* U = pass == 0 ? Uf : Ub;
    if (!U) continue;
    float k[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) k[r][s] = pass == 0 ? gk[r][s] : gk[2 - r][2 - s];
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      t[0][s] = k[0][s];
      t[1][s] = 0.5f * (k[0][s] + k[1][s] + k[2][s]);
      t[2][s] = 0.5f * (k[0][s] - k[1][s] + k[2][s]);
      t[3][s] = k[2][s];
    }
    const size_t plane = (size_t)O * I;
    const size_t at = pass == 0 ? (size_t)o * I + c : (size_t)c * O + o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      U[(r * 4 + 0) * plane + at] = t[r][0];
      U[(r * 4 + 1) * plane + at] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
      U[(r * 4 + 2) * plane + at] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
      U[(r * 4 + 3) * plane + at] = t[r][2];
    }
  }
}

// ---------------------------------------------------------------- batched GEMM  M_b[t][n] = sum_c V_b[t][c] U_b[n][c]
// Same machinery as conv3x3_igemm_kernel (pp_conv.hip): 128x128 tile, 2x2 32x32x2 MFMA tiles per wave, K-step 32,
// buffer loads with hardware bounds check, two LDS buffers, loads/stores interleaved in the MFMA shadow.
struct GemmArgs {
  const float* A; const float* B; float* C;
  int M, N, K;                    // A [batch][M][K], B [batch][N][K], C [batch][M][N]
  int m_tiles, n_tiles;
  unsigned a_bytes, b_bytes;      // per batch plane
  int nb;                         // planes (16 or 36)
};

template <int TM, int TN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void wino_gemm_kernel(GemmArgs a) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
  constexpr int RPP = NT / 8;
  constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * BM * WLD;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;
  const int q = tid & 7, r0 = tid >> 3;
  // block -> (batch, m tile, n tile): the n-tiles of one m-tile share an XCD (they re-read the same V rows)
  const int per_batch = a.m_tiles * a.n_tiles;
  int b = blockIdx.x;
  const int total = per_batch * a.nb;
  if ((total & 7) == 0) b = (b & 7) * (total >> 3) + (b >> 3);
  const int batch = b / per_batch;
  const int rem = b - batch * per_batch;
  const int mt = rem / a.n_tiles, nt = rem % a.n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const float* Ab = a.A + (size_t)batch * a.M * a.K;
  const float* Bb = a.B + (size_t)batch * a.N * a.K;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, a.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, a.b_bytes, 0x00020000);
  f32x4 ra[A_PASSES], rb[B_PASSES];
  const int n_it = (a.K + WBK - 1) / WBK;
  auto load_tile = [&](int it) {
    const int c = it * WBK + q * 4;
    const int cok = (int)(c < a.K);
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int m = m0 + r0 + i * RPP;
      const unsigned off = (cok & (int)(m < a.M)) ? (unsigned)(m * a.K + c) * 4u : 0xffffffffu;
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const int n = n0 + r0 + i * RPP;
      const unsigned off = (cok & (int)(n < a.N)) ? (unsigned)(n * a.K + c) * 4u : 0xffffffffu;
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, off, 0, 0));
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      *reinterpret_cast<f32x4*>(As + buf * BM * WLD + (r0 + i * RPP) * WLD + q * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i)
      *reinterpret_cast<f32x4*>(Bs + buf * BN * WLD + (r0 + i * RPP) * WLD + q * 4) = rb[i];
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int it = 0; it < n_it; ++it) {
    const int buf = it & 1;
    const bool more = it + 1 < n_it;
    const float* Ap = As + buf * BM * WLD + (wm * TM * 32 + lr) * WLD + lh * 4;
    const float* Bp = Bs + buf * BN * WLD + (wn * TN * 32 + lr) * WLD + lh * 4;
    float4 af[2][TM], bf[2][TN];
    auto read_frags = [&](int kk, int slot) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[slot][i] = *reinterpret_cast<const float4*>(Ap + i * 32 * WLD + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const float4*>(Bp + j * 32 * WLD + kk * 8);
    };
    auto mfma_block = [&](int slot) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][i].x, bf[slot][j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][i].y, bf[slot][j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][i].z, bf[slot][j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][i].w, bf[slot][j].w, acc[i][j], 0, 0, 0);
        }
    };
    read_frags(0, 0);
    read_frags(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) load_tile(it + 1);
    read_frags(2, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(1);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(3, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) store_tile(buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(1);
    __syncthreads();
  }
  float* Cb = a.C + (size_t)batch * a.M * a.N;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + lr;
    if (n >= a.N) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < a.M) Cb[(size_t)m * a.N + n] = acc[i][j][r];
      }
  }
}

template <int TM, int TN, int WAVES_M, int WAVES_N>
static int launch_gemm(GemmArgs a, hipStream_t s) {
  constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
  a.m_tiles = pp_cdiv(a.M, BM);
  a.n_tiles = pp_cdiv(a.N, BN);
  const size_t lds = (size_t)2 * (BM + BN) * WLD * sizeof(float);
  auto kern = wino_gemm_kernel<TM, TN, WAVES_M, WAVES_N>;
  {   // once per (kernel, device): pp_max_lds
    pp_max_lds(reinterpret_cast<const void*>(kern), (int)lds);
  }
  hipLaunchKernelGGL(kern, dim3(a.nb * a.m_tiles * a.n_tiles), dim3(WAVES_M * WAVES_N * 64), lds, s, a);
  return pp_launch_status("wino_gemm");
}

// ---------------------------------------------------------------- the same batched GEMM on the fp16 MFMA ("f16x3")
// M_b[t][n] = sum_c V_b[t][c] U_b[n][c] with BOTH operands pre-split into octets (see "pre-split operands"): a product
// is ah*bh + 2^-11 (ah*bl + al*bh), three v_mfma_f32_32x32x16_f16 per 16 k with fp32 accumulation in two accumulator
// sets (main / cross).  Nothing passes through VGPRs on its way to LDS: every wave issues `buffer_load ... lds`
// (LDS-DMA, 1 KB = 8 tile rows x 128 B per instruction) two K-steps ahead into a ring of three stages, waits with a
// COUNTED vmcnt for its own pieces of the current stage and meets the other waves at ONE raw s_barrier per K-step
// (no __syncthreads: its fence would drain the DMAs in flight).
// LDS image of a stage: [BM rows of A | BN rows of B], 128 B per row = 8 pieces of 16 B (piece p = 2 * octet + part,
// part 0 = hi, 1 = lo), stored at piece slot p ^ ((row >> 1) & 7).  LDS-DMA writes lane l at base + 16 l, so the
// swizzle is applied to the SOURCE address of each lane; the fragment reads apply it again.  Without it the 16 lanes of
// a ds_read_b128 group (rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31}, same piece) would fall on two 16-byte bank
// slots; with it they cover all 16.
struct GemmPsArgs {
  const char* A; const char* B; float* C;      // A [nb][M][K] octets, B [nb][N][K] octets, C [nb][M][N] fp32
  int M, N, K;
  int m_tiles, n_tiles, nb;
  unsigned a_bytes, b_bytes;                   // per plane
  const float* a_amax; float a_bound;          // the scale the producer of A applied (ps_scales of the same arguments)
};

// (r03 A/B, timing only: the same flops issued as pairs of v_mfma_f32_16x16x32_f16 ran the 512 / 1024-channel GEMMs 5-6 %
// faster -- the chip holds a higher clock on that shape -- not enough to pay for a second fragment layout)
__device__ __forceinline__ f32x16 ps_mfma(f16x8 a, f16x8 b, f32x16 c, int) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int TM, int TN, int WAVES_M, int WAVES_N, bool X1>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) __attribute__((amdgpu_waves_per_eu(WAVES_M* WAVES_N / 4)))
void wino_gemm_ps_kernel(GemmPsArgs a) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
  constexpr int STAGE = (BM + BN) * 128;                 // bytes
  constexpr int FA = BM / 8 / NW, FB = BN / 8 / NW;      // LDS-DMA instructions per wave and stage
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split evenly over the waves");
  extern __shared__ __attribute__((aligned(1024))) char smem_ps[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;
  const int per_batch = a.m_tiles * a.n_tiles;
  int b = blockIdx.x;
  const int total = per_batch * a.nb;
  if ((total & 7) == 0) b = (b & 7) * (total >> 3) + (b >> 3);      // the n-tiles of one m-tile share an XCD (same V rows)
  const int batch = b / per_batch;
  const int rem = b - batch * per_batch;
  const int mt = rem / a.n_tiles, nt = rem % a.n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  float s_in, s_out;
  ps_scales(a.a_amax, a.a_bound, s_in, s_out);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (size_t)batch * a.a_bytes), 0, a.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(a.B + (size_t)batch * a.b_bytes), 0, a.b_bytes, 0x00020000);
  // LDS-DMA geometry of this lane: instruction f of this wave covers tile rows (f * NW + wv) * 8 .. + 7; the lane brings
  // piece slot (lane & 7) of row (lane >> 3), i.e. the source piece (lane & 7) ^ swizzle(row)
  const int fr = lane >> 3, fq = lane & 7;
  const unsigned kbytes = (unsigned)a.K * 4u;
  unsigned offA[FA], offB[FB], pcA[FA], pcB[FB];
#pragma unroll
  for (int f = 0; f < FA; ++f) {
    const int row = (f * NW + wv) * 8 + fr;
    pcA[f] = (unsigned)((fq ^ ((row >> 1) & 7)) * 16);
    offA[f] = (m0 + row < a.M) ? (unsigned)(m0 + row) * kbytes + pcA[f] : 0xffffffffu;
  }
#pragma unroll
  for (int f = 0; f < FB; ++f) {
    const int row = (f * NW + wv) * 8 + fr;
    pcB[f] = (unsigned)((fq ^ ((row >> 1) & 7)) * 16);          // BM is a multiple of 16: the LDS row BM + row swizzles like row
    offB[f] = (n0 + row < a.N) ? (unsigned)(n0 + row) * kbytes + pcB[f] : 0xffffffffu;
  }
  // LDS-DMA instruction f (0 .. FA + FB - 1) of this wave for K-step `it` into ring stage `stage`
  auto fill_one = [&](int f, int it, int stage) {
    char* base = smem_ps + stage * STAGE + wv * 1024;
    const unsigned koff = (unsigned)it * 128u;
    if (f < FA) {
      const unsigned o = (offA[f < FA ? f : 0] != 0xffffffffu && koff + pcA[f < FA ? f : 0] < kbytes) ? offA[f < FA ? f : 0] + koff : 0xffffffffu;   // beyond K / M: zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr)(base + f * NW * 1024), 16, o, 0, 0, 0);
    } else {
      const int g = f - FA < FB ? f - FA : 0;
      const unsigned o = (offB[g] != 0xffffffffu && koff + pcB[g] < kbytes) ? offB[g] + koff : 0xffffffffu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr)(base + BM * 128 + g * NW * 1024), 16, o, 0, 0, 0);
    }
  };
  auto fill = [&](int it, int stage) {
#pragma unroll
    for (int f = 0; f < FA + FB; ++f) fill_one(f, it, stage);
  };
  f32x16 accm[TM][TN], accc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accc[i][j][r] = 0.f; }
  // fragment addresses: row (.. + lr) of the wave's A / B rows, piece (4 kb + 2 lh + part) ^ ((lr >> 1) & 7)
  const int sw = (lr >> 1) & 7;
  int qo[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int part = 0; part < 2; ++part) qo[kb][part] = ((4 * kb + 2 * lh + part) ^ sw) * 16;
  const int a_row = (wm * TM * 32 + lr) * 128, b_row = BM * 128 + (wn * TN * 32 + lr) * 128;
  const int n_it = (a.K + 31) / 32;
  // Software pipeline (one K-step = two 16-k halves X, Y of twelve MFMAs each):
  //   A  MFMA(X)  of stage it
  //   B  wait: this wave's pieces of stage it + 1 have landed, its fragment reads of stage it have returned; s_barrier
  //   C  fragment reads X <- stage it + 1
  //   D  MFMA(Y)  of stage it, the LDS-DMA instructions of K-step it + 3 between them: after B every wave is done with
  //      ring slot it % 3, so that slot is refilled at once -- three K-steps of prefetch in a three-slot ring
  //   E  fragment reads Y <- stage it + 1
  // The reads of a half are issued while the twelve MFMAs of the other half are queued, the barrier is crossed with
  // MFMAs in flight, and tiles past K read zeros through the buffer descriptor (uniform vmcnt counting).
  f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
  auto read_half = [&](int kb, int stage) {
    const char* Ap = smem_ps + stage * STAGE + a_row;
    const char* Bp = smem_ps + stage * STAGE + b_row;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[kb][i] = *reinterpret_cast<const f16x8*>(Ap + i * 32 * 128 + qo[kb][0]);
      if (!X1) al[kb][i] = *reinterpret_cast<const f16x8*>(Ap + i * 32 * 128 + qo[kb][1]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[kb][j] = *reinterpret_cast<const f16x8*>(Bp + j * 32 * 128 + qo[kb][0]);
      if (!X1) bl[kb][j] = *reinterpret_cast<const f16x8*>(Bp + j * 32 * 128 + qo[kb][1]);
    }
  };
  auto wait_landed = [&]() {             // all but the youngest FA + FB LDS-DMAs of this wave are done; no LDS read pending
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FA + FB) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  static_assert(FA + FB <= 16, "the counted wait assumes a handful of DMAs per wave and stage");
  fill(0, 0);
  fill(1, 1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FA + FB) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  fill(2, 2);
  read_half(0, 0);
  read_half(1, 0);
  int st_next = 1, st_free = 0;          // ring slots of stage it + 1 and of stage it (free after B)
  for (int it = 0; it < n_it; ++it) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        accm[i][j] = ps_mfma(ah[0][i], bh[0][j], accm[i][j], 0);
        if (!X1) {
          accc[i][j] = ps_mfma(ah[0][i], bl[0][j], accc[i][j], 0);
          accc[i][j] = ps_mfma(al[0][i], bh[0][j], accc[i][j], 0);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    wait_landed();
    read_half(0, st_next);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        accm[i][j] = ps_mfma(ah[1][i], bh[1][j], accm[i][j], 1);
        if (!X1) {
          accc[i][j] = ps_mfma(ah[1][i], bl[1][j], accc[i][j], 1);
          accc[i][j] = ps_mfma(al[1][i], bh[1][j], accc[i][j], 1);
        }
        constexpr int GROUPS = TM * TN;
        const int gidx = i * TN + j;
#pragma unroll
        for (int f = 0; f < FA + FB; ++f)
          if (f * GROUPS / (FA + FB) == gidx) fill_one(f, it + 3, st_free);
        __builtin_amdgcn_sched_barrier(0);
      }
    read_half(1, st_next);
    st_free = st_next;
    st_next = st_next == 2 ? 0 : st_next + 1;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // ghost fills / reads: no LDS-DMA may outlive the workgroup
  float* Cb = a.C + (size_t)batch * a.M * a.N;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + lr;
    if (n >= a.N) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < a.M) Cb[(size_t)m * a.N + n] = (X1 ? accm[i][j][r] : accm[i][j][r] + accc[i][j][r] * (1.f / F16_LO_SCALE)) * s_out;
      }
  }
}

// ---- Round 5: the same GEMM, PERSISTENT over its tiles with ONE software pipeline across tile boundaries ----
// The kernel above runs one 128 x 256 tile per block: fill two ring stages, wait, compute K / 32 steps, drain, store 128 KB, exit;
// the next block of that CU starts cold.  At K = 512 (16 K-steps of ~0.37 us) the cold start, the drain and the block turnaround
// are a third of a tile's life (SQ MFMA-busy 0.48).  Here a block walks over tiles b, b + grid, ... and the K-steps of ALL its
// tiles form one flattened iteration space: the LDS-DMA cursor runs three K-steps ahead of the MFMA cursor and simply crosses
// into the next tile (new row offsets / plane descriptors for the fills, computed once per tile), so the ring never drains; when
// the MFMA cursor finishes a tile it stores the accumulators and clears them while the next tile's first stages are already in
// LDS.  vmcnt bookkeeping: the 64 accumulator stores of a tile enter the same in-order counter as the DMAs.  The wait in front of
// the barrier must guarantee "the DMAs of stage it + 1 have landed"; the ops issued after those DMAs are the 6 DMAs of the
// following stage -- plus, for the two K-steps after a tile boundary, the 64 stores.  So those two waits use vmcnt(63) (>= 7 of
// the 70 younger ops may be outstanding, which still covers every older DMA) and do not stall on store completion; from the
// third K-step on the wait is vmcnt(6) again, by which time the stores have long drained.  (Needs >= 4 K-steps per tile, full
// tiles -- every lane issues every store -- and is launched for such shapes only.)
template <int TM, int TN, int WAVES_M, int WAVES_N, bool X1>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) __attribute__((amdgpu_waves_per_eu(WAVES_M* WAVES_N / 4)))
void wino_gemm_psp_kernel(GemmPsArgs a) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
  constexpr int STAGE = (BM + BN) * 128;                 // bytes
  constexpr int FA = BM / 8 / NW, FB = BN / 8 / NW;      // LDS-DMA instructions per wave and stage
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split evenly over the waves");
  static_assert(FA + FB <= 16, "the counted wait below assumes a handful of DMAs per wave and stage");
  constexpr int NST = TM * TN * 16;                      // accumulator stores per lane and tile (all unconditional)
  constexpr int WIN = NST + FA + FB < 63 ? NST + FA + FB : 63;      // vmcnt window of the two K-steps after a tile boundary
  extern __shared__ __attribute__((aligned(1024))) char smem_ps[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;
  const int per_batch = a.m_tiles * a.n_tiles;
  const int total = per_batch * a.nb;
  const int G = (int)gridDim.x;
  const int n_my = ((int)blockIdx.x < total) ? (total - (int)blockIdx.x + G - 1) / G : 0;
  if (n_my == 0) return;
  const bool xcd = (total & 7) == 0 && (G & 7) == 0;     // the n-tiles of one m-tile share an XCD (same V rows)
  auto tile_of = [&](int seq, int& batch, int& m0, int& n0) {
    const int it = (int)blockIdx.x + seq * G;
    const int b = xcd ? (it & 7) * (total >> 3) + (it >> 3) : it;
    batch = b / per_batch;
    const int rem = b - batch * per_batch;
    m0 = (rem / a.n_tiles) * BM;
    n0 = (rem % a.n_tiles) * BN;
  };
  float s_in, s_out;
  ps_scales(a.a_amax, a.a_bound, s_in, s_out);
  const int fr = lane >> 3, fq = lane & 7;
  const unsigned kbytes = (unsigned)a.K * 4u;
  const int n_it = (a.K + 31) / 32;
  // ---- fill cursor: tile sequence number f_seq, K-step f_it; per-lane source offsets of that tile
  unsigned pcA[FA], pcB[FB], offA[FA], offB[FB];
#pragma unroll
  for (int f = 0; f < FA; ++f) pcA[f] = (unsigned)((fq ^ ((((f * NW + wv) * 8 + fr) >> 1) & 7)) * 16);
#pragma unroll
  for (int f = 0; f < FB; ++f) pcB[f] = (unsigned)((fq ^ ((((f * NW + wv) * 8 + fr) >> 1) & 7)) * 16);
  int f_seq = 0, f_it = 0, f_batch = 0;
  auto fill_tile = [&](int seq) {                        // offsets of tile `seq` (beyond the block's last tile: zeros)
    int m0 = 0, n0 = 0;
    const bool live = seq < n_my;
    if (live) tile_of(seq, f_batch, m0, n0);
#pragma unroll
    for (int f = 0; f < FA; ++f) {
      const int row = (f * NW + wv) * 8 + fr;
      offA[f] = (live && m0 + row < a.M) ? (unsigned)(m0 + row) * kbytes + pcA[f] : 0xffffffffu;
    }
#pragma unroll
    for (int f = 0; f < FB; ++f) {
      const int row = (f * NW + wv) * 8 + fr;
      offB[f] = (live && n0 + row < a.N) ? (unsigned)(n0 + row) * kbytes + pcB[f] : 0xffffffffu;
    }
  };
  fill_tile(0);
  // LDS-DMA instruction f of this wave for the fill cursor's K-step into ring stage `stage`
  auto fill_one = [&](int f, int stage) {
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (size_t)f_batch * a.a_bytes), 0, a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(a.B + (size_t)f_batch * a.b_bytes), 0, a.b_bytes, 0x00020000);
    char* base = smem_ps + stage * STAGE + wv * 1024;
    const unsigned koff = (unsigned)f_it * 128u;
    if (f < FA) {
      const int g = f < FA ? f : 0;
      const unsigned o = (offA[g] != 0xffffffffu && koff + pcA[g] < kbytes) ? offA[g] + koff : 0xffffffffu;   // beyond K / M: zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr)(base + g * NW * 1024), 16, o, 0, 0, 0);
    } else {
      const int g = f - FA < FB ? f - FA : 0;
      const unsigned o = (offB[g] != 0xffffffffu && koff + pcB[g] < kbytes) ? offB[g] + koff : 0xffffffffu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr)(base + BM * 128 + g * NW * 1024), 16, o, 0, 0, 0);
    }
  };
  auto fill_advance = [&]() {                            // after the last DMA of a K-step
    if (++f_it == n_it) { f_it = 0; ++f_seq; fill_tile(f_seq); }
  };
  auto fill = [&](int stage) {
#pragma unroll
    for (int f = 0; f < FA + FB; ++f) fill_one(f, stage);
    fill_advance();
  };
  f32x16 accm[TM][TN], accc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accc[i][j][r] = 0.f; }
  const int sw = (lr >> 1) & 7;
  int qo[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int part = 0; part < 2; ++part) qo[kb][part] = ((4 * kb + 2 * lh + part) ^ sw) * 16;
  const int a_row = (wm * TM * 32 + lr) * 128, b_row = BM * 128 + (wn * TN * 32 + lr) * 128;
  f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
  auto read_half = [&](int kb, int stage) {
    const char* Ap = smem_ps + stage * STAGE + a_row;
    const char* Bp = smem_ps + stage * STAGE + b_row;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[kb][i] = *reinterpret_cast<const f16x8*>(Ap + i * 32 * 128 + qo[kb][0]);
      if (!X1) al[kb][i] = *reinterpret_cast<const f16x8*>(Ap + i * 32 * 128 + qo[kb][1]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[kb][j] = *reinterpret_cast<const f16x8*>(Bp + j * 32 * 128 + qo[kb][0]);
      if (!X1) bl[kb][j] = *reinterpret_cast<const f16x8*>(Bp + j * 32 * 128 + qo[kb][1]);
    }
  };
  int since_store = 2;                                   // K-steps since the last accumulator store (>= 2: the plain counted wait)
  auto wait_landed = [&]() {
    if (since_store < 2) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WIN) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FA + FB) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  fill(0);
  fill(1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FA + FB) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  fill(2);
  read_half(0, 0);
  read_half(1, 0);
  int st_next = 1, st_free = 0;          // ring slots of stage it + 1 and of stage it (free after the barrier)
  int c_seq = 0, c_it = 0;               // MFMA cursor
  const int total_it = n_my * n_it;
  for (int g = 0; g < total_it; ++g) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        accm[i][j] = ps_mfma(ah[0][i], bh[0][j], accm[i][j], 0);
        if (!X1) {
          accc[i][j] = ps_mfma(ah[0][i], bl[0][j], accc[i][j], 0);
          accc[i][j] = ps_mfma(al[0][i], bh[0][j], accc[i][j], 0);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    wait_landed();
    ++since_store;
    read_half(0, st_next);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        accm[i][j] = ps_mfma(ah[1][i], bh[1][j], accm[i][j], 1);
        if (!X1) {
          accc[i][j] = ps_mfma(ah[1][i], bl[1][j], accc[i][j], 1);
          accc[i][j] = ps_mfma(al[1][i], bh[1][j], accc[i][j], 1);
        }
        constexpr int GROUPS = TM * TN;
        const int gidx = i * TN + j;
#pragma unroll
        for (int f = 0; f < FA + FB; ++f)
          if (f * GROUPS / (FA + FB) == gidx) fill_one(f, st_free);
        __builtin_amdgcn_sched_barrier(0);
      }
    fill_advance();
    read_half(1, st_next);
    st_free = st_next;
    st_next = st_next == 2 ? 0 : st_next + 1;
    if (++c_it == n_it) {                // the tile is complete: store and clear the accumulators (uniform over the block)
      int batch, m0, n0;
      tile_of(c_seq, batch, m0, n0);
      float* Cb = a.C + (size_t)batch * a.M * a.N;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + lr;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float v = (X1 ? accm[i][j][r] : accm[i][j][r] + accc[i][j][r] * (1.f / F16_LO_SCALE)) * s_out;
            // UNCONDITIONAL: the host launches this kernel for full tiles only (M % BM == 0, N % BN == 0), and the vmcnt window
            // above counts on exactly NST stores per lane
            Cb[(size_t)m * a.N + n] = v;
            accm[i][j][r] = 0.f; accc[i][j][r] = 0.f;
          }
      }
      c_it = 0; ++c_seq; since_store = 0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // ghost fills / reads: no LDS-DMA may outlive the workgroup
}

template <int TM, int TN, int WAVES_M, int WAVES_N, bool X1>
static int launch_gemm_ps(GemmPsArgs a, hipStream_t s) {
  constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
  a.m_tiles = pp_cdiv(a.M, BM);
  a.n_tiles = pp_cdiv(a.N, BN);
  const size_t lds = (size_t)3 * (BM + BN) * 128;
  auto kern = wino_gemm_ps_kernel<TM, TN, WAVES_M, WAVES_N, X1>;
  pp_max_lds(reinterpret_cast<const void*>(kern), (int)lds);   // once per (kernel, device)
  static const int persist = getenv("PP_WINO_GEMM_PERSIST") ? atoi(getenv("PP_WINO_GEMM_PERSIST")) : 1;      // A/B knob: 0 = one tile per block (r03)
  const int total = a.nb * a.m_tiles * a.n_tiles;
  // persistent form: one block per CU (the ring takes 144 KB of LDS), full 128 x 256 / 256 x 128 tiles only (its accumulator
  // stores are unconditional per lane; the vmcnt(63) window needs their count) and at least two tiles per block to gain anything
  if (persist && a.M % BM == 0 && a.N % BN == 0 && a.K >= 128 && total >= 2 * 256) {
    auto kp = wino_gemm_psp_kernel<TM, TN, WAVES_M, WAVES_N, X1>;
    pp_max_lds(reinterpret_cast<const void*>(kp), (int)lds);
    hipLaunchKernelGGL(kp, dim3(256), dim3(WAVES_M * WAVES_N * 64), lds, s, a);
    return pp_launch_status("wino_gemm_psp");
  }
  hipLaunchKernelGGL(kern, dim3(total), dim3(WAVES_M * WAVES_N * 64), lds, s, a);
  return pp_launch_status("wino_gemm_ps");
}
template <bool X1>
static int launch_gemm_ps_any(const GemmPsArgs& a, hipStream_t s) {
  if (a.N % 256 == 0) return launch_gemm_ps<2, 2, 2, 4, X1>(a, s);       // 128 x 256
  if (a.N % 128 == 0) return launch_gemm_ps<2, 2, 4, 2, X1>(a, s);       // 256 x 128
  return launch_gemm_ps<2, 1, 4, 2, X1>(a, s);                            // 256 x 64 (the auxiliary bottleneck)
}

static inline int wino_blocks(long long total) {
  int b = pp_cdiv(total, 256);
  return b > 16384 ? 16384 : (b < 1 ? 1 : b);
}

static int wino_check(int C, int N, int B, int H, int W, int dil) {
  PP_CHECK_ARG(dil >= 1 && H % (2 * dil) == 0 && W % (2 * dil) == 0, "winograd: H, W must be multiples of 2*dilation");
  PP_CHECK_ARG(C % 4 == 0 && N % 4 == 0 && C > 0 && N > 0 && B > 0, "winograd: channel counts must be multiples of 4");
  const long long T = (long long)B * H * W / 4;       // upper bound (F(2x2)); F(4x4) has a quarter of the tiles
  PP_CHECK_ARG(T * C < 0x3fffffffLL && T * N < 0x3fffffffLL && (long long)N * C < 0x3fffffffLL,
               "winograd: plane exceeds the 4 GiB buffer-descriptor range");
  return 0;
}

#ifndef PP_ACT_16     // shape queries and weight packing do not depend on the activation type: one copy
extern "C" size_t pp_conv3x3_wino_workspace(int Cin, int Cout, int B, int H, int W, int dil) {
  const WinoGeom g = wino_geom(B, H, W, dil);
  return (size_t)g.nb * g.T * ((size_t)Cin + Cout) * sizeof(float) + 256;        // V + M (fwd / dgrad)
}

// elements of the transformed-input buffer a forward call can leave behind for the weight gradient
extern "C" size_t pp_conv3x3_wino_vkeep_elems(int Cin, int B, int H, int W, int dil) {
  const WinoGeom g = wino_geom(B, H, W, dil);
  return (size_t)g.nb * g.T * Cin + 4;          // + a tail slot: max |V| (operand scale of the split-fp16 GEMMs)
}

// tile = 2 -> Uf/Ub [16][..], tile = 4 -> [36][..]  (pp_conv3x3_wino_tile tells which one a layer shape uses)
extern "C" int pp_conv3x3_wino_tile(int H, int W, int dil) { return wino_tile(H, W, dil); }

extern "C" int pp_wino_pack_weights(const float* w_oihw, int O, int I, int tile, float* Uf, float* Ub, void* stream) {
  PP_CHECK_ARG(w_oihw && (Uf || Ub) && (tile == 2 || tile == 4), "wino_pack_weights: bad arguments");
  if (tile == 2)
    hipLaunchKernelGGL(wino_weight_kernel, dim3(pp_cdiv((long long)O * I, 256)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, O, I, Uf, Ub);
  else
    hipLaunchKernelGGL(wino4_weight_kernel, dim3(pp_cdiv((long long)O * I, 256)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, O, I, Uf, Ub);
  return pp_launch_status("wino_pack_weights");
}
#endif  // !PP_ACT_16

// max |x| of a tensor the caller brought no maximum for, into a scratch float
static int wino_own_amax(const act_t* x, int ld, int C, long long P, float* slot, hipStream_t s) {
  if (hipMemsetAsync(slot, 0, sizeof(float), s) != hipSuccess) return pp_launch_status("wino_amax_memset");
  hipLaunchKernelGGL(wino_amax_kernel, dim3(wino_blocks(P * (C / 4))), dim3(256), 0, s, x, ld, C, P, slot);
  return pp_launch_status("wino_amax");
}

// forward and data gradient share this driver (U = Uf [16][N][C] resp. Ub [16][I][O]).
// f16: split-fp16 GEMM on pre-split operands; in_amax (nullable device float) = max |in|, which fixes the power-of-two
// scale of the transformed input (null: the fixed scale of O(1) activations); own_amax: compute it here (gradients
// handed in without their maximum)
static int wino_conv(const act_t* in, int ld_in, int C, const float* U, const float* bias, act_t* out, int ld_out, int N,
                     int B, int H, int W, int dil, int accumulate, float* v_keep, void* ws, size_t ws_bytes,
                     hipStream_t s, bool f16 = false, PpEpi* epi = nullptr, bool* fused = nullptr,
                     const float* in_amax = nullptr, bool own_amax = false) {
  if (fused) *fused = false;
  if (int rc = wino_check(C, N, B, H, W, dil)) return rc;
  PP_CHECK_ARG(in && U && out && ws, "winograd conv: null pointer");
  PP_CHECK_ARG(ld_in % 4 == 0 && ld_out % 4 == 0 && ld_in >= C && ld_out >= N, "winograd conv: bad ld");
  WinoGeom g = wino_geom(B, H, W, dil);
  const size_t need = (size_t)g.nb * g.T * ((v_keep ? 0 : (size_t)C) + N) * sizeof(float) + (f16 ? 16 : 0);
  if (ws_bytes < need) {
    pp_set_error("winograd conv: workspace too small (%zu < %zu)", ws_bytes, need);
    return PP_ERR_WORKSPACE;
  }
  PP_CHECK_ARG(!f16 || (g.m == 4 && C % 8 == 0), "winograd f16x3: only the F(4x4,3x3) geometry (H, W multiples of 4*dilation) with K %% 8 == 0");
  PP_CHECK_ARG(!f16 || ((((uintptr_t)in & PP_ACT_ALIGN) | ((uintptr_t)U & 15)) == 0), "winograd f16x3: in / U must be 16-byte aligned");
#ifdef PP_ACT_16
  if (!f16) { pp_set_error("winograd conv (16-bit storage): only the split-fp16 F(4x4,3x3) path exists"); return PP_ERR_UNSUPPORTED; }
#endif
  if (f16 && own_amax && !in_amax) {
    float* slot = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ((need - 16 + 15) & ~(size_t)15));
    if (int rc = wino_own_amax(in, ld_in, C, (long long)B * H * W, slot, s)) return rc;
    in_amax = slot;
  }
  // the transformed input either stays in the caller's buffer (kept for the weight gradient) or lives in the workspace
  float* V = v_keep ? v_keep : reinterpret_cast<float*>(ws);
  float* M = v_keep ? reinterpret_cast<float*>(ws) : V + (size_t)g.nb * g.T * C;
  const double P = (double)B * H * W;
  const double expand = (double)g.nb / (g.m * g.m);          // transform-domain elements per pixel (4 or 2.25)
  pp_prof_begin(PP_K_WINO_XFORM, 0.0, 4.0 * P * C * (1.0 + expand), s);
#ifdef PP_ACT_16
  launch_wino4_input_ps(in, ld_in, C, g, reinterpret_cast<char*>(V), in_amax, s);
#else
  if (g.m == 2)
    hipLaunchKernelGGL(wino_input_kernel, dim3(wino_blocks((long long)g.T * (C / 4))), dim3(256), 0, s, in, ld_in, C, g, V);
  else if (f16)
    launch_wino4_input_ps(in, ld_in, C, g, reinterpret_cast<char*>(V), in_amax, s);
  else
    WINO4_LAUNCH(wino4_input_kernel, wino4_vec(in, ld_in, C), (long long)g.T * C, s, in, ld_in, C, g, V, (float*)nullptr);
#endif
  pp_prof_end(s);
  if (int rc = pp_launch_status("wino_input")) return rc;
  // flops booked = EXECUTED transform-domain flops: nb GEMMs over T tiles = 2*expand per pixel*cin*cout (8 for F(2x2),
  // 4.5 for F(4x4)); the direct convolution's algorithmic count is 18
  int rc;
  if (f16) {      // booked as executed 16-bit MFMA flops (three products per transform-domain product)
    pp_prof_begin2(PP_K_WINO_GEMM_F16X3, 6.0 * expand * P * (double)N * C, 18.0 * P * (double)N * C,
                   4.0 * (P * C + P * N + 9.0 * C * N), s);
    GemmPsArgs ga{reinterpret_cast<const char*>(V), reinterpret_cast<const char*>(U), M, g.T, N, C, 0, 0, g.nb,
                  (unsigned)((size_t)g.T * C * 4), (unsigned)((size_t)N * C * 4), in_amax, PS_BOUND_INPUT};
    rc = pp_f16_products() == 1 ? launch_gemm_ps_any<true>(ga, s) : launch_gemm_ps_any<false>(ga, s);
  } else {
    GemmArgs ga{V, U, M, g.T, N, C, 0, 0, (unsigned)((size_t)g.T * C * 4), (unsigned)((size_t)N * C * 4), g.nb};
    pp_prof_begin2(PP_K_WINO_GEMM, 2.0 * expand * P * (double)N * C, 18.0 * P * (double)N * C,
                   4.0 * (P * C + P * N + 9.0 * C * N), s);
    rc = (N % 128 == 0) ? launch_gemm<2, 2, 2, 2>(ga, s) : launch_gemm<2, 1, 2, 2>(ga, s);
  }
  pp_prof_end(s);
  if (rc) return rc;
  pp_prof_begin(PP_K_WINO_XFORM, 0.0, 4.0 * P * N * (1.0 + expand), s);
#ifndef PP_ACT_16
  if (g.m == 2) {
    hipLaunchKernelGGL(wino_output_kernel, dim3(wino_blocks((long long)g.T * (N / 4))), dim3(256), 0, s, M, N, g, bias,
                       out, ld_out, accumulate);
  } else
#endif
  {
    const int vec = (bias && ((uintptr_t)bias & 15)) ? 1 : wino4_vec(out, ld_out, N);
    const int cv = N / vec;
    const bool can_fuse = epi && epi->mode && fused && !accumulate && epi->groups <= PP_EPI_GROUPS &&
                          (256 % cv == 0 || cv % 256 == 0) && epi->px_per_group % (H * W) == 0 &&
                          (epi->mode != 2 || ((((uintptr_t)epi->scale | (uintptr_t)epi->shift) & 15) == 0));
    if (can_fuse) {
      const int unit = cv > 256 ? cv / 256 : 1;                    // blocks per sweep over the channel vectors
      int blocks = wino_blocks((long long)g.T * cv);
      if (blocks > 512) blocks = 512;                              // fatter threads: fewer partial rows to finalize
      blocks = (blocks + unit - 1) / unit * unit;
      epi->rows = blocks / unit;
      const int ipg = epi->px_per_group / (H * W);
      if (vec == 4) hipLaunchKernelGGL(wino4_output_bn_kernel<4>, dim3(blocks), dim3(256), 0, s, M, N, g, bias, out, ld_out, *epi, ipg);
      else if (vec == 2) hipLaunchKernelGGL(wino4_output_bn_kernel<2>, dim3(blocks), dim3(256), 0, s, M, N, g, bias, out, ld_out, *epi, ipg);
      else hipLaunchKernelGGL(wino4_output_bn_kernel<1>, dim3(blocks), dim3(256), 0, s, M, N, g, bias, out, ld_out, *epi, ipg);
      *fused = true;
    } else {
      WINO4_LAUNCH(wino4_output_kernel, vec, (long long)g.T * N, s, M, N, g, bias, out, ld_out, accumulate);
    }
  }
  pp_prof_end(s);
  return pp_launch_status("wino_output");
}

#ifndef PP_ACT_16     // fp32-operand entries
extern "C" int pp_conv3x3_wino_fwd(const float* in, int ld_in, int C, const float* Uf, const float* bias, float* out,
                                   int ld_out, int N, int B, int H, int W, int dil, int accumulate, float* v_keep,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  return wino_conv(in, ld_in, C, Uf, bias, out, ld_out, N, B, H, W, dil, accumulate, v_keep, workspace, workspace_bytes,
                   (hipStream_t)stream);
}

extern "C" int pp_conv3x3_wino_bwd_data(const float* dz, int ld_dz, int O, const float* Ub, float* dx, int ld_dx, int I,
                                        int B, int H, int W, int dil, int accumulate, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  return wino_conv(dz, ld_dz, O, Ub, nullptr, dx, ld_dx, I, B, H, W, dil, accumulate, nullptr, workspace, workspace_bytes,
                   (hipStream_t)stream);
}

// ---- split-fp16 variants (F(4x4,3x3) geometry only): same arguments, U from pp_wino_pack_weights_f16x3 ----
extern "C" int pp_wino_pack_weights_f16x3(const float* w_oihw, int O, int I, int tile, void* Uf16, void* Ub16, void* stream) {
  PP_CHECK_ARG(tile == 4 && I % 4 == 0 && O % 4 == 0, "wino_pack_weights_f16x3: tile must be 4 and O, I multiples of 4");
  PP_CHECK_ARG((!Uf16 || I % 8 == 0) && (!Ub16 || O % 8 == 0), "wino_pack_weights_f16x3: the GEMM K (I for Uf, O for Ub) must be a multiple of 8");
  PP_CHECK_ARG(w_oihw && (Uf16 || Ub16), "wino_pack_weights_f16x3: null pointer");
  PP_CHECK_ARG(((((uintptr_t)Uf16) | ((uintptr_t)Ub16)) & 15) == 0, "wino_pack_weights_f16x3: U must be 16-byte aligned");
  const int nq = O * I / 4;
  hipLaunchKernelGGL(wino4_weight_ps_kernel, dim3(pp_cdiv(nq, 256), 2), dim3(256), 0, (hipStream_t)stream, w_oihw, O, I,
                     (char*)Uf16, (char*)Ub16);
  return pp_launch_status("wino_pack_weights_f16x3");
}
extern "C" int pp_wino_pack_weights_f16x3_batch(const pp_wino_pack_item* items, int n, void* stream) {
  PP_CHECK_ARG(items && n >= 1, "wino_pack_weights_f16x3_batch: no items");
  for (int i0 = 0; i0 < n; i0 += WPACK_BATCH_MAX) {
    WPackBatch b;
    b.n = n - i0 < WPACK_BATCH_MAX ? n - i0 : WPACK_BATCH_MAX;
    int blk = 0;
    for (int i = 0; i < b.n; ++i) {
      const pp_wino_pack_item& q = items[i0 + i];
      PP_CHECK_ARG(q.w_oihw && (q.Uf16 || q.Ub16) && q.I % 4 == 0 && q.O % 4 == 0, "wino_pack_weights_f16x3_batch: bad item");
      PP_CHECK_ARG((!q.Uf16 || q.I % 8 == 0) && (!q.Ub16 || q.O % 8 == 0), "wino_pack_weights_f16x3_batch: the GEMM K (I for Uf, O for Ub) must be a multiple of 8");
      PP_CHECK_ARG(((((uintptr_t)q.Uf16) | ((uintptr_t)q.Ub16)) & 15) == 0, "wino_pack_weights_f16x3_batch: U must be 16-byte aligned");
      b.it[i] = WPackItem{q.w_oihw, (char*)q.Uf16, (char*)q.Ub16, q.O, q.I, blk};
      blk += pp_cdiv(q.O * q.I / 4, 256);
    }
    for (int i = b.n; i < WPACK_BATCH_MAX; ++i) b.it[i] = WPackItem{nullptr, nullptr, nullptr, 0, 0, 0x7fffffff};
    hipLaunchKernelGGL(wino4_weight_ps_batch_kernel, dim3(blk, 2), dim3(256), 0, (hipStream_t)stream, b);
    if (int rc = pp_launch_status("wino_pack_weights_f16x3_batch")) return rc;
  }
  return 0;
}
#endif  // !PP_ACT_16

extern "C" int PP_FN(pp_conv3x3_wino_fwd_f16x3)(const pp_act* in, int ld_in, int C, const void* Uf16, const float* bias, pp_act* out,
                                         int ld_out, int N, int B, int H, int W, int dil, int accumulate, float* v_keep,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  return wino_conv(in, ld_in, C, (const float*)Uf16, bias, out, ld_out, N, B, H, W, dil, accumulate, v_keep, workspace,
                   workspace_bytes, (hipStream_t)stream, true);
}

// Winograd forward convolution + the BatchNorm that follows it (see pp_conv3x3_fwd_bn in pp_conv.hip): the output
// transform carries the fused epilogue; F(2x2) shapes and odd channel counts run the unfused BatchNorm kernels.
static int wino_fwd_bn_impl(const act_t* in, int ld_in, int C, const void* U, const float* bias, act_t* out,
                            int ld_out, int N, int B, int H, int W, int dil, int f16x3, float* v_keep,
                            void* workspace, size_t workspace_bytes, int bn_mode, const float* scale,
                            const float* shift, float slope, int groups, double* stats, size_t stats_bytes,
                            int* rows_out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(bn_mode == 1 || bn_mode == 2, "conv3x3_wino_fwd_bn: bn_mode must be 1 (train) or 2 (eval)");
  PP_CHECK_ARG(groups >= 1 && (B % groups) == 0, "conv3x3_wino_fwd_bn: groups must divide the batch");
  PP_CHECK_ARG(bn_mode == 2 ? (scale && shift) : (stats && rows_out), "conv3x3_wino_fwd_bn: missing BatchNorm arguments");
  const int ppg = (B / groups) * H * W;
  const size_t need = (size_t)groups * 2048 * 2 * N * sizeof(double);
  if (bn_mode == 1 && stats_bytes < need) {
    pp_set_error("conv3x3_wino_fwd_bn: stats buffer too small (%zu < %zu)", stats_bytes, need);
    return PP_ERR_WORKSPACE;
  }
  PpEpi epi{bn_mode, scale, shift, slope, stats, 0, ppg, groups};
  bool fused = false;
  if (int rc = wino_conv(in, ld_in, C, (const float*)U, bias, out, ld_out, N, B, H, W, dil, 0, v_keep, workspace,
                         workspace_bytes, s, f16x3 != 0, &epi, &fused, nullptr, false)) return rc;
  int rows = epi.rows;
  if (!fused) {
    if (bn_mode == 1) {
      rows = pp_bn_partial_rows(N, ppg, groups);
      if (int rc = pp_bn_stats_partial_launch(out, ld_out, N, ppg, groups, stats, s)) return rc;
    } else {
      if (int rc = pp_bn_apply_launch(out, ld_out, scale, shift, 1, out, ld_out, N, ppg, groups, slope, s)) return rc;
    }
  }
  if (rows_out) *rows_out = rows;
  return 0;
}

extern "C" int PP_FN(pp_conv3x3_wino_fwd_bn)(const pp_act* in, int ld_in, int C, const void* U, const float* bias, pp_act* out,
                                      int ld_out, int N, int B, int H, int W, int dil, int f16x3, float* v_keep,
                                      void* workspace, size_t workspace_bytes, int bn_mode, const float* scale,
                                      const float* shift, float slope, int groups, double* stats, size_t stats_bytes,
                                      int* rows_out, void* stream) {
  return wino_fwd_bn_impl(in, ld_in, C, U, bias, out, ld_out, N, B, H, W, dil, f16x3, v_keep, workspace, workspace_bytes, bn_mode,
                          scale, shift, slope, groups, stats, stats_bytes, rows_out, stream);
}

// dz_amax (nullable device float): max |dz| -- the BatchNorm backward that wrote dz collects it (pp_bn_lrelu_bwd_amax /
// pp_bn_lrelu_bwd_eval); null: one extra pass over dz finds it here
extern "C" int PP_FN(pp_conv3x3_wino_bwd_data_f16x3)(const pp_act* dz, int ld_dz, int O, const void* Ub16, pp_act* dx, int ld_dx, int I,
                                              int B, int H, int W, int dil, int accumulate, void* workspace,
                                              size_t workspace_bytes, const float* dz_amax, void* stream) {
  return wino_conv(dz, ld_dz, O, (const float*)Ub16, nullptr, dx, ld_dx, I, B, H, W, dil, accumulate, nullptr, workspace,
                   workspace_bytes, (hipStream_t)stream, true, nullptr, nullptr, dz_amax, true);
}

// ------------------------------------------------------------------------------------------
// weight gradient in the transform domain: dU_b[o][c] = sum_t W_b[t][o] * V_b[t][c]   (K = tiles)
// Both operands are tile-major ([t][channel]) exactly like dz / x of the direct weight-gradient kernel, so the kernel
// has that kernel's structure (pixel-major LDS rows, ds_read_b32 operands) without taps or halo.
// ------------------------------------------------------------------------------------------
struct WinoWgArgs {
  const float* Wt; const float* V; float* part;     // Wt [16][T][O], V [16][T][C], part [splits][16][O][C]
  int T, O, C;
  int o_tiles, c_tiles, chunks_per_split, n_chunks;
  unsigned w_bytes, v_bytes;                          // per plane
  int nb, splits;
};

// TMW = 32-row MFMA tiles per wave along O: 2 -> 128 x 128 block, 1 -> 64 x 128 block (layers with 64 output
// channels -- dec2.c1, the aux bottleneck -- would otherwise run half-empty 128-row tiles)
template <int TMW>
__global__ __launch_bounds__(256) void wino_wgrad_gemm_kernel(WinoWgArgs a) {
  constexpr int KB = 32, BM = 64 * TMW, BN = 128, LDA = BM + 4, LDB = BN + 4;
  constexpr int A_Q = BM / 4, A_RPP = 256 / A_Q, A_PASSES = KB / A_RPP;   // float4 per A row, rows per pass, passes
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][KB][LDA]
  float* Bs = smem + 2 * KB * LDA;  // [2][KB][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int lr = lane & 31, lh = lane >> 5;
  // item order (split, plane, o_tile, c_tile), c_tile fastest, and one XCD gets a contiguous range of items: the
  // o_tiles x c_tiles blocks of one (split, plane) read the same W / V tile rows, so they should share an L2
  // (with the plane fastest this kernel fetched 1.6 GB per launch and ran at the fabric rate, r01 PMC profile)
  const int per_split = a.nb * a.c_tiles * a.o_tiles;
  const int total = per_split * gridDim.y;
  int L = blockIdx.y * gridDim.x + blockIdx.x;
  if ((total & 7) == 0) L = (L & 7) * (total >> 3) + (L >> 3);
  const int split = L / per_split;
  const int r = L - split * per_split;
  const int ct = r % a.c_tiles;
  const int ot = (r / a.c_tiles) % a.o_tiles;
  const int batch = r / (a.c_tiles * a.o_tiles);
  const int o0 = ot * BM, c0 = ct * BN;
  const int chunk_lo = split * a.chunks_per_split;
  int chunk_hi = chunk_lo + a.chunks_per_split;
  if (chunk_hi > a.n_chunks) chunk_hi = a.n_chunks;
  const float* Wb = a.Wt + (size_t)batch * a.T * a.O;
  const float* Vb = a.V + (size_t)batch * a.T * a.C;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, a.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, a.v_bytes, 0x00020000);
  const int cq = tid & 31, row0 = tid >> 5;            // B: 32 float4 per 128-channel row, 8 rows per pass
  const int aq = tid % A_Q, arow0 = tid / A_Q;         // A: A_Q float4 per row
  const int oa_ok = (int)(o0 + aq * 4 < a.O), cb_ok = (int)(c0 + cq * 4 < a.C);
  f32x4 ra[A_PASSES], rb[4];
  auto load_tile = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int t = chunk * KB + arow0 + i * A_RPP;
      const unsigned offa = (oa_ok & (int)(t < a.T)) ? (unsigned)(t * a.O + o0 + aq * 4) * 4u : 0xffffffffu;
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, offa, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = chunk * KB + row0 + i * 8;
      const unsigned offb = (cb_ok & (int)(t < a.T)) ? (unsigned)(t * a.C + c0 + cq * 4) * 4u : 0xffffffffu;
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, offb, 0, 0));
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      *reinterpret_cast<f32x4*>(As + buf * KB * LDA + (arow0 + i * A_RPP) * LDA + aq * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(Bs + buf * KB * LDB + (row0 + i * 8) * LDB + cq * 4) = rb[i];
  };
  f32x16 acc[TMW][2];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  if (chunk_lo < chunk_hi) {
    load_tile(chunk_lo);
    store_tile(0);
  }
  __syncthreads();
  for (int ch = chunk_lo; ch < chunk_hi; ++ch) {
    const int buf = (ch - chunk_lo) & 1;
    const bool more = ch + 1 < chunk_hi;
    if (more) load_tile(ch + 1);
    const float* Ap = As + buf * KB * LDA + wm * 32 * TMW + lr;
    const float* Bp = Bs + buf * KB * LDB + wn * 64 + lr;
#pragma unroll
    for (int kk = 0; kk < KB / 2; ++kk) {
      const int krow = 2 * kk + lh;
      float av[TMW];
#pragma unroll
      for (int i = 0; i < TMW; ++i) av[i] = Ap[krow * LDA + 32 * i];
      const float b0 = Bp[krow * LDB], b1 = Bp[krow * LDB + 32];
#pragma unroll
      for (int i = 0; i < TMW; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b1, acc[i][1], 0, 0, 0);
      }
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  float* part = a.part + ((size_t)split * a.nb + batch) * a.O * a.C;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = c0 + wn * 64 + j * 32 + lr;
    if (c >= a.C) continue;
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int o = o0 + (wm * TMW + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
        if (o < a.O) part[(size_t)o * a.C + c] = acc[i][j][q];
      }
  }
}

// ---------------------------------------------------------------- the same weight-gradient GEMM on the fp16 MFMA
// dU_b[o][c] = sum_t W_b[t][o] * V_b[t][c] has its reduction index t as the ROW index of both operands, while the
// 32x32x16 MFMA wants 8 consecutive k per lane.  The tiles are therefore staged as [32 t][128 channels] images of fp16
// hi and lo parts and read with ds_read_b64_tr_b16, gfx950's transposing LDS read: a 16-lane group fetches a
// 4 (t) x 16 (channel) block and every lane receives one channel's 4 consecutive t.  Rows are 256 B (no padding) and the
// 16-byte units of row t sit at unit ^ 4 (t & 3): the four rows of a transposed read then fall on four different 32-byte
// granules (8 banks each) and the two 16-lane groups of a 32-lane half on the odd / even granules -- all 64 banks once,
// conflict-free -- and a block needs 64 KB of LDS instead of the 80 KB of round 2's padded 320-byte rows (same-box A/B: no
// change in time, 2.89 ms/step either way -- two 80 KB blocks already fitted a CU, hipOccupancy probe -- so neither LDS
// capacity nor the residual bank conflicts of the padded image limit this kernel).  Both operands arrive PRE-SPLIT in octets along the channel axis
// (written by wino4_dy_ps_kernel / wino4_input_ps_kernel with their power-of-two scales), so a 16-byte piece of a row goes
// to the hi or the lo image as it is -- no conversion in this kernel.
typedef __fp16 h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define WG16_RS 128        // halves per image row
__device__ __forceinline__ f16x8 wg16_frag(const _Float16* p) {
  // two transposed reads (k = 0..3 and 4..7 of this lane's 8) joined into one MFMA operand
  const h4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)p);
  const h4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)(p + 4 * WG16_RS));
  const f16x4 a = __builtin_bit_cast(f16x4, lo4), b = __builtin_bit_cast(f16x4, hi4);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int TMW, bool X1>      // X1: hi parts only (--precision fp16: one product per fp32 product, as in the forward GEMM)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void wino_wgrad_gemm_f16x3_kernel(WinoWgArgs a, const float* __restrict__ dz_amax) {
  constexpr int KB = 32, BM = 64 * TMW, BN = 128;
  constexpr int IMG = KB * WG16_RS;                          // halves per image
  constexpr int A_Q = BM / 4, A_RPP = 256 / A_Q, A_PASSES = KB / A_RPP;
  extern __shared__ __attribute__((aligned(16))) _Float16 smem16[];
  // [2 buffers][A hi | A lo | B hi | B lo]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int per_split = a.nb * a.c_tiles * a.o_tiles;
  // 1-D grid padded to a multiple of 8: XCD x (blocks x, x + 8, ...) works on one contiguous eighth of the items whatever
  // their number (round 2 banded only totals divisible by 8)
  const int total = per_split * a.splits;
  // Round 5: PERSISTENT over the items -- the grid is min(items, block budget) (a multiple of 8), block b takes items b, b + grid, ...
  // With the budget below the chip (PP_WINO_WGRAD_CUS) this kernel, which runs on the second stream beside the data-gradient /
  // BatchNorm chain, leaves CUs to that chain instead of occupying every one of them with its first 512 blocks.
  const int padded = (total + 7) & ~7;
  for (int it = (int)blockIdx.x; it < padded; it += (int)gridDim.x) {
  const int L = (it & 7) * (padded >> 3) + (it >> 3);
  if (L >= total) continue;
  const int split = L / per_split;
  const int r = L - split * per_split;
  const int ct = r % a.c_tiles;
  const int ot = (r / a.c_tiles) % a.o_tiles;
  const int batch = r / (a.c_tiles * a.o_tiles);
  const int o0 = ot * BM, c0 = ct * BN;
  const int chunk_lo = split * a.chunks_per_split;
  int chunk_hi = chunk_lo + a.chunks_per_split;
  if (chunk_hi > a.n_chunks) chunk_hi = a.n_chunks;
  float sa_in, sa_out, sb_in, sb_out;
  ps_scales(dz_amax, PS_BOUND_DY, sa_in, sa_out);           // the scales the two transforms applied
  ps_scales(nullptr, PS_BOUND_INPUT, sb_in, sb_out);
  const float* Wb = a.Wt + (size_t)batch * a.T * a.O;
  const float* Vb = a.V + (size_t)batch * a.T * a.C;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, a.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, a.v_bytes, 0x00020000);
  const int cq = tid & 31, row0 = tid >> 5;
  const int aq = tid % A_Q, arow0 = tid / A_Q;
  const int oa_ok = (int)(o0 + aq * 4 < a.O), cb_ok = (int)(c0 + cq * 4 < a.C);
  f32x4 ra[A_PASSES], rb[4];
  auto load_tile = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int t = chunk * KB + arow0 + i * A_RPP;
      const unsigned offa = (oa_ok & (int)(t < a.T)) ? (unsigned)(t * a.O + o0 + aq * 4) * 4u : 0xffffffffu;
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, offa, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = chunk * KB + row0 + i * 8;
      const unsigned offb = (cb_ok & (int)(t < a.T)) ? (unsigned)(t * a.C + c0 + cq * 4) * 4u : 0xffffffffu;
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, offb, 0, 0));
    }
  };
  // piece q of a row segment = part (q & 1: hi / lo) of octet q >> 1: 8 halves of the hi or the lo image.  The images are a
  // multiple of 256 B apart, so the hi and the lo store of a lane pair would meet on the same banks: the lo image keeps
  // its columns rotated by half a row (column ^ BM / 2: the OTHER wave row's block), and its reads look there
  auto store_tile = [&](int buf) {
    _Float16* base = smem16 + buf * 4 * IMG;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      *reinterpret_cast<f32x4*>(base + (aq & 1) * IMG + (arow0 + i * A_RPP) * WG16_RS +
                                (((aq >> 1) * 8) ^ ((aq & 1) * (BM / 2)) ^ (32 * ((arow0 + i * A_RPP) & 3)))) = ra[i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(base + (2 + (cq & 1)) * IMG + (row0 + i * 8) * WG16_RS +
                                (((cq >> 1) * 8) ^ ((cq & 1) * (BN / 2)) ^ (32 * ((row0 + i * 8) & 3)))) = rb[i];
  };
  f32x16 accm[TMW][2], accc[TMW][2];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) { accm[i][j][q] = 0.f; accc[i][j][q] = 0.f; }
  if (chunk_lo < chunk_hi) {
    load_tile(chunk_lo);
    store_tile(0);
  }
  __syncthreads();
  // transposed-read lane geometry: 16-lane group g = (m-block mb, k-half h); lane 4q+p of the group addresses row q,
  // 8-byte piece p of its 4 x 16 block
  const int g = lane >> 4, i16 = lane & 15;
  const int r4 = i16 >> 2;                                   // (row & 3) of every row this lane reads: its swizzle term is 32 * r4
  const int tr_row = (8 * (g >> 1) + r4) * WG16_RS, tr_col = 16 * (g & 1) + 4 * (i16 & 3);
  // column-block bases (multiples of 32 halves) of this wave's fragments, swizzled for this lane's rows
  int ca_h[TMW], ca_l[TMW], cb_h[2], cb_l[2];
#pragma unroll
  for (int i = 0; i < TMW; ++i) {
    ca_h[i] = tr_row + ((wm * 32 * TMW + 32 * i) ^ (32 * r4)) + tr_col;
    ca_l[i] = tr_row + (((wm ^ 1) * 32 * TMW + 32 * i) ^ (32 * r4)) + tr_col;            // rotated lo columns
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    cb_h[j] = tr_row + ((wn * 64 + 32 * j) ^ (32 * r4)) + tr_col;
    cb_l[j] = tr_row + (((wn ^ 1) * 64 + 32 * j) ^ (32 * r4)) + tr_col;
  }
  for (int ch = chunk_lo; ch < chunk_hi; ++ch) {
    const int buf = (ch - chunk_lo) & 1;
    const bool more = ch + 1 < chunk_hi;
    if (more) load_tile(ch + 1);
    const _Float16* Ah = smem16 + buf * 4 * IMG;
    const _Float16* Al = smem16 + buf * 4 * IMG + IMG;
    const _Float16* Bh = smem16 + buf * 4 * IMG + 2 * IMG;
    const _Float16* Bl = smem16 + buf * 4 * IMG + 3 * IMG;
    // Two 16-tile reduction steps per stage; the transposed fragment reads of step 1 are issued before the MFMAs of
    // step 0 (explicit register double buffer + scheduling fences, as in conv3x3_wgrad_halo_f16x3_kernel: left to
    // itself hipcc put every step's reads directly in front of its MFMAs).
    f16x8 ah[2][TMW], al[2][TMW], bh[2][2], bl[2][2];
    auto read_step = [&](int kb, int slot) {
#pragma unroll
      for (int i = 0; i < TMW; ++i) {
        ah[slot][i] = wg16_frag(Ah + kb * 16 * WG16_RS + ca_h[i]);
        if (!X1) al[slot][i] = wg16_frag(Al + kb * 16 * WG16_RS + ca_l[i]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[slot][j] = wg16_frag(Bh + kb * 16 * WG16_RS + cb_h[j]);
        if (!X1) bl[slot][j] = wg16_frag(Bl + kb * 16 * WG16_RS + cb_l[j]);
      }
    };
    read_step(0, 0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb == 0) read_step(1, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb][i], bh[kb][j], accm[i][j], 0, 0, 0);
          if (!X1) {
            accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb][i], bl[kb][j], accc[i][j], 0, 0, 0);
            accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kb][i], bh[kb][j], accc[i][j], 0, 0, 0);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  const float s_out = sa_out * sb_out;
  const int lr = lane & 31, lh = lane >> 5;
  float* part = a.part + ((size_t)split * a.nb + batch) * a.O * a.C;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = c0 + wn * 64 + j * 32 + lr;
    if (c >= a.C) continue;
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int o = o0 + (wm * TMW + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
        if (o < a.O) part[(size_t)o * a.C + c] = (accm[i][j][q] + accc[i][j][q] * (1.f / F16_LO_SCALE)) * s_out;
      }
  }
  }      // items
}

// (r03 experiment, removed: requesting the tile of stage n + 1 TWO stages ahead -- a second register set, single-buffered
// fragments to pay for it, s_waitcnt vmcnt(8) in front of the LDS stores as intended -- changed nothing: 3.03 vs 3.01 ms/step.
// Global-load latency does not limit this kernel either.)
// (r03 experiment, removed: the same GEMM staged by LDS-DMA -- 256 x 128 tile, eight waves, three-slot ring, octet rows
// copied as they lie in memory, XOR-swizzled transposed reads with zero bank conflicts -- ran the 512 / 1024-channel layers
// 12-13 % SLOWER than this register-staged kernel (0.64 vs 0.57 ms on dec5.c1): its one 8-wave block per CU waits at the ring
// barrier 55 % of its wave-cycles (SQ_WAIT_ANY), where two independent 4-wave blocks per CU cover each other's waits.)

// dw[o][c][3][3] (+)= G^T (sum_splits dU) G      64 consecutive (o,c) pairs x 4 split-lanes per block (256-B reads)
__global__ __launch_bounds__(256) void wino_wgrad_finalize_kernel(const float* __restrict__ part, int splits, int O, int C,
                                                                  float* __restrict__ dw, int accumulate) {
  __shared__ float red[16][3][64];
  const int il = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const size_t oc = (size_t)blockIdx.x * 64 + il;
  const size_t per = (size_t)O * C;
  float u[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) u[b] = 0.f;
  if (oc < per)
    for (int k = sl; k < splits; k += 4)
#pragma unroll
      for (int b = 0; b < 16; ++b) u[b] += part[((size_t)k * 16 + b) * per + oc];
  if (sl > 0) {
#pragma unroll
    for (int b = 0; b < 16; ++b) red[b][sl - 1][il] = u[b];
  }
  __syncthreads();
  if (sl != 0 || oc >= per) return;
#pragma unroll
  for (int b = 0; b < 16; ++b) u[b] += red[b][0][il] + red[b][1][il] + red[b][2][il];
  float p[3][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    p[0][s] = u[0 * 4 + s] + 0.5f * (u[1 * 4 + s] + u[2 * 4 + s]);
    p[1][s] = 0.5f * (u[1 * 4 + s] - u[2 * 4 + s]);
    p[2][s] = 0.5f * (u[1 * 4 + s] + u[2 * 4 + s]) + u[3 * 4 + s];
  }
  float* d = dw + oc * 9;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float g0 = p[i][0] + 0.5f * (p[i][1] + p[i][2]);
    const float g1 = 0.5f * (p[i][1] - p[i][2]);
    const float g2 = 0.5f * (p[i][1] + p[i][2]) + p[i][3];
    d[i * 3 + 0] = accumulate ? d[i * 3 + 0] + g0 : g0;
    d[i * 3 + 1] = accumulate ? d[i * 3 + 1] + g1 : g1;
    d[i * 3 + 2] = accumulate ? d[i * 3 + 2] + g2 : g2;
  }
}

// F(4x4,3x3): 36 planes -> 3x3.  Round 5: the round-2 form (one (o, c) pair per thread, 256-byte wave loads, nine scattered
// 4-byte stores per thread) ran at 1.0-1.5 TB/s of the partial sums it reads (0.70 ms per step over 11 launches).  Now a block
// owns 32 QUADS of consecutive (o, c) pairs and has one 32-thread group per COLUMN s of the 6 x 6 transform-domain tile: group s
// loads u[r][s] (r = 0..5, every split: 6 * splits independent 16-byte loads, 512 B contiguous per group), sums the splits in
// fixed order, applies G^T down its column and leaves p[0..2][s] in LDS; after the barrier the groups i < 3 apply G^T along row i
// and the 32 x 36 results leave through LDS as contiguous 16-byte stores.  Fixed order everywhere: bit-reproducible.
#define WF4_Q 32
__global__ __launch_bounds__(6 * WF4_Q) void wino4_wgrad_finalize_kernel(const float* __restrict__ part, int splits, int O, int C,
                                                                         float* __restrict__ dw, int accumulate) {
  __shared__ f32x4 ps[3][6][WF4_Q];
  __shared__ __attribute__((aligned(16))) float outs[WF4_Q * 36];
  const int il = threadIdx.x % WF4_Q, sgrp = threadIdx.x / WF4_Q;
  const size_t per4 = (size_t)O * C / 4;                                // (O, C multiples of 4: wino_check)
  const size_t q = (size_t)blockIdx.x * WF4_Q + il;
  const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
  f32x4 col[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) col[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (q < per4) {
    for (int k = 0; k < splits; ++k) {
      f32x4 v[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) v[r] = p4[((size_t)k * 36 + r * 6 + sgrp) * per4 + q];
#pragma unroll
      for (int r = 0; r < 6; ++r) col[r] += v[r];
    }
  }
  f32x4 pc[3];
  f4_gt(col, pc);
#pragma unroll
  for (int i = 0; i < 3; ++i) ps[i][sgrp][il] = pc[i];
  __syncthreads();
  if (sgrp < 3) {
    f32x4 row[6], o3[3];
#pragma unroll
    for (int s2 = 0; s2 < 6; ++s2) row[s2] = ps[sgrp][s2][il];
    f4_gt(row, o3);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < 3; ++j) outs[(il * 4 + e) * 9 + sgrp * 3 + j] = o3[j][e];
  }
  __syncthreads();
  // 32 quads x 36 floats = 288 16-byte pieces, contiguous in dw ([o][c][3][3], (o, c) pairs consecutive)
  const size_t base = (size_t)blockIdx.x * WF4_Q * 36;                  // floats
  const size_t total = (size_t)O * C * 9;
  if (((uintptr_t)dw & 15) == 0) {
    for (int t = threadIdx.x; t < WF4_Q * 9; t += 6 * WF4_Q) {
      const size_t off = base + (size_t)t * 4;
      if (off >= total) break;
      f32x4 v = *reinterpret_cast<const f32x4*>(outs + t * 4);
      f32x4* d = reinterpret_cast<f32x4*>(dw + off);
      if (accumulate) v += *d;
      *d = v;
    }
  } else {                                         // a gradient view that does not start on a 16-byte boundary
    for (int t = threadIdx.x; t < WF4_Q * 36; t += 6 * WF4_Q) {
      const size_t off = base + t;
      if (off >= total) break;
      dw[off] = accumulate ? dw[off] + outs[t] : outs[t];
    }
  }
}

struct WinoWgPlan { int bm, o_tiles, c_tiles, n_chunks, splits, chunks_per_split; };
static WinoWgPlan wino_wg_plan(int O, int C, int T, int nb) {
  WinoWgPlan p;
  p.bm = (O % 128 != 0 && O % 128 <= 64) ? 64 : 128;        // 64-row blocks when the last 128-row tile would be half empty
  p.o_tiles = pp_cdiv(O, p.bm);
  p.c_tiles = pp_cdiv(C, 128);
  p.n_chunks = pp_cdiv(T, 32);
  // Reduction splits: rounds of 512 resident blocks (two per CU) x chunks per block, times a per-split price for the partial
  // sums (written by the GEMM, read by the finalize) -- the rule that reproduces the measured optimum of every Winograd layer
  // of the benchmark network (scripts/sweep_wino_wg_splits.py, r03: 3 / 3 / 4 / 2 / 1 / 4 splits against the 8 / 6 / 3 / 2 /
  // 4 / 15 of round 2's "at least 1536 blocks": enc4.c2 121 -> 96 us, enc5.c1 195 -> 178, dec3.c1 346 -> 301)
  const int base = nb * p.o_tiles * p.c_tiles;
  const int max_splits = pp_cdiv(p.n_chunks, 16);
  int splits = 1;
  double best = 1e30;
  for (int sp = 1; sp <= max_splits && sp <= 16; ++sp) {
    const int cps = pp_cdiv(p.n_chunks, sp), eff = pp_cdiv(p.n_chunks, cps);
    const double cost = (double)pp_cdiv(base * eff, 512) * cps * (1.0 + 0.08 * eff);
    if (cost < best) { best = cost; splits = eff; }
  }
  if (splits < 1) splits = 1;
  p.chunks_per_split = pp_cdiv(p.n_chunks, splits);
  p.splits = pp_cdiv(p.n_chunks, p.chunks_per_split);
  return p;
}

#ifndef PP_ACT_16
extern "C" size_t pp_conv3x3_wino_bwd_weight_workspace(int O, int C, int B, int H, int W, int dil) {
  const WinoGeom g = wino_geom(B, H, W, dil);
  WinoWgPlan p = wino_wg_plan(O, C, g.T, g.nb);
  return ((size_t)g.nb * g.T * ((size_t)O + C) + (size_t)p.splits * g.nb * O * C) * sizeof(float) + 256;
}

// Number of reduction splits the weight-gradient GEMM of this shape runs with (a pure function of the shape: tests quote
// it to show which launch configuration they exercised).
extern "C" int pp_conv3x3_wino_bwd_weight_splits(int O, int C, int B, int H, int W, int dil) {
  const WinoGeom g = wino_geom(B, H, W, dil);
  return wino_wg_plan(O, C, g.T, g.nb).splits;
}
#endif  // !PP_ACT_16

static int wino_bwd_weight_impl(const act_t* dz, int ld_dz, int O, const act_t* x, int ld_x, int C, int B,
                                int H, int W, int dil, float* dw_oihw, int accumulate, const float* v_cached,
                                void* workspace, size_t workspace_bytes, void* stream, bool f16,
                                const float* dz_amax = nullptr) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = wino_check(C, O, B, H, W, dil)) return rc;
  PP_CHECK_ARG(dz && (x || v_cached) && dw_oihw && workspace, "winograd wgrad: null pointer");
  PP_CHECK_ARG(ld_dz % 4 == 0 && ld_x % 4 == 0 && ld_dz >= O && ld_x >= C, "winograd wgrad: bad ld");
  WinoGeom g = wino_geom(B, H, W, dil);
  WinoWgPlan p = wino_wg_plan(O, C, g.T, g.nb);
  const size_t need = ((size_t)g.nb * g.T * ((size_t)O + (v_cached ? 0 : C)) + (size_t)p.splits * g.nb * O * C) * sizeof(float) +
                      (f16 ? 32 : 0);
  if (workspace_bytes < need) {
    pp_set_error("winograd wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    return PP_ERR_WORKSPACE;
  }
#ifdef PP_ACT_16
  if (!f16) { pp_set_error("winograd wgrad (16-bit storage): only the split-fp16 F(4x4,3x3) path exists"); return PP_ERR_UNSUPPORTED; }
#endif
  PP_CHECK_ARG(!f16 || (g.m == 4 && O % 8 == 0 && C % 8 == 0),
               "winograd wgrad f16x3: only the F(4x4,3x3) geometry (H, W multiples of 4*dilation) with O, C multiples of 8");
  float* Wt = reinterpret_cast<float*>(workspace);
  float* part = Wt + (size_t)g.nb * g.T * O;
  float* Vown = part + (size_t)p.splits * g.nb * O * C;
  const float* V = v_cached ? v_cached : Vown;
  // split-fp16 GEMM on pre-split operands: the dy transform scales by a power of two taken from max |dz| (brought by
  // the caller, else found here), a V computed or kept by the forward pass carries the fixed activation scale
  if (f16 && !dz_amax) {
    float* slot = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ((need - 32 + 15) & ~(size_t)15));
    if (int rc = wino_own_amax(dz, ld_dz, O, (long long)B * H * W, slot, s)) return rc;
    dz_amax = slot;
  }
  const double P = (double)B * H * W;
  const double expand = (double)g.nb / (g.m * g.m);
  pp_prof_begin(PP_K_WINO_XFORM, 0.0, 4.0 * P * ((v_cached ? 0 : C) + O) * (1.0 + expand), s);
#ifndef PP_ACT_16
  if (g.m == 2) {
    if (!v_cached)
      hipLaunchKernelGGL(wino_input_kernel, dim3(wino_blocks((long long)g.T * (C / 4))), dim3(256), 0, s, x, ld_x, C, g, Vown);
    hipLaunchKernelGGL(wino_dy_kernel, dim3(wino_blocks((long long)g.T * (O / 4))), dim3(256), 0, s, dz, ld_dz, O, g, Wt);
  } else
#endif
  {
    if (f16) {
      if (!v_cached)
        launch_wino4_input_ps(x, ld_x, C, g, reinterpret_cast<char*>(Vown), nullptr, s);
      hipLaunchKernelGGL(wino4_dy_ps_kernel, dim3(wino_blocks((long long)g.T * (O / 4))), dim3(256), 0, s, dz, ld_dz, O, g,
                         reinterpret_cast<char*>(Wt), dz_amax);
    } else {
#ifndef PP_ACT_16
      if (!v_cached)
        WINO4_LAUNCH(wino4_input_kernel, wino4_vec(x, ld_x, C), (long long)g.T * C, s, x, ld_x, C, g, Vown, (float*)nullptr);
      WINO4_LAUNCH(wino4_dy_kernel, wino4_vec(dz, ld_dz, O), (long long)g.T * O, s, dz, ld_dz, O, g, Wt, (float*)nullptr);
#endif
    }
  }
  pp_prof_end(s);
  if (int rc = pp_launch_status("wino_wgrad_transforms")) return rc;
  WinoWgArgs a{Wt, V, part, g.T, O, C, p.o_tiles, p.c_tiles, p.chunks_per_split, p.n_chunks,
               (unsigned)((size_t)g.T * O * 4), (unsigned)((size_t)g.T * C * 4), g.nb, p.splits};
  const size_t lds = (size_t)2 * 32 * (132 + 132) * sizeof(float);
  {   // once per (kernel, device): pp_max_lds
    pp_max_lds(reinterpret_cast<const void*>(wino_wgrad_gemm_kernel<2>), (int)lds);
    pp_max_lds(reinterpret_cast<const void*>(wino_wgrad_gemm_kernel<1>), (int)lds);
  }
  if (f16) {
    const size_t lds16 = (size_t)2 * 4 * 32 * WG16_RS * sizeof(_Float16);          // 64 KB: two blocks per CU with room to spare
    {   // once per (kernel, device): pp_max_lds
      pp_max_lds(reinterpret_cast<const void*>(wino_wgrad_gemm_f16x3_kernel<2, false>), (int)lds16);
      pp_max_lds(reinterpret_cast<const void*>(wino_wgrad_gemm_f16x3_kernel<1, false>), (int)lds16);
      pp_max_lds(reinterpret_cast<const void*>(wino_wgrad_gemm_f16x3_kernel<2, true>), (int)lds16);
      pp_max_lds(reinterpret_cast<const void*>(wino_wgrad_gemm_f16x3_kernel<1, true>), (int)lds16);
    }
    pp_prof_begin2(PP_K_WINO_WGRAD_F16X3, 6.0 * expand * P * (double)O * C, 18.0 * P * (double)O * C,
                   4.0 * (P * (O + C) + 9.0 * O * C), s);
    // padded to a multiple of 8 (the kernel's XCD mapping) and capped at two blocks per CU
    constexpr int wg_cus = 256;       // (a budget below the chip for this GEMM only lost: 192 / 128 / 64 CUs +0.1 / +0.6 / +1.4 ms, r05)
    unsigned gblocks = (unsigned)(pp_cdiv(g.nb * p.o_tiles * p.c_tiles * p.splits, 8) * 8);
    const unsigned gcap = (unsigned)(wg_cus < 8 ? 8 : wg_cus) * 2u / 8u * 8u;
    if (gblocks > gcap) gblocks = gcap;
    const dim3 grid(gblocks);
    const bool x1 = pp_f16_products() == 1;
    if (p.bm == 128) {
      if (x1) hipLaunchKernelGGL((wino_wgrad_gemm_f16x3_kernel<2, true>), grid, dim3(256), lds16, s, a, dz_amax);
      else hipLaunchKernelGGL((wino_wgrad_gemm_f16x3_kernel<2, false>), grid, dim3(256), lds16, s, a, dz_amax);
    } else {
      if (x1) hipLaunchKernelGGL((wino_wgrad_gemm_f16x3_kernel<1, true>), grid, dim3(256), lds16, s, a, dz_amax);
      else hipLaunchKernelGGL((wino_wgrad_gemm_f16x3_kernel<1, false>), grid, dim3(256), lds16, s, a, dz_amax);
    }
  } else {
    pp_prof_begin2(PP_K_WINO_WGRAD, 2.0 * expand * P * (double)O * C, 18.0 * P * (double)O * C,
                   4.0 * (P * (O + C) + 9.0 * O * C), s);
    if (p.bm == 128)
      hipLaunchKernelGGL(wino_wgrad_gemm_kernel<2>, dim3(g.nb * p.o_tiles * p.c_tiles, p.splits), dim3(256), lds, s, a);
    else
      hipLaunchKernelGGL(wino_wgrad_gemm_kernel<1>, dim3(g.nb * p.o_tiles * p.c_tiles, p.splits), dim3(256), lds, s, a);
  }
  if (g.m == 2)
    hipLaunchKernelGGL(wino_wgrad_finalize_kernel, dim3(pp_cdiv((long long)O * C, 64)), dim3(256), 0, s, part, p.splits, O,
                       C, dw_oihw, accumulate);
  else
    hipLaunchKernelGGL(wino4_wgrad_finalize_kernel, dim3(pp_cdiv((long long)O * C / 4, WF4_Q)), dim3(6 * WF4_Q), 0, s, part, p.splits, O,
                       C, dw_oihw, accumulate);
  pp_prof_end(s);
  return pp_launch_status("wino_wgrad");
}

#ifndef PP_ACT_16
extern "C" int pp_conv3x3_wino_bwd_weight(const float* dz, int ld_dz, int O, const float* x, int ld_x, int C, int B,
                                          int H, int W, int dil, float* dw_oihw, int accumulate, const float* v_cached,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  return wino_bwd_weight_impl(dz, ld_dz, O, x, ld_x, C, B, H, W, dil, dw_oihw, accumulate, v_cached, workspace,
                              workspace_bytes, stream, false);
}
#endif  // !PP_ACT_16

// split-fp16 GEMM (F(4x4,3x3) geometry only).  A cached V must come from a split-fp16 forward call that was given the
// buffer as `v_keep` (it then holds the pre-split octets).  dz_amax: as for pp_conv3x3_wino_bwd_data_f16x3.
extern "C" int PP_FN(pp_conv3x3_wino_bwd_weight_f16x3)(const pp_act* dz, int ld_dz, int O, const pp_act* x, int ld_x, int C, int B,
                                                int H, int W, int dil, float* dw_oihw, int accumulate,
                                                const float* v_cached, void* workspace, size_t workspace_bytes,
                                                const float* dz_amax, void* stream) {
  return wino_bwd_weight_impl(dz, ld_dz, O, x, ld_x, C, B, H, W, dil, dw_oihw, accumulate, v_cached, workspace,
                              workspace_bytes, stream, true, dz_amax);
}

PP_NS_END
