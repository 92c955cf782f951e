// Spatial glue kernels for NHWC fp32 activations (all HBM-bound, 16 B per lane along channels):
//   * NCHW image -> zero-padded NHWC                          (module boundary, train_chaos.py:269)
//   * MaxPool2d(2,2) forward / backward                        (models/unet.py:109)
//   * bilinear resize, align_corners=True, forward / backward  (models/unet.py:144, aux_path_memory.py:52,75)
//   * 1x1 convolution head NHWC -> NCHW logits fwd / bwd       (models/unet.py:60, aux_path_memory.py:32)
// Backward passes are written as gathers (every output element is produced by exactly one thread):
// no atomics, bit-reproducible.
#include "pp_common.h"

PP_NS_BEGIN

#define SP_THREADS 256
#define SP_MAX_BLOCKS 8192

static inline int sp_blocks(long long total) {
  int b = pp_cdiv(total, SP_THREADS);
  return b > SP_MAX_BLOCKS ? SP_MAX_BLOCKS : (b < 1 ? 1 : b);
}

// ---------------------------------------------------------------- image packing
__global__ void pack_image_kernel(const float* __restrict__ src, int N, int C, int HW, act_t* __restrict__ dst,
                                  int ld, int Cpad) {
  const long long total = (long long)N * HW * Cpad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const long long p = i / Cpad;
    const int n = (int)(p / HW), hw = (int)(p % HW);
    dst[p * ld + c] = c < C ? src[((size_t)n * C + c) * HW + hw] : 0.f;
  }
}

extern "C" int PP_FN(pp_pack_image_nchw_to_nhwc)(const float* src, int N, int C, int H, int W, pp_act* dst, int ld_dst,
                                          int Cpad, void* stream) {
  PP_CHECK_ARG(src && dst && Cpad >= C && ld_dst >= Cpad, "pack_image: bad arguments");
  const long long total = (long long)N * H * W * Cpad;
  hipLaunchKernelGGL(pack_image_kernel, dim3(sp_blocks(total)), dim3(SP_THREADS), 0, (hipStream_t)stream, src, N, C,
                     H * W, dst, ld_dst, Cpad);
  return pp_launch_status("pack_image");
}

// Blocks are dealt round-robin over the 8 XCDs (block b and b + 8 share an L2).  Kernels whose NEIGHBOURING rows read the
// same input rows (bilinear: two input rows per output row, four output rows per input row) renumber their row blocks so
// that every XCD works on one contiguous band of rows: the shared rows are then fetched into ONE L2 instead of up to eight
// (r02 PMC: bilinear_fwd / bwd moved 1.9x their algorithmic bytes).
__device__ __forceinline__ int xcd_band_row(int b, int rows) {
  return (rows & 7) == 0 ? (b & 7) * (rows >> 3) + (b >> 3) : b;
}

// ---------------------------------------------------------------- max pool 2x2 / stride 2
// lz (all kernels below that take one): x is a LAZY tensor (pp_common.h) -- BatchNorm + LeakyReLU of the layer that produced it are
// applied to every loaded value; the result these kernels write is an ordinary tensor.  LAZY is a template flag: the
// ordinary instantiations are the round-3 kernels instruction for instruction (a run-time test of lz.coef cost 0.2 ms per step)
#define SP_LAZY4(v, n_img)                                                                   \
  if (LAZY) {                                                                                \
    pp_f32x4 l_sc, l_sh, l_sl;                                                               \
    pp_lazy_rows4(lz, n_img, cq * 4, l_sc, l_sh, l_sl);                                      \
    _Pragma("unroll") for (int q_ = 0; q_ < (int)(sizeof(v) / sizeof(v[0])); ++q_) {        \
      const pp_f32x4 t_ = pp_lazy_apply4(pp_f32x4{v[q_].x, v[q_].y, v[q_].z, v[q_].w}, l_sc, l_sh, l_sl); \
      v[q_] = make_float4(t_[0], t_[1], t_[2], t_[3]);                                       \
    }                                                                                        \
  }

__global__ void maxpool2_fwd_kernel(const act_t* __restrict__ x, int ld_x, act_t* __restrict__ y, int ld_y, int C,
                                    int N, int H, int W) {
  // grid = (output rows, row segments): no 64-bit index arithmetic per element
  const int Ho = H >> 1, Wo = W >> 1, c4n = C >> 2;
  const int e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e < Wo * c4n) {
    const int xo = e / c4n, cq = e - xo * c4n;
    const int row = xcd_band_row(blockIdx.x, gridDim.x);
    const int n = row / Ho, yo = row - n * Ho;
    const size_t po = (size_t)row * Wo + xo;
    const size_t pi = ((size_t)n * H + 2 * yo) * W + 2 * xo;
    float4 v4[4];
    v4[0] = act_ld4f(x + pi * ld_x + cq * 4);
    v4[1] = act_ld4f(x + (pi + 1) * ld_x + cq * 4);
    v4[2] = act_ld4f(x + (pi + W) * ld_x + cq * 4);
    v4[3] = act_ld4f(x + (pi + W + 1) * ld_x + cq * 4);
    const float4 a = v4[0], b = v4[1], c = v4[2], d = v4[3];
    float4 o;
    o.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x));
    o.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
    o.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z));
    o.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
    act_st4f(y + (size_t)po * ld_y + cq * 4, o);
  }
}

// the first maximum in window order (0,0),(0,1),(1,0),(1,1) receives the gradient (PyTorch's index rule)
__device__ __forceinline__ void pool_route(float a, float b, float c, float d, float g, float& ga, float& gb,
                                           float& gc, float& gd) {
  int k = 0;
  float m = a;
  if (b > m) { m = b; k = 1; }
  if (c > m) { m = c; k = 2; }
  if (d > m) { m = d; k = 3; }
  ga = k == 0 ? g : 0.f; gb = k == 1 ? g : 0.f; gc = k == 2 ? g : 0.f; gd = k == 3 ? g : 0.f;
}

__global__ void maxpool2_bwd_kernel(const act_t* __restrict__ x, int ld_x, const act_t* __restrict__ dy, int ld_dy,
                                    act_t* __restrict__ dx, int ld_dx, int C, int N, int H, int W, int accumulate) {
  const int Ho = H >> 1, Wo = W >> 1, c4n = C >> 2;
  const int e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e < Wo * c4n) {
    const int xo = e / c4n, cq = e - xo * c4n;
    const int row = xcd_band_row(blockIdx.x, gridDim.x);
    const int n = row / Ho, yo = row - n * Ho;
    const size_t po = (size_t)row * Wo + xo;
    const size_t pi = ((size_t)n * H + 2 * yo) * W + 2 * xo;
    const size_t o0 = pi, o1 = pi + 1, o2 = pi + W, o3 = pi + W + 1;
    float4 v4[4];
    v4[0] = act_ld4f(x + o0 * ld_x + cq * 4);
    v4[1] = act_ld4f(x + o1 * ld_x + cq * 4);
    v4[2] = act_ld4f(x + o2 * ld_x + cq * 4);
    v4[3] = act_ld4f(x + o3 * ld_x + cq * 4);
    const float4 g = act_ld4f(dy + (size_t)po * ld_dy + cq * 4);
    const float4 a = v4[0], b = v4[1], c = v4[2], d = v4[3];
    float4 ga, gb, gc, gd;
    pool_route(a.x, b.x, c.x, d.x, g.x, ga.x, gb.x, gc.x, gd.x);
    pool_route(a.y, b.y, c.y, d.y, g.y, ga.y, gb.y, gc.y, gd.y);
    pool_route(a.z, b.z, c.z, d.z, g.z, ga.z, gb.z, gc.z, gd.z);
    pool_route(a.w, b.w, c.w, d.w, g.w, ga.w, gb.w, gc.w, gd.w);
    act_t* pa = dx + o0 * ld_dx + cq * 4;
    act_t* pb = dx + o1 * ld_dx + cq * 4;
    act_t* pc = dx + o2 * ld_dx + cq * 4;
    act_t* pd = dx + o3 * ld_dx + cq * 4;
    if (accumulate) {
      float4 t;
      t = act_ld4f(pa); ga.x += t.x; ga.y += t.y; ga.z += t.z; ga.w += t.w;
      t = act_ld4f(pb); gb.x += t.x; gb.y += t.y; gb.z += t.z; gb.w += t.w;
      t = act_ld4f(pc); gc.x += t.x; gc.y += t.y; gc.z += t.z; gc.w += t.w;
      t = act_ld4f(pd); gd.x += t.x; gd.y += t.y; gd.z += t.z; gd.w += t.w;
    }
    act_st4f(pa, ga); act_st4f(pb, gb); act_st4f(pc, gc); act_st4f(pd, gd);
  }
}

static int sp_check(const void* a, const void* b, int C, int lda, int ldb) {
  PP_CHECK_ARG(a && b, "spatial: null pointer");
  PP_CHECK_ARG(C > 0 && C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && lda >= C && ldb >= C, "spatial: C=%d ld=%d/%d", C, lda, ldb);
  PP_CHECK_ARG(((uintptr_t)a & PP_ACT_ALIGN) == 0 && ((uintptr_t)b & PP_ACT_ALIGN) == 0, "spatial: tensors must be 16-byte aligned");
  return 0;
}

static int sp_lazy(const pp_lazy_in* in, int C, int N, PpLazy& lz) {
  lz = pp_lazy_none();
  if (!in || !in->coef) return 0;
  PP_CHECK_ARG(in->groups >= 1 && N % in->groups == 0 && in->ld % 4 == 0 && in->ld >= C && ((uintptr_t)in->coef & 15) == 0,
               "spatial: bad lazy-input descriptor (groups=%d ld=%d)", in->groups, in->ld);
  lz = PpLazy{in->coef, in->ld, N / in->groups};
  return 0;
}

static int maxpool2_fwd_impl(const pp_act* x, int ld_x, pp_act* y, int ld_y, int C, int N, int H, int W, hipStream_t s) {
  if (int rc = sp_check(x, y, C, ld_x, ld_y)) return rc;
  PP_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "maxpool2: H and W must be even (H=%d W=%d)", H, W);
  pp_prof_begin(PP_K_SPATIAL, 0.0, 5.0 * N * (double)H * W * C, s);
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(N * (H / 2), pp_cdiv((W / 2) * (C / 4), SP_THREADS)), dim3(SP_THREADS), 0, s, x, ld_x, y, ld_y, C, N, H, W);
  pp_prof_end(s);
  return pp_launch_status("maxpool2_fwd");
}

extern "C" int PP_FN(pp_maxpool2_fwd)(const pp_act* x, int ld_x, pp_act* y, int ld_y, int C, int N, int H, int W,
                               void* stream) {
  return maxpool2_fwd_impl(x, ld_x, y, ld_y, C, N, H, W, (hipStream_t)stream);
}

static int maxpool2_bwd_impl(const pp_act* x, int ld_x, const pp_act* dy, int ld_dy, pp_act* dx, int ld_dx, int C,
                             int N, int H, int W, int accumulate, hipStream_t s) {
  if (int rc = sp_check(x, dy, C, ld_x, ld_dy)) return rc;
  if (int rc = sp_check(x, dx, C, ld_x, ld_dx)) return rc;
  PP_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "maxpool2: H and W must be even (H=%d W=%d)", H, W);
  pp_prof_begin(PP_K_SPATIAL, 0.0, 9.0 * N * (double)H * W * C, s);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(N * (H / 2), pp_cdiv((W / 2) * (C / 4), SP_THREADS)), dim3(SP_THREADS), 0, s, x, ld_x, dy, ld_dy, dx, ld_dx,
                     C, N, H, W, accumulate);
  pp_prof_end(s);
  return pp_launch_status("maxpool2_bwd");
}

extern "C" int PP_FN(pp_maxpool2_bwd)(const pp_act* x, int ld_x, const pp_act* dy, int ld_dy, pp_act* dx, int ld_dx, int C,
                               int N, int H, int W, int accumulate, void* stream) {
  return maxpool2_bwd_impl(x, ld_x, dy, ld_dy, dx, ld_dx, C, N, H, W, accumulate, (hipStream_t)stream);
}

// ---------------------------------------------------------------- bilinear, align_corners=True
// src = dst * (in-1)/(out-1);  i0 = floor(src);  i1 = min(i0+1, in-1);  l1 = src - i0;  l0 = 1 - l1
__device__ __forceinline__ void lin_coeff(int o, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
  const float src = scale * (float)o;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}
static inline float lin_scale(int in_size, int out_size) {
  return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}

__global__ void bilinear_fwd_kernel(const act_t* __restrict__ x, int ld_x, act_t* __restrict__ y, int ld_y, int C,
                                    int N, int Hi, int Wi, int Ho, int Wo, float sy, float sx) {
  const int c4n = C >> 2;
  const int e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e < Wo * c4n) {
    const int xo = e / c4n, cq = e - xo * c4n;
    const int row = xcd_band_row(blockIdx.x, gridDim.x);
    const int n = row / Ho, yo = row - n * Ho;
    const size_t po = (size_t)row * Wo + xo;
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    lin_coeff(yo, sy, Hi, y0, y1, wy0, wy1);
    lin_coeff(xo, sx, Wi, x0, x1, wx0, wx1);
    const act_t* base = x + (size_t)n * Hi * Wi * ld_x + cq * 4;
    float4 v4[4];
    v4[0] = act_ld4f(base + ((size_t)y0 * Wi + x0) * ld_x);
    v4[1] = act_ld4f(base + ((size_t)y0 * Wi + x1) * ld_x);
    v4[2] = act_ld4f(base + ((size_t)y1 * Wi + x0) * ld_x);
    v4[3] = act_ld4f(base + ((size_t)y1 * Wi + x1) * ld_x);
    const float4 a = v4[0], b = v4[1], c = v4[2], d = v4[3];
    // the roundings are spelled out (product, fma; product, fma; product, fma): every form of this kernel must interpolate bit
    // for bit alike -- a last-bit difference in an activation flips LeakyReLU branches downstream
#define SP_LERP(A, B, C_, D) __builtin_fmaf(wy1, __builtin_fmaf(wx1, D, wx0 * C_), wy0 * __builtin_fmaf(wx1, B, wx0 * A))
    float4 o;
    o.x = SP_LERP(a.x, b.x, c.x, d.x);
    o.y = SP_LERP(a.y, b.y, c.y, d.y);
    o.z = SP_LERP(a.z, b.z, c.z, d.z);
    o.w = SP_LERP(a.w, b.w, c.w, d.w);
#undef SP_LERP
    act_st4f(y + (size_t)po * ld_y + cq * 4, o);
  }
}

// Adjoint of the forward gather, with the forward's OWN float arithmetic deciding which outputs touch an input (lin_coeff
// again, so that weights agree bit for bit): output o reads inputs i0(o) = floor(scale * o) and i0 + 1, i0 is monotone in
// o, so the outputs touching input i are the contiguous run with i0(o) in {i - 1, i}; first_touch() finds its start.
__device__ __forceinline__ int first_touch(int i, float scale, int in_size, int out_size) {
  if (scale <= 0.f) return 0;
  int o = (int)floorf((float)(i - 1) / scale) - 2;
  if (o < 0) o = 0;
  for (; o < out_size; ++o) {
    int i0, i1; float l0, l1;
    lin_coeff(o, scale, in_size, i0, i1, l0, l1);
    if (i0 >= i - 1) break;
  }
  return o;
}
// weight with which output o feeds input i (0 when it does not touch it or lies outside the image)
__device__ __forceinline__ float touch_weight(int o, int i, float scale, int in_size, int out_size) {
  if (o >= out_size) return 0.f;
  int i0, i1; float l0, l1;
  lin_coeff(o, scale, in_size, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

// Up-sampling factors up to 2 (every decoder stage of the network): at most BT = 5 consecutive outputs per dimension touch one
// input (the run spans 2 / scale <= 4.1 output positions).  (Round 3's kernel for this case loaded all 25 taps of a result first --
// 1.33 TB/s of its algorithmic bytes, bound by the load path; the column walker below replaced it in round 5 and it was removed in
// round 6 together with its switch.)
#define BT 5
// Round 5: COLUMN WALKER.  The round-3 taps kernel issued 25 16-byte loads per result (1.33 TB/s of its algorithmic bytes:
// bound by the load path, not by HBM -- PMC traffic is within 1.15x of algorithmic).  Here a thread owns XP adjacent input
// pixels of one channel quad and walks DOWN a band of RB input rows: per output row it loads the NT = 2 XP + 3 taps of its x-run
// once, reduces them along x (t = sum_k wx[k] g[k]) and adds wy0 * t / wy1 * t to the two input rows the output row touches.
// i0(yo) is monotone, so two running accumulators (row cur, row cur + 1) suffice and a finished row is stored as soon as the walk
// leaves it -- all control flow is uniform over the block (every thread of a block shares the image and the row band).
// Loads per result: (2 RB + 3) / RB * NT / XP = 7.7 for XP = 2, RB = 16 (25 before); the next output row's taps are requested
// before the current row is reduced.  Same taps and weights as the forward (lin_coeff), fixed order: bit-reproducible.
template <int XP>
__global__ __launch_bounds__(SP_THREADS) void bilinear_bwd_col_kernel(const act_t* __restrict__ dy, int ld_dy, act_t* __restrict__ dx,
                                                                      int ld_dx, int C, int N, int Hi, int Wi, int Ho, int Wo, float sy,
                                                                      float sx, int accumulate, int RB, int bands) {
  constexpr int NT = 2 * XP + 3;
  const int c4n = C >> 2;
  const int nxq = (Wi + XP - 1) / XP;
  const int e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e >= nxq * c4n) return;
  const int xq = e / c4n, cq = e - xq * c4n;
  const int xi0 = xq * XP;
  const int blk = xcd_band_row(blockIdx.x, gridDim.x);
  const int n = blk / bands, band = blk - n * bands;
  const int yi0 = band * RB, yi1 = min(yi0 + RB, Hi) - 1;
  // x-run of this thread and the weight of every tap for each of its XP pixels
  const int xa = first_touch(xi0, sx, Wi, Wo);
  int ox[NT];
  float wx[XP][NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    ox[k] = min(xa + k, Wo - 1);
#pragma unroll
    for (int p = 0; p < XP; ++p) wx[p][k] = (xi0 + p < Wi) ? touch_weight(xa + k, xi0 + p, sx, Wi, Wo) : 0.f;
  }
  const act_t* base = dy + (size_t)n * Ho * Wo * ld_dy + cq * 4;
  act_t* obase = dx + (size_t)n * Hi * Wi * ld_dx + cq * 4;
  float4 a0[XP], a1[XP];
#pragma unroll
  for (int p = 0; p < XP; ++p) a0[p] = a1[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  int cur = yi0 - 1;
  auto flush = [&]() {                 // row `cur` is complete: store it (rows outside the band are scratch), shift the accumulators
    if (cur >= yi0) {
#pragma unroll
      for (int p = 0; p < XP; ++p)
        if (xi0 + p < Wi) {
          act_t* o = obase + ((size_t)cur * Wi + xi0 + p) * ld_dx;
          float4 v = a0[p];
          if (accumulate) { const float4 t = act_ld4f(o); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
          act_st4f(o, v);
        }
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) { a0[p] = a1[p]; a1[p] = make_float4(0.f, 0.f, 0.f, 0.f); }
    ++cur;
  };
  int yo = first_touch(yi0, sy, Hi, Ho);
  int i0, i1; float l0, l1;
  bool live = false;
  if (yo < Ho) { lin_coeff(yo, sy, Hi, i0, i1, l0, l1); live = i0 <= yi1; }
  float4 g[NT];
  if (live) {
#pragma unroll
    for (int k = 0; k < NT; ++k) g[k] = act_ld4f(base + ((size_t)yo * Wo + ox[k]) * ld_dy);
  }
  while (live) {
    // request the next output row before this one is reduced
    int n0 = 0, n1 = 0; float m0 = 0.f, m1 = 0.f;
    bool more = yo + 1 < Ho;
    if (more) { lin_coeff(yo + 1, sy, Hi, n0, n1, m0, m1); more = n0 <= yi1; }
    float4 gn[NT];
    if (more) {
#pragma unroll
      for (int k = 0; k < NT; ++k) gn[k] = act_ld4f(base + ((size_t)(yo + 1) * Wo + ox[k]) * ld_dy);
    }
    while (cur < i0) flush();
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const float w = wx[p][k];
        if (w != 0.f) { t.x += w * g[k].x; t.y += w * g[k].y; t.z += w * g[k].z; t.w += w * g[k].w; }   // (a zero weight must not touch the tap: 0 * inf)
      }
      a0[p].x += l0 * t.x; a0[p].y += l0 * t.y; a0[p].z += l0 * t.z; a0[p].w += l0 * t.w;
      if (i1 == i0) { a0[p].x += l1 * t.x; a0[p].y += l1 * t.y; a0[p].z += l1 * t.z; a0[p].w += l1 * t.w; }
      else          { a1[p].x += l1 * t.x; a1[p].y += l1 * t.y; a1[p].z += l1 * t.z; a1[p].w += l1 * t.w; }
    }
    live = more;
    if (more) {
      ++yo; i0 = n0; i1 = n1; l0 = m0; l1 = m1;
#pragma unroll
      for (int k = 0; k < NT; ++k) g[k] = gn[k];
    }
  }
  while (cur <= yi1) flush();
}

// (Round 4 tried a 2 x 2 block of input pixels per thread -- the union of their taps walked row by row, 12.25 loads per result
// instead of 25: 550 us against 416 us on the 64-channel 256 -> 128 launch.  Seven loads in flight and a quarter of the threads
// lose more than the halved L1 traffic wins; profiles/r04_experiments/bilinear_bwd_block2.log.  Removed again.)
// output rows whose taps can touch input row `i`: every o with floor(scale*o) in {i-1, i}
__device__ __forceinline__ void touch_range(int i, float scale, int out_size, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out_size - 1; return; }
  lo = (int)floorf((float)(i - 1) / scale) - 1;
  hi = (int)ceilf((float)(i + 1) / scale) + 1;
  if (lo < 0) lo = 0;
  if (hi > out_size - 1) hi = out_size - 1;
}

// any scale: walks a conservative window
__global__ void bilinear_bwd_kernel(const act_t* __restrict__ dy, int ld_dy, act_t* __restrict__ dx, int ld_dx, int C,
                                    int N, int Hi, int Wi, int Ho, int Wo, float sy, float sx, int accumulate) {
  const int c4n = C >> 2;
  const int e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e < Wi * c4n) {
    const int xi = e / c4n, cq = e - xi * c4n;
    const int row = xcd_band_row(blockIdx.x, gridDim.x);
    const int n = row / Hi, yi = row - n * Hi;
    const size_t pi = (size_t)row * Wi + xi;
    int ylo, yhi, xlo, xhi;
    touch_range(yi, sy, Ho, ylo, yhi);
    touch_range(xi, sx, Wo, xlo, xhi);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const act_t* base = dy + (size_t)n * Ho * Wo * ld_dy + cq * 4;
    for (int yo = ylo; yo <= yhi; ++yo) {
      int y0, y1; float wy0, wy1;
      lin_coeff(yo, sy, Hi, y0, y1, wy0, wy1);
      const float wy = (y0 == yi ? wy0 : 0.f) + (y1 == yi ? wy1 : 0.f);
      if (wy == 0.f) continue;
      for (int xo = xlo; xo <= xhi; ++xo) {
        int x0, x1; float wx0, wx1;
        lin_coeff(xo, sx, Wi, x0, x1, wx0, wx1);
        const float wx = (x0 == xi ? wx0 : 0.f) + (x1 == xi ? wx1 : 0.f);
        if (wx == 0.f) continue;
        const float4 g = act_ld4f(base + ((size_t)yo * Wo + xo) * ld_dy);
        const float w = wy * wx;
        acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
      }
    }
    act_t* o = dx + (size_t)pi * ld_dx + cq * 4;
    if (accumulate) { const float4 t = act_ld4f(o); acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
    act_st4f(o, acc);
  }
}

// (Round 5 tried a column walker here as well -- a thread keeps the x-interpolated values of two input rows and walks down a band
// of output rows, 0.9 loads per store instead of 4, bit-identical results: 481 us against the gather kernel's 400 us on the
// 64-channel 128 -> 256 launch (profiles/r05_experiments/bilinear_fwd_col_walker_microbench.log).  This kernel writes four bytes
// for every byte it reads, and gfx950's single in-order vmcnt makes every wait for a prefetched row wait for the stores issued
// in front of it: the walker serialises on store latency where the gather kernel simply has 64x more independent threads.
// Removed again; the BACKWARD walker below reads four bytes per byte it writes and gains 1.45x.)
static int bilinear_fwd_impl(const pp_act* x, int ld_x, pp_act* y, int ld_y, int C, int N, int Hi, int Wi, int Ho,
                             int Wo, hipStream_t s) {
  if (int rc = sp_check(x, y, C, ld_x, ld_y)) return rc;
  PP_CHECK_ARG(Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "bilinear: bad sizes");
  pp_prof_begin(PP_K_SPATIAL, 0.0, 4.0 * N * C * ((double)Hi * Wi + (double)Ho * Wo), s);
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(N * Ho, pp_cdiv(Wo * (C / 4), SP_THREADS)), dim3(SP_THREADS), 0, s, x, ld_x, y, ld_y, C, N, Hi, Wi,
                     Ho, Wo, lin_scale(Hi, Ho), lin_scale(Wi, Wo));
  pp_prof_end(s);
  return pp_launch_status("bilinear_fwd");
}

extern "C" int PP_FN(pp_bilinear_fwd)(const pp_act* x, int ld_x, pp_act* y, int ld_y, int C, int N, int Hi, int Wi, int Ho,
                               int Wo, void* stream) {
  return bilinear_fwd_impl(x, ld_x, y, ld_y, C, N, Hi, Wi, Ho, Wo, (hipStream_t)stream);
}

extern "C" int PP_FN(pp_bilinear_bwd)(const pp_act* dy, int ld_dy, pp_act* dx, int ld_dx, int C, int N, int Hi, int Wi, int Ho,
                               int Wo, int accumulate, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = sp_check(dy, dx, C, ld_dy, ld_dx)) return rc;
  PP_CHECK_ARG(Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "bilinear: bad sizes");
  pp_prof_begin(PP_K_SPATIAL, 0.0, 4.0 * N * C * ((double)Hi * Wi + (double)Ho * Wo), s);
  const float sy = lin_scale(Hi, Ho), sx = lin_scale(Wi, Wo);
  // the run of outputs touching one input spans 2 / scale positions: <= 4.1 -> at most BT = 5 of them
  const bool taps = sy > 0.f && sx > 0.f && 2.f / sy <= 4.1f && 2.f / sx <= 4.1f;
  if (taps) {                      // column walker, two input pixels per thread (one measured 4 % slower, r05)
    const int RB = Hi >= 128 ? 16 : (Hi >= 32 ? 8 : Hi);
    const int bands = pp_cdiv(Hi, RB);
    const dim3 grid(N * bands, pp_cdiv(pp_cdiv(Wi, 2) * (C / 4), SP_THREADS));
    hipLaunchKernelGGL(bilinear_bwd_col_kernel<2>, grid, dim3(SP_THREADS), 0, s, dy, ld_dy, dx, ld_dx, C, N, Hi, Wi, Ho, Wo, sy, sx, accumulate, RB, bands);
  }
  else
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(N * Hi, pp_cdiv(Wi * (C / 4), SP_THREADS)), dim3(SP_THREADS), 0, s, dy, ld_dy, dx, ld_dx, C, N, Hi,
                       Wi, Ho, Wo, sy, sx, accumulate);
  pp_prof_end(s);
  return pp_launch_status("bilinear_bwd");
}

// ---------------------------------------------------------------- copy a channel slab (used for scale_factor=1 "upsample")
__global__ void copy_slab_kernel(const act_t* __restrict__ x, int ld_x, act_t* __restrict__ y, int ld_y, int C,
                                 long long P, int accumulate) {
  const int c4n = C >> 2;
  const long long total = P * c4n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4n);
    const long long p = i / c4n;
    float4 v = act_ld4f(x + (size_t)p * ld_x + cq * 4);
    act_t* o = y + (size_t)p * ld_y + cq * 4;
    if (accumulate) { const float4 t = act_ld4f(o); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    act_st4f(o, v);
  }
}

extern "C" int PP_FN(pp_copy_slab)(const pp_act* x, int ld_x, pp_act* y, int ld_y, int C, long long P, int accumulate,
                            void* stream) {
  if (int rc = sp_check(x, y, C, ld_x, ld_y)) return rc;
  hipLaunchKernelGGL(copy_slab_kernel, dim3(sp_blocks(P * (C / 4))), dim3(SP_THREADS), 0, (hipStream_t)stream, x, ld_x,
                     y, ld_y, C, P, accumulate);
  return pp_launch_status("copy_slab");
}

// y[n][p][c] (+)= x[n][p][c] * scale[n][c]: nn.Dropout2d (whole channels of a sample dropped, survivors scaled by
// 1/(1-p); models/aux_path_memory.py:22,31) and its backward -- the mask (0 or 1/(1-p)) is drawn by the caller and kept.
__global__ void channel_scale_kernel(const act_t* __restrict__ x, int ld_x, act_t* __restrict__ y, int ld_y,
                                     const float* __restrict__ scale, int C, int N, int HW, int accumulate) {
  const int c4n = C >> 2;
  const long long total = (long long)N * HW * c4n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4n);
    const long long p = i / c4n;
    const int n = (int)(p / HW);
    const float4 m = *reinterpret_cast<const float4*>(scale + (size_t)n * C + cq * 4);
    float4 v = act_ld4f(x + (size_t)p * ld_x + cq * 4);
    v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
    act_t* o = y + (size_t)p * ld_y + cq * 4;
    if (accumulate) { const float4 t = act_ld4f(o); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    act_st4f(o, v);
  }
}

extern "C" int PP_FN(pp_channel_scale)(const pp_act* x, int ld_x, pp_act* y, int ld_y, const float* scale, int C, int N, int HW,
                                int accumulate, void* stream) {
  if (int rc = sp_check(x, y, C, ld_x, ld_y)) return rc;
  PP_CHECK_ARG(scale && N > 0 && HW > 0 && ((uintptr_t)scale & 15) == 0, "channel_scale: bad scale / shape");
  hipLaunchKernelGGL(channel_scale_kernel, dim3(sp_blocks((long long)N * HW * (C / 4))), dim3(SP_THREADS), 0,
                     (hipStream_t)stream, x, ld_x, y, ld_y, scale, C, N, HW, accumulate);
  return pp_launch_status("channel_scale");
}

// ---------------------------------------------------------------- 1x1 head: NHWC features -> NCHW logits
#define HEAD_MAXK 8
#define HEAD_MAXC 128
// A block stages a tile of TP pixels x C channels in LDS with coalesced float4 loads (a pixel's channels are
// contiguous), then every thread reduces one pixel's row against the K weight rows (row stride C + 1 floats: the
// threads of a wave walk different banks) and writes its K logits to the NCHW planes (coalesced across pixels).
// The first version had each thread stream its own pixel straight from global memory: lanes 128 B apart, 2.7x the
// algorithmic HBM traffic (r01 PMC profile).
template <int TP, bool LAZY>
__global__ __launch_bounds__(TP) void conv1x1_fwd_kernel(const act_t* __restrict__ x, int ld_x, int C,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ logits, int K, int N, int HW, PpLazy lz) {
  extern __shared__ float sm[];
  float* ws = sm;                          // [K][C]
  float* bs = ws + K * C;                  // [K]
  float* xs = bs + HEAD_MAXK;              // [TP][C + 1]
  for (int i = threadIdx.x; i < K * C; i += TP) ws[i] = w[i];
  if (threadIdx.x < K) bs[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
  const long long P = (long long)N * HW;
  const int c4n = C >> 2, ldx = C + 1;
  for (long long p0 = (long long)blockIdx.x * TP; p0 < P; p0 += (long long)gridDim.x * TP) {
    __syncthreads();
    for (int i = threadIdx.x; i < TP * c4n; i += TP) {
      const int pp = i / c4n, cq = i - pp * c4n;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p0 + pp < P) {
        v = act_ld4f(x + (size_t)(p0 + pp) * ld_x + cq * 4);
        float4 v1[1] = {v};
        SP_LAZY4(v1, (int)((p0 + pp) / HW))
        v = v1[0];
      }
      float* d = xs + pp * ldx + cq * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const long long p = p0 + threadIdx.x;
    if (p < P) {
      float acc[HEAD_MAXK];
#pragma unroll
      for (int k = 0; k < HEAD_MAXK; ++k) acc[k] = 0.f;
      const float* xp = xs + threadIdx.x * ldx;
      for (int c = 0; c < C; ++c) {
        const float v = xp[c];
#pragma unroll
        for (int k = 0; k < HEAD_MAXK; ++k)
          if (k < K) acc[k] += v * ws[k * C + c];
      }
      const int n = (int)(p / HW), hw = (int)(p % HW);
#pragma unroll
      for (int k = 0; k < HEAD_MAXK; ++k)
        if (k < K) logits[((size_t)n * K + k) * HW + hw] = acc[k] + bs[k];
    }
  }
}

// Streaming form (round 3) for C / 4 a power of two >= K: thread -> (pixel lane, channel quad); one coalesced float4 load
// per pixel, the K partial dot products are summed over the quad lanes of the pixel by a butterfly, and lane `quad == k`
// stores class k (a wave stores all K planes of its 64 / c4n pixels with one instruction).  No LDS tile, no barrier.
// The LDS-tiled kernel above ran the 32 -> 5 head at 256^2 x 64 images at 2.4 TB/s of its 0.62 GB.
template <bool LAZY>
__global__ __launch_bounds__(SP_THREADS) void conv1x1_fwd_stream_kernel(const act_t* __restrict__ x, int ld_x, int C,
                                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                                        float* __restrict__ logits, int K, int N, int HW,
                                                                        int pix_per_block, PpLazy lz) {
  const int c4n = C >> 2, ppl = SP_THREADS / c4n;
  const int cq = threadIdx.x % c4n, pl = threadIdx.x / c4n;
  const int P = N * HW;
  const int p_lo = blockIdx.x * pix_per_block;
  const int p_hi = min(P, p_lo + pix_per_block);
  float wr[HEAD_MAXK][4];
#pragma unroll
  for (int k = 0; k < HEAD_MAXK; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[k][j] = k < K ? w[k * C + cq * 4 + j] : 0.f;
  const float bv = (bias && cq < K) ? bias[cq] : 0.f;
  auto pixel = [&](int p, float* out, int n_img) {           // out = logits + (n K HW + hw) of pixel p
    float4 xv1[1] = {act_ld4f(x + (size_t)p * ld_x + cq * 4)};
    SP_LAZY4(xv1, n_img)
    const float4 xv = xv1[0];
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < HEAD_MAXK; ++k)
      if (k < K) {
        float t = xv.x * wr[k][0] + xv.y * wr[k][1] + xv.z * wr[k][2] + xv.w * wr[k][3];
        t = pp_group_sum(t, c4n);
        if (cq == k) mine = t;
      }
    if (cq < K) out[(size_t)cq * HW] = mine + bv;
  };
  int p = p_lo + pl;
  int n = p / HW, hw = p - n * HW;
  if (HW % pix_per_block == 0) {
    float* out = logits + (size_t)n * K * HW + hw;
#pragma unroll 4
    for (; p < p_hi; p += ppl, out += ppl) pixel(p, out, n);
  } else {
    // every lane of a pixel's quad group must run the butterfly: the trip count is uniform per group (same p)
    for (; p < p_hi; p += ppl) {
      pixel(p, logits + (size_t)n * K * HW + hw, n);
      hw += ppl;
      while (hw >= HW) { hw -= HW; ++n; }
    }
  }
}

static int conv1x1_fwd_impl(const pp_act* x, int ld_x, int C, const float* w, const float* bias,
                            float* logits, int K, int N, int HW, PpLazy lz, hipStream_t s) {
  PP_CHECK_ARG(x && w && logits, "conv1x1_fwd: null pointer");
  PP_CHECK_ARG(K >= 1 && K <= HEAD_MAXK && C % 4 == 0 && C <= HEAD_MAXC && ld_x % 4 == 0 && ld_x >= C,
               "conv1x1_fwd: K=%d (<=8) C=%d (<=128, %%4) ld=%d", K, C, ld_x);
  PP_CHECK_ARG(((uintptr_t)x & PP_ACT_ALIGN) == 0, "conv1x1_fwd: x must be 16-byte aligned");
  const long long P = (long long)N * HW;
  pp_prof_begin(PP_K_SPATIAL, 2.0 * P * K * C, 4.0 * P * (C + K), s);
  const int tp = C > 64 ? 128 : 256;
  const size_t lds = (size_t)(K * C + HEAD_MAXK + tp * (C + 1)) * sizeof(float);
  int blocks = pp_cdiv(P, tp);
  if (blocks > SP_MAX_BLOCKS) blocks = SP_MAX_BLOCKS;
  {   // once per (kernel, device): pp_max_lds
    pp_max_lds(reinterpret_cast<const void*>(conv1x1_fwd_kernel<128, false>), (int)((HEAD_MAXK * HEAD_MAXC + HEAD_MAXK + 128 * (HEAD_MAXC + 1)) * sizeof(float)));
    pp_max_lds(reinterpret_cast<const void*>(conv1x1_fwd_kernel<256, false>), (int)((HEAD_MAXK * 64 + HEAD_MAXK + 256 * 65) * sizeof(float)));
    pp_max_lds(reinterpret_cast<const void*>(conv1x1_fwd_kernel<128, true>), (int)((HEAD_MAXK * HEAD_MAXC + HEAD_MAXK + 128 * (HEAD_MAXC + 1)) * sizeof(float)));
    pp_max_lds(reinterpret_cast<const void*>(conv1x1_fwd_kernel<256, true>), (int)((HEAD_MAXK * 64 + HEAD_MAXK + 256 * 65) * sizeof(float)));
  }
  const int c4n = C / 4;
  if ((c4n & (c4n - 1)) == 0 && c4n >= K && P < 0x7fffffffLL) {
    int ppb = (int)pp_cdiv(P, 2048);                                   // <= 2048 blocks, whole pixel-lane groups per block
    if (ppb < 1024) ppb = 1024;
    ppb = pp_cdiv(ppb, SP_THREADS) * SP_THREADS;
    if (lz.coef) hipLaunchKernelGGL(conv1x1_fwd_stream_kernel<true>, dim3(pp_cdiv(P, ppb)), dim3(SP_THREADS), 0, s, x, ld_x, C, w, bias, logits, K, N, HW, ppb, lz);
    else hipLaunchKernelGGL(conv1x1_fwd_stream_kernel<false>, dim3(pp_cdiv(P, ppb)), dim3(SP_THREADS), 0, s, x, ld_x, C, w, bias, logits, K, N, HW, ppb, lz);
  } else if (tp == 128)
  {
    if (lz.coef) hipLaunchKernelGGL((conv1x1_fwd_kernel<128, true>), dim3(blocks), dim3(128), lds, s, x, ld_x, C, w, bias, logits, K, N, HW, lz);
    else hipLaunchKernelGGL((conv1x1_fwd_kernel<128, false>), dim3(blocks), dim3(128), lds, s, x, ld_x, C, w, bias, logits, K, N, HW, lz);
  }
  else
  {
    if (lz.coef) hipLaunchKernelGGL((conv1x1_fwd_kernel<256, true>), dim3(blocks), dim3(256), lds, s, x, ld_x, C, w, bias, logits, K, N, HW, lz);
    else hipLaunchKernelGGL((conv1x1_fwd_kernel<256, false>), dim3(blocks), dim3(256), lds, s, x, ld_x, C, w, bias, logits, K, N, HW, lz);
  }
  pp_prof_end(s);
  return pp_launch_status("conv1x1_fwd");
}

extern "C" int PP_FN(pp_conv1x1_nhwc_to_nchw_fwd)(const pp_act* x, int ld_x, int C, const float* w, const float* bias,
                                           float* logits, int K, int N, int HW, void* stream) {
  return conv1x1_fwd_impl(x, ld_x, C, w, bias, logits, K, N, HW, pp_lazy_none(), (hipStream_t)stream);
}

extern "C" int PP_FN(pp_conv1x1_nhwc_to_nchw_fwd_lazy)(const pp_act* x, int ld_x, int C, const float* w, const float* bias,
                                                float* logits, int K, int N, int HW, const pp_lazy_in* lazy_x, void* stream) {
  PpLazy lz;
  if (int rc = sp_lazy(lazy_x, C, N, lz)) return rc;
  return conv1x1_fwd_impl(x, ld_x, C, w, bias, logits, K, N, HW, lz, (hipStream_t)stream);
}

// backward: dx[p][c] = sum_k dl[k][p] * w[k][c];  dw[k][c] = sum_p dl[k][p] * x[p][c];  db[k] = sum_p dl[k][p]
// dw/db: per-block partial sums (each thread owns a set of (k,c) outputs and walks the block's pixel range
// through LDS tiles), then a fixed-order finalize.
#define HEAD_TP 64       // pixels per LDS tile
template <bool LAZY>
__global__ __launch_bounds__(SP_THREADS) void conv1x1_bwd_kernel(const float* __restrict__ dl, const act_t* __restrict__ x,
                                                                 int ld_x, int C, const float* __restrict__ w,
                                                                 act_t* __restrict__ dx, int ld_dx, int K, int N, int HW,
                                                                 int pix_per_block, int accumulate_dx,
                                                                 float* __restrict__ partial /*[blocks][K*(C+1)]*/, PpLazy lz) {
  __shared__ float ws[HEAD_MAXK * HEAD_MAXC];
  __shared__ float xs[HEAD_TP * (HEAD_MAXC + 1)];
  __shared__ float ds[HEAD_TP * HEAD_MAXK];
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) ws[i] = w[i];
  const long long P = (long long)N * HW;
  const long long p_lo = (long long)blockIdx.x * pix_per_block;
  long long p_hi = p_lo + pix_per_block;
  if (p_hi > P) p_hi = P;
  const int nout = K * (C + 1);              // (k, c) pairs plus the bias column c == C
  // each thread accumulates up to ceil(nout / 256) outputs
  constexpr int NACC = (HEAD_MAXK * (HEAD_MAXC + 1) + SP_THREADS - 1) / SP_THREADS;
  double acc[NACC];      // across-tile accumulation in double: the bias gradient is a sum of N*H*W terms that largely cancel
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = 0.0;
  const int c4n = C >> 2;
  __syncthreads();
  for (long long pt = p_lo; pt < p_hi; pt += HEAD_TP) {
    const int np = (int)((p_hi - pt) < HEAD_TP ? (p_hi - pt) : HEAD_TP);
    // stage dl tile [np][K] and x tile [np][C]
    for (int i = threadIdx.x; i < HEAD_TP * K; i += blockDim.x) {
      const int pp = i % HEAD_TP, k = i / HEAD_TP;
      float v = 0.f;
      if (pp < np) {
        const long long p = pt + pp;
        const int n = (int)(p / HW), hw = (int)(p % HW);
        v = dl[((size_t)n * K + k) * HW + hw];
      }
      ds[pp * HEAD_MAXK + k] = v;
    }
    for (int i = threadIdx.x; i < HEAD_TP * c4n; i += blockDim.x) {
      const int cq = i % c4n, pp = i / c4n;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pp < np) {
        float4 v1[1] = {act_ld4f(x + (size_t)(pt + pp) * ld_x + cq * 4)};
        SP_LAZY4(v1, (int)((pt + pp) / HW))
        v = v1[0];
      }
      float* d = xs + pp * (HEAD_MAXC + 1) + cq * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    // dx for this tile: thread -> (pixel, channel quad)
    if (dx) {
      for (int i = threadIdx.x; i < np * c4n; i += blockDim.x) {
        const int cq = i % c4n, pp = i / c4n;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < K; ++k) {
          const float g = ds[pp * HEAD_MAXK + k];
          const float* wk = ws + k * C + cq * 4;
          o.x += g * wk[0]; o.y += g * wk[1]; o.z += g * wk[2]; o.w += g * wk[3];
        }
        act_t* po = dx + (size_t)(pt + pp) * ld_dx + cq * 4;
        if (accumulate_dx) { const float4 t = act_ld4f(po); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
        act_st4f(po, o);
      }
    }
    // dw / db partials
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      const int o = threadIdx.x + j * SP_THREADS;
      if (o < nout) {
        const int k = o / (C + 1), c = o % (C + 1);
        float s = 0.f;
        if (c < C) for (int pp = 0; pp < np; ++pp) s += ds[pp * HEAD_MAXK + k] * xs[pp * (HEAD_MAXC + 1) + c];
        else       for (int pp = 0; pp < np; ++pp) s += ds[pp * HEAD_MAXK + k];
        acc[j] += (double)s;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NACC; ++j) {
    const int o = threadIdx.x + j * SP_THREADS;
    if (o < nout) partial[(size_t)blockIdx.x * nout + o] = (float)acc[j];
  }
}

// Streaming form (round 3) for C / 4 a power of two: thread -> (pixel lane, channel quad); per pixel ONE coalesced float4 load of
// x, K (wave-broadcast) loads of dl, 4 K FMAs for dx and 4 K for dw, one coalesced float4 store -- no LDS staging and no
// barrier inside the loop.  The LDS-tiled kernel above ran the 32 -> 5 head at 256^2 x 64 images at 2.4 TB/s of its
// 1.16 GB (three barriers per 64-pixel tile, 165 of 256 threads busy in the dw phase).  Per-thread fp32 sums over its
// <= 64 pixels, fp32 across the lanes of a wave, double across waves and blocks (fixed order: deterministic).
template <bool LAZY>
__global__ __launch_bounds__(SP_THREADS) void conv1x1_bwd_stream_kernel(const float* __restrict__ dl, const act_t* __restrict__ x,
                                                                        int ld_x, int C, const float* __restrict__ w,
                                                                        act_t* __restrict__ dx, int ld_dx, int K, int N, int HW,
                                                                        int pix_per_block, int accumulate_dx,
                                                                        float* __restrict__ partial /*[blocks][K*(C+1)]*/, PpLazy lz) {
  __shared__ float red[SP_THREADS / 64][HEAD_MAXK][HEAD_MAXC + 4];
  const int c4n = C >> 2, ppl = SP_THREADS / c4n;
  const int cq = threadIdx.x % c4n, pl = threadIdx.x / c4n;
  const int P = N * HW;
  const int p_lo = blockIdx.x * pix_per_block;
  const int p_hi = min(P, p_lo + pix_per_block);
  float wr[HEAD_MAXK][4], acc[HEAD_MAXK][4], accb[HEAD_MAXK];
#pragma unroll
  for (int k = 0; k < HEAD_MAXK; ++k) {
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[k][j] = k < K ? w[k * C + cq * 4 + j] : 0.f;     // w sits in the flat parameter slab: 4-byte aligned only
    acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.f;
    accb[k] = 0.f;
  }
  auto pixel = [&](int p, const float* dp, int n_img) {       // dp = dl + (n K HW + hw) of pixel p
    float4 xv1[1] = {act_ld4f(x + (size_t)p * ld_x + cq * 4)};
    SP_LAZY4(xv1, n_img)
    const float4 xv = xv1[0];
    float g[HEAD_MAXK];
#pragma unroll
    for (int k = 0; k < HEAD_MAXK; ++k) g[k] = k < K ? dp[(size_t)k * HW] : 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < HEAD_MAXK; ++k)
      if (k < K) {
        o.x += g[k] * wr[k][0]; o.y += g[k] * wr[k][1]; o.z += g[k] * wr[k][2]; o.w += g[k] * wr[k][3];
        acc[k][0] += g[k] * xv.x; acc[k][1] += g[k] * xv.y; acc[k][2] += g[k] * xv.z; acc[k][3] += g[k] * xv.w;
        accb[k] += g[k];
      }
    if (dx) {
      act_t* po = dx + (size_t)p * ld_dx + cq * 4;
      if (accumulate_dx) { const float4 t = act_ld4f(po); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
      act_st4f(po, o);
    }
  };
  int p = p_lo + pl;
  int n = p / HW, hw = p - n * HW;
  if (HW % pix_per_block == 0) {          // the block stays inside one image: no wrap test, four pixels' loads in flight
    const float* dp = dl + (size_t)n * K * HW + hw;
#pragma unroll 4
    for (; p < p_hi; p += ppl, dp += ppl) pixel(p, dp, n);
  } else {
    for (; p < p_hi; p += ppl) {
      pixel(p, dl + (size_t)n * K * HW + hw, n);
      hw += ppl;
      while (hw >= HW) { hw -= HW; ++n; }
    }
  }
  // lanes of a wave that share the channel quad (lane = pixel lane * c4n + quad): butterfly over the pixel-lane bits
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < HEAD_MAXK; ++k)
    if (k < K) {
      for (int d = c4n; d < 64; d <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] += __shfl_xor(acc[k][j], d, 64);
        accb[k] += __shfl_xor(accb[k], d, 64);
      }
      if (lane < c4n) {               // c4n <= 32 < 64: lanes 0 .. c4n-1 hold pixel lane 0 of the wave for every quad
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wv][k][cq * 4 + j] = acc[k][j];
        if (cq == 0) red[wv][k][C] = accb[k];
      }
    }
  __syncthreads();
  const int nout = K * (C + 1);
  for (int o = threadIdx.x; o < nout; o += SP_THREADS) {
    const int k = o / (C + 1), c = o - k * (C + 1);
    double t = 0.0;
#pragma unroll
    for (int v = 0; v < SP_THREADS / 64; ++v) t += (double)red[v][k][c];
    partial[(size_t)blockIdx.x * nout + o] = (float)t;
  }
}

// 16 outputs x 16 partial-lanes per block; each lane strides over the per-block partials, LDS combine in fixed order
__global__ __launch_bounds__(256) void conv1x1_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks,
                                                                   int K, int C, float* dw, float* db,
                                                                   int accumulate) {
  __shared__ double red[16][17];
  const int ol = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + ol;
  const int nout = K * (C + 1);
  double s = 0.0;
  if (o < nout) {
    // eight independent loads in flight per lane: the dependent load -> add chain of the first form (2048 partial rows over 16
    // lanes = 128 round trips to memory) made this launch 51 us at the benchmark shape
    int b = bl;
    for (; b + 7 * 16 < nblocks; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = partial[(size_t)(b + j * 16) * nout + o];
      s += (((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3])) + (((double)v[4] + (double)v[5]) + ((double)v[6] + (double)v[7]));
    }
    for (; b < nblocks; b += 16) s += (double)partial[(size_t)b * nout + o];
  }
  red[bl][ol] = s;
  __syncthreads();
  if (bl != 0 || o >= nout) return;
  double t = 0.0;
  for (int i = 0; i < 16; ++i) t += red[i][ol];
  const int k = o / (C + 1), c = o % (C + 1);
  if (c < C) {
    if (dw) dw[k * C + c] = (accumulate ? dw[k * C + c] : 0.f) + (float)t;
  } else {
    if (db) db[k] = (accumulate ? db[k] : 0.f) + (float)t;
  }
}

static int head_blocks(long long P, int* pix_per_block) {
  int blocks = pp_cdiv(P, 1024);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  int ppb = pp_cdiv(P, blocks);
  ppb = pp_cdiv(ppb, HEAD_TP) * HEAD_TP;
  *pix_per_block = ppb;
  return pp_cdiv(P, ppb);
}

extern "C" size_t PP_FN(pp_conv1x1_bwd_workspace)(int K, int C, int N, int HW) {
  int ppb;
  const int blocks = head_blocks((long long)N * HW, &ppb);
  return (size_t)blocks * K * (C + 1) * sizeof(float);
}

static int conv1x1_bwd_impl(const float* dlogits, const pp_act* x, int ld_x, int C, const float* w,
                            pp_act* dx, int ld_dx, float* dw, float* dbias, int K, int N, int HW,
                            int accumulate_dx, int accumulate_param_grads, void* workspace,
                            size_t workspace_bytes, PpLazy lz, hipStream_t s) {
  PP_CHECK_ARG(dlogits && x && w && workspace, "conv1x1_bwd: null pointer");
  PP_CHECK_ARG(K >= 1 && K <= HEAD_MAXK && C % 4 == 0 && C <= HEAD_MAXC && ld_x % 4 == 0 && ld_x >= C,
               "conv1x1_bwd: K=%d (<=8) C=%d (<=128, %%4) ld=%d", K, C, ld_x);
  PP_CHECK_ARG(!dx || (ld_dx % 4 == 0 && ld_dx >= C && ((uintptr_t)dx & PP_ACT_ALIGN) == 0), "conv1x1_bwd: bad dx");
  PP_CHECK_ARG(((uintptr_t)x & PP_ACT_ALIGN) == 0, "conv1x1_bwd: x must be 16-byte aligned");
  if (workspace_bytes < pp_conv1x1_bwd_workspace(K, C, N, HW)) {
    pp_set_error("conv1x1_bwd: workspace too small");
    return PP_ERR_WORKSPACE;
  }
  const long long P = (long long)N * HW;
  int ppb;
  const int blocks = head_blocks(P, &ppb);
  pp_prof_begin(PP_K_SPATIAL, 4.0 * P * K * C, 4.0 * P * (2.0 * C + K), s);
  const int c4n = C / 4;
  if ((c4n & (c4n - 1)) == 0 && P < 0x7fffffffLL)
  {
    if (lz.coef) hipLaunchKernelGGL(conv1x1_bwd_stream_kernel<true>, dim3(blocks), dim3(SP_THREADS), 0, s, dlogits, x, ld_x, C, w, dx, ld_dx, K, N,
                       HW, ppb, accumulate_dx, (float*)workspace, lz);
    else hipLaunchKernelGGL(conv1x1_bwd_stream_kernel<false>, dim3(blocks), dim3(SP_THREADS), 0, s, dlogits, x, ld_x, C, w, dx, ld_dx, K, N,
                       HW, ppb, accumulate_dx, (float*)workspace, lz);
  }
  else
  {
    if (lz.coef) hipLaunchKernelGGL(conv1x1_bwd_kernel<true>, dim3(blocks), dim3(SP_THREADS), 0, s, dlogits, x, ld_x, C, w, dx, ld_dx, K, N,
                       HW, ppb, accumulate_dx, (float*)workspace, lz);
    else hipLaunchKernelGGL(conv1x1_bwd_kernel<false>, dim3(blocks), dim3(SP_THREADS), 0, s, dlogits, x, ld_x, C, w, dx, ld_dx, K, N,
                       HW, ppb, accumulate_dx, (float*)workspace, lz);
  }
  hipLaunchKernelGGL(conv1x1_bwd_finalize_kernel, dim3(pp_cdiv(K * (C + 1), 16)), dim3(256), 0, s, (const float*)workspace,
                     blocks, K, C, dw, dbias, accumulate_param_grads);
  pp_prof_end(s);
  return pp_launch_status("conv1x1_bwd");
}

extern "C" int PP_FN(pp_conv1x1_nchw_to_nhwc_bwd)(const float* dlogits, const pp_act* x, int ld_x, int C, const float* w,
                                           pp_act* dx, int ld_dx, float* dw, float* dbias, int K, int N, int HW,
                                           int accumulate_dx, int accumulate_param_grads, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  return conv1x1_bwd_impl(dlogits, x, ld_x, C, w, dx, ld_dx, dw, dbias, K, N, HW, accumulate_dx, accumulate_param_grads, workspace,
                          workspace_bytes, pp_lazy_none(), (hipStream_t)stream);
}

// lazy x: dw is taken against y = LeakyReLU(BN(x)), evaluated while x is loaded (dx does not depend on x)
extern "C" int PP_FN(pp_conv1x1_nchw_to_nhwc_bwd_lazy)(const float* dlogits, const pp_act* x, int ld_x, int C, const float* w,
                                                pp_act* dx, int ld_dx, float* dw, float* dbias, int K, int N, int HW,
                                                int accumulate_dx, int accumulate_param_grads, void* workspace,
                                                size_t workspace_bytes, const pp_lazy_in* lazy_x, void* stream) {
  PpLazy lz;
  if (int rc = sp_lazy(lazy_x, C, N, lz)) return rc;
  return conv1x1_bwd_impl(dlogits, x, ld_x, C, w, dx, ld_dx, dw, dbias, K, N, HW, accumulate_dx, accumulate_param_grads, workspace,
                          workspace_bytes, lz, (hipStream_t)stream);
}

PP_NS_END

#ifndef PP_ACT_16       // fp32-only sections: scribble synthesis, strided / transposed convolution (--strided_unet)
// ---------------------------------------------------------------- synthetic scribbles (utils/utils_artificial_scribbles.py)
// skimage.morphology.skeletonize (2-D, Zhang-Suen thinning [Zha84]) as the reference uses it at
// utils_artificial_scribbles.py:19,33: two sub-iterations per sweep, every pixel of a sub-iteration is judged on the
// SAME input image and the deletions are applied together -- a parallel algorithm by construction.  With the 8
// neighbours P2..P9 clockwise from north, B = their sum, A = number of 0 -> 1 steps around the ring:
//   delete in sub-iteration 1 when 2 <= B <= 6, A == 1, P2*P4*P6 == 0 and P4*P6*P8 == 0,
//   delete in sub-iteration 2 when 2 <= B <= 6, A == 1, P2*P4*P8 == 0 and P2*P6*P8 == 0;   until nothing changes.
// One block per mask keeps the (zero-bordered) image twice in LDS: masks up to about 280 x 280 (256 x 256 slices fit).
#define SK_MAXPIX (322 * 322)
__global__ __launch_bounds__(1024) void skeletonize_kernel(unsigned char* __restrict__ masks, int H, int W, int max_sweeps) {
  extern __shared__ unsigned char sk[];             // [2][(H+2)*(W+2)]
  const int Wp = W + 2, Np = (H + 2) * Wp;
  unsigned char* cur = sk;
  unsigned char* nxt = sk + Np;
  unsigned char* m = masks + (size_t)blockIdx.x * H * W;
  __shared__ int changed;
  for (int i = threadIdx.x; i < Np; i += blockDim.x) {
    const int y = i / Wp - 1, x = i % Wp - 1;
    const unsigned char v = (y >= 0 && y < H && x >= 0 && x < W) ? (m[y * W + x] != 0) : 0;
    cur[i] = v; nxt[i] = v;
  }
  __syncthreads();
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    if (threadIdx.x == 0) changed = 0;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
      int local = 0;
      for (int p = threadIdx.x; p < H * W; p += blockDim.x) {
        const int i = (p / W + 1) * Wp + p % W + 1;
        if (!cur[i]) continue;
        const int p2 = cur[i - Wp], p3 = cur[i - Wp + 1], p4 = cur[i + 1], p5 = cur[i + Wp + 1];
        const int p6 = cur[i + Wp], p7 = cur[i + Wp - 1], p8 = cur[i - 1], p9 = cur[i - Wp - 1];
        const int B = p2 + p3 + p4 + p5 + p6 + p7 + p8 + p9;
        const int A = (!p2 && p3) + (!p3 && p4) + (!p4 && p5) + (!p5 && p6) + (!p6 && p7) + (!p7 && p8) + (!p8 && p9) + (!p9 && p2);
        const bool common = B >= 2 && B <= 6 && A == 1;
        const bool del = common && (pass == 0 ? (p2 * p4 * p6 == 0 && p4 * p6 * p8 == 0) : (p2 * p4 * p8 == 0 && p2 * p6 * p8 == 0));
        if (del) { nxt[i] = 0; local = 1; }
      }
      if (local) changed = 1;
      __syncthreads();
      for (int i = threadIdx.x; i < Np; i += blockDim.x) cur[i] = nxt[i];
      __syncthreads();
    }
    if (!changed) break;
    __syncthreads();
  }
  for (int p = threadIdx.x; p < H * W; p += blockDim.x) m[p] = cur[(p / W + 1) * Wp + p % W + 1];
}

// scipy.ndimage.binary_dilation(seed, structure = anti-diagonal 3x3, iterations, mask): a pixel of the mask joins when
// its north-east or south-west neighbour is set (utils_artificial_scribbles.py:31, background-only images).
__global__ __launch_bounds__(1024) void dilate_antidiag_kernel(unsigned char* __restrict__ seeds, const unsigned char* __restrict__ masks,
                                                              int H, int W, int iterations) {
  extern __shared__ unsigned char sk[];
  const int Wp = W + 2, Np = (H + 2) * Wp;
  unsigned char* cur = sk;
  unsigned char* nxt = sk + Np;
  unsigned char* s = seeds + (size_t)blockIdx.x * H * W;
  const unsigned char* mk = masks + (size_t)blockIdx.x * H * W;
  for (int i = threadIdx.x; i < Np; i += blockDim.x) {
    const int y = i / Wp - 1, x = i % Wp - 1;
    cur[i] = (y >= 0 && y < H && x >= 0 && x < W) ? (s[y * W + x] != 0) : 0;
  }
  __syncthreads();
  for (int it = 0; it < iterations; ++it) {
    for (int p = threadIdx.x; p < H * W; p += blockDim.x) {
      const int i = (p / W + 1) * Wp + p % W + 1;
      nxt[i] = cur[i] | ((mk[p] != 0) & (cur[i - Wp + 1] | cur[i + Wp - 1]));
    }
    __syncthreads();
    for (int p = threadIdx.x; p < H * W; p += blockDim.x) {
      const int i = (p / W + 1) * Wp + p % W + 1;
      cur[i] = nxt[i];
    }
    __syncthreads();
  }
  for (int p = threadIdx.x; p < H * W; p += blockDim.x) s[p] = cur[(p / W + 1) * Wp + p % W + 1];
}

// endpoints of a one-pixel-wide curve (utils_shorten_scribble_length.py:64-75): set pixels with exactly one set 8-neighbour
__global__ void endpoints_kernel(const unsigned char* __restrict__ img, unsigned char* __restrict__ out, int M, int H, int W) {
  const long long total = (long long)M * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % ((long long)H * W)), y = p / W, x = p % W;
    const unsigned char* im = img + (i - p);
    int nb = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx)
        if ((dy || dx) && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W) nb += im[(y + dy) * W + x + dx] != 0;
    out[i] = (im[p] != 0) && nb == 1;
  }
}

extern "C" int pp_skeletonize(unsigned char* masks, int M, int H, int W, void* stream) {
  PP_CHECK_ARG(masks && M >= 1 && H >= 1 && W >= 1 && (H + 2) * (W + 2) <= SK_MAXPIX, "skeletonize: bad arguments");
  const int lds = 2 * (H + 2) * (W + 2);
  PP_CHECK_ARG(lds <= 160 * 1024 - 64, "skeletonize: image does not fit the LDS");
  pp_max_lds(reinterpret_cast<const void*>(skeletonize_kernel), lds);
  hipLaunchKernelGGL(skeletonize_kernel, dim3(M), dim3(1024), lds, (hipStream_t)stream, masks, H, W, H + W);
  return pp_launch_status("skeletonize");
}

extern "C" int pp_dilate_antidiagonal(unsigned char* seeds, const unsigned char* masks, int M, int H, int W, int iterations,
                                      void* stream) {
  PP_CHECK_ARG(seeds && masks && M >= 1 && iterations >= 0 && 2 * (H + 2) * (W + 2) <= 160 * 1024 - 64, "dilate_antidiagonal: bad arguments");
  const int lds = 2 * (H + 2) * (W + 2);
  pp_max_lds(reinterpret_cast<const void*>(dilate_antidiag_kernel), lds);
  hipLaunchKernelGGL(dilate_antidiag_kernel, dim3(M), dim3(1024), lds, (hipStream_t)stream, seeds, masks, H, W, iterations);
  return pp_launch_status("dilate_antidiagonal");
}

extern "C" int pp_curve_endpoints(const unsigned char* img, unsigned char* out, int M, int H, int W, void* stream) {
  PP_CHECK_ARG(img && out && M >= 1 && H >= 1 && W >= 1, "curve_endpoints: bad arguments");
  hipLaunchKernelGGL(endpoints_kernel, dim3(sp_blocks((long long)M * H * W)), dim3(SP_THREADS), 0, (hipStream_t)stream, img, out,
                     M, H, W);
  return pp_launch_status("curve_endpoints");
}

// ================================================================ strided / transposed convolution variants of the U-Net
// `--is_stride_conv / --is_trans_conv` (models/unet.py:100-152): down-sampling by the first convolution of a stage with
// stride 2 instead of MaxPool2d, up-sampling by ConvTranspose2d(lower, skip, k, k, bias=False) instead of bilinear.  No
// reference configuration uses them (the flags are `type=bool` with default False), so they are built for correctness on top
// of the existing kernels, not for speed:
//   * a stride-2, padding-1 3x3 convolution is the stride-1 convolution sampled at the even pixels (pp_stride2_gather); its
//     gradients are those of the stride-1 convolution for dz scattered back to the even pixels (pp_stride2_scatter);
//   * ConvTranspose2d with kernel == stride == k is k*k independent pointwise products:
//       out[n, k y + a, k x + b, o] = sum_c x[n, y, x, c] * w[c, o, a, b]       (k = 1: a plain 1x1 convolution)
//     i.e. one GEMM  [pixels] x [Cin] x [k k Cout]  whose output columns are scattered to the k x k sub-positions; its data
//     gradient is the same GEMM with the gathered gradient as A and w as B^T, its weight gradient reduces over pixels.
__global__ void stride2_gather_kernel(const float* __restrict__ full, int ld_f, float* __restrict__ out, int ld_o, int C, int Ho,
                                      int Wo, long long total) {
  const int c4n = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4n);
    const long long p = i / c4n;                          // output pixel (n, yo, xo)
    const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho);
    const long long n = p / ((long long)Wo * Ho);
    const long long pf = (n * (2 * Ho) + 2 * yo) * (2 * Wo) + 2 * xo;
    *reinterpret_cast<float4*>(out + p * ld_o + cq * 4) = *reinterpret_cast<const float4*>(full + pf * ld_f + cq * 4);
  }
}

__global__ void stride2_scatter_kernel(const float* __restrict__ dz, int ld_z, float* __restrict__ full, int ld_f, int C, int Ho,
                                       int Wo, long long total) {
  const int c4n = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % c4n);
    const long long pf = i / c4n;                         // full-resolution pixel (n, y, x)
    const int x = (int)(pf % (2 * Wo)), y = (int)((pf / (2 * Wo)) % (2 * Ho));
    const long long n = pf / ((long long)4 * Wo * Ho);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(x & 1) && !(y & 1)) v = *reinterpret_cast<const float4*>(dz + ((n * Ho + (y >> 1)) * Wo + (x >> 1)) * ld_z + cq * 4);
    *reinterpret_cast<float4*>(full + pf * ld_f + cq * 4) = v;
  }
}

extern "C" int pp_stride2_gather(const float* full, int ld_full, float* out, int ld_out, int C, int N, int Ho, int Wo, void* stream) {
  PP_CHECK_ARG(full && out && C > 0 && C % 4 == 0 && ld_full % 4 == 0 && ld_out % 4 == 0 && ld_full >= C && ld_out >= C && N > 0 && Ho > 0 && Wo > 0,
               "stride2_gather: bad arguments");
  const long long total = (long long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(stride2_gather_kernel, dim3(sp_blocks(total)), dim3(SP_THREADS), 0, (hipStream_t)stream, full, ld_full, out, ld_out, C,
                     Ho, Wo, total);
  return pp_launch_status("stride2_gather");
}

extern "C" int pp_stride2_scatter(const float* dz, int ld_dz, float* full, int ld_full, int C, int N, int Ho, int Wo, void* stream) {
  PP_CHECK_ARG(dz && full && C > 0 && C % 4 == 0 && ld_full % 4 == 0 && ld_dz % 4 == 0 && ld_full >= C && ld_dz >= C && N > 0 && Ho > 0 && Wo > 0,
               "stride2_scatter: bad arguments");
  const long long total = (long long)N * 4 * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(stride2_scatter_kernel, dim3(sp_blocks(total)), dim3(SP_THREADS), 0, (hipStream_t)stream, dz, ld_dz, full, ld_full, C,
                     Ho, Wo, total);
  return pp_launch_status("stride2_scatter");
}

// ---- ConvTranspose2d(Cin, Cout, k, k, bias=False), k = 1 or 2: one 64 x 64 tiled fp32 GEMM kernel, three addressings ----
// MODE 0 forward:      C[p][j] = sum_c x[p][c] * w[c][j],            j = (o, a, b) as stored (w is [Cin][Cout][k][k])
// MODE 1 data grad:    C[p][c] = sum_j g(p, j) * w[c][j]             g(p, j) = dout at the sub-position of j
// MODE 2 weight grad:  C[c][j] = sum_{p in split} x[p][c] * g(p, j)  (per-split partials, fixed-order finalize)
struct CtArgs {
  const float* x; int ld_x;        // (N, H, W, Cin)
  const float* w;                  // [Cin][Cout * k * k]
  float* out; int ld_out;          // forward: (N, kH, kW, Cout);  data grad: dx (N, H, W, Cin);  weight grad: partial [splits][Cin][J]
  const float* g; int ld_g;        // (N, kH, kW, Cout) gradient (modes 1, 2)
  int Cin, Cout, k, N, H, W, accumulate, p_per_split;
};
__device__ __forceinline__ size_t ct_sub_pixel(const CtArgs& a, int p, int ab) {     // pixel index of sub-position ab of input pixel p
  const int x = p % a.W, y = (p / a.W) % a.H, n = p / (a.W * a.H);
  return ((size_t)n * a.H * a.k + (size_t)y * a.k + ab / a.k) * (a.W * a.k) + (size_t)x * a.k + ab % a.k;
}
template <int MODE>
__global__ __launch_bounds__(256) void convtranspose_gemm_kernel(CtArgs a) {
  __shared__ float As[16][64 + 4], Bs[16][64 + 4];
  const int P = a.N * a.H * a.W, kk = a.k * a.k, J = a.Cout * kk;
  const int M = MODE == 2 ? a.Cin : P, Nn = MODE == 1 ? a.Cin : J;
  int k_lo = 0, k_hi = MODE == 0 ? a.Cin : (MODE == 1 ? J : P);
  if (MODE == 2) { k_lo = blockIdx.z * a.p_per_split; k_hi = min(P, k_lo + a.p_per_split); }
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 x 16 threads, 4 x 4 outputs each
  float acc[4][4] = {};
  // element (m, kidx) of A and (kidx, n) of B in each mode
  auto A_at = [&](int m, int kidx) -> float {
    if (m >= M || kidx >= k_hi) return 0.f;
    if (MODE == 0) return a.x[(size_t)m * a.ld_x + kidx];
    if (MODE == 1) return a.g[ct_sub_pixel(a, m, kidx % kk) * a.ld_g + kidx / kk];      // j = o * kk + ab
    return a.x[(size_t)kidx * a.ld_x + m];                                                // x[p][c], m = c
  };
  auto B_at = [&](int kidx, int n) -> float {
    if (n >= Nn || kidx >= k_hi) return 0.f;
    if (MODE == 0) return a.w[(size_t)kidx * J + n];
    if (MODE == 1) return a.w[(size_t)n * J + kidx];
    return a.g[ct_sub_pixel(a, kidx, n % kk) * a.ld_g + n / kk];
  };
  for (int kb = k_lo; kb < k_hi; kb += 16) {
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      const int r = e / 64, c = e % 64;
      As[r][c] = A_at(m0 + c, kb + r);
      Bs[r][c] = B_at(kb + r, n0 + c);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = As[r][ty * 4 + i]; bv[i] = Bs[r][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m >= M || n >= Nn) continue;
      if (MODE == 0) {
        float* o = a.out + ct_sub_pixel(a, m, n % kk) * a.ld_out + n / kk;
        *o = acc[i][j];
      } else if (MODE == 1) {
        float* o = a.out + (size_t)m * a.ld_out + n;
        *o = a.accumulate ? *o + acc[i][j] : acc[i][j];
      } else {
        a.out[((size_t)blockIdx.z * a.Cin + m) * J + n] = acc[i][j];
      }
    }
}

__global__ void convtranspose_wgrad_finalize_kernel(const float* __restrict__ part, int splits, long long n, float* __restrict__ dw, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];                 // fixed order: deterministic
  dw[i] = accumulate ? dw[i] + s : s;
}

static int ct_check(const CtArgs& a) {
  PP_CHECK_ARG(a.x && a.w && a.out && (a.k == 1 || a.k == 2) && a.Cin > 0 && a.Cout > 0 && a.N > 0 && a.H > 0 && a.W > 0,
               "convtranspose: bad arguments (kernel == stride must be 1 or 2)");
  PP_CHECK_ARG((long long)a.N * a.H * a.W * a.k * a.k < 0x7fffffffLL, "convtranspose: too many pixels");
  return 0;
}
static int ct_splits(int P) { int s = pp_cdiv(P, 4096); return s < 1 ? 1 : (s > 64 ? 64 : s); }

extern "C" size_t pp_convtranspose_bwd_weight_workspace(int Cin, int Cout, int k, int N, int H, int W) {
  return (size_t)ct_splits(N * H * W) * Cin * Cout * k * k * sizeof(float) + 256;
}

extern "C" int pp_convtranspose_fwd(const float* x, int ld_x, int Cin, const float* w, float* out, int ld_out, int Cout, int k, int N,
                                    int H, int W, void* stream) {
  CtArgs a{x, ld_x, w, out, ld_out, nullptr, 0, Cin, Cout, k, N, H, W, 0, 0};
  if (int rc = ct_check(a)) return rc;
  PP_CHECK_ARG(ld_x >= Cin && ld_out >= Cout, "convtranspose_fwd: bad ld");
  hipLaunchKernelGGL(convtranspose_gemm_kernel<0>, dim3(pp_cdiv(N * H * W, 64), pp_cdiv(Cout * k * k, 64)), dim3(256), 0, (hipStream_t)stream, a);
  return pp_launch_status("convtranspose_fwd");
}

extern "C" int pp_convtranspose_bwd_data(const float* dout, int ld_g, int Cout, const float* w, float* dx, int ld_dx, int Cin, int k,
                                         int N, int H, int W, int accumulate, void* stream) {
  CtArgs a{dout, ld_g, w, dx, ld_dx, dout, ld_g, Cin, Cout, k, N, H, W, accumulate, 0};
  if (int rc = ct_check(a)) return rc;
  PP_CHECK_ARG(ld_g >= Cout && ld_dx >= Cin, "convtranspose_bwd_data: bad ld");
  hipLaunchKernelGGL(convtranspose_gemm_kernel<1>, dim3(pp_cdiv(N * H * W, 64), pp_cdiv(Cin, 64)), dim3(256), 0, (hipStream_t)stream, a);
  return pp_launch_status("convtranspose_bwd_data");
}

extern "C" int pp_convtranspose_bwd_weight(const float* dout, int ld_g, int Cout, const float* x, int ld_x, int Cin, int k, int N, int H,
                                           int W, float* dw, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  CtArgs a{x, ld_x, dw, reinterpret_cast<float*>(workspace), 0, dout, ld_g, Cin, Cout, k, N, H, W, accumulate, 0};
  if (int rc = ct_check(a)) return rc;
  PP_CHECK_ARG(dout && workspace && ld_g >= Cout && ld_x >= Cin, "convtranspose_bwd_weight: bad arguments");
  const int P = N * H * W, splits = ct_splits(P);
  if (workspace_bytes < pp_convtranspose_bwd_weight_workspace(Cin, Cout, k, N, H, W)) {
    pp_set_error("convtranspose_bwd_weight: workspace too small");
    return PP_ERR_WORKSPACE;
  }
  a.p_per_split = pp_cdiv(P, splits);
  const int J = Cout * k * k;
  hipLaunchKernelGGL(convtranspose_gemm_kernel<2>, dim3(pp_cdiv(Cin, 64), pp_cdiv(J, 64), splits), dim3(256), 0, (hipStream_t)stream, a);
  const long long n = (long long)Cin * J;
  hipLaunchKernelGGL(convtranspose_wgrad_finalize_kernel, dim3(pp_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float*>(workspace), splits, n, dw, accumulate);
  return pp_launch_status("convtranspose_bwd_weight");
}
#endif  // !PP_ACT_16
