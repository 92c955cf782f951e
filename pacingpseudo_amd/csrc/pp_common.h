// Shared host/device helpers for the pacingpseudo HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include "pacingpseudo_hip.h"      // the public C ABI: every extern "C" definition is checked against its declaration

#define PP_WAVE 64

// ---- activation element type of this translation unit (round 4: 16-bit storage, BASELINE config 5) ----
// Every source file is compiled twice: once with fp32 activations in HBM (the entry points of include/pacingpseudo_hip.h) and
// once with -DPP_ACT_H16, where every NHWC activation / activation-gradient tensor (y, the pre-BatchNorm z, dy, dz, the
// network input) is stored as IEEE fp16 and the entry points carry the suffix _h16 (include/pacingpseudo_hip_h16.h).  Only
// the loads and stores change (act_ld4 / act_st4 / act_buf_ld4 ...): arithmetic, accumulators, BatchNorm statistics,
// Winograd-domain operands, weights, logits and every gradient of a parameter stay fp32.  `ld` arguments count ELEMENTS.
// Round 6: a third compilation with -DPP_ACT_BF16 stores the same tensors as bfloat16 (BASELINE.json configs[4] names bf16; entry
// points suffixed _bf16, include/pacingpseudo_hip_bf16.h).  The kernels are the 16-bit build's: a bf16 number has 8 significand
// bits and fp32's exponent range, so it converts EXACTLY into the fp16 hi operand of the split-operand products wherever it lies
// in fp16's normal range -- activations behind BatchNorm + LeakyReLU are O(1), gradients are scaled by a power of two from their
// maximum before they are staged (f16_scales) -- and the low part is zero as for fp16 storage.  PP_ACT_16 marks both 16-bit builds.
#if defined(PP_ACT_BF16)
typedef __bf16 act_t;
typedef __bf16 pp_act;                    // C callers see void* (pacingpseudo_hip_bf16.h)
typedef __bf16 pp_a16x4 __attribute__((ext_vector_type(4)));
#define PP_ACT_16 1
#define PP_ACT_BYTES 2
#define PP_ACT_ALIGN 7
#define PP_ACT_LO 0
#define PP_FN(name) name##_bf16
#define PP_NS_BEGIN namespace pp_bf16 {
#define PP_NS_END }
#elif defined(PP_ACT_H16)
typedef _Float16 act_t;
typedef _Float16 pp_act;                  // activation pointers in the C ABI of the _h16 entry points (C callers see void*: pacingpseudo_hip_h16.h)
typedef _Float16 pp_a16x4 __attribute__((ext_vector_type(4)));
#define PP_ACT_16 1
#define PP_ACT_BYTES 2
#define PP_ACT_ALIGN 7                    // a 4-element vector access needs 8-byte alignment
#define PP_ACT_LO 0                       // an fp16 activation IS its high part: the split-operand kernels drop the low-part products
#define PP_FN(name) name##_h16
#define PP_NS_BEGIN namespace pp_h16 {
#define PP_NS_END }
#else
typedef float act_t;
typedef float pp_act;
#define PP_ACT_BYTES 4
#define PP_ACT_ALIGN 15
#define PP_ACT_LO 1
#define PP_FN(name) name
#define PP_NS_BEGIN
#define PP_NS_END
#endif

// ---- error plumbing (thread-local last-error string, see include/pacingpseudo_hip.h) ----
extern "C" const char* pp_last_error(void);
void pp_set_error(const char* fmt, ...);

#define PP_ERR_ARG (-1)
#define PP_ERR_UNSUPPORTED (-2)
#define PP_ERR_WORKSPACE (-3)

#define PP_CHECK_ARG(cond, ...)                     \
  do {                                              \
    if (!(cond)) {                                  \
      pp_set_error(__VA_ARGS__);                    \
      return PP_ERR_ARG;                            \
    }                                               \
  } while (0)

// Check the launch that was just enqueued; returns the positive hipError_t on failure.
static inline int pp_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pp_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// hipFuncAttributeMaxDynamicSharedMemorySize, applied once per (kernel, device) under a lock (pp_runtime.cpp): launchers
// may be entered concurrently from the forward thread and the autograd thread, and on several devices of one process.
void pp_max_lds(const void* kernel, int bytes);

// ---- fused BatchNorm side of a forward convolution (conv -> BatchNorm2d -> LeakyReLU, models/unet.py:188-193) ----
// mode 1 (BN train): the conv kernel also emits per-block partial sums of its raw output z,
//         stats[((g * rows + row) * 2 + {0: sum z, 1: sum z^2}) * N + n]  (double) -- the pass over z that
//         bn_stats_partial_kernel made is gone; bn_stats_finalize_kernel consumes the rows as before.
// mode 2 (BN eval): out = leaky_relu(z * scale[n] + shift[n], slope): y is written directly, z never reaches HBM.
struct PpEpi {
  int mode;
  const float* scale; const float* shift; float slope;
  double* stats; int rows; int px_per_group; int groups;
};
#define PP_EPI_GROUPS 2        // the siamese step's weak | strong halves; more groups run the unfused kernels
// unfused fall-backs (pp_norm.hip) for convolution variants without a fused epilogue
PP_NS_BEGIN
int pp_bn_partial_rows(int C, int P_per_group, int groups);
int pp_bn_stats_partial_launch(const act_t* z, int ld, int C, int P_per_group, int groups, double* partial, hipStream_t s);
int pp_bn_apply_launch(const act_t* z, int ld_z, const float* scale, const float* shift, int coef_groups, act_t* y, int ld_y,
                       int C, int P_per_group, int groups, float slope, hipStream_t s);
PP_NS_END

static inline int pp_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- "lazy" activations: BatchNorm + LeakyReLU applied by the CONSUMER while it loads (round 4) ----
// In train mode the batch statistics exist only after the whole convolution output z has been written, so the normalised
// activation y = lrelu(z * scale + shift) used to cost one more pass over the tensor (bn_lrelu_fwd_kernel: 8 B per
// element, 9.7 GB per benchmark step).  A lazy tensor keeps z in HBM -- in the buffer y would have occupied -- together
// with per-(group, channel) coefficient rows, and every kernel that reads it evaluates y on the fly:
//   coef[(g * 3 + 0) * ld + c] = scale,  [(g * 3 + 1) * ld + c] = shift,  [(g * 3 + 2) * ld + c] = negative slope
// (g = image / imgs_per_group: the weak | strong halves of the siamese batch have their own statistics).  Channels of a
// concatenation buffer that already hold final values carry the identity row (1, 0, 1).  Zero padding applies to y, not
// to z: loaders must produce 0 -- not lrelu(shift) -- outside the image.
struct PpLazy {
  const float* coef;            // null: the tensor holds final values
  int ld;                       // row length of coef (channels of the buffer the view is a slice of)
  int imgs_per_group;           // images per statistics group (>= 1)
};
static inline PpLazy pp_lazy_none() { return PpLazy{nullptr, 0, 1}; }

// ---- optional per-launch profiling (HIP events on the launch stream) ----
// kind: index into the kernel-family table (see pp_prof_* in the header).
void pp_prof_begin(int kind, double flops, double bytes, hipStream_t s);
// same, with the ALGORITHMIC flop count of the operation when it differs from the executed one (Winograd)
void pp_prof_begin2(int kind, double flops, double alg_flops, double bytes, hipStream_t s);
void pp_prof_end(hipStream_t s);

enum PpProfKind {
  PP_K_CONV_IGEMM = 0,
  PP_K_CONV_WGRAD = 1,
  PP_K_BN = 2,
  PP_K_SPATIAL = 3,
  PP_K_LOSS = 4,
  PP_K_OPTIM = 5,
  PP_K_MISC = 6,
  PP_K_WINO_GEMM = 7,       // Winograd-domain batched GEMM (fwd / dgrad); flops = EXECUTED (8 per pixel*cin*cout)
  PP_K_WINO_WGRAD = 8,      // Winograd-domain weight-gradient GEMM; flops = executed
  PP_K_WINO_XFORM = 9,      // input / output / gradient transforms (HBM-bound)
  PP_K_CONV_F16X3 = 10,     // split-fp16 implicit-GEMM convolution (conv3x3_igemm_f16x3_kernel, fwd / dgrad); flops = executed 16-bit MFMA flops (3x algorithmic)
  PP_K_WINO_GEMM_F16X3 = 11, // Winograd-domain GEMM on the fp16 MFMA with split operands; flops = executed (3x)
  PP_K_WINO_WGRAD_F16X3 = 12, // Winograd-domain weight-gradient GEMM on the fp16 MFMA; flops = executed (3x)
  PP_K_CONV_WGRAD_F16X3 = 13, // direct weight gradient on the fp16 MFMA (narrow layers); flops = executed (3x)
  PP_K_CONV_HALO_F16X3 = 14,  // split-fp16 persistent halo-tile convolution (conv3x3_halo_f16x3_kernel, fwd / dgrad of the narrow layers)
  PP_K_COUNT = 15
};

#ifdef __HIPCC__
// ---- device helpers ----
__device__ __forceinline__ float pp_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double pp_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over aligned groups of `n` adjacent lanes (n a power of two <= 64), every lane of the group gets it -- the butterfly of
// __shfl_xor(v, 1), (v, 2), ... with the first four steps as DPP adds inside the VALU (quad permutes, row_half_mirror, row_mirror:
// after the quad steps all lanes of a quad hold the same sum, so mirroring the 8 / 16-lane row pairs the same partial sums an
// xor-4 / xor-8 exchange would -- bit-identical results) instead of ds_bpermute round trips through the LDS crossbar, which
// bounded the 32 -> 5 head (15 of them per pixel: 2.3 TB/s of its input).
template <int CTRL>
__device__ __forceinline__ float pp_dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float pp_group_sum(float v, int n) {
  if (n >= 2) v = pp_dpp_add<0xB1>(v);              // quad_perm [1,0,3,2]
  if (n >= 4) v = pp_dpp_add<0x4E>(v);              // quad_perm [2,3,0,1]
  if (n >= 8) v = pp_dpp_add<0x141>(v);             // row_half_mirror
  if (n >= 16) v = pp_dpp_add<0x140>(v);            // row_mirror
  if (n >= 32) v += __shfl_xor(v, 16, 64);
  if (n >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
// Block-wide sum for blockDim.x multiple of 64 (<= 1024); `sh` holds >= 16 floats. Result valid in all threads.
__device__ __forceinline__ float pp_block_sum(float v, float* sh) {
  v = pp_wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];   // fixed order -> deterministic
  return r;
}
__device__ __forceinline__ float pp_lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }
// y = lrelu(z * sc + sh) with a per-channel slope (four channels at once)
typedef float pp_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pp_f32x4 pp_lazy_apply4(pp_f32x4 z, pp_f32x4 sc, pp_f32x4 sh, pp_f32x4 sl) {
  // ONE rounding (fma), spelled out: the forward branch of an activation (here) and the backward branch (bn_bwd_*_kernel,
  // pp_norm.hip) must be decided by the same arithmetic in every kernel, whatever the compiler's contraction choice
  pp_f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float v = __builtin_fmaf(z[e], sc[e], sh[e]);
    o[e] = v > 0.f ? v : v * sl[e];
  }
  return o;
}
__device__ __forceinline__ float pp_lazy_apply1(float z, float sc, float sh, float sl) {
  const float v = __builtin_fmaf(z, sc, sh);
  return v > 0.f ? v : v * sl;
}
// pre-activation of BatchNorm + LeakyReLU, the one expression every kernel uses for it (see pp_lazy_apply4)
__device__ __forceinline__ float pp_bn_pre(float z, float sc, float sh) { return __builtin_fmaf(z, sc, sh); }
// the three coefficient rows of channels c..c+3 for image n
__device__ __forceinline__ void pp_lazy_rows4(const PpLazy& L, int n, int c, pp_f32x4& sc, pp_f32x4& sh, pp_f32x4& sl) {
  const float* r = L.coef + (size_t)(n / L.imgs_per_group) * 3 * L.ld + c;
  sc = *reinterpret_cast<const pp_f32x4*>(r);
  sh = *reinterpret_cast<const pp_f32x4*>(r + L.ld);
  sl = *reinterpret_cast<const pp_f32x4*>(r + 2 * L.ld);
}
__device__ __forceinline__ double* pp_epi_row(const PpEpi& e, int g, int row, int which, int N) {
  return e.stats + ((size_t)(g * e.rows + row) * 2 + which) * N;
}

// ---- activation loads / stores (act_t = float, _Float16 or __bf16, see the top of this file) ----
typedef _Float16 pp_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pp_f32x4 act_ld4(const act_t* p) {
#ifdef PP_ACT_16
  return __builtin_convertvector(*reinterpret_cast<const pp_a16x4*>(p), pp_f32x4);
#else
  return *reinterpret_cast<const pp_f32x4*>(p);
#endif
}
__device__ __forceinline__ void act_st4(act_t* p, pp_f32x4 v) {
#ifdef PP_ACT_16
  *reinterpret_cast<pp_a16x4*>(p) = __builtin_convertvector(v, pp_a16x4);
#else
  *reinterpret_cast<pp_f32x4*>(p) = v;
#endif
}
// Two-step form for loads under a condition (halo pixels): the RAW bits are selected (load : zero) and converted afterwards.
// With the conversion inside the conditional expression every load of a tile sat in its own basic block followed by
// s_waitcnt vmcnt(0) + v_cvt: 36 serialised round trips in the Winograd input transform (+45 % kernel time, r04 trace).
#ifdef PP_ACT_16
typedef unsigned act_raw4 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ act_raw4 act_ld4_raw(const act_t* p) { return *reinterpret_cast<const act_raw4*>(p); }
__device__ __forceinline__ pp_f32x4 act_cvt4(act_raw4 r) { return __builtin_convertvector(__builtin_bit_cast(pp_a16x4, r), pp_f32x4); }
#else
typedef pp_f32x4 act_raw4;
__device__ __forceinline__ act_raw4 act_ld4_raw(const act_t* p) { return *reinterpret_cast<const act_raw4*>(p); }
__device__ __forceinline__ pp_f32x4 act_cvt4(act_raw4 r) { return r; }
#endif
__device__ __forceinline__ act_raw4 act_raw4_zero() { act_raw4 z; for (int i = 0; i < (int)(sizeof(act_raw4) / 4); ++i) z[i] = 0; return z; }
// the same with HIP's float4 struct, which the streaming kernels use
__device__ __forceinline__ float4 act_ld4f(const act_t* p) {
  const pp_f32x4 v = act_ld4(p);
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void act_st4f(act_t* p, float4 v) { act_st4(p, pp_f32x4{v.x, v.y, v.z, v.w}); }
__device__ __forceinline__ float act_ld1(const act_t* p) { return (float)*p; }
__device__ __forceinline__ void act_st1(act_t* p, float v) { *p = (act_t)v; }
// through a buffer descriptor (hardware bounds check: an offset beyond the extent reads 0 / drops the store); byte offsets
__device__ __forceinline__ pp_f32x4 act_buf_ld4(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
#ifdef PP_ACT_16
  typedef unsigned pp_u32x2 __attribute__((ext_vector_type(2)));
  const pp_u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
  return __builtin_convertvector(__builtin_bit_cast(pp_a16x4, r), pp_f32x4);
#else
  return __builtin_bit_cast(pp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
#endif
}
// the RAW bits of 4 elements through a buffer descriptor (converted later by act_cvt4: prefetch registers hold 8 bytes, not 16)
__device__ __forceinline__ act_raw4 act_buf_ld4_raw(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
#ifdef PP_ACT_16
  return __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
#else
  return __builtin_bit_cast(pp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
#endif
}
__device__ __forceinline__ float act_buf_ld1(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
#ifdef PP_ACT_16
  return (float)__builtin_bit_cast(act_t, __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0));
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
#endif
}
__device__ __forceinline__ void act_buf_st1(float v, __amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
#ifdef PP_ACT_16
  __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (act_t)v), rs, voff, soff, 0);
#else
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs, voff, soff, 0);
#endif
}

// ---- split-fp16 ("f16x3") operands, shared by pp_conv.hip and pp_wino.hip ----
// Matrix products per fp32 product in the forward / data-gradient matrix kernels: 3 (split operands, fp32-grade: the
// default) or 1 (hi parts only: fp16 operands, 11 significand bits, fp32 accumulation -- the "mixed precision" mode of
// BASELINE config 5, `--precision fp16`).  Set per process through pp_set_matrix_products (pp_runtime.cpp; the initial value
// comes from PP_F16_PRODUCTS); read at launch time.
int pp_f16_products();
#define PP_WGRAD_CUS_MAX 1024       // largest budget pp_set_wgrad_cus accepts; workspace queries size for max(256, ...) below
int pp_wgrad_cus();                 // CU budget of the persistent direct weight-gradient kernels (pp_set_wgrad_cus; per thread)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define H_LD 72            // halves per LDS row: 32 hi + 32 lo + 8 pad (144 B: ds_read_b128 conflict-free as for fp32)
#define F16_LO_SCALE 2048.f

// power-of-two scale that brings `amax` into [2^9, 2^10) (1 when amax is 0 / not finite)
__device__ __forceinline__ void f16_scales(const float* amax, float& s_in, float& s_out) {
  s_in = 1.f; s_out = 1.f;
  if (amax) {
    const float m = *amax;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      (void)frexpf(m, &e);                       // m = f * 2^e, f in [0.5, 1)
      s_in = ldexpf(1.f, 10 - e);
      s_out = ldexpf(1.f, e - 10);
    }
  }
}
#endif
