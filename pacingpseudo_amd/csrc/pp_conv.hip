// 3x3 (dilated) convolution for NHWC fp32 activations on gfx950, as implicit GEMMs on the
// f32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32 fma chains, 256 FLOP/clk/CU).
//
//   forward : out[p][n]  = sum_{tap,c} in[p + off(tap)][c] * Wf[n][tap][c] + bias[n]
//             GEMM M = pixels, N = Cout, K = 9*Cin       (reference: nn.Conv2d, models/unet.py:188)
//   dgrad   : the same kernel run on dz with Wb[c][8-tap][n] (taps flipped, channels swapped)
//   wgrad   : dW[n][tap][c] = sum_p dz[p][n] * x[p + off(tap)][c]
//             GEMM M = Cout, N = Cin (per tap), K = pixels, split over pixel ranges
//
// Both GEMM operands are staged through LDS by registers (global_load_dwordx4 -> ds_write_b128,
// next tile's loads issued before the current tile's MFMAs), one barrier per K-step, two LDS
// buffers.  LDS rows are padded by one 16-B slot so the ds_read_b128 fragment reads of the
// 32-row MFMA operand are bank-conflict free (row stride 36 floats: 36*m mod 64 hits 16 slots).
#include "pp_common.h"
#include <stdlib.h>

PP_NS_BEGIN

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 32          // K elements (channels of one tap, or pixels for wgrad) per LDS stage
#define LDS_LD 36      // padded row length in floats for the K-contiguous tiles

struct ConvArgs {
  const act_t* in; int ld_in; int C;
  const float* w;            // [N][9][C]
  const float* bias;         // [N] or null
  act_t* out; int ld_out; int N;
  int P, H, W, dil, accumulate;
  int m_tiles, n_tiles;
  unsigned in_bytes, w_bytes;      // extents for the buffer descriptors (hardware bounds check)
  PpEpi epi;                       // fused BatchNorm epilogue of a forward call (mode 0: none), see pp_common.h
  unsigned out_bytes;              // extent of `out` for buffer stores (conv3x3_halo2_f16x3_kernel); 0 = exceeds 4 GiB
  PpLazy lazy;                     // `in` is a lazy tensor (pp_common.h): only the two-half halo kernel reads one
};

// Map a linear block id to (m_tile, n_tile).  Blocks b, b + 8, ... run on the same XCD and share its 4 MB L2.  An
// m-tile is 128 consecutive pixels, i.e. (part of) an image row, and its nine taps read the rows above and below:
// every XCD therefore gets a contiguous BAND of m-tiles (so the three uses of an input row meet in one L2 instead of
// three), and inside the band the n-tiles of one m-tile are adjacent.  The r01 mapping interleaved the m-tiles over the
// XCDs (m = 8 k + xcd): the 128 x 64-tile launches then moved 3.9 GB for 1.07 GB of operands (r02 PMC profile).
__device__ __forceinline__ void tile_of_block(int b, int m_tiles, int n_tiles, int& mt, int& nt) {
  if ((m_tiles & 7) == 0) {
    const int xcd = b & 7, slot = b >> 3;
    nt = slot % n_tiles;
    mt = xcd * (m_tiles >> 3) + slot / n_tiles;
  } else {
    nt = b % n_tiles;
    mt = b / n_tiles;
  }
}

// wgrad work item -> (tap, c_tile, o_tile, split).  Items are enumerated (split, o_tile, c_tile, tap) and each XCD
// (blocks b, b+8, ... share one) gets a contiguous range of them, so the blocks that re-read one dz tile and the
// overlapping x tiles of its nine taps hit the same L2 instead of eight different ones.
__device__ __forceinline__ void wgrad_item(int c_tiles, int o_tiles, int& tap, int& ct, int& ot, int& split) {
  const int per_split = 9 * c_tiles * o_tiles;
  const int total = per_split * gridDim.y;
  int L = blockIdx.y * gridDim.x + blockIdx.x;
  if ((total & 7) == 0) L = (L & 7) * (total >> 3) + (L >> 3);
  split = L / per_split;
  const int r = L - split * per_split;
  tap = r % 9;
  ct = (r / 9) % c_tiles;
  ot = r / (9 * c_tiles);
}

#ifndef PP_ACT_16     // the fp32-MFMA kernels (PP_PRECISION=fp32) exist for fp32 activations only
template <int TM, int TN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void conv3x3_igemm_kernel(ConvArgs a) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = 32 * TM * WAVES_M;
  constexpr int BN = 32 * TN * WAVES_N;
  constexpr int RPP = NT / 8;               // tile rows covered per load pass (8 float4 per row)
  constexpr int A_PASSES = BM / RPP;
  constexpr int B_PASSES = (BN + RPP - 1) / RPP;
  static_assert(BM % RPP == 0, "BM must be a multiple of the rows per pass");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                          // [2][BM][LDS_LD]
  float* Bs = smem + 2 * BM * LDS_LD;        // [2][BN][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;
  const int q = tid & 7, r0 = tid >> 3;

  int mt, nt;
  tile_of_block(blockIdx.x, a.m_tiles, a.n_tiles, mt, nt);
  const int m0 = mt * BM, n0 = nt * BN;

  // per-thread bookkeeping of the A rows (pixels) this thread stages
  int py[A_PASSES], px[A_PASSES], pbase[A_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int p = m0 + r0 + i * RPP;
    if (p < a.P) {
      px[i] = p % a.W;
      py[i] = (p / a.W) % a.H;
      pbase[i] = p * a.ld_in;
    } else {
      px[i] = -0x40000000; py[i] = -0x40000000; pbase[i] = 0;   // fails every bounds check
    }
  }

  const int n_cchunks = (a.C + BK - 1) / BK;
  const int n_it = 9 * n_cchunks;

  // Tile loader.  Loads go through buffer descriptors: a lane that must read zero (halo pixel, ragged channel or row
  // tail) gets an out-of-range offset and the hardware bounds check returns 0 -- no divergent branch around the load,
  // so the whole K-step stays one basic block that the scheduler can interleave with the MFMAs.
  // (Measured alternative, r01: pointing masked lanes at a zero page made hipcc wait for the loads before the MFMAs
  // and cost 25-45 %.)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
  f32x4 ra[A_PASSES], rb[B_PASSES];
  auto load_tile = [&](int it) {
    const int tap = it / n_cchunks;
    const int c = (it - tap * n_cchunks) * BK + q * 4;
    const int dy = (tap / 3 - 1) * a.dil, dx = (tap % 3 - 1) * a.dil;
    const int shift = (dy * a.W + dx) * a.ld_in + c;
    const bool cok = c < a.C;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      // bitwise (not short-circuit) logic: keeps the predicate a data dependency instead of a branch
      const int ok = (int)cok & (int)((unsigned)(py[i] + dy) < (unsigned)a.H) & (int)((unsigned)(px[i] + dx) < (unsigned)a.W);
      const unsigned off = ok ? (unsigned)(pbase[i] + shift) * (unsigned)PP_ACT_BYTES : 0xffffffffu;
      ra[i] = act_buf_ld4(rs_in, off, 0);
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const int r = r0 + i * RPP;
      const int n = n0 + r;
      const int ok = (int)cok & (int)(r < BN) & (int)(n < a.N);
      const unsigned off = ok ? (unsigned)((n * 9 + tap) * a.C + c) * 4u : 0xffffffffu;
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));
    }
  };
  auto store_tile = [&](int buf) {
    float* Ab = As + buf * BM * LDS_LD;
    float* Bb = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      *reinterpret_cast<f32x4*>(Ab + (r0 + i * RPP) * LDS_LD + q * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const int r = r0 + i * RPP;
      if (r < BN) *reinterpret_cast<f32x4*>(Bb + r * LDS_LD + q * 4) = rb[i];
    }
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

  // One K-step = 4 blocks of (fragment reads + TM*TN*4 MFMAs).  The next tile's global loads are issued after the
  // first MFMA block and its LDS writes after the third, so both sit in the shadow of this wave's own MFMAs
  // instead of in front of / behind them (the other LDS buffer is free from the previous barrier on).
  for (int it = 0; it < n_it; ++it) {
    const int buf = it & 1;
    const bool more = it + 1 < n_it;
    const float* Ab = As + buf * BM * LDS_LD + (wm * TM * 32 + lr) * LDS_LD + lh * 4;
    const float* Bb = Bs + buf * BN * LDS_LD + (wn * TN * 32 + lr) * LDS_LD + lh * 4;
    // register double-buffered fragments: the reads of block kk+1 are issued before the MFMAs of block kk
    f32x4 af[2][TM], bf[2][TN];
    auto read_frags = [&](int kk, int slot) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[slot][i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDS_LD + kk * 8);
    };
    auto mfma_block = [&](int slot) {
      // (k-outer / tile-inner issue order, i.e. consecutive MFMAs on different accumulators, measured the same: r01)
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
#ifndef PP_SCHED
#define PP_SCHED 0
#endif
#if PP_SCHED == 0
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
#elif PP_SCHED == 1
    // no fences: let the machine scheduler interleave
    read_frags(0, 0);
    read_frags(1, 1);
    if (more) load_tile(it + 1);
    mfma_block(0);
    read_frags(2, 0);
    mfma_block(1);
    read_frags(3, 1);
    mfma_block(0);
    if (more) store_tile(buf ^ 1);
    mfma_block(1);
#else
    // explicit interleave: the tile loader's VALU/VMEM are spread over the first MFMA blocks
    read_frags(0, 0);
    read_frags(1, 1);
    if (more) load_tile(it + 1);
    mfma_block(0);
    read_frags(2, 0);
    mfma_block(1);
    read_frags(3, 1);
    mfma_block(0);
    if (more) store_tile(buf ^ 1);
    mfma_block(1);
#pragma unroll
    for (int g = 0; g < TM * TN * 8; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);      // up to 5 VALU
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
    }
#endif
    __syncthreads();
  }

  // epilogue: D[row = (r&3) + 8*(r>>2) + 4*lh][col = lr]
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + lr;
    if (n >= a.N) continue;
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float old[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) old[r] = 0.f;
      if (a.accumulate) {                    // reads first (see conv3x3_igemm_f16x3_kernel)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (p < a.P) old[r] = a.out[(size_t)p * a.ld_out + n];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (p < a.P) a.out[(size_t)p * a.ld_out + n] = acc[i][j][r] + bv + old[r];
      }
    }
  }
}

template <int TM, int TN, int WAVES_M, int WAVES_N>
static int launch_igemm(ConvArgs a, hipStream_t s) {
  constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
  a.m_tiles = pp_cdiv(a.P, BM);
  a.n_tiles = pp_cdiv(a.N, BN);
  const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
  auto kern = conv3x3_igemm_kernel<TM, TN, WAVES_M, WAVES_N>;
  {   // once per (kernel, device): pp_max_lds
    pp_max_lds(reinterpret_cast<const void*>(kern), (int)lds);
  }
  hipLaunchKernelGGL(kern, dim3(a.m_tiles * a.n_tiles), dim3(WAVES_M * WAVES_N * 64), lds, s, a);
  return pp_launch_status("conv3x3_igemm");
}
#endif  // !PP_ACT_16

// ------------------------------------------------------------------------------------------
// Split-fp16 implicit GEMM ("f16x3"): the same convolution on v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate).
// Every fp32 operand x is split into two fp16 numbers, x ~ hi + lo * 2^-11 (hi = rne16(x), lo = rne16((x - hi) * 2^11):
// 22 significand bits, the pre-scaling keeps lo out of the fp16 subnormals), and the product is evaluated as
//     a*b ~ ah*bh + 2^-11 * (ah*bl + al*bh)                        (al*bl ~ 2^-22 |a*b| is dropped)
// with fp32 accumulation in two accumulator sets (main, cross).  Products of fp16 numbers are exact in the MFMA's fp32
// accumulation, so the only errors are the 2^-23-relative representation error and the dropped term -- the same size
// as fp32 rounding.  Measured on the whole network (scripts/split_precision_study.py): logits differ from fp64 by
// 1.7e-5 with this path against 1.8e-5 with plain fp32 (the 3-product bf16 split gives 3.3e-4 and fails the 1e-4 bar).
// LDS rows hold [32 hi | 32 lo] halves = the same 128 B (+16 pad) as the fp32 tile, so staging traffic is unchanged;
// weights are split once per step by pack_weights_f16x3_kernel, activations on the fly while staging.
// Dynamic range: fp16 tops out at 65504 and loses precision below 6e-5, so an operand whose magnitude is not O(1)
// (gradients) is multiplied by a power of two taken from its amax (a device scalar written by the producer) and the
// result is scaled back in the epilogue.
// ------------------------------------------------------------------------------------------
template <int TM, int TN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) __attribute__((amdgpu_waves_per_eu(2)))
void conv3x3_igemm_f16x3_kernel(ConvArgs a, const float* in_amax) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = 32 * TM * WAVES_M;
  constexpr int BN = 32 * TN * WAVES_N;
  constexpr int RPP = NT / 8;
  constexpr int A_PASSES = BM / RPP;
  constexpr int B_PASSES = (BN + RPP - 1) / RPP;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) _Float16 smem16[];
  _Float16* As = smem16;                       // [2][BM][H_LD]
  _Float16* Bs = smem16 + 2 * BM * H_LD;       // [2][BN][H_LD]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;
  const int q = tid & 7, r0 = tid >> 3;
  int mt, nt;
  tile_of_block(blockIdx.x, a.m_tiles, a.n_tiles, mt, nt);
  const int m0 = mt * BM, n0 = nt * BN;
  float s_in, s_out;
  f16_scales(in_amax, s_in, s_out);

  int py[A_PASSES], px[A_PASSES], pbase[A_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int p = m0 + r0 + i * RPP;
    if (p < a.P) {
      px[i] = p % a.W;
      py[i] = (p / a.W) % a.H;
      pbase[i] = p * a.ld_in;
    } else {
      px[i] = -0x40000000; py[i] = -0x40000000; pbase[i] = 0;
    }
  }
  const int n_cchunks = (a.C + BK - 1) / BK;
  const int n_it = 9 * n_cchunks;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
  f32x4 ra[A_PASSES], rb[B_PASSES];
  auto load_tile = [&](int it) {                 // it == n_it (one past the end): every lane reads zeros, no branch
    const int tap = it / n_cchunks;
    const int c = (it - tap * n_cchunks) * BK + q * 4;
    const int dy = (tap / 3 - 1) * a.dil, dx = (tap % 3 - 1) * a.dil;
    const int shift = (dy * a.W + dx) * a.ld_in + c;
    const bool cok = (c < a.C) & (it < n_it);
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int ok = (int)cok & (int)((unsigned)(py[i] + dy) < (unsigned)a.H) & (int)((unsigned)(px[i] + dx) < (unsigned)a.W);
      const unsigned off = ok ? (unsigned)(pbase[i] + shift) * (unsigned)PP_ACT_BYTES : 0xffffffffu;
      ra[i] = act_buf_ld4(rs_in, off, 0);
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const int r = r0 + i * RPP;
      const int n = n0 + r;
      const int ok = (int)cok & (int)(r < BN) & (int)(n < a.N);
      const unsigned off = ok ? (unsigned)((n * 9 + tap) * a.C + c) * 4u : 0xffffffffu;   // [hi4 | lo4] per channel quad
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));
    }
  };
  auto store_tile = [&](int buf) {
    _Float16* Ab = As + buf * BM * H_LD;
    _Float16* Bb = Bs + buf * BN * H_LD;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const f32x4 v = ra[i] * s_in;
      const f16x4 hi = __builtin_convertvector(v, f16x4);
      _Float16* d = Ab + (r0 + i * RPP) * H_LD + q * 4;
      *reinterpret_cast<f16x4*>(d) = hi;
      if (PP_ACT_LO) {
        const f16x4 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * F16_LO_SCALE, f16x4);
        *reinterpret_cast<f16x4*>(d + 32) = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const int r = r0 + i * RPP;
      if (r < BN) {
        _Float16* d = Bb + r * H_LD + q * 4;
        *reinterpret_cast<f32x2*>(d) = __builtin_shufflevector(rb[i], rb[i], 0, 1);          // hi0..hi3
        *reinterpret_cast<f32x2*>(d + 32) = __builtin_shufflevector(rb[i], rb[i], 2, 3);     // lo0..lo3
      }
    }
  };

  f32x16 accm[TM][TN], accc[TM][TN];             // main (hi*hi) and cross (hi*lo + lo*hi, weight 2^-11) accumulators
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.f; accc[i][j][r] = 0.f; }

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int it = 0; it < n_it; ++it) {
    const int buf = it & 1;
    const _Float16* Ap = As + buf * BM * H_LD + (wm * TM * 32 + lr) * H_LD + lh * 8;
    const _Float16* Bp = Bs + buf * BN * H_LD + (wn * TN * 32 + lr) * H_LD + lh * 8;
    // Two 16-channel MFMA steps per 32-channel stage.  All fragments of the stage are read up front, the next tile's
    // global loads are issued in front of the first MFMA block, and its fp32 -> hi/lo conversion + LDS writes are
    // interleaved with the second block's MFMAs (one MFMA : a few VALU : one DS write) instead of running behind
    // them with the matrix pipe idle (r02 profile: 26-37 % MFMA utilisation with the serial order).
    f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[kb][i] = *reinterpret_cast<const f16x8*>(Ap + i * 32 * H_LD + kb * 16);
        if (PP_ACT_LO) al[kb][i] = *reinterpret_cast<const f16x8*>(Ap + i * 32 * H_LD + kb * 16 + 32);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[kb][j] = *reinterpret_cast<const f16x8*>(Bp + j * 32 * H_LD + kb * 16);
        bl[kb][j] = *reinterpret_cast<const f16x8*>(Bp + j * 32 * H_LD + kb * 16 + 32);
      }
    }
    load_tile(it + 1);                           // unconditional (see load_tile): the K-step stays ONE basic block
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][i], bh[0][j], accm[i][j], 0, 0, 0);
        accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][i], bl[0][j], accc[i][j], 0, 0, 0);
        if (PP_ACT_LO) accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0][i], bh[0][j], accc[i][j], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    store_tile(buf ^ 1);                         // after the last step this writes zeros nobody reads
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][i], bh[1][j], accm[i][j], 0, 0, 0);
        accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][i], bl[1][j], accc[i][j], 0, 0, 0);
        if (PP_ACT_LO) accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[1][i], bh[1][j], accc[i][j], 0, 0, 0);
      }
#pragma unroll
    for (int g = 0; g < TM * TN * 3; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);      // up to 6 VALU of the conversion
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);      // up to 2 DS writes
    }
    __syncthreads();
  }

  float st_s[TN], st_q[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    st_s[j] = 0.f; st_q[j] = 0.f;
    const int n = n0 + (wn * TN + j) * 32 + lr;
    if (n >= a.N) continue;
    const float bv = a.bias ? a.bias[n] : 0.f;
    const float e_sc = a.epi.mode == 2 ? a.epi.scale[n] : 1.f, e_sh = a.epi.mode == 2 ? a.epi.shift[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float old[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) old[r] = 0.f;
      if (a.accumulate) {                    // all reads of the tile first: read-add-write per element serialises 16 round trips
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (p < a.P) old[r] = a.out[(size_t)p * a.ld_out + n];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (p < a.P) {
          act_t* o = a.out + (size_t)p * a.ld_out + n;
          float v = (accm[i][j][r] + accc[i][j][r] * (1.f / F16_LO_SCALE)) * s_out + bv;
          if (a.epi.mode == 1) { st_s[j] += v; st_q[j] += v * v; }
          if (a.epi.mode == 2) { v = v * e_sc + e_sh; v = fmaxf(v, v * a.epi.slope); }
          *o = (act_t)(v + old[r]);
        }
      }
    }
  }
  if (a.epi.mode == 1) {
    // one partial row per m-tile (the host guarantees that an m-tile never straddles two groups): fold the two lane
    // halves, then the WAVES_M waves that share this block's channels, in a fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem16);             // [WAVES_M][2][BN]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float ss = st_s[j] + __shfl_xor(st_s[j], 32, 64), qq = st_q[j] + __shfl_xor(st_q[j], 32, 64);
      if (lh == 0) {
        red[(wm * 2 + 0) * BN + (wn * TN + j) * 32 + lr] = ss;
        red[(wm * 2 + 1) * BN + (wn * TN + j) * 32 + lr] = qq;
      }
    }
    __syncthreads();
    const int tiles_per_group = a.epi.px_per_group / BM;
    const int g = mt / tiles_per_group, row = mt - g * tiles_per_group;
    for (int e = tid; e < 2 * BN; e += NT) {
      const int which = e / BN, c = e - which * BN;
      if (n0 + c < a.N) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) acc += (double)red[(w * 2 + which) * BN + c];
        pp_epi_row(a.epi, g, row, which, a.N)[n0 + c] = acc;
      }
    }
  }
}

template <int TM, int TN, int WAVES_M, int WAVES_N>
static int launch_igemm_f16x3(ConvArgs a, const float* in_amax, hipStream_t s) {
  constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
  a.m_tiles = pp_cdiv(a.P, BM);
  a.n_tiles = pp_cdiv(a.N, BN);
  const size_t lds = (size_t)2 * (BM + BN) * H_LD * sizeof(_Float16);
  auto kern = conv3x3_igemm_f16x3_kernel<TM, TN, WAVES_M, WAVES_N>;
  {   // once per (kernel, device): pp_max_lds
    pp_max_lds(reinterpret_cast<const void*>(kern), (int)lds);
  }
  hipLaunchKernelGGL(kern, dim3(a.m_tiles * a.n_tiles), dim3(WAVES_M * WAVES_N * 64), lds, s, a, in_amax);
  return pp_launch_status("conv3x3_igemm_f16x3");
}

// ------------------------------------------------------------------------------------------
// Narrow layers (Cin <= 96, high resolution): persistent halo-tile kernel.
// The implicit-GEMM kernel above re-stages every input pixel once per tap and has a 9..27 step K loop per block, so on
// the 32..96-channel 256^2 / 128^2 layers its prologue / epilogue and L2->LDS traffic show (MFMA pipes 65 % busy,
// r01 PMC profile).  Here a block keeps ALL taps of its 32 output channels' weights in LDS for its whole life and
// walks over 4 x 32-pixel output tiles: per tile and 32-channel chunk the 6 x 34 halo patch is staged ONCE
// (prefetched into registers during the previous stage's MFMAs) and the nine taps read their A fragments from it at
// shifted addresses -- 144 MFMAs per wave between barriers, no re-reads, stores overlapped with the next tile.
// Same operands as conv3x3_igemm_kernel (w = [N][9][C]), so forward and dgrad both use it.
// ------------------------------------------------------------------------------------------
#define HT_ROWS 4
#define HT_COLS 32
#define HT_HC (HT_COLS + 2)
#define HT_PIX ((HT_ROWS + 2) * HT_HC)             // 204 halo pixels
#define HT_APASS ((HT_PIX * 8 + 255) / 256)         // float4 loads per thread per stage (7)

#ifndef PP_ACT_16
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(ConvArgs a, int n_chunks, int tiles_x, int tiles_y,
                                                           int n_tiles) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bs = smem;                                        // [n_chunks][9][32][LDS_LD]
  float* As = smem + n_chunks * 9 * 32 * LDS_LD;           // [HT_PIX][LDS_LD]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * 32;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // weights of this block's 32 output channels, all taps and chunks: staged once
  {
    const int total = n_chunks * 9 * 32 * 8;
    for (int e = tid; e < total; e += 256) {
      const int q = e & 7, row = e >> 3;                   // row = (chunk * 9 + tap) * 32 + n
      const int n = row & 31, ct = row >> 5, tap = ct % 9, chunk = ct / 9;
      const unsigned off = (n0 + n < a.N) ? (unsigned)(((n0 + n) * 9 + tap) * a.C + chunk * 32 + q * 4) * 4u : 0xffffffffu;
      *reinterpret_cast<f32x4*>(Bs + row * LDS_LD + q * 4) =
          __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));
    }
  }

  // per-thread halo bookkeeping (tile independent)
  int hy[HT_APASS], hx[HT_APASS], rel[HT_APASS], lds_off[HT_APASS];
#pragma unroll
  for (int i = 0; i < HT_APASS; ++i) {
    const int e = tid + 256 * i, pix = e >> 3, q = e & 7;
    if (pix < HT_PIX) {
      hy[i] = pix / HT_HC;
      hx[i] = pix - hy[i] * HT_HC;
      rel[i] = (hy[i] * a.W + hx[i]) * a.ld_in + q * 4;
      lds_off[i] = pix * LDS_LD + q * 4;
    } else {
      hy[i] = -0x40000000; hx[i] = -0x40000000; rel[i] = 0; lds_off[i] = -1;
    }
  }
  f32x4 ra[HT_APASS];
  auto load_patch = [&](int t, int chunk) {
    const int tx = t % tiles_x, r = t / tiles_x, ty = r % tiles_y, img = r / tiles_y;
    const int y0 = ty * HT_ROWS - 1, x0 = tx * HT_COLS - 1;
    const int base = ((img * a.H + y0) * a.W + x0) * a.ld_in + chunk * 32;
#pragma unroll
    for (int i = 0; i < HT_APASS; ++i) {
      const int ok = (int)((unsigned)(y0 + hy[i]) < (unsigned)a.H) & (int)((unsigned)(x0 + hx[i]) < (unsigned)a.W);
      const unsigned off = ok ? (unsigned)(base + rel[i]) * (unsigned)PP_ACT_BYTES : 0xffffffffu;
      ra[i] = act_buf_ld4(rs_in, off, 0);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < HT_APASS; ++i)
      if (lds_off[i] >= 0) *reinterpret_cast<f32x4*>(As + lds_off[i]) = ra[i];
  };

  const float bv = (a.bias && n0 + lr < a.N) ? a.bias[n0 + lr] : 0.f;
  const float* Ab = As + (wv * HT_HC + lr) * LDS_LD + lh * 4;
  const bool n_ok = n0 + lr < a.N;
  auto out_row = [&](int t) -> float* {
    const int tx = t % tiles_x, rr = t / tiles_x, ty = rr % tiles_y, img = rr / tiles_y;
    return a.out + ((size_t)(img * a.H + ty * HT_ROWS + wv) * a.W + tx * HT_COLS) * a.ld_out + n0 + lr;
  };
  // The finished tile's accumulators are written out one stage LATE (from `pend`, right after the next stage's
  // barriers) and, in accumulate mode, the old output values are fetched one stage EARLY (`old`): hipcc's
  // __syncthreads() waits vmcnt(0), i.e. also for outstanding stores, so neither sits directly in front of a barrier.
  f32x16 pend, old;
  int pend_t = -1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { pend[r] = 0.f; old[r] = 0.f; }
  auto write_pending = [&]() {
    if (pend_t >= 0 && n_ok) {
      float* orow = out_row(pend_t);
#pragma unroll
      for (int r = 0; r < 16; ++r) orow[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * a.ld_out] = pend[r] + bv + old[r];
    }
    pend_t = -1;
  };
  auto fetch_old = [&](int t) {
    if (a.accumulate && n_ok) {
      const float* orow = out_row(t);
#pragma unroll
      for (int r = 0; r < 16; ++r) old[r] = orow[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * a.ld_out];
    }
  };

  int t = blockIdx.x;
  if (t < n_tiles) load_patch(t, 0);
  for (; t < n_tiles; t += gridDim.x) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
      __syncthreads();                       // every wave is done reading the previous patch (and, first time, Bs is written)
      store_patch();
      __syncthreads();
      // prefetch the next stage's patch into registers; it is consumed after this stage's 144 MFMAs
      {
        const bool last_chunk = chunk + 1 == n_chunks;
        const int nt = last_chunk ? t + (int)gridDim.x : t;
        if (nt < n_tiles) load_patch(nt, last_chunk ? 0 : chunk + 1);
      }
      if (chunk == 0) write_pending();
      if (chunk + 1 == n_chunks) fetch_old(t);
      const float* Bb = Bs + (chunk * 9 * 32 + lr) * LDS_LD + lh * 4;
      f32x4 af[2], bf[2];
      af[0] = *reinterpret_cast<const f32x4*>(Ab);
      bf[0] = *reinterpret_cast<const f32x4*>(Bb);
#pragma unroll
      for (int st = 0; st < 36; ++st) {      // st = tap * 4 + k-block
        const int cur = st & 1;
        if (st + 1 < 36) {
          const int tap = (st + 1) >> 2, kk = (st + 1) & 3;
          af[cur ^ 1] = *reinterpret_cast<const f32x4*>(Ab + ((tap / 3) * HT_HC + tap % 3) * LDS_LD + kk * 8);
          bf[cur ^ 1] = *reinterpret_cast<const f32x4*>(Bb + tap * 32 * LDS_LD + kk * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        // one accumulator chain is enough: two independent chains measured the same (r01)
        for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][k], bf[cur][k], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    pend = acc;          // D[row = pixel x = (r&3) + 8*(r>>2) + 4*lh][col = channel lr]
    pend_t = t;
  }
  write_pending();
}
#endif  // !PP_ACT_16

// ------------------------------------------------------------------------------------------
// First layer (image padded to 4 channels, models/unet.py:188 with in_ch = 1): K = 36 is far too short for the
// implicit-GEMM tile (a 32-channel K-step is 7/8 zeros; 0.82 ms per launch at the benchmark shape, r01), and the
// layer is HBM-bound anyway (reads 16 B, writes 4*Cout B per pixel).  On v_mfma_f32_16x16x4_f32 one tap is exactly
// one instruction: A[row n][k = c] = w[n][tap][c] (all taps resident in 9 VGPRs per 16 output channels),
// B[k = c][col = pixel] = x[pixel + off(tap)][c] loaded straight from global (16 pixels x 16 B contiguous), and
// D[row n][col pixel] leaves each lane with 4 consecutive channels of one pixel -> float4 stores.  No LDS.
// ------------------------------------------------------------------------------------------
template <int MT>              // MT = N / 16 (1..4)
__global__ __launch_bounds__(256) void conv3x3_c4_fwd_kernel(ConvArgs a, int groups_per_wave) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l16 = lane & 15, k = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  float wa[MT][9];
  f32x4 binit[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) wa[mt][tap] = a.w[((size_t)(mt * 16 + l16) * 9 + tap) * 4 + k];
#pragma unroll
    for (int r = 0; r < 4; ++r) binit[mt][r] = a.bias ? a.bias[mt * 16 + 4 * k + r] : 0.f;
  }
  f32x4 e_sc[MT], e_sh[MT], st_s0[MT], st_q0[MT], st_s1[MT], st_q1[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      e_sc[mt][r] = a.epi.mode == 2 ? a.epi.scale[mt * 16 + 4 * k + r] : 1.f;
      e_sh[mt][r] = a.epi.mode == 2 ? a.epi.shift[mt * 16 + 4 * k + r] : 0.f;
      st_s0[mt][r] = 0.f; st_q0[mt][r] = 0.f; st_s1[mt][r] = 0.f; st_q1[mt][r] = 0.f;
    }
  }
  int boff[9], bdy[9], bdx[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    bdy[tap] = (tap / 3 - 1) * a.dil;
    bdx[tap] = (tap % 3 - 1) * a.dil;
    boff[tap] = (bdy[tap] * a.W + bdx[tap]) * a.ld_in + k;
  }
  const int n_groups = a.P >> 4;                   // W % 16 == 0: a group of 16 pixels never straddles a row
  const long long w_id = (long long)blockIdx.x * 4 + wv;
  const int g0 = (int)(w_id * groups_per_wave);
  int g1 = g0 + groups_per_wave;
  if (g1 > n_groups) g1 = n_groups;
  for (int g = g0; g < g1; g += 2) {
    float bv[2][9];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q0 = (g + u) << 4;
      const int live = g + u < g1;
      const int x0 = q0 % a.W, y0 = (q0 / a.W) % a.H;
      const int p = q0 + l16;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ok = live & (int)((unsigned)(y0 + bdy[tap]) < (unsigned)a.H) & (int)((unsigned)(x0 + l16 + bdx[tap]) < (unsigned)a.W);
        const unsigned off = ok ? (unsigned)(p * a.ld_in + boff[tap]) * (unsigned)PP_ACT_BYTES : 0xffffffffu;
        bv[u][tap] = act_buf_ld1(rs_in, off, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (g + u >= g1) break;
      const int p = ((g + u) << 4) + l16;
      const bool second = a.epi.mode == 1 && ((g + u) << 4) >= a.epi.px_per_group;     // group of these 16 pixels
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4 acc = binit[mt];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][tap], bv[u][tap], acc, 0, 0, 0);
        if (a.epi.mode == 1) {
          if (second) { st_s1[mt] += acc; st_q1[mt] += acc * acc; } else { st_s0[mt] += acc; st_q0[mt] += acc * acc; }
        }
        if (a.epi.mode == 2) {
          acc = acc * e_sc[mt] + e_sh[mt];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], acc[r] * a.epi.slope);
        }
        act_t* o = a.out + (size_t)p * a.ld_out + mt * 16 + 4 * k;   // D[row = 4*k + r][col = l16]
        act_st4(o, a.accumulate ? act_ld4(o) + acc : acc);
      }
    }
  }
  if (a.epi.mode == 1) {
    // fold the 16 pixel lanes of every channel quad, then the four waves: one partial row per block
    __shared__ float red[4][PP_EPI_GROUPS][2][MT * 16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v0 = st_s0[mt][r], v1 = st_q0[mt][r], v2 = st_s1[mt][r], v3 = st_q1[mt][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          v0 += __shfl_xor(v0, o, 64); v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64); v3 += __shfl_xor(v3, o, 64);
        }
        if (l16 == 0) {
          const int c = mt * 16 + 4 * k + r;
          red[wv][0][0][c] = v0; red[wv][0][1][c] = v1; red[wv][1][0][c] = v2; red[wv][1][1][c] = v3;
        }
      }
    __syncthreads();
    for (int e = threadIdx.x; e < a.epi.groups * 2 * MT * 16; e += 256) {
      const int c = e % (MT * 16), which = (e / (MT * 16)) & 1, gg = e / (2 * MT * 16);
      const double acc = ((double)red[0][gg][which][c] + (double)red[1][gg][which][c]) + ((double)red[2][gg][which][c] + (double)red[3][gg][which][c]);
      pp_epi_row(a.epi, gg, blockIdx.x, which, a.N)[c] = acc;
    }
  }
}

static inline bool c4_eligible(const ConvArgs& a) {
  return a.C == 4 && a.N % 16 == 0 && a.N <= 64 && a.W % 16 == 0 && a.ld_out % 4 == 0 && ((uintptr_t)a.out & PP_ACT_ALIGN) == 0;
}

static int c4_blocks(const ConvArgs& a, int* gpw_out) {
  const int n_groups = a.P / 16;
  int waves = pp_cdiv(n_groups, 8);                // >= 8 groups (128 pixels) per wave
  constexpr int cap = 256 * 32;
  if (waves > cap) waves = cap;
  const int gpw = pp_cdiv(pp_cdiv(n_groups, waves), 2) * 2;
  if (gpw_out) *gpw_out = gpw;
  return pp_cdiv(pp_cdiv(n_groups, gpw), 4);
}

static int launch_c4(ConvArgs a, hipStream_t s) {
  int gpw;
  const int blocks = c4_blocks(a, &gpw);
  switch (a.N / 16) {
    case 1: hipLaunchKernelGGL(conv3x3_c4_fwd_kernel<1>, dim3(blocks), dim3(256), 0, s, a, gpw); break;
    case 2: hipLaunchKernelGGL(conv3x3_c4_fwd_kernel<2>, dim3(blocks), dim3(256), 0, s, a, gpw); break;
    case 3: hipLaunchKernelGGL(conv3x3_c4_fwd_kernel<3>, dim3(blocks), dim3(256), 0, s, a, gpw); break;
    default: hipLaunchKernelGGL(conv3x3_c4_fwd_kernel<4>, dim3(blocks), dim3(256), 0, s, a, gpw); break;
  }
  return pp_launch_status("conv3x3_c4_fwd");
}

// ------------------------------------------------------------------------------------------
// The halo-tile kernel on the fp16 MFMA with split operands ("f16x3", see conv3x3_igemm_f16x3_kernel): the 128 x 32
// f16x3 implicit-GEMM tile re-fetches every input pixel for each tap (3.06 GB of HBM traffic per launch on the
// 32-channel 256^2 layers against 1.07 GB algorithmic, r01 PMC profile); here the patch is staged once per tile and
// the pre-split weights of the block's 32 output channels stay in LDS.  TMR = output rows per wave (tile = 4*TMR rows
// x 32 columns): 2 halves the weight-fragment reads per MFMA; 1 is used when three channel chunks of weights leave no
// room for the larger patch.
// ------------------------------------------------------------------------------------------
// -DPP_HALO_TRACE (tests/studies/halo_phase_trace.py builds a second library with it): wave 0 of block (0, 0) accumulates
// the shader-clock cycles of every phase of a stage and leaves them in pp_halo_trace for pp_debug_halo_trace().
#ifdef PP_HALO_TRACE
__device__ long long pp_halo_trace[16];
#define HT_TRK(k) { const long long now_ = __builtin_readcyclecounter(); tr[k] += now_ - t_prev; t_prev = now_; }
#else
#define HT_TRK(k)
#endif
template <int TMR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void conv3x3_halo_f16x3_kernel(ConvArgs a, int n_chunks, int tiles_x, int tiles_y, int n_tiles, const float* in_amax, int chunk0) {
  // chunk0: first 32-channel chunk of this launch (split-K: a layer with more than 96 input channels runs as two launches over
  // channel chunks [0, n) and [n, 2n), the second accumulating into the first's output; a.C stays the layer's channel count)
  constexpr int ROWS = 4 * TMR, PIX = (ROWS + 2) * HT_HC, APASS = (PIX * 8 + 255) / 256;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) _Float16 smem16[];
  _Float16* Bs = smem16;                                   // [n_chunks][9][32][H_LD]  pre-split weights
  _Float16* As = smem16 + n_chunks * 9 * 32 * H_LD;        // [PIX][H_LD]              patch, hi | lo per row
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * 32;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
  float s_in, s_out;
  f16_scales(in_amax, s_in, s_out);
  {
    const int total = n_chunks * 9 * 32 * 8;
    for (int e = tid; e < total; e += 256) {
      const int q = e & 7, row = e >> 3;                   // row = (chunk * 9 + tap) * 32 + n
      const int n = row & 31, ct = row >> 5, tap = ct % 9, chunk = ct / 9;
      const unsigned off = (n0 + n < a.N) ? (unsigned)(((n0 + n) * 9 + tap) * a.C + (chunk0 + chunk) * 32 + q * 4) * 4u : 0xffffffffu;
      const f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));   // [hi4 | lo4]
      _Float16* d = Bs + row * H_LD + q * 4;
      *reinterpret_cast<f32x2*>(d) = __builtin_shufflevector(w, w, 0, 1);
      *reinterpret_cast<f32x2*>(d + 32) = __builtin_shufflevector(w, w, 2, 3);
    }
  }
  int hy[APASS], hx[APASS], rel[APASS], lds_off[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int e = tid + 256 * i, pix = e >> 3, q = e & 7;
    if (pix < PIX) {
      hy[i] = pix / HT_HC;
      hx[i] = pix - hy[i] * HT_HC;
      rel[i] = (hy[i] * a.W + hx[i]) * a.ld_in + q * 4;
      lds_off[i] = pix * H_LD + q * 4;
    } else {
      hy[i] = -0x40000000; hx[i] = -0x40000000; rel[i] = 0; lds_off[i] = -1;
    }
  }
  // Two register sets of prefetched patches: the loads for stage s + 2 are issued during stage s.  One stage of MFMAs
  // (54 x 32 cycles ~ 0.9 us) is shorter than the HBM latency under load, and the 64 / 96-channel layers fit only ONE
  // block per CU (weights of 2-3 chunks in LDS), so with a prefetch distance of one stage the matrix pipe sat idle for
  // two thirds of every stage (r02 per-dispatch trace: 64 -> 64 at 128^2 took 331 us against a 115 us MFMA bound).
  f32x4 ra0[APASS], ra1[APASS];
  // stages of this block in order: s = (k-th tile of the block) * n_chunks + chunk.  The stage loop is unrolled by two
  // (one register set each) and EVERY path through it issues the same loads -- a stage past the end is a "ghost" whose
  // loads are all out of range (the buffer descriptor returns zeros without touching memory) and which writes nothing
  // -- because hipcc's s_waitcnt insertion merges control-flow paths conservatively: with a conditional load in the
  // prologue or the loop it waited for vmcnt(0) in front of every patch store, i.e. for the set issued one stage ago.
  const int my_tiles = (int)blockIdx.x < n_tiles ? (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int total = my_tiles * n_chunks;
  auto stage_tile = [&](int s) { return (int)blockIdx.x + (s / n_chunks) * (int)gridDim.x; };
  auto load_patch = [&](f32x4 (&ra)[APASS], int s) {
    const bool live = s < total;
    const int t = live ? stage_tile(s) : 0, chunk = s % n_chunks;
    const int tx = t % tiles_x, r = t / tiles_x, ty = r % tiles_y, img = r / tiles_y;
    const int y0 = ty * ROWS - 1, x0 = tx * HT_COLS - 1;
    const int base = ((img * a.H + y0) * a.W + x0) * a.ld_in + (chunk0 + chunk) * 32;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int ok = (int)((unsigned)(y0 + hy[i]) < (unsigned)a.H) & (int)((unsigned)(x0 + hx[i]) < (unsigned)a.W) & (int)live;
      const unsigned off = ok ? (unsigned)(base + rel[i]) * (unsigned)PP_ACT_BYTES : 0xffffffffu;
      ra[i] = act_buf_ld4(rs_in, off, 0);
    }
  };
  auto store_patch = [&](f32x4 (&ra)[APASS]) {
#pragma unroll
    for (int i = 0; i < APASS; ++i)
      if (lds_off[i] >= 0) {
        const f32x4 v = ra[i] * s_in;
        const f16x4 hi = __builtin_convertvector(v, f16x4);
        *reinterpret_cast<f16x4*>(As + lds_off[i]) = hi;
        if (PP_ACT_LO) {
          const f16x4 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * F16_LO_SCALE, f16x4);
          *reinterpret_cast<f16x4*>(As + lds_off[i] + 32) = lo;
        }
      }
  };
  const bool n_ok = n0 + lr < a.N;
  const float bv = (a.bias && n_ok) ? a.bias[n0 + lr] : 0.f;
  const float e_sc = (a.epi.mode == 2 && n_ok) ? a.epi.scale[n0 + lr] : 1.f;
  const float e_sh = (a.epi.mode == 2 && n_ok) ? a.epi.shift[n0 + lr] : 0.f;
  float st_s0 = 0.f, st_q0 = 0.f, st_s1 = 0.f, st_q1 = 0.f;
  const _Float16* Ab = As + (wv * TMR * HT_HC + lr) * H_LD + lh * 8;
  auto out_row = [&](int t, int i) -> act_t* {
    const int tx = t % tiles_x, rr = t / tiles_x, ty = rr % tiles_y, img = rr / tiles_y;
    return a.out + ((size_t)(img * a.H + ty * ROWS + wv * TMR + i) * a.W + tx * HT_COLS) * a.ld_out + n0 + lr;
  };
  f32x16 pend[TMR];
  int pend_t = -1;
#pragma unroll
  for (int i = 0; i < TMR; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) pend[i][r] = 0.f;
  // finished tiles are written out one stage late, right after the next stage's barriers (no store in front of a barrier)
  auto write_pending = [&]() {
    if (pend_t >= 0 && n_ok) {
#pragma unroll
      for (int i = 0; i < TMR; ++i) {
        act_t* orow = out_row(pend_t, i);
        if (a.accumulate && !a.epi.mode) {   // all 16 reads first: read-add-write per element is 16 serial round trips
          float old[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) old[r] = orow[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * a.ld_out];
#pragma unroll
          for (int r = 0; r < 16; ++r) orow[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * a.ld_out] = (act_t)(old[r] + pend[i][r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) orow[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * a.ld_out] = (act_t)pend[i][r];
        }
      }
    }
    pend_t = -1;
  };
  f32x16 accm[TMR], accc[TMR];
#ifdef PP_HALO_TRACE
  long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_begin = __builtin_readcyclecounter(), w_begin = __builtin_amdgcn_s_memrealtime();
  long long t_prev = t_begin;
#endif
  auto stage = [&](int s, f32x4 (&ra)[APASS]) {
    const bool live = s < total;
    const int t = stage_tile(s), chunk = s % n_chunks;
    __syncthreads();
    HT_TRK(0)
    store_patch(ra);
    HT_TRK(1)
    __syncthreads();
    HT_TRK(2)
    if (chunk == 0) {
      write_pending();
#pragma unroll
      for (int i = 0; i < TMR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accm[i][r] = 0.f; accc[i][r] = 0.f; }
    }
    HT_TRK(3)
    load_patch(ra, s + 2);
    HT_TRK(4)
    const _Float16* Bb = Bs + (chunk * 9 * 32 + lr) * H_LD + lh * 8;
    // 18 steps (tap, 16-channel block) of 3 MFMAs per output row.  Register double-buffered fragments: the four
    // ds_read_b128 of step s+1 are issued before the MFMAs of step s (hipcc left to itself issued each step's reads
    // directly in front of its MFMAs and waited for them: 36 % of the matrix pipe, r02 profile).
    f16x8 ah[2][TMR], al[2][TMR], bh[2], bl[2];
    auto read_step = [&](int st, int slot) {
      const int tap = st >> 1, kb = st & 1;
      bh[slot] = *reinterpret_cast<const f16x8*>(Bb + tap * 32 * H_LD + kb * 16);
      bl[slot] = *reinterpret_cast<const f16x8*>(Bb + tap * 32 * H_LD + kb * 16 + 32);
#pragma unroll
      for (int i = 0; i < TMR; ++i) {
        const _Float16* ap = Ab + ((i + tap / 3) * HT_HC + tap % 3) * H_LD + kb * 16;
        ah[slot][i] = *reinterpret_cast<const f16x8*>(ap);
        if (PP_ACT_LO) al[slot][i] = *reinterpret_cast<const f16x8*>(ap + 32);
      }
    };
    read_step(0, 0);
#pragma unroll
    for (int st = 0; st < 18; ++st) {      // st = tap * 2 + 16-channel block
      const int cur = st & 1;
      if (st + 1 < 18) read_step(st + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TMR; ++i) {
        accm[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bh[cur], accm[i], 0, 0, 0);
        accc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bl[cur], accc[i], 0, 0, 0);
        if (PP_ACT_LO) accc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][i], bh[cur], accc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    HT_TRK(5)
    if (chunk + 1 == n_chunks && live) {
      float ts = 0.f, tq = 0.f;
      float prev[TMR][16];
      if (a.accumulate && a.epi.mode && n_ok) {     // second split-K launch with a fused epilogue: the epilogue needs the full sum
#pragma unroll
        for (int i = 0; i < TMR; ++i) {
          const act_t* orow = out_row(t, i);
#pragma unroll
          for (int r = 0; r < 16; ++r) prev[i][r] = orow[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * a.ld_out];
        }
      } else {
#pragma unroll
        for (int i = 0; i < TMR; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) prev[i][r] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < TMR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = (accm[i][r] + accc[i][r] * (1.f / F16_LO_SCALE)) * s_out + bv + prev[i][r];
          if (a.epi.mode == 1) { ts += v; tq += v * v; }                       // BatchNorm batch statistics of z
          if (a.epi.mode == 2) { v = v * e_sc + e_sh; v = fmaxf(v, v * a.epi.slope); }   // eval-mode BN + LeakyReLU
          pend[i][r] = v;
        }
      if (a.epi.mode == 1) {                 // group (weak | strong half of the batch) of this tile's image
        if ((t / (tiles_x * tiles_y)) * (a.H * a.W) >= a.epi.px_per_group) { st_s1 += ts; st_q1 += tq; }
        else { st_s0 += ts; st_q0 += tq; }
      }
      pend_t = t;
    }
    HT_TRK(6)
  };
  load_patch(ra0, 0);
  load_patch(ra1, 1);
  for (int s = 0; s < total; s += 2) {
    stage(s, ra0);
    stage(s + 1, ra1);
  }
#ifdef PP_HALO_TRACE
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    for (int k = 0; k < 7; ++k) pp_halo_trace[k] = tr[k];
    for (int k = 7; k < 10; ++k) pp_halo_trace[k] = 0;
    pp_halo_trace[10] = total;
    pp_halo_trace[11] = __builtin_readcyclecounter() - t_begin;
    pp_halo_trace[12] = __builtin_amdgcn_s_memrealtime() - w_begin;       // 100 MHz
    pp_halo_trace[13] = 1;                                                 // kernel id
  }
#endif
  write_pending();
  if (a.epi.mode == 1) {
    // per-channel partial sums of this block: lanes lr / lr + 32 hold the same channel, the four waves four rows
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem16);             // [4 waves][2 groups][2][32]
#pragma unroll
    for (int g = 0; g < PP_EPI_GROUPS; ++g) {
      const float sg = g ? st_s1 : st_s0, qg = g ? st_q1 : st_q0;
      const float ss = sg + __shfl_xor(sg, 32, 64), qq = qg + __shfl_xor(qg, 32, 64);
      if (lh == 0) { red[((wv * 2 + g) * 2 + 0) * 32 + lr] = ss; red[((wv * 2 + g) * 2 + 1) * 32 + lr] = qq; }
    }
    __syncthreads();
    if (tid < 128 && (tid >> 6) < a.epi.groups && n0 + (tid & 31) < a.N) {
      const int g = tid >> 6, which = (tid >> 5) & 1, c = tid & 31;
      double acc = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) acc += (double)red[((w * 2 + g) * 2 + which) * 32 + c];
      pp_epi_row(a.epi, g, blockIdx.x, which, a.N)[n0 + c] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------
// conv3x3_halo2_f16x3_kernel: the halo-tile kernel with TWO half-blocks of four waves working in anti-phase.
//
// Phase trace of conv3x3_halo_f16x3_kernel (tests/studies/halo_phase_trace.py, r02, 64 -> 64 at 128^2): of the 4984
// cycles of a stage only 1815 are the 54 MFMAs; patch conversion + LDS stores take 1045, tile / address arithmetic for
// the prefetch 806, the output stores 545, finalisation 394, barriers 370 -- and with the weights of two channel chunks
// resident only ONE block (one wave per SIMD) fits a CU, so none of that overlaps with matrix work.  Here a block has
// eight waves: half 0 (waves 0-3) and half 1 (waves 4-7) share the resident weights, own one patch buffer each and
// alternate -- while one half runs its "P" phase (barrier, patch -> LDS, barrier, output stores, prefetch issue) the
// other runs its "M" phase (the MFMAs) on the same SIMDs.  s_barrier is workgroup-wide, so every phase executes
// exactly two barriers (the M phase two fence-less ones) and half 1 starts one phase late.
// The scalar work is also cheaper than in the one-half kernel: tile coordinates advance incrementally in SGPRs (no
// integer division per stage), halo validity comes from four per-thread bit masks and four uniform edge flags, and the
// outputs go through buffer stores with a uniform row offset (no 64-bit address arithmetic per element).
// Used for Cin <= 64 (weights of two chunks + two patches = 142 KB of LDS); Cin = 96 keeps the one-half kernel.
// ------------------------------------------------------------------------------------------
// halves per patch row of the two-half kernel: [32 hi | 32 lo | 8 pad]; with 16-bit storage [32 | 8 pad] (80-byte rows: the
// 16 rows a ds_read_b128 quarter-wave touches start 20 dwords apart -- 16 distinct groups of four banks, conflict-free)
#define HALO2_P_LD (PP_ACT_LO ? H_LD : 40)
// the M phase executes its second barrier after step PP_HALO2_MSPLIT of its 18 (tap, 16-channel block) steps: the first segment
// runs beside the other half's patch staging, the second beside its output stores + prefetch issue
#ifndef PP_HALO2_MSPLIT
#define PP_HALO2_MSPLIT 7
#endif
struct HaloCursor {                          // all uniform: position of one half in its stage sequence
  int t, chunk, tx, ty, img;
};

// X1: one fp16 product per fp32 product (hi parts only) -- the PP_F16_PRODUCTS=1 "mixed precision" mode: fp16 operands
// (11 significand bits, dynamic range through the amax scaling), fp32 accumulation, fp32 tensors in HBM.
// TMR: output rows per wave (tile = 4 TMR rows x 32 columns per half).  2 is built for 16-bit storage only: a wave then applies
// every weight fragment it reads to two pixel rows -- the M phase of the one-row form reads 3 KB of LDS for 2 MFMAs per wave
// (fp32 storage: 4 KB for 3), 192 B per clock and CU against the 128 B the LDS delivers, i.e. it is LDS-bound at two thirds of
// the matrix rate; with two rows it is 4 KB for 4 MFMAs.  fp32 storage has no registers left for it (two prefetched patch sets
// of 11 float4 next to 64 accumulators; fp16 patches are half that).
// LAZY: a.in holds the raw convolution output z of the layer in front (train-mode BatchNorm, pp_common.h: PpLazy); BatchNorm +
// LeakyReLU are applied while the patch goes to LDS -- the P phase spends most of its time waiting for memory (r04 phase trace),
// the three VALU operations per element ride along (+4 % on the kernel when EVERY launch did them, same-box A/B), and the
// bn_lrelu_fwd pass over the mid tensor of a DoubleConv disappears.  Coefficient rows of the block's channels sit in LDS.
template <bool X1, int TMR, bool LAZY>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv3x3_halo2_f16x3_kernel(ConvArgs a, int n_chunks, int tiles_x, int tiles_y, int n_tiles, const float* in_amax, int chunk0) {
  // chunk0: first 32-channel chunk of this launch (split-K over several launches, see conv3x3_halo_f16x3_kernel; a.C stays the layer's)
  constexpr int ROWS = 4 * TMR, PIX = (ROWS + 2) * HT_HC, APASS = (PIX * 8 + 255) / 256;
  constexpr int P_LD = HALO2_P_LD;             // halves per patch row in LDS (no low part, no room for it, with 16-bit storage)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) _Float16 smem16[];
  const int tid = threadIdx.x, htid = tid & 255, lane = tid & 63;
  // wave-uniform, and the compiler has to know it: they feed scalar offsets of buffer instructions
  const int half = __builtin_amdgcn_readfirstlane(tid >> 8), wv = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int lr = lane & 31, lh = lane >> 5;
  _Float16* Bs = smem16;                                                   // [n_chunks][9][32][H_LD]  pre-split weights
  _Float16* As = smem16 + n_chunks * 9 * 32 * H_LD + half * PIX * P_LD;    // this half's patch [PIX][P_LD]
  float* Lz = reinterpret_cast<float*>(smem16 + n_chunks * 9 * 32 * H_LD + 2 * PIX * P_LD);   // LAZY: [2 groups][n_chunks][3][32]
  const int n0 = blockIdx.y * 32;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);
  float s_in, s_out;
  f16_scales(in_amax, s_in, s_out);
  {
    const int total = n_chunks * 9 * 32 * 8;
    for (int e = tid; e < total; e += 512) {
      const int q = e & 7, row = e >> 3;                   // row = (chunk * 9 + tap) * 32 + n
      const int n = row & 31, ct = row >> 5, tap = ct % 9, chunk = ct / 9;
      const unsigned off = (n0 + n < a.N) ? (unsigned)(((n0 + n) * 9 + tap) * a.C + (chunk0 + chunk) * 32 + q * 4) * 4u : 0xffffffffu;
      const f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));   // [hi4 | lo4]
      _Float16* d = Bs + row * H_LD + q * 4;
      *reinterpret_cast<f32x2*>(d) = __builtin_shufflevector(w, w, 0, 1);
      *reinterpret_cast<f32x2*>(d + 32) = __builtin_shufflevector(w, w, 2, 3);
    }
  }
  if (LAZY) {
    const int groups = (a.P / (a.H * a.W) + a.lazy.imgs_per_group - 1) / a.lazy.imgs_per_group;       // <= 2 (host)
    for (int e = tid; e < 2 * n_chunks * 96; e += 512) {
      const int c = e & 31, row = (e >> 5) % 3, gc = e / 96, chunk = gc % n_chunks, g = gc / n_chunks;
      Lz[e] = g < groups ? a.lazy.coef[(size_t)(g * 3 + row) * a.lazy.ld + (chunk0 + chunk) * 32 + c] : (row == 1 ? 0.f : 1.f);
    }
  }
  __syncthreads();                           // the phase barriers of the waiting half carry no fence: publish the weights here
  // per-thread constants of the patch staging: byte offset of (patch pixel, channel quad) relative to the tile's first
  // halo pixel, LDS offset, and one bit per pass in four edge masks (+ m_dead: pass beyond the patch)
  int relb[APASS], lds_off[APASS];
  unsigned m_top = 0, m_bot = 0, m_left = 0, m_right = 0, m_dead = 0;
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int e = htid + 256 * i, pix = e >> 3, q = e & 7;
    const int hy = pix / HT_HC, hx = pix - hy * HT_HC;
    relb[i] = ((hy * a.W + hx) * a.ld_in + q * 4) * PP_ACT_BYTES;
    lds_off[i] = pix * P_LD + q * 4;
    if (pix >= PIX) { m_dead |= 1u << i; relb[i] = 0; lds_off[i] = 0; }
    if (hy == 0) m_top |= 1u << i;
    if (hy == ROWS + 1) m_bot |= 1u << i;
    if (hx == 0) m_left |= 1u << i;
    if (hx == HT_HC - 1) m_right |= 1u << i;
  }
  // stage sequence of this half: its tiles are t = blockIdx.x + (2 k + half) * gridDim.x; both halves run R rounds
  const int G = (int)gridDim.x, G2 = 2 * G;
  const int d_tx = G2 % tiles_x, d_q = G2 / tiles_x, d_ty = d_q % tiles_y, d_img = d_q / tiles_y;
  auto cursor_at = [&](int t) __attribute__((always_inline)) {
    HaloCursor c;
    c.t = t; c.chunk = 0;
    c.tx = t % tiles_x;
    const int r = t / tiles_x;
    c.ty = r % tiles_y; c.img = r / tiles_y;
    return c;
  };
  auto advance = [&](HaloCursor& c) __attribute__((always_inline)) {
    if (++c.chunk == n_chunks) {
      c.chunk = 0;
      c.t += G2;
      c.tx += d_tx; if (c.tx >= tiles_x) { c.tx -= tiles_x; ++c.ty; }
      c.ty += d_ty; if (c.ty >= tiles_y) { c.ty -= tiles_y; ++c.img; }
      c.img += d_img;
    }
  };
  const int tiles_h0 = (int)blockIdx.x < n_tiles ? (n_tiles - (int)blockIdx.x + G2 - 1) / G2 : 0;      // half 0 has the most
  const int R = (tiles_h0 * n_chunks + 1) & ~1;                                                    // rounds, even
  act_raw4 ra0[APASS], ra1[APASS];            // prefetched patches as loaded (fp16 storage: 8 bytes per quad)
  auto load_patch = [&](act_raw4 (&ra)[APASS], const HaloCursor& c) __attribute__((always_inline)) {
    const bool live = c.t < n_tiles;
    // first halo pixel = (row 4 ty - 1, column 32 tx - 1): may lie one row / column outside the image, where the
    // byte offset is meaningless -- those passes are masked, as are all passes of a ghost stage
    const int sbase = (((c.img * a.H + c.ty * ROWS - 1) * a.W + c.tx * HT_COLS - 1) * a.ld_in + (chunk0 + c.chunk) * 32) * PP_ACT_BYTES;
    const __amdgpu_buffer_rsrc_t rs_in_l = TMR == 1 ? rs_in : __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    unsigned bad = m_dead;
    if (!live) bad = ~0u;
    if (c.ty == 0) bad |= m_top;
    if (c.ty == tiles_y - 1) bad |= m_bot;
    if (c.tx == 0) bad |= m_left;
    if (c.tx == tiles_x - 1) bad |= m_right;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const unsigned off = ((bad >> i) & 1u) ? 0xffffffffu : (unsigned)(sbase + relb[i]);
      ra[i] = act_buf_ld4_raw(rs_in_l, off, 0);
    }
  };
  // cs: the stage this patch belongs to (LAZY: its image selects the coefficient group, its position the zero-padded halo)
  auto store_patch = [&](act_raw4 (&ra)[APASS], const HaloCursor& cs) __attribute__((always_inline)) {
    f32x4 l_sc, l_sh, l_sl;
    unsigned pad = 0;
    if (LAZY) {
      const float* r = Lz + ((cs.img >= a.lazy.imgs_per_group ? n_chunks : 0) + cs.chunk) * 96 + (htid & 7) * 4;
      l_sc = *reinterpret_cast<const f32x4*>(r);
      l_sh = *reinterpret_cast<const f32x4*>(r + 32);
      l_sl = *reinterpret_cast<const f32x4*>(r + 64);
      if (cs.ty == 0) pad |= m_top;           // zero padding applies to y, not to z: halo pixels outside the image stay 0
      if (cs.ty == tiles_y - 1) pad |= m_bot;
      if (cs.tx == 0) pad |= m_left;
      if (cs.tx == tiles_x - 1) pad |= m_right;
    }
#pragma unroll
    for (int i = 0; i < APASS; ++i)
      if (!((m_dead >> i) & 1u)) {
        f32x4 v = act_cvt4(ra[i]);
        if (LAZY) v = ((pad >> i) & 1u) ? f32x4{0.f, 0.f, 0.f, 0.f} : pp_lazy_apply4(v, l_sc, l_sh, l_sl);
        v = v * s_in;
        const f16x4 hi = __builtin_convertvector(v, f16x4);
        *reinterpret_cast<f16x4*>(As + lds_off[i]) = hi;
        if (!X1 && PP_ACT_LO) {
          const f16x4 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * F16_LO_SCALE, f16x4);
          *reinterpret_cast<f16x4*>(As + lds_off[i] + 32) = lo;
        }
      }
  };
  const bool n_ok = n0 + lr < a.N;
  const float bv = (a.bias && n_ok) ? a.bias[n0 + lr] : 0.f;
  const float e_sc = (a.epi.mode == 2 && n_ok) ? a.epi.scale[n0 + lr] : 1.f;
  const float e_sh = (a.epi.mode == 2 && n_ok) ? a.epi.shift[n0 + lr] : 0.f;
  float st_s0 = 0.f, st_q0 = 0.f, st_s1 = 0.f, st_q1 = 0.f;
  const _Float16* Ab = As + (wv * TMR * HT_HC + lr) * P_LD + lh * 8;
  // output element r of this lane: pixel row (r & 3) + 8 (r >> 2) + 4 lh of the wave's 32-pixel output row, channel n0 + lr
  const unsigned o_lane = n_ok ? (unsigned)((4 * lh * a.ld_out + n0 + lr) * PP_ACT_BYTES) : 0xffffffffu;
  f32x16 pend[TMR];
  int pend_img = -1, pend_ty = 0, pend_tx = 0;
#pragma unroll
  for (int i = 0; i < TMR; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) pend[i][r] = 0.f;
  auto write_pending = [&]() __attribute__((always_inline)) {
    if (pend_img >= 0) {
      const __amdgpu_buffer_rsrc_t rs_out_l = TMR == 1 ? rs_out : __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < TMR; ++i) {
        const int o_tile = (((pend_img * a.H + pend_ty * ROWS + wv * TMR + i) * a.W + pend_tx * HT_COLS) * a.ld_out) * PP_ACT_BYTES;
        if (a.accumulate) {
          float old[16];
#pragma unroll
          for (int r = 0; r < 16; ++r)
            old[r] = act_buf_ld1(rs_out_l, o_lane, o_tile + ((r & 3) + 8 * (r >> 2)) * a.ld_out * PP_ACT_BYTES);
#pragma unroll
          for (int r = 0; r < 16; ++r)
            act_buf_st1(old[r] + pend[i][r], rs_out_l, o_lane, o_tile + ((r & 3) + 8 * (r >> 2)) * a.ld_out * PP_ACT_BYTES);
        } else {
          // separate path: a shared store loop over (old = 0 | loaded) made hipcc wait for vmcnt(0) -- i.e. for the prefetch
          // just issued -- before zeroing `old`.  The whole vector is bit-cast once: with a per-element
          // __builtin_bit_cast(int, pend[r]) this loop was compiled into 16 stores of pend[0] (hipcc 7.2).
#ifdef PP_ACT_16
#pragma unroll
          for (int r = 0; r < 16; ++r) act_buf_st1(pend[i][r], rs_out_l, o_lane, o_tile + ((r & 3) + 8 * (r >> 2)) * a.ld_out * PP_ACT_BYTES);
#else
          typedef int i32x16 __attribute__((ext_vector_type(16)));
          const i32x16 pi = __builtin_bit_cast(i32x16, pend[i]);
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(pi[r], rs_out_l, o_lane, o_tile + ((r & 3) + 8 * (r >> 2)) * a.ld_out * 4, 0);
#endif
        }
      }
    }
    pend_img = -1;
  };
  HaloCursor cc = cursor_at((int)blockIdx.x + half * G), cl = cc;
  f32x16 accm[TMR], accc[TMR];
#pragma unroll
  for (int i = 0; i < TMR; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { accm[i][r] = 0.f; accc[i][r] = 0.f; }
  // P phase: this half's patch goes to LDS, the finished tile to memory, the prefetch two stages ahead is issued
#ifdef PP_HALO_TRACE
  long long tr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_begin = __builtin_readcyclecounter(), w_begin = __builtin_amdgcn_s_memrealtime();
  long long t_prev = t_begin;
#endif
  auto phase_p = [&](act_raw4 (&ra)[APASS]) __attribute__((always_inline)) {
    __syncthreads();
    HT_TRK(0)
    store_patch(ra, cc);
    HT_TRK(1)
    __syncthreads();
    HT_TRK(2)
    if (cc.chunk == 0) {
      write_pending();                       // stores first (the other order -- prefetch first -- measured 1 % slower: gfx950
#pragma unroll                                // counts loads and stores in ONE in-order vmcnt, so every later wait for a
      for (int i = 0; i < TMR; ++i)           // prefetch also waits for the stores in front of it)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accm[i][r] = 0.f; accc[i][r] = 0.f; }
    }
    HT_TRK(3)
    load_patch(ra, cl);
    advance(cl);
    HT_TRK(4)
  };
  // M phase: 18 steps (tap, 16-channel block) of 3 MFMAs with register double-buffered fragments, and the two
  // barriers the other half's P phase is executing meanwhile
  auto phase_m = [&]() __attribute__((always_inline)) {
    const bool live = cc.t < n_tiles;
    const _Float16* Bb = Bs + (cc.chunk * 9 * 32 + lr) * H_LD + lh * 8;
    f16x8 ah[2][TMR], al[2][TMR], bh[2], bl[2];
    auto read_step = [&](int st, int slot) __attribute__((always_inline)) {
      const int tap = st >> 1, kb = st & 1;
      bh[slot] = *reinterpret_cast<const f16x8*>(Bb + tap * 32 * H_LD + kb * 16);
      if (!X1) bl[slot] = *reinterpret_cast<const f16x8*>(Bb + tap * 32 * H_LD + kb * 16 + 32);
#pragma unroll
      for (int i = 0; i < TMR; ++i) {
        const _Float16* ap = Ab + ((i + tap / 3) * HT_HC + tap % 3) * P_LD + kb * 16;
        ah[slot][i] = *reinterpret_cast<const f16x8*>(ap);
        if (!X1 && PP_ACT_LO) al[slot][i] = *reinterpret_cast<const f16x8*>(ap + 32);
      }
    };
    __builtin_amdgcn_s_barrier();
    HT_TRK(5)
    read_step(0, 0);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      const int cur = st & 1;
      if (st + 1 < 18) read_step(st + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TMR; ++i) {
        if (!X1) accc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bl[cur], accc[i], 0, 0, 0);
        accm[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bh[cur], accm[i], 0, 0, 0);   // between the two dependent ones
        if (!X1 && PP_ACT_LO) accc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][i], bh[cur], accc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (st == PP_HALO2_MSPLIT) {
        HT_TRK(6)
        __builtin_amdgcn_s_barrier();
        HT_TRK(7)
      }
    }
    HT_TRK(8)
    if (cc.chunk + 1 == n_chunks && live) {
      float ts = 0.f, tq = 0.f;
#pragma unroll
      for (int i = 0; i < TMR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = (X1 ? accm[i][r] : accm[i][r] + accc[i][r] * (1.f / F16_LO_SCALE)) * s_out + bv;
          if (a.epi.mode == 1) { ts += v; tq += v * v; }                       // BatchNorm batch statistics of z
          if (a.epi.mode == 2) { v = v * e_sc + e_sh; v = fmaxf(v, v * a.epi.slope); }   // eval-mode BN + LeakyReLU
          pend[i][r] = v;
        }
      if (a.epi.mode == 1) {                 // group (weak | strong half of the batch) of this tile's image
        if (cc.img * (a.H * a.W) >= a.epi.px_per_group) { st_s1 += ts; st_q1 += tq; }
        else { st_s0 += ts; st_q0 += tq; }
      }
      pend_img = cc.img; pend_ty = cc.ty; pend_tx = cc.tx;
    }
    advance(cc);
    HT_TRK(9)
  };
  load_patch(ra0, cl); advance(cl);
  load_patch(ra1, cl); advance(cl);
  if (half == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }     // starts one phase late
  for (int r = 0; r < R; r += 2) {
    phase_p(ra0); phase_m();
    phase_p(ra1); phase_m();
  }
  if (half == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }     // half 1's last M phase
#ifdef PP_HALO_TRACE
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    for (int k = 0; k < 10; ++k) pp_halo_trace[k] = tr[k];
    pp_halo_trace[10] = R;
    pp_halo_trace[11] = __builtin_readcyclecounter() - t_begin;
    pp_halo_trace[12] = __builtin_amdgcn_s_memrealtime() - w_begin;       // 100 MHz
    pp_halo_trace[13] = 2;                                                 // kernel id
  }
#endif
  write_pending();
  if (a.epi.mode == 1) {
    // per-channel partial sums of this block: lanes lr / lr + 32 hold the same channel, the eight waves eight rows
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem16);             // [8 waves][2 groups][2][32]
    const int w8 = tid >> 6;
#pragma unroll
    for (int g = 0; g < PP_EPI_GROUPS; ++g) {
      const float sg = g ? st_s1 : st_s0, qg = g ? st_q1 : st_q0;
      const float ss = sg + __shfl_xor(sg, 32, 64), qq = qg + __shfl_xor(qg, 32, 64);
      if (lh == 0) { red[((w8 * 2 + g) * 2 + 0) * 32 + lr] = ss; red[((w8 * 2 + g) * 2 + 1) * 32 + lr] = qq; }
    }
    __syncthreads();
    if (tid < 128 && (tid >> 6) < a.epi.groups && n0 + (tid & 31) < a.N) {
      const int g = tid >> 6, which = (tid >> 5) & 1, c = tid & 31;
      double acc = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) acc += (double)red[((w * 2 + g) * 2 + which) * 32 + c];
      pp_epi_row(a.epi, g, blockIdx.x, which, a.N)[n0 + c] = acc;
    }
  }
}


static bool wgrad_lazy_ok(int O, int C, int H, int W, int dil);
static inline int halo_f16_rows(const ConvArgs& a) {           // 0 = not eligible, else output rows per wave (1 or 2)
  // up to 192 output channels (six weight-resident blocks per tile column): the 64 -> 192 data gradient of dec2.c1 runs
  // 1.4x faster here than in the implicit-GEMM kernel (r02 A/B on one box: step 38.47 -> 38.02 ms)
  constexpr int max_n = 192;
  if (a.dil != 1 || a.C % 32 != 0 || a.C > 96 || a.N % 32 != 0 || a.N > max_n || a.W % HT_COLS != 0 || a.H % 4 != 0) return 0;
  // (two rows per wave in the ONE-half kernel spilled -- 256 VGPRs -- and measured 1.8x slower on the 32 / 64-channel layers, r01:
  // that instantiation was removed in round 6; the two-half kernel of the 16-bit build has its own two-row form)
  return 1;
}

// LDS bytes of the two-half kernel: resident weights of all chunks + one patch per half (rows2 = 4 * rows per wave)
static inline size_t halo2_lds(int n_chunks, int rows_per_wave, bool lazy = false) {
  return ((size_t)n_chunks * 9 * 32 * H_LD + (size_t)2 * (4 * rows_per_wave + 2) * HT_HC * HALO2_P_LD) * sizeof(_Float16) +
         (lazy ? (size_t)2 * n_chunks * 96 * sizeof(float) : 0);
}
// The two-half kernel: 0 = not for this call, else its output rows per wave.  Needs out addressable with 32-bit offsets and
// weights + two patches within 160 KB of LDS: Cin <= 64 with fp32 storage; with 16-bit storage (patch rows without a low part)
// Cin = 96 fits as well, and the one- and two-chunk layers run two rows per wave (see the kernel).
static inline int halo2_ok(const ConvArgs& a, int tmr) {
  if (tmr != 1 || a.C > 96 || ((long long)(a.P - 1) * a.ld_out + a.N) * 4 >= 0xffffffffLL) return 0;
  int t = 1;
#ifdef PP_ACT_16
  if (a.H % 8 == 0) t = 2;
#endif
  const bool lz = a.lazy.coef != nullptr;
  if (t == 2 && halo2_lds(a.C / 32, 2, lz) > 163840) t = 1;
  return halo2_lds(a.C / 32, t, lz) <= 163840 ? t : 0;
}

// the two-half kernel for a split-K launch over `n_chunks_launch` chunks (0: not a split launch -> halo2_ok of the whole layer);
// split launches carry no fused epilogue there (its accumulate path adds the previous value when the tile is WRITTEN, after the
// epilogue has seen the partial sum)
static inline int halo2_ok_launch(const ConvArgs& a, int tmr, int n_chunks_launch) {
  if (!n_chunks_launch) return halo2_ok(a, tmr);
  if (a.epi.mode || a.lazy.coef) return 0;
  ConvArgs h = a;
  h.C = n_chunks_launch * 32;
  return halo2_ok(h, tmr);
}

static int halo_f16x3_grid_x(const ConvArgs& a, int tmr, int n_chunks_launch = 0) {
  if (const int t2 = halo2_ok_launch(a, tmr, n_chunks_launch)) {
    const int n_tiles = (a.P / (a.H * a.W)) * (a.W / HT_COLS) * (a.H / (4 * t2));
    int gx = 256 / (a.N / 32);
    if (gx < 1) gx = 1;
    const int want = (n_tiles + 1) / 2;      // two halves per block
    return gx > want ? want : gx;
  }
  const int n_chunks = n_chunks_launch ? n_chunks_launch : a.C / 32, rows = 4 * tmr;
  const int n_tiles = (a.P / (a.H * a.W)) * (a.W / HT_COLS) * (a.H / rows);
  const size_t lds = (size_t)(n_chunks * 9 * 32 + (rows + 2) * HT_HC) * H_LD * sizeof(_Float16);
  int per_cu = (int)(163840 / lds);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 2) per_cu = 2;
  int gx = (256 * per_cu) / (a.N / 32);
  if (gx < 1) gx = 1;
  return gx > n_tiles ? n_tiles : gx;
}

// chunk0 / n_chunks_launch: split-K launch over the channel chunks [chunk0, chunk0 + n_chunks_launch) (0 = the whole layer)
static int launch_halo_f16x3(ConvArgs a, const float* in_amax, int tmr, hipStream_t s, int chunk0 = 0, int n_chunks_launch = 0) {
  const int n_chunks = n_chunks_launch ? n_chunks_launch : a.C / 32;
  const int rows = 4 * tmr;
  const int tiles_x = a.W / HT_COLS, tiles_y = a.H / rows;
  const int n_tiles = (a.P / (a.H * a.W)) * tiles_x * tiles_y;
  const size_t lds = (size_t)(n_chunks * 9 * 32 + (rows + 2) * HT_HC) * H_LD * sizeof(_Float16);
  {   // once per (kernel, device): pp_max_lds
    pp_max_lds(reinterpret_cast<const void*>(conv3x3_halo_f16x3_kernel<1>), (int)((3 * 9 * 32 + 6 * HT_HC) * H_LD * sizeof(_Float16)));
  }
  const int gy = a.N / 32;
  const int gx = halo_f16x3_grid_x(a, tmr, n_chunks_launch);
  if (const int t2 = halo2_ok_launch(a, tmr, n_chunks_launch)) {
    const size_t lds2 = halo2_lds(n_chunks, t2, a.lazy.coef != nullptr);
    const int tiles_y2 = a.H / (4 * t2), n_tiles2 = (a.P / (a.H * a.W)) * tiles_x * tiles_y2;
    a.out_bytes = (unsigned)(((long long)(a.P - 1) * a.ld_out + a.N) * PP_ACT_BYTES);
    const bool x1 = pp_f16_products() == 1;
#define HALO2_LAUNCH_L(X1, T, LZ)                                                                                                 \
    do {                                                                                                                           \
      pp_max_lds(reinterpret_cast<const void*>(conv3x3_halo2_f16x3_kernel<X1, T, LZ>), 163840);                                   \
      hipLaunchKernelGGL((conv3x3_halo2_f16x3_kernel<X1, T, LZ>), dim3(gx, gy), dim3(512), lds2, s, a, n_chunks, tiles_x, tiles_y2, n_tiles2, in_amax, chunk0); \
    } while (0)
#define HALO2_LAUNCH(X1, T) do { if (a.lazy.coef) HALO2_LAUNCH_L(X1, T, true); else HALO2_LAUNCH_L(X1, T, false); } while (0)
#ifdef PP_ACT_16
    if (t2 == 2) { if (x1) HALO2_LAUNCH(true, 2); else HALO2_LAUNCH(false, 2); } else
#endif
    { if (x1) HALO2_LAUNCH(true, 1); else HALO2_LAUNCH(false, 1); }
#undef HALO2_LAUNCH
#undef HALO2_LAUNCH_L
    return pp_launch_status("conv3x3_halo2_f16x3");
  }
  if (a.lazy.coef) {
    pp_set_error("conv3x3: a lazy input needs the two-half halo kernel (pp_conv3x3_lazy_ok tells)");
    return PP_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(conv3x3_halo_f16x3_kernel<1>, dim3(gx, gy), dim3(256), lds, s, a, n_chunks, tiles_x, tiles_y, n_tiles, in_amax, chunk0);
  return pp_launch_status("conv3x3_halo_f16x3");
}

#ifndef PP_ACT_16
static inline bool halo_eligible(const ConvArgs& a) {
  return a.dil == 1 && a.C % 32 == 0 && a.C <= 96 && a.N % 32 == 0 && a.W % HT_COLS == 0 && a.H % HT_ROWS == 0;
}

static int launch_halo(ConvArgs a, hipStream_t s) {
  const int n_chunks = a.C / 32;
  const int tiles_x = a.W / HT_COLS, tiles_y = a.H / HT_ROWS;
  const int n_tiles = (a.P / (a.H * a.W)) * tiles_x * tiles_y;
  const size_t lds = (size_t)(n_chunks * 9 * 32 + HT_PIX) * LDS_LD * sizeof(float);
  {   // once per (kernel, device): pp_max_lds
    pp_max_lds(reinterpret_cast<const void*>(conv3x3_halo_kernel), (int)((3 * 9 * 32 + HT_PIX) * LDS_LD * sizeof(float)));
  }
  const int gy = a.N / 32;
  const int per_cu = (int)(163840 / lds) < 1 ? 1 : (int)(163840 / lds);
  int gx = (256 * (per_cu > 2 ? 2 : per_cu)) / gy;
  if (gx < 1) gx = 1;
  if (gx > n_tiles) gx = n_tiles;
  hipLaunchKernelGGL(conv3x3_halo_kernel, dim3(gx, gy), dim3(256), lds, s, a, n_chunks, tiles_x, tiles_y, n_tiles);
  return pp_launch_status("conv3x3_halo");
}
#endif  // !PP_ACT_16

// `fused` (nullable): set to whether the selected kernel executed a.epi itself; when it is null or the variant has no
// fused epilogue, a.epi is cleared and the caller runs the unfused BatchNorm kernels.
static int conv_dispatch(ConvArgs a, hipStream_t s, bool* fused = nullptr, int* epi_rows = nullptr) {
  if (fused) *fused = false;
  PP_CHECK_ARG(a.in && a.w && a.out, "conv3x3: null pointer");
  PP_CHECK_ARG(a.C > 0 && a.C % 4 == 0 && a.ld_in % 4 == 0, "conv3x3: C (%d) and ld_in (%d) must be multiples of 4", a.C, a.ld_in);
  PP_CHECK_ARG(((uintptr_t)a.in & PP_ACT_ALIGN) == 0 && ((uintptr_t)a.w & 15) == 0, "conv3x3: in/w must be 16-byte aligned");
  PP_CHECK_ARG(a.N > 0 && a.P > 0 && a.H > 0 && a.W > 0 && a.P % (a.H * a.W) == 0, "conv3x3: bad shape P=%d H=%d W=%d N=%d", a.P, a.H, a.W, a.N);
  PP_CHECK_ARG(a.dil >= 1 && a.ld_out >= a.N && a.ld_in >= a.C, "conv3x3: bad dil/ld");
  PP_CHECK_ARG((long long)a.P * a.ld_in < 0x3fffffffLL && (long long)a.P * a.ld_out < 0x7fffffffLL &&
                   (long long)a.N * 9 * a.C < 0x3fffffffLL, "conv3x3: tensor exceeds the 4 GiB buffer-descriptor range");
  a.in_bytes = (unsigned)(((long long)(a.P - 1) * a.ld_in + a.C) * PP_ACT_BYTES);
  a.w_bytes = (unsigned)((long long)a.N * 9 * a.C * 4);
  const double flops = 2.0 * a.P * (double)a.N * 9.0 * a.C;
  const double bytes = 4.0 * ((double)a.P * a.C + (double)a.P * a.N + 9.0 * a.C * a.N);
  pp_prof_begin(PP_K_CONV_IGEMM, flops, bytes, s);
  int rc;
  int v = 0;
#ifdef PP_ACT_16
  v = c4_eligible(a) ? 9 : -1;               // 16-bit storage: the first-layer kernel only; everything else is f16x3
  if (v < 0) { pp_set_error("conv3x3 (16-bit storage): only the first-layer shape has an fp32-MFMA kernel; use the f16x3 entry points"); return PP_ERR_UNSUPPORTED; }
#else
  if (v == 0 && halo_eligible(a)) v = 8;
  if (v == 0 && c4_eligible(a)) v = 9;
  if (v == 0) v = (a.N % 128 == 0) ? 1 : ((a.N % 64 == 0) ? 2 : 4);   // measured per layer: scripts/bench_conv.py
#endif
  if (a.epi.mode && fused && v == 9 && a.epi.groups <= PP_EPI_GROUPS && a.epi.px_per_group % 16 == 0 && !a.accumulate) {
    a.epi.rows = c4_blocks(a, nullptr);
    *fused = true;
    if (epi_rows) *epi_rows = a.epi.rows;
  } else {
    a.epi.mode = 0;
  }
  switch (v) {
#ifndef PP_ACT_16
    case 1: rc = launch_igemm<2, 2, 2, 2>(a, s); break;       // 128 x 128
    case 2: rc = launch_igemm<2, 1, 2, 2>(a, s); break;       // 128 x 64
    case 4: rc = launch_igemm<1, 1, 4, 1>(a, s); break;       // 128 x 32
    case 8: rc = launch_halo(a, s); break;                    // persistent halo tiles (narrow layers)
#endif
    case 9: rc = launch_c4(a, s); break;                      // first layer (4-channel padded image)
    default: pp_set_error("conv3x3: no kernel variant %d", v); return PP_ERR_ARG;
  }
  pp_prof_end(s);
  return rc;
}

static int conv_dispatch_f16x3(ConvArgs a, const float* in_amax, hipStream_t s, bool* fused = nullptr, int* epi_rows = nullptr) {
  if (fused) *fused = false;
  PP_CHECK_ARG(a.in && a.w && a.out, "conv3x3_f16x3: null pointer");
  PP_CHECK_ARG(a.C > 0 && a.C % 4 == 0 && a.ld_in % 4 == 0, "conv3x3_f16x3: C (%d) and ld_in (%d) must be multiples of 4", a.C, a.ld_in);
  PP_CHECK_ARG(((uintptr_t)a.in & PP_ACT_ALIGN) == 0 && ((uintptr_t)a.w & 15) == 0, "conv3x3_f16x3: in/w must be 16-byte aligned");
  PP_CHECK_ARG(a.N > 0 && a.P > 0 && a.H > 0 && a.W > 0 && a.P % (a.H * a.W) == 0, "conv3x3_f16x3: bad shape");
  PP_CHECK_ARG(a.dil >= 1 && a.ld_out >= a.N && a.ld_in >= a.C, "conv3x3_f16x3: bad dil/ld");
  PP_CHECK_ARG((long long)a.P * a.ld_in < 0x3fffffffLL && (long long)a.P * a.ld_out < 0x7fffffffLL &&
                   (long long)a.N * 9 * a.C < 0x3fffffffLL, "conv3x3_f16x3: tensor exceeds the 4 GiB buffer-descriptor range");
  a.in_bytes = (unsigned)(((long long)(a.P - 1) * a.ld_in + a.C) * PP_ACT_BYTES);
  a.w_bytes = (unsigned)((long long)a.N * 9 * a.C * 4);
  const double flops = 2.0 * a.P * (double)a.N * 9.0 * a.C;
  const double bytes = 4.0 * ((double)a.P * a.C + (double)a.P * a.N + 9.0 * a.C * a.N);
  int rc;
  int v = (a.N % 128 == 0) ? 1 : ((a.N % 64 == 0) ? 2 : 4);
  int tmr = halo_f16_rows(a);
  // split-K over two launches of the one-row halo kernel for 128 < C <= 192 (dec2.c1 forward, 192 -> 64 at 128^2: the 128 x 64
  // implicit-GEMM tile fetched 6.3 GB for its 0.8 GB input, r03 PMC profile; Winograd measured slower there)
  constexpr int splitk_max = 192;
  bool splitk = false;
  if (!tmr && a.C > 128 && a.C <= splitk_max && a.C % 64 == 0) {
    ConvArgs h = a;
    h.C = a.C / 2;
    splitk = halo_f16_rows(h) == 1;
    if (splitk) { tmr = 1; v = 9; }
  }
  // ... or, round 5, as C / 64 accumulating launches of the TWO-HALF kernel over 64 channels each (0.9 PF/s executed against the 0.58
  // of the one-row kernel with three resident chunks; no fused epilogue: the caller's unfused BatchNorm passes run behind it)
  bool splitk2 = false;
#ifndef PP_ACT_16
  if (splitk) {
    ConvArgs h = a;
    h.epi.mode = 0;
    splitk2 = halo2_ok_launch(h, 1, 2) != 0;
    if (splitk2) v = 10;
  }
#endif
  if (tmr && !splitk) v = 8;
  // (Round 5 built the weight-streaming form of the two-half kernel for Cin >= 96 -- a ring of two weight slots fed by LDS-DMA from
  // the M phase, nothing through registers -- and measured it EQUAL to the kernels it replaced, layer by layer (enc3.c2 0.350 ->
  // 0.333 ms, dec1.c1 0.912 -> 0.916, dec2.c1 0.907 -> 0.900): all of them run at the rate of the patch staging.  As a persistent
  // one-block-per-CU kernel it also took the CUs the second stream's weight gradients share with the implicit GEMM in the backward
  // pass (+0.9 ms there).  Removed again; profiles/r05_experiments/ab_step_weight_streaming_halo2s.log)
  if (a.epi.mode && fused && v != 10 && a.epi.groups <= PP_EPI_GROUPS && !a.accumulate &&
      (v >= 8 ? (tmr == 1 && a.epi.px_per_group % (a.H * a.W) == 0) : ((v == 1 || v == 2 || v == 4) && a.epi.px_per_group % 128 == 0))) {
    a.epi.rows = v == 9 ? halo_f16x3_grid_x(a, tmr, a.C / 64) : v == 8 ? halo_f16x3_grid_x(a, tmr) : a.epi.px_per_group / 128;
    *fused = true;
    if (epi_rows) *epi_rows = a.epi.rows;
  } else {
    a.epi.mode = 0;
  }
  if (a.lazy.coef && !(v == 8 && halo2_ok(a, tmr))) {
    pp_set_error("conv3x3_f16x3: a lazy input needs the two-half halo kernel for this shape (pp_conv3x3_lazy_ok tells)");
    return PP_ERR_UNSUPPORTED;
  }
  // executes three 16-bit products per fp32 product; the two kernels are profiled as separate kinds
  pp_prof_begin2(v >= 8 ? PP_K_CONV_HALO_F16X3 : PP_K_CONV_F16X3, 3.0 * flops, flops, bytes, s);
  switch (v) {
    case 8: rc = launch_halo_f16x3(a, in_amax, tmr, s); break;             // persistent halo tiles (narrow layers)
    case 9: {                                                              // the same, two launches over half the channels each
      ConvArgs h = a;
      h.epi.mode = 0;                      // first half: plain store (or accumulate, as the caller asked), no epilogue
      rc = launch_halo_f16x3(h, in_amax, 1, s, 0, a.C / 64);
      if (rc) break;
      h = a;
      h.accumulate = 1;
      h.bias = nullptr;                    // added by the first launch
      rc = launch_halo_f16x3(h, in_amax, 1, s, a.C / 64, a.C / 64);
      break;
    }
    case 10: {                                                             // C / 64 launches of the two-half kernel, 64 channels each
      rc = 0;
      for (int k = 0; k < a.C / 64 && !rc; ++k) {
        ConvArgs h = a;
        h.epi.mode = 0;
        if (k > 0) { h.accumulate = 1; h.bias = nullptr; }               // the bias was added by the first launch
        rc = launch_halo_f16x3(h, in_amax, 1, s, 2 * k, 2);
      }
      break;
    }
    case 1: rc = launch_igemm_f16x3<2, 2, 2, 2>(a, in_amax, s); break;     // 128 x 128 (256 x 128 with 8 waves measured no gain, r02)
    case 2: rc = launch_igemm_f16x3<2, 1, 2, 2>(a, in_amax, s); break;     // 128 x 64
    case 4: rc = launch_igemm_f16x3<1, 1, 4, 1>(a, in_amax, s); break;     // 128 x 32
    default: pp_set_error("conv3x3_f16x3: no kernel variant %d", v); return PP_ERR_ARG;
  }
  pp_prof_end(s);
  return rc;
}

extern "C" int PP_FN(pp_conv3x3_fwd_f16x3)(const pp_act* in, int ld_in, int C, const void* wf16, const float* bias, pp_act* out,
                                    int ld_out, int N, int B, int H, int W, int dil, int accumulate, const float* in_amax,
                                    void* stream) {
  ConvArgs a{in, ld_in, C, (const float*)wf16, bias, out, ld_out, N, B * H * W, H, W, dil, accumulate, 0, 0, 0, 0};
  return conv_dispatch_f16x3(a, in_amax, (hipStream_t)stream);
}

extern "C" int PP_FN(pp_conv3x3_bwd_data_f16x3)(const pp_act* dz, int ld_dz, int O, const void* wb16, pp_act* dx, int ld_dx, int I,
                                         int B, int H, int W, int dil, int accumulate, const float* dz_amax, void* stream) {
  ConvArgs a{dz, ld_dz, O, (const float*)wb16, nullptr, dx, ld_dx, I, B * H * W, H, W, dil, accumulate, 0, 0, 0, 0};
  return conv_dispatch_f16x3(a, dz_amax, (hipStream_t)stream);
}

extern "C" int PP_FN(pp_conv3x3_fwd)(const pp_act* in, int ld_in, int C, const float* wf, const float* bias, pp_act* out,
                              int ld_out, int N, int B, int H, int W, int dil, int accumulate, void* stream) {
  ConvArgs a{in, ld_in, C, wf, bias, out, ld_out, N, B * H * W, H, W, dil, accumulate, 0, 0, 0, 0};
  return conv_dispatch(a, (hipStream_t)stream);
}

extern "C" int PP_FN(pp_conv3x3_bwd_data)(const pp_act* dz, int ld_dz, int O, const float* wb, pp_act* dx, int ld_dx, int I,
                                   int B, int H, int W, int dil, int accumulate, void* stream) {
  ConvArgs a{dz, ld_dz, O, wb, nullptr, dx, ld_dx, I, B * H * W, H, W, dil, accumulate, 0, 0, 0, 0};
  return conv_dispatch(a, (hipStream_t)stream);
}

// Forward convolution with the BatchNorm that follows it (models/unet.py:188-193) fused into the epilogue where the
// selected kernel supports it, and the unfused BatchNorm kernels behind it where not -- same results either way.
//   bn_mode 1 (train): out = z = conv + bias, and stats[groups][*rows_out][2][N] (double) = per-block (sum z, sum z^2),
//                      to be handed to pp_bn_train_finalize(stats, *rows_out, ...)
//   bn_mode 2 (eval) : out = y = leaky_relu(z * scale[n] + shift[n], slope)
extern "C" size_t PP_FN(pp_conv3x3_bn_stats_bytes)(int N, int B, int H, int W, int groups) {
  long long rows = ((long long)B * H * W / (groups > 0 ? groups : 1)) / 128 + 1;
  if (rows < 2048) rows = 2048;
  return (size_t)(groups > 0 ? groups : 1) * rows * 2 * N * sizeof(double);
}

static int conv3x3_fwd_bn_impl(const pp_act* in, int ld_in, int C, const void* wf, const float* bias, pp_act* out,
                               int ld_out, int N, int B, int H, int W, int dil, int f16x3, const float* in_amax,
                               int bn_mode, const float* scale, const float* shift, float slope, int groups,
                               double* stats, size_t stats_bytes, int* rows_out, PpLazy lazy, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(bn_mode == 1 || bn_mode == 2, "conv3x3_fwd_bn: bn_mode must be 1 (train) or 2 (eval)");
  PP_CHECK_ARG(groups >= 1 && (B % groups) == 0, "conv3x3_fwd_bn: groups must divide the batch");
  PP_CHECK_ARG(bn_mode == 2 ? (scale && shift) : (stats && rows_out), "conv3x3_fwd_bn: missing BatchNorm arguments");
  const int ppg = (B / groups) * H * W;
  ConvArgs a{in, ld_in, C, (const float*)wf, bias, out, ld_out, N, B * H * W, H, W, dil, 0, 0, 0, 0, 0};
  a.epi = PpEpi{bn_mode, scale, shift, slope, stats, 0, ppg, groups};
  a.lazy = lazy;
  PP_CHECK_ARG(!lazy.coef || f16x3, "conv3x3_fwd_bn_lazy: a lazy input needs the split-fp16 kernels");
  if (bn_mode == 1 && stats_bytes < PP_FN(pp_conv3x3_bn_stats_bytes)(N, B, H, W, groups)) {
    pp_set_error("conv3x3_fwd_bn: stats buffer too small (%zu < %zu)", stats_bytes, PP_FN(pp_conv3x3_bn_stats_bytes)(N, B, H, W, groups));
    return PP_ERR_WORKSPACE;
  }
  bool fused = false;
  int rows = 0;
  if (int rc = f16x3 ? conv_dispatch_f16x3(a, in_amax, s, &fused, &rows) : conv_dispatch(a, s, &fused, &rows)) return rc;
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

extern "C" int PP_FN(pp_conv3x3_fwd_bn)(const pp_act* in, int ld_in, int C, const void* wf, const float* bias, pp_act* out,
                                 int ld_out, int N, int B, int H, int W, int dil, int f16x3, const float* in_amax,
                                 int bn_mode, const float* scale, const float* shift, float slope, int groups,
                                 double* stats, size_t stats_bytes, int* rows_out, void* stream) {
  return conv3x3_fwd_bn_impl(in, ld_in, C, wf, bias, out, ld_out, N, B, H, W, dil, f16x3, in_amax, bn_mode, scale, shift, slope, groups,
                             stats, stats_bytes, rows_out, pp_lazy_none(), stream);
}

// 1 when pp_conv3x3_fwd_bn_lazy and pp_conv3x3_bwd_weight_f16x3_lazy accept a lazy input for this layer shape (the two-half halo
// kernel forward, the halo-tile kernels for the weight gradient), else 0: a pure function of the shape, asked once per plan.
extern "C" int PP_FN(pp_conv3x3_lazy_ok)(int C, int N, int B, int H, int W, int dil) {
  ConvArgs a{nullptr, C, C, nullptr, nullptr, nullptr, N, N, B * H * W, H, W, dil, 0, 0, 0, 0, 0};
  static float dummy;
  a.lazy = PpLazy{&dummy, C, B};
  return (C % 4 == 0 && halo_f16_rows(a) == 1 && halo2_ok(a, 1) > 0 && wgrad_lazy_ok(N, C, H, W, dil)) ? 1 : 0;
}

// the same with a LAZY input tensor (pp_lazy_in): `in` holds the raw convolution output of the layer in front, BatchNorm +
// LeakyReLU are applied while the two-half halo kernel stages its patches (at most two statistics groups)
extern "C" int PP_FN(pp_conv3x3_fwd_bn_lazy)(const pp_act* in, int ld_in, int C, const void* wf, const float* bias, pp_act* out,
                                      int ld_out, int N, int B, int H, int W, int dil, int f16x3, const float* in_amax,
                                      int bn_mode, const float* scale, const float* shift, float slope, int groups,
                                      double* stats, size_t stats_bytes, int* rows_out, const pp_lazy_in* lazy_in, void* stream) {
  PpLazy lz = pp_lazy_none();
  if (lazy_in && lazy_in->coef) {
    PP_CHECK_ARG(lazy_in->groups >= 1 && lazy_in->groups <= PP_EPI_GROUPS && B % lazy_in->groups == 0,
                 "conv3x3_fwd_bn_lazy: 1 or 2 lazy groups that divide the batch");
    PP_CHECK_ARG(lazy_in->ld % 4 == 0 && lazy_in->ld >= C && ((uintptr_t)lazy_in->coef & 15) == 0, "conv3x3_fwd_bn_lazy: bad coefficient rows");
    lz = PpLazy{lazy_in->coef, lazy_in->ld, B / lazy_in->groups};
  }
  return conv3x3_fwd_bn_impl(in, ld_in, C, wf, bias, out, ld_out, N, B, H, W, dil, f16x3, in_amax, bn_mode, scale, shift, slope, groups,
                             stats, stats_bytes, rows_out, lz, stream);
}

// ------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------
#undef BK      // from here on the stage depth is the wgrad kernel's template parameter

struct WgradArgs {
  const act_t* dz; int ld_dz; int O;
  const act_t* x; int ld_x; int C;          // C = padded input channels (multiple of 4)
  float* part;                              // [splits][O][9][C]
  int P, H, W, dil;
  int o_tiles, c_tiles, chunks_per_split, n_chunks;
  unsigned dz_bytes, x_bytes;
};

template <int TM, int TN, int WAVES_M, int WAVES_N, int WAVES_K, int BKP>
__global__ __launch_bounds__(WAVES_M* WAVES_N* WAVES_K * 64) void conv3x3_wgrad_kernel(WgradArgs a) {
  constexpr int BK = BKP;                   // pixels per LDS stage (shadows the file-level BK)
  constexpr int NT = WAVES_M * WAVES_N * WAVES_K * 64;
  constexpr int BM = 32 * TM * WAVES_M;     // output channels per block
  constexpr int BN = 32 * TN * WAVES_N;     // input channels per block
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int F4A = BM / 4, F4B = BN / 4;
  constexpr int RPPA = NT / F4A, RPPB = NT / F4B;
  constexpr int PASSA = BK / RPPA, PASSB = BK / RPPB;
  static_assert(NT % F4A == 0 && NT % F4B == 0 && BK % RPPA == 0 && BK % RPPB == 0, "bad wgrad tile");
  static_assert((BK / 2) % WAVES_K == 0, "k-pairs must split evenly over WAVES_K");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                          // [2][BK][LDA]  dz rows (pixel-major)
  float* Bs = smem + 2 * BK * LDA;           // [2][BK][LDB]  shifted x rows

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wk = wv % WAVES_K, wmn = wv / WAVES_K;
  const int wm = wmn / WAVES_N, wn = wmn % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;

  int tap, ct, ot, split;
  wgrad_item(a.c_tiles, a.o_tiles, tap, ct, ot, split);
  const int o0 = ot * BM, c0 = ct * BN;
  const int dy = (tap / 3 - 1) * a.dil, dx = (tap % 3 - 1) * a.dil;
  const int shift = dy * a.W + dx;

  const int ca = tid % F4A, ra0 = tid / F4A;
  const int cb = tid % F4B, rb0 = tid / F4B;
  const bool oa_ok = o0 + ca * 4 < a.O;      // O is a multiple of 4 whenever this is a partial tile
  const bool cb_ok = c0 + cb * 4 < a.C;

  const int chunk_lo = split * a.chunks_per_split;
  int chunk_hi = chunk_lo + a.chunks_per_split;
  if (chunk_hi > a.n_chunks) chunk_hi = a.n_chunks;

  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, a.dz_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  f32x4 ra[PASSA], rb[PASSB];
  // image coordinates of the x rows this thread stages, advanced incrementally chunk by chunk (no division
  // in the loop: the address arithmetic of a stage must stay small next to its MFMAs)
  int bx[PASSB], by[PASSB];
#pragma unroll
  for (int i = 0; i < PASSB; ++i) {
    const int p = chunk_lo * BK + rb0 + i * RPPB;
    bx[i] = p % a.W;
    by[i] = (p / a.W) % a.H;
  }
  const int step_y = (BK / a.W) % a.H, step_x = BK % a.W;     // advance of BK pixels in (y, x)
  // masked lanes (ragged tails, halo) get an out-of-range offset: the buffer bounds check returns 0, no branch
  auto load_tile = [&](int chunk) {
    const int pk = chunk * BK;
#pragma unroll
    for (int i = 0; i < PASSA; ++i) {
      const int p = pk + ra0 + i * RPPA;
      const int ok = (int)oa_ok & (int)(p < a.P);
      const unsigned off = ok ? (unsigned)(p * a.ld_dz + o0 + ca * 4) * (unsigned)PP_ACT_BYTES : 0xffffffffu;
      ra[i] = act_buf_ld4(rs_dz, off, 0);
    }
#pragma unroll
    for (int i = 0; i < PASSB; ++i) {
      const int p = pk + rb0 + i * RPPB;
      const int ok = (int)cb_ok & (int)(p < a.P) & (int)((unsigned)(by[i] + dy) < (unsigned)a.H) &
                     (int)((unsigned)(bx[i] + dx) < (unsigned)a.W);
      const unsigned off = ok ? (unsigned)((p + shift) * a.ld_x + c0 + cb * 4) * (unsigned)PP_ACT_BYTES : 0xffffffffu;
      rb[i] = act_buf_ld4(rs_x, off, 0);
      bx[i] += step_x;
      by[i] += step_y;
      if (bx[i] >= a.W) { bx[i] -= a.W; by[i] += 1; }
      if (by[i] >= a.H) by[i] -= a.H;
    }
  };
  auto store_tile = [&](int buf) {
    float* Ab = As + buf * BK * LDA;
    float* Bb = Bs + buf * BK * LDB;
#pragma unroll
    for (int i = 0; i < PASSA; ++i) *reinterpret_cast<f32x4*>(Ab + (ra0 + i * RPPA) * LDA + ca * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < PASSB; ++i) *reinterpret_cast<f32x4*>(Bb + (rb0 + i * RPPB) * LDB + cb * 4) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (chunk_lo < chunk_hi) {
    load_tile(chunk_lo);
    store_tile(0);
  }
  __syncthreads();
  for (int ch = chunk_lo; ch < chunk_hi; ++ch) {
    const int buf = (ch - chunk_lo) & 1;
    if (ch + 1 < chunk_hi) load_tile(ch + 1);
    const float* Ab = As + buf * BK * LDA + wm * TM * 32 + lr;
    const float* Bb = Bs + buf * BK * LDB + wn * TN * 32 + lr;
    constexpr int KP = BK / 2 / WAVES_K;       // k-pairs per wave per chunk
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      const int krow = 2 * (wk * KP + kk) + lh;
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = Ab[krow * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bb[krow * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (ch + 1 < chunk_hi) store_tile(buf ^ 1);
    __syncthreads();
  }

  // reduce the WAVES_K partial accumulators through LDS (fixed order -> deterministic)
  if (WAVES_K > 1) {
    float* red = smem;                         // reuse: [WAVES_K-1][WAVES_M*WAVES_N][TM*TN*16][64]
    constexpr int PER_WAVE = TM * TN * 16 * 64;
    if (wk > 0) {
      float* dst = red + ((wk - 1) * (WAVES_M * WAVES_N) + wmn) * PER_WAVE;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wk == 0) {
      for (int k = 1; k < WAVES_K; ++k) {
        const float* src = red + ((k - 1) * (WAVES_M * WAVES_N) + wmn) * PER_WAVE;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * TN + j) * 16 + r) * 64 + lane];
      }
    }
  }
  if (wk != 0) return;

  float* part = a.part + (size_t)split * a.O * 9 * a.C;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int c = c0 + (wn * TN + j) * 32 + lr;
    if (c >= a.C) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (o < a.O) part[((size_t)o * 9 + tap) * a.C + c] = acc[i][j][r];
      }
  }
}

// (Measured and dropped, r01: a variant of this kernel that transposes while staging -- each lane loading one channel
//  for 16 consecutive pixels with dword loads, [channel][pixel] LDS image, the forward kernel's ds_read_b128 MFMA loop
//  -- ran 88-97 TFLOP/s against 98-107 for the pixel-major form below on the 256..1024-channel layers.)

// ------------------------------------------------------------------------------------------
// Tap-fused weight gradient for the high-resolution, few-channel layers (dilation 1, W % 64 == 0).
// The per-tap kernel above re-reads dz and x once per tap: at 256x256 with 32 channels that is 9 GB of HBM reads per
// launch (profiles/r01_hbm_traffic_per_launch.json) and the kernel runs at the HBM rate, not the MFMA rate.  Here a
// block stages ONE 64-pixel row segment of dz and the 3 x 66 pixel halo of x in LDS and produces all nine taps from
// it: lane (c, k) of the B operand of tap (ty,tx) is xs[ty][k + tx][c].  The four waves split the 64 pixels (16
// each), keep 9 accumulator tiles, and each wave writes its own split-K partial (no cross-wave reduction).
// ------------------------------------------------------------------------------------------
#ifndef PP_ACT_16     // the tap-fused fp32 weight-gradient kernel: fp32 activations only
struct Wgrad9Args {
  const float* dz; int ld_dz; int O;
  const float* x; int ld_x; int C;
  float* part;                              // [splits][O][9][C]
  int P, H, W;
  int o_tiles, c_tiles, segs_per_split, n_segs;
  unsigned dz_bytes, x_bytes;
};

#define W9_SEG 64
#define W9_LD 36
__global__ __launch_bounds__(256) void conv3x3_wgrad9_kernel(Wgrad9Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int DZ_F = W9_SEG * W9_LD;                  // floats per dz stage
  constexpr int XS_F = 3 * (W9_SEG + 2) * W9_LD;        // floats per x halo stage
  float* dzs = smem;                                    // [2][64][36]
  float* xs = smem + 2 * DZ_F;                          // [2][3][66][36]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int ct = blockIdx.x % a.c_tiles, ot = blockIdx.x / a.c_tiles;
  const int o0 = ot * 32, c0 = ct * 32;
  const int split = blockIdx.y;
  const int seg_lo = split * a.segs_per_split;
  int seg_hi = seg_lo + a.segs_per_split;
  if (seg_hi > a.n_segs) seg_hi = a.n_segs;
  const int segs_per_row = a.W / W9_SEG;
  const int q = tid & 7, r0 = tid >> 3;                 // 8 float4 per 32-channel row, 32 rows per pass
  const bool o_ok = o0 + q * 4 < a.O, c_ok = c0 + q * 4 < a.C;

  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, a.dz_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  f32x4 rdz[2], rx[7];
  // per-thread constants of the halo rows this thread stages: (tap row, column) of each of its 7 float4s
  int hty[7], hxx[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int r = r0 + i * 32;
    hty[i] = r / (W9_SEG + 2);
    hxx[i] = r - hty[i] * (W9_SEG + 2);
  }
  auto load_seg = [&](int seg) {
    const int row = seg / segs_per_row;                 // global image row index (n*H + y)
    const int x0 = (seg - row * segs_per_row) * W9_SEG;
    const int y = row % a.H;
    const int p0 = row * a.W + x0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = r0 + i * 32;
      const unsigned off = o_ok ? (unsigned)((p0 + r) * a.ld_dz + o0 + q * 4) * 4u : 0xffffffffu;
      rdz[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dz, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int yy = y + hty[i] - 1, gx = x0 + hxx[i] - 1;
      const int ok = (int)c_ok & (int)(hty[i] < 3) & (int)((unsigned)yy < (unsigned)a.H) & (int)((unsigned)gx < (unsigned)a.W);
      const unsigned off = ok ? (unsigned)(((row + hty[i] - 1) * a.W + gx) * a.ld_x + c0 + q * 4) * 4u : 0xffffffffu;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
    }
  };
  auto store_seg = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<f32x4*>(dzs + buf * DZ_F + (r0 + i * 32) * W9_LD + q * 4) = rdz[i];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int r = r0 + i * 32;
      if (r < 3 * (W9_SEG + 2)) *reinterpret_cast<f32x4*>(xs + buf * XS_F + r * W9_LD + q * 4) = rx[i];
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (seg_lo < seg_hi) {
    load_seg(seg_lo);
    store_seg(0);
  }
  __syncthreads();
  for (int seg = seg_lo; seg < seg_hi; ++seg) {
    const int buf = (seg - seg_lo) & 1;
    const bool more = seg + 1 < seg_hi;
    if (more) load_seg(seg + 1);
    const float* A = dzs + buf * DZ_F + lr;
    const float* B = xs + buf * XS_F + lr;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k = wv * 16 + 2 * kk + lh;              // pixel of the segment this lane supplies
      const float av = A[k * W9_LD];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float bv = B[((t / 3) * (W9_SEG + 2) + k + (t % 3)) * W9_LD];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
    if (more) store_seg(buf ^ 1);
    __syncthreads();
  }
  // cross-wave reduction, one tap tile at a time through LDS (fixed order -> deterministic)
  float* red = smem;                                    // [3][1024]
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
    if (wv > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wv - 1) * 1024 + r * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc[t][r] += red[r * 64 + lane] + red[1024 + r * 64 + lane] + red[2048 + r * 64 + lane];
    }
  }
  if (wv != 0) return;
  float* part = a.part + (size_t)split * a.O * 9 * a.C;
  const int c = c0 + lr;
  if (c < a.C) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (o < a.O) part[((size_t)o * 9 + t) * a.C + c] = acc[t][r];
      }
  }
}

struct Wgrad9Plan { int o_tiles, c_tiles, n_segs, splits, segs_per_split; };
static bool wgrad9_applicable(int O, int C, int H, int W, int dil) {
  return dil == 1 && W % W9_SEG == 0 && O % 4 == 0 && O <= 64 && C <= 192;
}
static Wgrad9Plan wgrad9_plan(int O, int C, int P) {
  Wgrad9Plan p;
  p.o_tiles = pp_cdiv(O, 32);
  p.c_tiles = pp_cdiv(C, 32);
  p.n_segs = P / W9_SEG;
  int splits = pp_cdiv(768, p.o_tiles * p.c_tiles);     // ~3 blocks per CU over the launch
  const int max_splits = pp_cdiv(p.n_segs, 16);         // >= 16 segments (1024 pixels) per block
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.segs_per_split = pp_cdiv(p.n_segs, splits);
  p.splits = pp_cdiv(p.n_segs, p.segs_per_split);
  return p;
}

#endif  // !PP_ACT_16

// dw_oihw[o][c][tap] (+)= sum_s part[s][o][tap][c]   for c < I_true
// Round 5: 16 QUADS of consecutive partial elements x 16 split-lanes per block, four independent 16-byte loads in flight per
// thread (the round-1 form -- one element per thread, 64-byte reads -- ran at 1 TB/s of the 38-45 MB of per-block partials the
// persistent weight-gradient kernels leave behind: 0.53 ms per step over 12 launches); LDS combine in fixed order.
__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const float* __restrict__ part, int splits, int O, int Cpad,
                                                             int I_true, float* dw, int accumulate) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  __shared__ v4 red[16][16];
  const size_t per4 = (size_t)O * 9 * Cpad / 4;                         // (Cpad is a multiple of 4)
  const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const size_t q = (size_t)blockIdx.x * 16 + il;
  v4 s = v4{0.f, 0.f, 0.f, 0.f};
  if (q < per4) {
    const v4* p = reinterpret_cast<const v4*>(part) + q;
    int k = sl;
    for (; k + 48 < splits; k += 64) {
      const v4 a = p[(size_t)k * per4], b = p[(size_t)(k + 16) * per4], c = p[(size_t)(k + 32) * per4], d = p[(size_t)(k + 48) * per4];
      s += a; s += b; s += c; s += d;
    }
    for (; k < splits; k += 16) s += p[(size_t)k * per4];
  }
  red[sl][il] = s;
  __syncthreads();
  if (sl != 0 || q >= per4) return;
  v4 t = red[0][il];
#pragma unroll
  for (int i = 1; i < 16; ++i) t += red[i][il];
  const size_t idx = q * 4;
  const int c = (int)(idx % Cpad);
  const int tap = (int)((idx / Cpad) % 9);
  const int o = (int)(idx / ((size_t)Cpad * 9));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (c + j >= I_true) break;
    float* d = dw + ((size_t)o * I_true + c + j) * 9 + tap;
    *d = accumulate ? *d + t[j] : t[j];
  }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the first layer (x padded to 4 channels): dW[n][tap][c] = sum_p dz[p][n] * x[p + off(tap)][c]
// is a (Cout x 36) output with K = all pixels.  On v_mfma_f32_16x16x4_f32 one instruction consumes 4 pixels:
// A[row n][k = pixel] straight from dz (16 channels x 4 pixels per wave load), B[k = pixel][col = (tap, c)] gathered
// from x through the L1 (3 column tiles of 16 cover the 36 columns) -- no LDS staging at all.  Each wave owns a
// contiguous pixel range; a block sums its 4 waves through LDS and writes one [Cout][9][4] partial (fixed-order
// finalize as for the other weight-gradient kernels).  The tap-fused kernel took 0.78 ms here (32-wide MFMA tiles
// that are 7/8 zeros), this form is bound by reading dz once.
// ------------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));
struct WgradC4Args {
  const act_t* dz; int ld_dz; int O;
  const act_t* x; int ld_x;
  float* part;                 // [blocks][O][9][4]
  int P, H, W, dil;
  int px_per_wave;             // multiple of 16
  unsigned dz_bytes, x_bytes;
};

template <int MT>              // MT = O / 16 (1..4)
__global__ __launch_bounds__(256) void conv3x3_c4_wgrad_kernel(WgradC4Args a) {
  __shared__ float red[4][MT * 3][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l16 = lane & 15, k = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, a.dz_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  // this lane's three B columns: (tap, c) = (col >> 2, col & 3), col = nt * 16 + l16 (cols >= 36 are padding)
  int bdy[3], bdx[3], boff[3], bok[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    const int col = nt * 16 + l16, tap = col >> 2, c = col & 3;
    bok[nt] = col < 36;
    bdy[nt] = (tap / 3 - 1) * a.dil;
    bdx[nt] = (tap % 3 - 1) * a.dil;
    boff[nt] = (bdy[nt] * a.W + bdx[nt]) * a.ld_x + c;
  }
  f32x4_t acc[MT][3];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const long long w_id = (long long)blockIdx.x * 4 + wv;
  long long pb = w_id * a.px_per_wave, pe = pb + a.px_per_wave;
  if (pe > a.P) pe = a.P;
  for (int p0 = (int)pb; p0 < (int)pe; p0 += 16) {
    // 4 steps of 4 pixels per iteration: 20 independent loads in flight per lane before the first MFMA
    float av[4][MT], bv[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q0 = p0 + 4 * u;                   // W % 4 == 0: the 4 pixels of a step share a row
      const int live = q0 < (int)pe;
      const int x0 = q0 % a.W, y0 = (q0 / a.W) % a.H;
      const int p = q0 + k;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const unsigned off = live ? (unsigned)(p * a.ld_dz + mt * 16 + l16) * (unsigned)PP_ACT_BYTES : 0xffffffffu;
        av[u][mt] = act_buf_ld1(rs_dz, off, 0);
      }
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) {
        const int ok = live & bok[nt] & (int)((unsigned)(y0 + bdy[nt]) < (unsigned)a.H) &
                       (int)((unsigned)(x0 + k + bdx[nt]) < (unsigned)a.W);
        const unsigned off = ok ? (unsigned)(p * a.ld_x + boff[nt]) * (unsigned)PP_ACT_BYTES : 0xffffffffu;
        bv[u][nt] = act_buf_ld1(rs_x, off, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mt], bv[u][nt], acc[mt][nt], 0, 0, 0);
  }
  // block reduction in fixed wave order, then one partial per block
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
      *reinterpret_cast<f32x4_t*>(&red[wv][mt * 3 + nt][lane][0]) = acc[mt][nt];
  __syncthreads();
  float* part = a.part + (size_t)blockIdx.x * a.O * 36;
  for (int e = tid; e < MT * 3 * 64 * 4; e += 256) {
    const int r = e & 3, ln = (e >> 2) & 63, tile = e >> 8;
    const int mt = tile / 3, nt = tile % 3;
    const float v = ((red[0][tile][ln][r] + red[1][tile][ln][r]) + red[2][tile][ln][r]) + red[3][tile][ln][r];
    const int n = mt * 16 + 4 * (ln >> 4) + r;     // D[row = 4*(lane>>4) + r][col = lane & 15]
    const int col = nt * 16 + (ln & 15);
    if (col < 36) part[(size_t)n * 36 + col] = v;
  }
}

static bool wgrad_c4_applicable(int O, int Cpad, int W) {
  return Cpad == 4 && O % 16 == 0 && O <= 64 && W % 4 == 0;
}
struct WgradC4Plan { int blocks, px_per_wave; };
static WgradC4Plan wgrad_c4_plan(int P) {
  WgradC4Plan q;
  int waves = pp_cdiv(P, 256);                     // >= 256 pixels per wave
  if (waves > 4096) waves = 4096;
  if (waves < 4) waves = 4;
  q.px_per_wave = pp_cdiv(pp_cdiv(P, waves), 16) * 16;
  q.blocks = pp_cdiv(pp_cdiv(P, q.px_per_wave), 4);
  return q;
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the narrow layers on the fp16 MFMA with split operands (dilation 1, W % 32 == 0, H % 4 == 0,
// O % 32 == 0, C % 32 == 0).  A block owns one (32 output channels) x (32 input channels) pair and walks over 4 x 32-pixel
// tiles like the halo-tile forward kernel: the dz tile and the 6 x 34 halo patch of x are staged once as [pixel][channel]
// fp16 hi / lo images -- exactly as they lie in memory -- and the reduction-major MFMA operands are produced by
// ds_read_b64_tr_b16 (see wino_wgrad_gemm_f16x3_kernel, pp_wino.hip): the nine taps read the same patch at shifted
// pixel rows.  64-byte image rows put the bank of (pixel q, 16-channel block mb, 8-byte piece p) at 16 q + 8 mb + 2 p:
// conflict-free without padding.  Wave w reduces over row w of every tile into nine accumulator tiles that live for
// the whole kernel; each wave writes its own partial and the fixed-order finalize sums them (deterministic).
// One accumulator set instead of the forward kernels' two: dz is scaled into [2^9, 2^10) by its amax, which keeps
// its low part above the fp16 subnormals unscaled, and the 2^-11 weight of the x low part (stored times 2^11, x is
// not rescaled) is applied to the dz fragment instead:  dz*x ~ dzh*xh + (dzh * 2^-11) * xl' + dzl * xh.
// ------------------------------------------------------------------------------------------
struct WgradH16Args {
  const act_t* dz; int ld_dz; int O;
  const act_t* x; int ld_x; int C;
  float* part;                 // [gridDim.x * 4][O][9][C]
  int P, H, W;
  int c_tiles, tiles_x, tiles_y, n_tiles;
  unsigned dz_bytes, x_bytes;
  int walkers, per_walker;     // 1-D grid of walkers * per_walker blocks, see wh_walker_pair
  PpLazy lazy;                 // x is a lazy tensor (pp_common.h): BatchNorm + LeakyReLU applied while the patch goes to LDS
};
// Block -> (tile walker, channel pair / group).  The `per_walker` blocks that walk the SAME tile sequence with different
// channel pairs re-read the same dz tile and x patch; blocks L, L + 8, ... share an XCD and its L2, so they get consecutive
// slots of one XCD (walkers is a multiple of 8).  With the 2-D grid of round 3 they sat gridDim.x blocks apart -- on other XCDs
// whenever that was not a multiple of 8 -- and conv3x3_wgrad_halo_f16x3_kernel fetched 1.9 GB per launch for 0.7 GB of
// operands (r03 PMC profile: 96 -> 32 channels at 256^2 read dz three times from HBM).
__device__ __forceinline__ void wh_walker_pair(const WgradH16Args& a, int& w, int& p) {
  const int L = blockIdx.x, slot = L >> 3;
  w = (L & 7) + 8 * (slot / a.per_walker);
  p = slot % a.per_walker;
}
#define WH_RS 32                                   // halves per image row (64 B)
#define WH_DZ_PIX (HT_ROWS * HT_COLS)              // 128
#define WH_THREADS 512                             // 8 waves: wave -> (tile row = wv & 3, 16-pixel half = wv >> 2)
#define WH_DZ_PASS (WH_DZ_PIX * 8 / WH_THREADS)    // 2 float4 loads per thread
#define WH_X_PASS ((HT_PIX * 8 + WH_THREADS - 1) / WH_THREADS)   // 4
template <bool LAZY>
__global__ __launch_bounds__(WH_THREADS) __attribute__((amdgpu_waves_per_eu(2)))
void conv3x3_wgrad_halo_f16x3_kernel(WgradH16Args a, const float* __restrict__ dz_amax) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef __fp16 h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
  extern __shared__ __attribute__((aligned(16))) _Float16 smem16[];
  _Float16* Dh = smem16;                           // dz hi [128][32]
  _Float16* Dl = Dh + WH_DZ_PIX * WH_RS;           // dz lo (unscaled)
  _Float16* Xh = Dl + WH_DZ_PIX * WH_RS;           // x hi [204][32]
  _Float16* Xl = Xh + HT_PIX * WH_RS;              // x lo * 2^11
  float* Lz = reinterpret_cast<float*>(Xl + HT_PIX * WH_RS);          // LAZY: [2 groups][3][32] coefficient rows of this block's channels
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row = wv & 3, hc = wv >> 2;
  int wk, pr;
  wh_walker_pair(a, wk, pr);
  const int ct = pr % a.c_tiles, ot = pr / a.c_tiles;
  const int o0 = ot * 32, c0 = ct * 32;
  float s_in, s_out;
  f16_scales(dz_amax, s_in, s_out);
  const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, a.dz_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  // staging: thread -> (pixel, channel quad); pass i handles pixel (tid >> 3) + 64 i.  Everything but two offsets is
  // recomputed per tile (a few integer ops in the shadow of the MFMAs): per-pass tables in registers pushed the
  // 4-wave version of this kernel over 256 VGPRs, and its spills (of the prefetched tile itself) serialised the
  // prefetch -- 66 % of the wave time parked in waits, 27 % of the matrix pipe (r02 profile).  Eight waves share one
  // staged tile, so every thread prefetches 6 float4 instead of 11.
  const int q4 = tid & 7, pix0 = tid >> 3;         // pix0 in [0, 64)
  const int lds0 = pix0 * WH_RS + q4 * 4;          // + 64 * WH_RS per pass
  f32x4 rx[WH_X_PASS], rd[WH_DZ_PASS];
  unsigned s_ok = 0;                               // LAZY: which passes of the staged patch lie inside the image, and its group
  int s_grp = 0;
  if (LAZY) {
    const int groups = (a.P / (a.H * a.W) + a.lazy.imgs_per_group - 1) / a.lazy.imgs_per_group;       // <= 2 (host)
    if (tid < 192) {
      const int c = tid & 31, rw = (tid >> 5) % 3, g = tid / 96;
      Lz[tid] = g < groups ? a.lazy.coef[(size_t)(g * 3 + rw) * a.lazy.ld + c0 + c] : (rw == 1 ? 0.f : 1.f);
    }
  }
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    const int tx = t % a.tiles_x, r = t / a.tiles_x, ty = r % a.tiles_y, img = r / a.tiles_y;
    if (LAZY) { s_ok = 0; s_grp = img >= a.lazy.imgs_per_group ? 1 : 0; }
    const int y0 = ty * HT_ROWS, x0 = tx * HT_COLS;
    const int pbase = (img * a.H + y0) * a.W + x0;
#pragma unroll
    for (int i = 0; i < WH_DZ_PASS; ++i) {         // dz tile: pixel (row 2 i + (pix0 >> 5), column pix0 & 31)
      const unsigned off = (unsigned)((pbase + (2 * i + (pix0 >> 5)) * a.W + (pix0 & 31)) * a.ld_dz + o0 + q4 * 4) * (unsigned)PP_ACT_BYTES;
      rd[i] = act_buf_ld4(rs_dz, off, 0);
    }
    const int xbase = (pbase - a.W - 1) * a.ld_x + c0 + q4 * 4;
#pragma unroll
    for (int i = 0; i < WH_X_PASS; ++i) {          // 6 x 34 halo patch of x
      const int pix = pix0 + 64 * i;
      const int hy = pix / HT_HC, hx = pix - hy * HT_HC;
      const int ok = (int)(pix < HT_PIX) & (int)((unsigned)(y0 - 1 + hy) < (unsigned)a.H) & (int)((unsigned)(x0 - 1 + hx) < (unsigned)a.W);
      const unsigned off = ok ? (unsigned)(xbase + (hy * a.W + hx) * a.ld_x) * (unsigned)PP_ACT_BYTES : 0xffffffffu;
      rx[i] = act_buf_ld4(rs_x, off, 0);
      if (LAZY) s_ok |= (unsigned)ok << i;
    }
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WH_DZ_PASS; ++i) {
      const f32x4 v = rd[i] * s_in;
      const f16x4 hi = __builtin_convertvector(v, f16x4);
      *reinterpret_cast<f16x4*>(Dh + lds0 + i * 64 * WH_RS) = hi;
      if (PP_ACT_LO) {
        const f16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), f16x4);
        *reinterpret_cast<f16x4*>(Dl + lds0 + i * 64 * WH_RS) = lo;
      }
    }
    f32x4 l_sc, l_sh, l_sl;
    if (LAZY) {
      const float* r = Lz + s_grp * 96 + q4 * 4;
      l_sc = *reinterpret_cast<const f32x4*>(r);
      l_sh = *reinterpret_cast<const f32x4*>(r + 32);
      l_sl = *reinterpret_cast<const f32x4*>(r + 64);
    }
#pragma unroll
    for (int i = 0; i < WH_X_PASS; ++i)
      if (pix0 + 64 * i < HT_PIX) {
        f32x4 v = rx[i];
        if (LAZY) v = ((s_ok >> i) & 1u) ? pp_lazy_apply4(v, l_sc, l_sh, l_sl) : f32x4{0.f, 0.f, 0.f, 0.f};      // zero padding of y
        const f16x4 hi = __builtin_convertvector(v, f16x4);
        *reinterpret_cast<f16x4*>(Xh + lds0 + i * 64 * WH_RS) = hi;
        if (PP_ACT_LO) {
          const f16x4 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * F16_LO_SCALE, f16x4);
          *reinterpret_cast<f16x4*>(Xl + lds0 + i * 64 * WH_RS) = lo;
        }
      }
  };
  auto frag = [&](const _Float16* img, int pixel0) __attribute__((always_inline)) -> f16x8 {       // 16 pixels x 32 channels, reduction-major
    // 16-lane group g = (channel block mb, k-half h); lane 4q+p of the group addresses pixel row q, piece p
    const int g = lane >> 4, i16 = lane & 15;
    const _Float16* p0 = img + (pixel0 + 8 * (g >> 1) + (i16 >> 2)) * WH_RS + 16 * (g & 1) + 4 * (i16 & 3);
    const h4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)p0);
    const h4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)(p0 + 4 * WH_RS));
    return __builtin_shufflevector(__builtin_bit_cast(f16x4, v0), __builtin_bit_cast(f16x4, v1), 0, 1, 2, 3, 4, 5, 6, 7);
  };
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const f16x8 two_m11 = {(_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f,
                         (_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f};
  int t = wk;
  if (t < a.n_tiles) load_tile(t);
  for (; t < a.n_tiles; t += a.walkers) {
    __syncthreads();                               // every wave is done with the previous tile's images
    store_tile();
    __syncthreads();
    if (t + a.walkers < a.n_tiles) load_tile(t + a.walkers);
    // This wave's 16-pixel reduction block (tile row `row`, columns 16 hc ..) x nine taps, three MFMAs per tap.
    // Software pipeline: the x fragments of tap t+1 are read while the MFMAs of tap t run (left to itself hipcc
    // issued every tap's four transposed reads directly in front of its MFMAs and waited for them).
    f16x8 bh[2], bl[2];
    const f16x8 ah = frag(Dh, row * 32 + 16 * hc);
    f16x8 al = ah;
    if (PP_ACT_LO) al = frag(Dl, row * 32 + 16 * hc);
    bh[0] = frag(Xh, row * HT_HC + 16 * hc);
    if (PP_ACT_LO) bl[0] = frag(Xl, row * HT_HC + 16 * hc);
    const f16x8 ahs = ah * two_m11;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int cur = tap & 1;
      if (tap + 1 < 9) {
        const int pix0n = (row + (tap + 1) / 3) * HT_HC + 16 * hc + (tap + 1) % 3;
        bh[cur ^ 1] = frag(Xh, pix0n);
        if (PP_ACT_LO) bl[cur ^ 1] = frag(Xl, pix0n);
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[cur], acc[tap], 0, 0, 0);
      if (PP_ACT_LO) {
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahs, bl[cur], acc[tap], 0, 0, 0);
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[cur], acc[tap], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // The two waves of a tile row (hc = 0, 1) add their accumulators through LDS, tap by tap (4 KB per wave), in a
  // fixed order; then one partial per row: D[row = o = (r&3) + 8*(r>>2) + 4*lh][col = c = lr]
  const int lr = lane & 31, lh = lane >> 5;
  float* xch = reinterpret_cast<float*>(smem16) + (size_t)row * 1024;          // [16][64] floats per row pair
  float* part = a.part + ((size_t)(wk * 4 + row) * a.O) * 9 * a.C;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    __syncthreads();                               // images / the previous tap's exchange are no longer read
    if (hc == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[r * 64 + lane] = acc[tap][r];
    }
    __syncthreads();
    if (hc == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        part[((size_t)o * 9 + tap) * a.C + c0 + lr] = (acc[tap][r] + xch[r * 64 + lane]) * s_out;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// conv3x3_wgrad_halo_mp_f16x3_kernel: the same weight gradient with SEVERAL (32 outputs x 32 inputs) pairs per block.
// In the one-pair kernel above a 64 x 64-channel layer is four blocks that each stage the same dz tile and the same x
// patch (global loads, fp32 -> hi / lo conversion, LDS stores) for 27 MFMAs per wave: staging dominates (r02 trace:
// 4300 cycles per tile, 864 of them MFMAs).  Here a block owns OB x CB = (32 OBK) x (32 CBK) channels, stages the
// (wider) tile once, and its eight waves are (pair, pixel split): 2 pairs x 4 splits as launched (the template also
// covers 4 x 2, which needs more registers than a wave has).  A wave reduces
// over 128 / splits pixels (units of 16) for its pair, i.e. 27 MFMAs per unit; staging per MFMA drops by
// 2 OBK CBK / (OBK + CBK).  Images stay one [pixels][32 channels] block of 64-byte rows per 32-channel group, so the
// transposed fragment reads are the conflict-free ones of the one-pair kernel.  The pixel splits of a pair are summed
// through LDS in a fixed order: one partial per block.
// ------------------------------------------------------------------------------------------
template <int OBK, int CBK, bool LAZY>
__global__ __launch_bounds__(WH_THREADS) __attribute__((amdgpu_waves_per_eu(2)))
void conv3x3_wgrad_halo_mp_f16x3_kernel(WgradH16Args a, const float* __restrict__ dz_amax) {
  constexpr int PAIRS = OBK * CBK, SPLITS = 8 / PAIRS, UNITS = 8 / SPLITS;
  constexpr int DZ_PASS = WH_DZ_PIX * 8 * OBK / WH_THREADS;                      // 2 OBK
  constexpr int X_PASS = (HT_PIX * 8 * CBK + WH_THREADS - 1) / WH_THREADS;       // 4 (CBK = 1) or 7
  constexpr int D_IMG = WH_DZ_PIX * WH_RS, X_IMG = HT_PIX * WH_RS;               // halves per 32-channel image
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef __fp16 h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
  extern __shared__ __attribute__((aligned(16))) _Float16 smem16[];
  _Float16* Dh = smem16;                           // dz hi [OBK][128][32]
  _Float16* Dl = Dh + OBK * D_IMG;                 // dz lo (unscaled)
  _Float16* Xh = Dl + OBK * D_IMG;                 // x hi [CBK][204][32]
  _Float16* Xl = Xh + CBK * X_IMG;                 // x lo * 2^11
  float* Lz = reinterpret_cast<float*>(Xl + CBK * X_IMG);             // LAZY: [2 groups][3][32 CBK] coefficient rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wv % PAIRS, split = wv / PAIRS, po = pair / CBK, pc = pair % CBK;
  const int c_groups = (a.c_tiles + CBK - 1) / CBK;    // the last group of an odd chunk count (96 channels, CBK = 2) is half empty
  int wk, pr;
  wh_walker_pair(a, wk, pr);
  const int og = pr / c_groups, cg = pr % c_groups;
  const int o0 = og * 32 * OBK, c0 = cg * 32 * CBK;
  const bool pair_live = c0 + pc * 32 < a.C;           // this wave's channel pair exists (uniform per wave)
  float s_in, s_out;
  f16_scales(dz_amax, s_in, s_out);
  const __amdgpu_buffer_rsrc_t rs_dz = __builtin_amdgcn_make_buffer_rsrc((void*)a.dz, 0, a.dz_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  // staging constants: pass i of a thread handles quad e = tid + 512 i of the tile -> (pixel, 32-channel image, quad)
  // (512 is a multiple of the quads per pixel, so the quad of a thread is the same in every pass and the pixel advances by
  // DZ_STEP / X_STEP: LDS offsets and the dz offsets are linear in the pass -- one register each instead of one per pass)
  constexpr int DZ_STEP = WH_THREADS / (8 * OBK), X_STEP = WH_THREADS / (8 * CBK);     // pixels per pass
  static_assert(DZ_STEP % 32 == 0, "a dz pass must cover whole tile rows");
  const int dpix0 = tid / (8 * OBK), dq = tid % (8 * OBK), xpix0 = tid / (8 * CBK), xq = tid % (8 * CBK);
  const bool x_dead = c0 + (xq >> 3) * 32 >= a.C;      // this thread stages a 32-channel image beyond the layer's channels: zeros
  const int dz_rel0 = (((dpix0 >> 5) * a.W + (dpix0 & 31)) * a.ld_dz + o0 + dq * 4) * PP_ACT_BYTES, dz_rel_step = (DZ_STEP / 32) * a.W * a.ld_dz * PP_ACT_BYTES;
  const int dz_lds0 = (dq >> 3) * D_IMG + dpix0 * WH_RS + (dq & 7) * 4;
  const int x_lds0 = (xq >> 3) * X_IMG + xpix0 * WH_RS + (xq & 7) * 4;
  int x_rel[X_PASS];
  unsigned m_top = 0, m_bot = 0, m_left = 0, m_right = 0, m_dead = 0;
#pragma unroll
  for (int i = 0; i < X_PASS; ++i) {
    const int pix = xpix0 + X_STEP * i;
    const int hy = pix / HT_HC, hx = pix - hy * HT_HC;
    x_rel[i] = ((hy * a.W + hx) * a.ld_x + c0 + xq * 4) * PP_ACT_BYTES;
    if (pix >= HT_PIX || x_dead) { m_dead |= 1u << i; x_rel[i] = 0; }
    if (hy == 0) m_top |= 1u << i;
    if (hy == HT_ROWS + 1) m_bot |= 1u << i;
    if (hx == 0) m_left |= 1u << i;
    if (hx == HT_HC - 1) m_right |= 1u << i;
  }
  // tile cursor (uniform, advanced without divisions): tiles blockIdx.x, + gridDim.x, ...
  const int G = a.walkers;
  const int d_tx = G % a.tiles_x, d_q = G / a.tiles_x, d_ty = d_q % a.tiles_y, d_img = d_q / a.tiles_y;
  int t_next = wk, n_tx = t_next % a.tiles_x, n_ty = (t_next / a.tiles_x) % a.tiles_y, n_img = t_next / (a.tiles_x * a.tiles_y);
  act_raw4 rd[DZ_PASS], rx[X_PASS];      // prefetched tiles as loaded (16-bit storage: 8 bytes per quad, converted when staged)
  unsigned s_bad = 0;                              // LAZY: halo passes of the staged patch outside the image, and its group
  int s_grp = 0;
  if (LAZY) {
    const int groups = (a.P / (a.H * a.W) + a.lazy.imgs_per_group - 1) / a.lazy.imgs_per_group;       // <= 2 (host)
    for (int e = tid; e < 2 * 3 * 32 * CBK; e += WH_THREADS) {
      const int c = e % (32 * CBK), rw = (e / (32 * CBK)) % 3, g = e / (96 * CBK);
      Lz[e] = (g < groups && c0 + c < a.C) ? a.lazy.coef[(size_t)(g * 3 + rw) * a.lazy.ld + c0 + c] : (rw == 1 ? 0.f : 1.f);
    }
  }
  auto load_tile = [&]() __attribute__((always_inline)) {                         // tile (n_img, n_ty, n_tx); out of range past the last tile
    const bool live = t_next < a.n_tiles;
    const int pbase = (n_img * a.H + n_ty * HT_ROWS) * a.W + n_tx * HT_COLS;
    const int dbase = pbase * a.ld_dz * PP_ACT_BYTES, xbase = (pbase - a.W - 1) * a.ld_x * PP_ACT_BYTES;
    unsigned bad = m_dead;
    if (!live) bad = ~0u;
    if (n_ty == 0) bad |= m_top;
    if (n_ty == a.tiles_y - 1) bad |= m_bot;
    if (n_tx == 0) bad |= m_left;
    if (n_tx == a.tiles_x - 1) bad |= m_right;
    if (LAZY) { s_bad = bad; s_grp = n_img >= a.lazy.imgs_per_group ? 1 : 0; }
#pragma unroll
    for (int i = 0; i < DZ_PASS; ++i)
      rd[i] = act_buf_ld4_raw(rs_dz, live ? (unsigned)(dbase + dz_rel0 + i * dz_rel_step) : 0xffffffffu, 0);
#pragma unroll
    for (int i = 0; i < X_PASS; ++i)
      rx[i] = act_buf_ld4_raw(rs_x, ((bad >> i) & 1u) ? 0xffffffffu : (unsigned)(xbase + x_rel[i]), 0);
    t_next += G;
    n_tx += d_tx; if (n_tx >= a.tiles_x) { n_tx -= a.tiles_x; ++n_ty; }
    n_ty += d_ty; if (n_ty >= a.tiles_y) { n_ty -= a.tiles_y; ++n_img; }
    n_img += d_img;
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < DZ_PASS; ++i) {
      const f32x4 v = act_cvt4(rd[i]) * s_in;
      const f16x4 hi = __builtin_convertvector(v, f16x4);
      *reinterpret_cast<f16x4*>(Dh + dz_lds0 + i * DZ_STEP * WH_RS) = hi;
      if (PP_ACT_LO) {
        const f16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), f16x4);
        *reinterpret_cast<f16x4*>(Dl + dz_lds0 + i * DZ_STEP * WH_RS) = lo;
      }
    }
    f32x4 l_sc, l_sh, l_sl;
    if (LAZY) {
      const float* r = Lz + s_grp * 96 * CBK + xq * 4;
      l_sc = *reinterpret_cast<const f32x4*>(r);
      l_sh = *reinterpret_cast<const f32x4*>(r + 32 * CBK);
      l_sl = *reinterpret_cast<const f32x4*>(r + 64 * CBK);
    }
#pragma unroll
    for (int i = 0; i < X_PASS; ++i)
      if (!((m_dead >> i) & 1u)) {
        f32x4 v = act_cvt4(rx[i]);
        if (LAZY) {
          v = pp_lazy_apply4(v, l_sc, l_sh, l_sl);
          const float keep = ((s_bad >> i) & 1u) ? 0.f : 1.f;      // zero padding of y
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = keep != 0.f ? v[e] : 0.f;
        }
        const f16x4 hi = __builtin_convertvector(v, f16x4);
        *reinterpret_cast<f16x4*>(Xh + x_lds0 + i * X_STEP * WH_RS) = hi;
        if (PP_ACT_LO) {
          const f16x4 lo = __builtin_convertvector((v - __builtin_convertvector(hi, f32x4)) * F16_LO_SCALE, f16x4);
          *reinterpret_cast<f16x4*>(Xl + x_lds0 + i * X_STEP * WH_RS) = lo;
        }
      }
  };
  auto frag = [&](const _Float16* img, int pixel0) __attribute__((always_inline)) -> f16x8 {       // 16 pixels x 32 channels, reduction-major
    const int g = lane >> 4, i16 = lane & 15;
    const _Float16* p0 = img + (pixel0 + 8 * (g >> 1) + (i16 >> 2)) * WH_RS + 16 * (g & 1) + 4 * (i16 & 3);
    const h4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)p0);
    const h4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4_t*)(p0 + 4 * WH_RS));
    return __builtin_shufflevector(__builtin_bit_cast(f16x4, v0), __builtin_bit_cast(f16x4, v1), 0, 1, 2, 3, 4, 5, 6, 7);
  };
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const f16x8 two_m11 = {(_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f,
                         (_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f, (_Float16)4.8828125e-4f};
  const _Float16 *dh = Dh + po * D_IMG, *dl = Dl + po * D_IMG, *xh = Xh + pc * X_IMG, *xl = Xl + pc * X_IMG;
  const int my_tiles = wk < a.n_tiles ? (a.n_tiles - wk + G - 1) / G : 0;
  load_tile();
  for (int k = 0; k < my_tiles; ++k) {
    __syncthreads();                               // every wave is done with the previous tile's images
    store_tile();
    __syncthreads();
    load_tile();                                   // unconditional (masked past the end): keeps the s_waitcnt counting exact
#pragma unroll 1
    for (int uu = 0; uu < (pair_live ? UNITS : 0); ++uu) {           // this wave's 16-pixel units: (tile row, half) = (u >> 1, u & 1)
      const int u = split + uu * SPLITS, row = u >> 1, hc = u & 1;
      constexpr int NB = PAIRS == 4 ? 1 : 2;         // four pairs: no room for double-buffered x fragments (256 VGPRs)
      f16x8 bh[NB], bl[NB];
      const f16x8 ah = frag(dh, row * 32 + 16 * hc);
      f16x8 al = ah;
      if (PP_ACT_LO) al = frag(dl, row * 32 + 16 * hc);
      bh[0] = frag(xh, row * HT_HC + 16 * hc);
      if (PP_ACT_LO) bl[0] = frag(xl, row * HT_HC + 16 * hc);
      const f16x8 ahs = ah * two_m11;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int cur = NB == 2 ? (tap & 1) : 0;
        if (NB == 2 && tap + 1 < 9) {
          const int pix0n = (row + (tap + 1) / 3) * HT_HC + 16 * hc + (tap + 1) % 3;
          bh[(NB - 1) & (cur ^ 1)] = frag(xh, pix0n);
          if (PP_ACT_LO) bl[(NB - 1) & (cur ^ 1)] = frag(xl, pix0n);
        }
        if (NB == 1 && tap > 0) {
          const int pix0n = (row + tap / 3) * HT_HC + 16 * hc + tap % 3;
          bh[0] = frag(xh, pix0n);
          if (PP_ACT_LO) bl[0] = frag(xl, pix0n);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[cur], acc[tap], 0, 0, 0);
        if (PP_ACT_LO) {
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahs, bl[cur], acc[tap], 0, 0, 0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[cur], acc[tap], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // the SPLITS waves of a pair add their accumulators through LDS tap by tap in a fixed order (4 KB per wave and tap);
  // one partial per block: D[row = o = (r&3) + 8*(r>>2) + 4*lh][col = c = lr]
  const int lr = lane & 31, lh = lane >> 5;
  float* xch = reinterpret_cast<float*>(smem16) + (size_t)pair * (SPLITS - 1) * 1024;      // [SPLITS-1][16][64] floats per pair
  float* part = a.part + (size_t)wk * a.O * 9 * a.C;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    __syncthreads();                               // images / the previous tap's exchange are no longer read
    if (split > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[(split - 1) * 1024 + r * 64 + lane] = acc[tap][r];
    }
    __syncthreads();
    if (split == 0 && pair_live) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[tap][r];
#pragma unroll
        for (int s2 = 1; s2 < SPLITS; ++s2) v += xch[(s2 - 1) * 1024 + r * 64 + lane];
        const int o = o0 + po * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        part[((size_t)o * 9 + tap) * a.C + c0 + pc * 32 + lr] = v * s_out;
      }
    }
  }
}

static bool wgrad_h16_applicable(int O, int C, int H, int W, int dil);
static bool wgrad_lazy_ok(int O, int C, int H, int W, int dil) { return wgrad_h16_applicable(O, C, H, W, dil); }
static bool wgrad_h16_applicable(int O, int C, int H, int W, int dil) {
  return dil == 1 && O % 32 == 0 && C % 32 == 0 && O <= 256 && C <= 192 && W % HT_COLS == 0 && H % HT_ROWS == 0;
}
static int wgrad_h16_walkers(int per_walker, int cus = 0) {   // tile walkers for `per_walker` blocks each: one 8-wave block per CU,
  if (cus <= 0) cus = pp_wgrad_cus();                 // CUs the weight gradient may fill (pp_set_wgrad_cus; PP_WGRAD_CUS)
  int g = (cus / per_walker) / 8 * 8;                 // a multiple of 8 (wh_walker_pair); a walker without tiles writes zeros
  return g < 8 ? 8 : g;
}
static int wgrad_h16_blocks(int O, int C, int B, int H, int W, int cus = 0) {     // walkers of the one-pair kernel (each writes 4 partials)
  (void)B; (void)H; (void)W;
  return wgrad_h16_walkers((O / 32) * (C / 32), cus);
}

struct WgradPlan { int tile; int bk; int o_tiles, c_tiles, splits, chunks_per_split, n_chunks; };

static WgradPlan wgrad_plan(int O, int C, int P) {
  WgradPlan p;
  p.tile = (O % 128 == 0 && C % 128 == 0) ? 128 : ((O % 64 == 0 && C % 64 == 0) ? 64 : 32);
  p.bk = p.tile == 128 ? 32 : (p.tile == 64 ? 64 : 128);   // pixels per LDS stage: >= 16 MFMAs per wave per barrier
  p.o_tiles = pp_cdiv(O, p.tile);
  p.c_tiles = pp_cdiv(C, p.tile);
  p.n_chunks = pp_cdiv(P, p.bk);
  const int tiles = 9 * p.o_tiles * p.c_tiles;
  int splits = pp_cdiv(1536, tiles);                    // aim at ~6 blocks per CU over the launch
  const int max_splits = pp_cdiv(p.n_chunks, 512 / p.bk > 1 ? 512 / p.bk : 1);   // >= 512 pixels per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.chunks_per_split = pp_cdiv(p.n_chunks, splits);
  p.splits = pp_cdiv(p.n_chunks, p.chunks_per_split);
  return p;
}

#ifndef PP_ACT_16     // shape query, independent of the storage type: one copy
extern "C" size_t pp_conv3x3_bwd_weight_workspace(int O, int Cpad, int B, int H, int W) {
  WgradPlan p = wgrad_plan(O, Cpad, B * H * W);
  size_t need = (size_t)p.splits * O * 9 * Cpad * sizeof(float);
  if (wgrad9_applicable(O, Cpad, H, W, 1)) {           // the caller may run this layer with dilation 1
    Wgrad9Plan q = wgrad9_plan(O, Cpad, B * H * W);
    const size_t n9 = (size_t)q.splits * O * 9 * Cpad * sizeof(float);
    if (n9 > need) need = n9;
  }
  if (wgrad_c4_applicable(O, Cpad, W)) {
    const size_t n4 = (size_t)wgrad_c4_plan(B * H * W).blocks * O * 36 * sizeof(float);
    if (n4 > need) need = n4;
  }
  if (wgrad_h16_applicable(O, Cpad, H, W, 1)) {
    // sized for the LARGEST budget a launch may run under (the budget is a per-thread launch setting, the query must not depend
    // on what ran before: ADVICE r05); walkers grow with the budget
    const size_t n16 = (size_t)wgrad_h16_blocks(O, Cpad, B, H, W, PP_WGRAD_CUS_MAX) * 4 * O * 9 * Cpad * sizeof(float);
    if (n16 > need) need = n16;
  }
  return need;
}
#endif  // !PP_ACT_16

template <int TM, int TN, int WAVES_M, int WAVES_N, int WAVES_K, int BKP>
static int launch_wgrad(WgradArgs a, int splits, hipStream_t s) {
  constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
  size_t lds = (size_t)2 * BKP * (BM + 4 + BN + 4) * sizeof(float);
  const size_t red = (size_t)(WAVES_K - 1) * WAVES_M * WAVES_N * TM * TN * 16 * 64 * sizeof(float);
  if (red > lds) lds = red;
  auto kern = conv3x3_wgrad_kernel<TM, TN, WAVES_M, WAVES_N, WAVES_K, BKP>;
  {   // once per (kernel, device): pp_max_lds
    pp_max_lds(reinterpret_cast<const void*>(kern), (int)lds);
  }
  hipLaunchKernelGGL(kern, dim3(9 * a.o_tiles * a.c_tiles, splits), dim3(WAVES_M * WAVES_N * WAVES_K * 64), lds, s, a);
  return pp_launch_status("conv3x3_wgrad");
}

extern "C" int PP_FN(pp_conv3x3_bwd_weight)(const pp_act* dz, int ld_dz, int O, const pp_act* x, int ld_x, int Cpad,
                                     int I_true, int B, int H, int W, int dil, float* dw_oihw, int accumulate,
                                     float* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(dz && x && dw_oihw && workspace, "wgrad: null pointer");
  PP_CHECK_ARG(Cpad % 4 == 0 && ld_x % 4 == 0 && ld_dz % 4 == 0 && O % 4 == 0, "wgrad: O, Cpad, ld must be multiples of 4");
  PP_CHECK_ARG(I_true > 0 && I_true <= Cpad && ld_x >= Cpad && ld_dz >= O, "wgrad: bad channel counts");
  PP_CHECK_ARG(((uintptr_t)dz & PP_ACT_ALIGN) == 0 && ((uintptr_t)x & PP_ACT_ALIGN) == 0, "wgrad: dz/x must be 16-byte aligned");
  const int P = B * H * W;
  PP_CHECK_ARG((long long)P * ld_x < 0x3fffffffLL && (long long)P * ld_dz < 0x3fffffffLL,
               "wgrad: tensor exceeds the 4 GiB buffer-descriptor range");
  if (wgrad_c4_applicable(O, Cpad, W)) {
    const WgradC4Plan q = wgrad_c4_plan(P);
    const size_t need4 = (size_t)q.blocks * O * 36 * sizeof(float);
    if (workspace_bytes < need4) {
      pp_set_error("wgrad: workspace too small (%zu < %zu)", workspace_bytes, need4);
      return PP_ERR_WORKSPACE;
    }
    WgradC4Args a4{dz, ld_dz, O, x, ld_x, workspace, P, H, W, dil, q.px_per_wave,
                   (unsigned)(((long long)(P - 1) * ld_dz + O) * PP_ACT_BYTES), (unsigned)(((long long)(P - 1) * ld_x + Cpad) * PP_ACT_BYTES)};
    pp_prof_begin(PP_K_CONV_WGRAD, 2.0 * P * (double)O * 9.0 * Cpad, 4.0 * ((double)P * (O + Cpad) + 9.0 * O * Cpad), s);
    switch (O / 16) {
      case 1: hipLaunchKernelGGL(conv3x3_c4_wgrad_kernel<1>, dim3(q.blocks), dim3(256), 0, s, a4); break;
      case 2: hipLaunchKernelGGL(conv3x3_c4_wgrad_kernel<2>, dim3(q.blocks), dim3(256), 0, s, a4); break;
      case 3: hipLaunchKernelGGL(conv3x3_c4_wgrad_kernel<3>, dim3(q.blocks), dim3(256), 0, s, a4); break;
      default: hipLaunchKernelGGL(conv3x3_c4_wgrad_kernel<4>, dim3(q.blocks), dim3(256), 0, s, a4); break;
    }
    pp_prof_end(s);
    if (int rc = pp_launch_status("conv3x3_c4_wgrad")) return rc;
    hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(pp_cdiv((size_t)O * 9, 16)), dim3(256), 0, s, workspace, q.blocks, O, Cpad,
                       I_true, dw_oihw, accumulate);
    return pp_launch_status("wgrad_finalize");
  }
#ifndef PP_ACT_16
  if (wgrad9_applicable(O, Cpad, H, W, dil)) {
    Wgrad9Plan q = wgrad9_plan(O, Cpad, P);
    const size_t need9 = (size_t)q.splits * O * 9 * Cpad * sizeof(float);
    if (workspace_bytes < need9) {
      pp_set_error("wgrad: workspace too small (%zu < %zu)", workspace_bytes, need9);
      return PP_ERR_WORKSPACE;
    }
    Wgrad9Args a9{dz, ld_dz, O, x, ld_x, Cpad, workspace, P, H, W, q.o_tiles, q.c_tiles, q.segs_per_split, q.n_segs,
                  (unsigned)(((long long)(P - 1) * ld_dz + O) * 4), (unsigned)(((long long)(P - 1) * ld_x + Cpad) * 4)};
    const size_t lds = (size_t)2 * (W9_SEG * W9_LD + 3 * (W9_SEG + 2) * W9_LD) * sizeof(float);
    {   // once per (kernel, device): pp_max_lds
      pp_max_lds(reinterpret_cast<const void*>(conv3x3_wgrad9_kernel), (int)lds);
    }
    pp_prof_begin(PP_K_CONV_WGRAD, 2.0 * P * (double)O * 9.0 * Cpad, 4.0 * ((double)P * (O + Cpad) + 9.0 * O * Cpad), s);
    hipLaunchKernelGGL(conv3x3_wgrad9_kernel, dim3(q.o_tiles * q.c_tiles, q.splits), dim3(256), lds, s, a9);
    pp_prof_end(s);
    if (int rc = pp_launch_status("conv3x3_wgrad9")) return rc;
    const size_t per9 = (size_t)O * 9 * Cpad;
    hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(pp_cdiv(per9 / 4, 16)), dim3(256), 0, s, workspace, q.splits, O, Cpad,
                       I_true, dw_oihw, accumulate);
    return pp_launch_status("wgrad_finalize");
  }
#endif
  WgradPlan p = wgrad_plan(O, Cpad, P);
  const size_t need = (size_t)p.splits * O * 9 * Cpad * sizeof(float);
  if (workspace_bytes < need) {
    pp_set_error("wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    return PP_ERR_WORKSPACE;
  }
  WgradArgs a{dz, ld_dz, O, x, ld_x, Cpad, workspace, P, H, W, dil, p.o_tiles, p.c_tiles, p.chunks_per_split, p.n_chunks,
              (unsigned)(((long long)(P - 1) * ld_dz + O) * PP_ACT_BYTES), (unsigned)(((long long)(P - 1) * ld_x + Cpad) * PP_ACT_BYTES)};
  pp_prof_begin(PP_K_CONV_WGRAD, 2.0 * P * (double)O * 9.0 * Cpad, 4.0 * ((double)P * (O + Cpad) + 9.0 * O * Cpad), s);
  int rc;
  if (p.tile == 128)
    rc = launch_wgrad<2, 2, 2, 2, 1, 32>(a, p.splits, s);
  else if (p.tile == 64)
    rc = launch_wgrad<1, 1, 2, 2, 1, 64>(a, p.splits, s);
  else
    rc = launch_wgrad<1, 1, 1, 1, 4, 128>(a, p.splits, s);
  pp_prof_end(s);
  if (rc) return rc;
  const size_t per = (size_t)O * 9 * Cpad;
  hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(pp_cdiv(per / 4, 16)), dim3(256), 0, s, workspace, p.splits, O, Cpad,
                     I_true, dw_oihw, accumulate);
  return pp_launch_status("wgrad_finalize");
}

// split-fp16 form of pp_conv3x3_bwd_weight for the narrow layers (see conv3x3_wgrad_halo_f16x3_kernel); falls back
// to the fp32 kernels when the shape does not qualify.  dz_amax: device float, max |dz| (pp_bn_lrelu_bwd_amax).
static int bwd_weight_f16x3_impl(const pp_act* dz, int ld_dz, int O, const pp_act* x, int ld_x, int Cpad,
                                 int I_true, int B, int H, int W, int dil, float* dw_oihw, int accumulate,
                                 float* workspace, size_t workspace_bytes, const float* dz_amax, PpLazy lazy, void* stream) {
  if (!wgrad_h16_applicable(O, Cpad, H, W, dil) || !dz_amax) {
    if (lazy.coef) {
      pp_set_error("wgrad_f16x3: a lazy x needs the halo-tile kernels (pp_conv3x3_lazy_ok tells) and dz_amax");
      return PP_ERR_UNSUPPORTED;
    }
    return PP_FN(pp_conv3x3_bwd_weight)(dz, ld_dz, O, x, ld_x, Cpad, I_true, B, H, W, dil, dw_oihw, accumulate, workspace,
                                 workspace_bytes, stream);
  }
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_ARG(dz && x && dw_oihw && workspace, "wgrad_f16x3: null pointer");
  PP_CHECK_ARG(ld_x % 4 == 0 && ld_dz % 4 == 0 && I_true > 0 && I_true <= Cpad && ld_x >= Cpad && ld_dz >= O, "wgrad_f16x3: bad ld / channels");
  PP_CHECK_ARG(((uintptr_t)dz & PP_ACT_ALIGN) == 0 && ((uintptr_t)x & PP_ACT_ALIGN) == 0, "wgrad_f16x3: dz/x must be 16-byte aligned");
  const int P = B * H * W;
  PP_CHECK_ARG((long long)P * ld_x < 0x3fffffffLL && (long long)P * ld_dz < 0x3fffffffLL,
               "wgrad_f16x3: tensor exceeds the 4 GiB buffer-descriptor range");
  const int gx = wgrad_h16_blocks(O, Cpad, B, H, W);
  const size_t need = (size_t)gx * 4 * O * 9 * Cpad * sizeof(float);
  if (workspace_bytes < need) {
    pp_set_error("wgrad_f16x3: workspace too small (%zu < %zu)", workspace_bytes, need);
    return PP_ERR_WORKSPACE;
  }
  WgradH16Args a{dz, ld_dz, O, x, ld_x, Cpad, workspace, P, H, W, Cpad / 32, W / HT_COLS, H / HT_ROWS,
                 B * (H / HT_ROWS) * (W / HT_COLS),
                 (unsigned)(((long long)(P - 1) * ld_dz + O) * PP_ACT_BYTES), (unsigned)(((long long)(P - 1) * ld_x + Cpad) * PP_ACT_BYTES), 0, 0};
  a.lazy = lazy;
  const bool lz = lazy.coef != nullptr;
  const size_t lds = (size_t)2 * (WH_DZ_PIX + HT_PIX) * WH_RS * sizeof(_Float16) + (lz ? 2 * 96 * sizeof(float) : 0);
  pp_prof_begin2(PP_K_CONV_WGRAD_F16X3, 6.0 * P * (double)O * 9.0 * Cpad, 2.0 * P * (double)O * 9.0 * Cpad,
                 4.0 * ((double)P * (O + Cpad) + 9.0 * O * Cpad), s);
  constexpr int mp = 1;
  // two pairs per block (four would need 11 prefetched float4 per thread next to 144 accumulator registers: spills).
  // Sharing the x patch (204 pixels) between two output blocks saves more staging than sharing the dz tile (128).
  // (96 / 160 input channels, round 5: two-pair blocks with a half-empty last group -- 32 outputs x 96 inputs at 256^2 ran as three
  // one-pair blocks per tile, each staging the dz tile again for 27 MFMAs per wave)
  constexpr int odd = 1;
  const int obk = (mp && O % 64 == 0) ? 2 : 1, cbk = (mp && obk == 1 && (Cpad % 64 == 0 || (odd && Cpad > 32))) ? 2 : 1;
  int slabs = gx * 4;
  if (obk * cbk > 1) {                             // several (32 x 32) pairs per block: the tile is staged once for all of them
    const int groups = (O / (32 * obk)) * ((Cpad / 32 + cbk - 1) / cbk);
    const int gmp = wgrad_h16_walkers(groups);
    slabs = gmp;                                   // <= gx * 4: the workspace bound above covers it
    a.walkers = gmp; a.per_walker = groups;
    const size_t lmp = (size_t)2 * (obk * WH_DZ_PIX + cbk * HT_PIX) * WH_RS * sizeof(_Float16) + (lz ? (size_t)2 * 96 * cbk * sizeof(float) : 0);
#define WGMP_LAUNCH(OB, CB, LZ)                                                                                              \
    do {                                                                                                                       \
      pp_max_lds(reinterpret_cast<const void*>(conv3x3_wgrad_halo_mp_f16x3_kernel<OB, CB, LZ>), (int)lmp);                    \
      hipLaunchKernelGGL((conv3x3_wgrad_halo_mp_f16x3_kernel<OB, CB, LZ>), dim3(gmp * groups), dim3(WH_THREADS), lmp, s, a, dz_amax); \
    } while (0)
    if (obk == 2) { if (lz) WGMP_LAUNCH(2, 1, true); else WGMP_LAUNCH(2, 1, false); }
    else { if (lz) WGMP_LAUNCH(1, 2, true); else WGMP_LAUNCH(1, 2, false); }
#undef WGMP_LAUNCH
  } else {
    a.walkers = gx; a.per_walker = (O / 32) * (Cpad / 32);
    if (lz) hipLaunchKernelGGL(conv3x3_wgrad_halo_f16x3_kernel<true>, dim3(gx * a.per_walker), dim3(WH_THREADS), lds, s, a, dz_amax);
    else hipLaunchKernelGGL(conv3x3_wgrad_halo_f16x3_kernel<false>, dim3(gx * a.per_walker), dim3(WH_THREADS), lds, s, a, dz_amax);
  }
  pp_prof_end(s);
  if (int rc = pp_launch_status("conv3x3_wgrad_halo_f16x3")) return rc;
  hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(pp_cdiv((size_t)O * 9 * Cpad / 4, 16)), dim3(256), 0, s, workspace, slabs, O, Cpad,
                     I_true, dw_oihw, accumulate);
  return pp_launch_status("wgrad_finalize");
}

extern "C" int PP_FN(pp_conv3x3_bwd_weight_f16x3)(const pp_act* dz, int ld_dz, int O, const pp_act* x, int ld_x, int Cpad,
                                           int I_true, int B, int H, int W, int dil, float* dw_oihw, int accumulate,
                                           float* workspace, size_t workspace_bytes, const float* dz_amax, void* stream) {
  return bwd_weight_f16x3_impl(dz, ld_dz, O, x, ld_x, Cpad, I_true, B, H, W, dil, dw_oihw, accumulate, workspace, workspace_bytes,
                               dz_amax, pp_lazy_none(), stream);
}
// the same with a LAZY x (pp_lazy_in): x holds the raw convolution output of the layer in front; shapes: pp_conv3x3_lazy_ok
extern "C" int PP_FN(pp_conv3x3_bwd_weight_f16x3_lazy)(const pp_act* dz, int ld_dz, int O, const pp_act* x, int ld_x, int Cpad,
                                                int I_true, int B, int H, int W, int dil, float* dw_oihw, int accumulate,
                                                float* workspace, size_t workspace_bytes, const float* dz_amax,
                                                const pp_lazy_in* lazy_x, void* stream) {
  PpLazy lz = pp_lazy_none();
  if (lazy_x && lazy_x->coef) {
    PP_CHECK_ARG(lazy_x->groups >= 1 && lazy_x->groups <= PP_EPI_GROUPS && B % lazy_x->groups == 0,
                 "wgrad_f16x3_lazy: 1 or 2 lazy groups that divide the batch");
    PP_CHECK_ARG(lazy_x->ld % 4 == 0 && lazy_x->ld >= Cpad && ((uintptr_t)lazy_x->coef & 15) == 0, "wgrad_f16x3_lazy: bad coefficient rows");
    lz = PpLazy{lazy_x->coef, lazy_x->ld, B / lazy_x->groups};
  }
  return bwd_weight_f16x3_impl(dz, ld_dz, O, x, ld_x, Cpad, I_true, B, H, W, dil, dw_oihw, accumulate, workspace, workspace_bytes,
                               dz_amax, lz, stream);
}

#ifndef PP_ACT_16       // weight packing and the MFMA probe do not touch activations: one copy, in the fp32 build
// ------------------------------------------------------------------------------------------
// f16x3 weight packing: the same two layouts as pack_weights_kernel, every group of 4 consecutive K elements stored
// as 16 bytes [hi0..hi3 | lo0..lo3] (fp16) -- same size and indexing as the fp32 tensors, split done once per step.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_weights_f16x3_body(const float* w, int O, int I, int Ipad, _Float16* wf, _Float16* wb, size_t idx) {
  const size_t total = (size_t)O * 9 * Ipad;
  if (idx >= total) return;
  const int c = (int)(idx % Ipad);
  const int tap = (int)((idx / Ipad) % 9);
  const int o = (int)(idx / ((size_t)Ipad * 9));
  const float v = c < I ? w[((size_t)o * I + c) * 9 + tap] : 0.f;
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)((v - (float)hi) * F16_LO_SCALE);
  if (wf) {
    _Float16* d = wf + (((size_t)o * 9 + tap) * Ipad + (c & ~3)) * 2;      // 8 halves per channel quad
    d[c & 3] = hi;
    d[4 + (c & 3)] = lo;
  }
  if (wb && c < I) {
    _Float16* d = wb + (((size_t)c * 9 + (8 - tap)) * O + (o & ~3)) * 2;
    d[o & 3] = hi;
    d[4 + (o & 3)] = lo;
  }
}

__global__ void pack_weights_f16x3_kernel(const float* w, int O, int I, int Ipad, _Float16* wf, _Float16* wb) {
  pack_weights_f16x3_body(w, O, I, Ipad, wf, wb, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Every layer's pack in ONE launch (round 5): the per-layer launches are 5 - 6 us each behind one another at the start of every
// step for a few microseconds of work.  Block -> layer through the first-block table (uniform scan over <= 24 entries).
#define PACK_BATCH_MAX 24
struct PackItem { const float* w; _Float16* wf; _Float16* wb; int O, I, Ipad, blk0; };
struct PackBatch { PackItem it[PACK_BATCH_MAX]; int n; };
__global__ void pack_weights_f16x3_batch_kernel(PackBatch b) {
  int k = 0;
  for (int i = 1; i < b.n; ++i)
    if ((int)blockIdx.x >= b.it[i].blk0) k = i;
  const PackItem it = b.it[k];
  pack_weights_f16x3_body(it.w, it.O, it.I, it.Ipad, it.wf, it.wb, (size_t)(blockIdx.x - it.blk0) * blockDim.x + threadIdx.x);
}

extern "C" int pp_pack_conv3x3_weights_f16x3(const float* w_oihw, int O, int I, int Ipad, void* wf16, void* wb16,
                                             void* stream) {
  PP_CHECK_ARG(w_oihw && (wf16 || wb16) && Ipad >= I && Ipad % 4 == 0, "pack_weights_f16x3: bad arguments");
  PP_CHECK_ARG(!wb16 || O % 4 == 0, "pack_weights_f16x3: the dgrad layout needs O %% 4 == 0");
  const size_t total = (size_t)O * 9 * Ipad;
  hipLaunchKernelGGL(pack_weights_f16x3_kernel, dim3(pp_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, O, I,
                     Ipad, (_Float16*)wf16, (_Float16*)wb16);
  return pp_launch_status("pack_weights_f16x3");
}

extern "C" int pp_pack_conv3x3_weights_f16x3_batch(const pp_pack_item* items, int n, void* stream) {
  PP_CHECK_ARG(items && n >= 1, "pack_weights_f16x3_batch: no items");
  for (int i0 = 0; i0 < n; i0 += PACK_BATCH_MAX) {
    PackBatch b;
    b.n = n - i0 < PACK_BATCH_MAX ? n - i0 : PACK_BATCH_MAX;
    int blk = 0;
    for (int i = 0; i < b.n; ++i) {
      const pp_pack_item& q = items[i0 + i];
      PP_CHECK_ARG(q.w_oihw && (q.wf16 || q.wb16) && q.Ipad >= q.I && q.Ipad % 4 == 0, "pack_weights_f16x3_batch: bad item");
      PP_CHECK_ARG(!q.wb16 || q.O % 4 == 0, "pack_weights_f16x3_batch: the dgrad layout needs O %% 4 == 0");
      b.it[i] = PackItem{q.w_oihw, (_Float16*)q.wf16, (_Float16*)q.wb16, q.O, q.I, q.Ipad, blk};
      blk += (int)pp_cdiv((size_t)q.O * 9 * q.Ipad, 256);
    }
    for (int i = b.n; i < PACK_BATCH_MAX; ++i) b.it[i] = PackItem{nullptr, nullptr, nullptr, 0, 0, 4, 0x7fffffff};
    hipLaunchKernelGGL(pack_weights_f16x3_batch_kernel, dim3(blk), dim3(256), 0, (hipStream_t)stream, b);
    if (int rc = pp_launch_status("pack_weights_f16x3_batch")) return rc;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------
// weight packing: OIHW -> Wf[O][9][Ipad] (forward) and Wb[I][9][O] with flipped taps (dgrad)
// ------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* w, int O, int I, int Ipad, float* wf, float* wb) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)O * 9 * Ipad;
  if (idx >= total) return;
  const int c = (int)(idx % Ipad);
  const int tap = (int)((idx / Ipad) % 9);
  const int o = (int)(idx / ((size_t)Ipad * 9));
  const float v = c < I ? w[((size_t)o * I + c) * 9 + tap] : 0.f;
  wf[idx] = v;
  if (wb && c < I) wb[((size_t)c * 9 + (8 - tap)) * O + o] = v;
}

extern "C" int pp_pack_conv3x3_weights(const float* w_oihw, int O, int I, int Ipad, float* wf, float* wb,
                                       void* stream) {
  PP_CHECK_ARG(w_oihw && wf, "pack_weights: null pointer");
  PP_CHECK_ARG(Ipad >= I && Ipad % 4 == 0, "pack_weights: Ipad must be a multiple of 4 and >= I");
  const size_t total = (size_t)O * 9 * Ipad;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(pp_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, O, I,
                     Ipad, wf, wb);
  return pp_launch_status("pack_weights");
}

// ------------------------------------------------------------------------------------------
// diagnostic: what the chip sustains on bare v_mfma_f32_32x32x2_f32 (4 independent accumulators per wave, operands in
// registers) -- the ceiling the convolution kernels are measured against on THIS device at ITS clock under load.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* out, int iters, float seed) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
    }
    a = -a;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Runs `blocks` x 4 waves x iters x 16 MFMAs; returns the flop count so the caller can divide by its event time.
extern "C" int pp_mfma_probe(float* out, int blocks, int iters, double* flops, void* stream) {
  PP_CHECK_ARG(out && blocks > 0 && iters > 0, "mfma_probe: bad arguments");
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 0.37f);
  if (flops) *flops = (double)blocks * 4.0 * iters * 16.0 * (2.0 * 32 * 32 * 2);
  return pp_launch_status("mfma_probe");
}

#ifdef PP_HALO_TRACE
// study builds only: phase cycles of the last conv3x3_halo_f16x3_kernel<1> launch (see HT_TRK)
extern "C" int pp_debug_halo_trace(long long* out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return -4;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pp_halo_trace), sizeof(long long) * (n < 16 ? n : 16)) == hipSuccess ? 0 : -4;
}
#endif
#endif  // !PP_ACT_16

PP_NS_END
