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

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define WLD 36          // padded LDS row (floats), see pp_conv.hip
#define WBK 32

struct WinoGeom {
  int N, H, W, dil;     // images, image size, dilation
  int Hs, Ws;           // sub-image size  (H/dil, W/dil)
  int th, tw;           // tiles per sub-image (Hs/2, Ws/2)
  int T;                // total tiles = N * dil*dil * th * tw
};
static inline WinoGeom wino_geom(int N, int H, int W, int dil) {
  WinoGeom g{N, H, W, dil, H / dil, W / dil, H / dil / 2, W / dil / 2, 0};
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
  const int total = per_batch * 16;
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
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(16 * a.m_tiles * a.n_tiles), dim3(WAVES_M * WAVES_N * 64), lds, s, a);
  return pp_launch_status("wino_gemm");
}

static inline int wino_blocks(long long total) {
  int b = pp_cdiv(total, 256);
  return b > 16384 ? 16384 : (b < 1 ? 1 : b);
}

static int wino_check(int C, int N, int B, int H, int W, int dil) {
  PP_CHECK_ARG(dil >= 1 && H % (2 * dil) == 0 && W % (2 * dil) == 0, "winograd: H, W must be multiples of 2*dilation");
  PP_CHECK_ARG(C % 4 == 0 && N % 4 == 0 && C > 0 && N > 0 && B > 0, "winograd: channel counts must be multiples of 4");
  const long long T = (long long)B * H * W / 4;
  PP_CHECK_ARG(T * C < 0x3fffffffLL && T * N < 0x3fffffffLL && (long long)N * C < 0x3fffffffLL,
               "winograd: plane exceeds the 4 GiB buffer-descriptor range");
  return 0;
}

extern "C" size_t pp_conv3x3_wino_workspace(int Cin, int Cout, int B, int H, int W) {
  const size_t T = (size_t)B * H * W / 4;
  const size_t fwd = 16 * T * ((size_t)Cin + Cout) * sizeof(float);             // V + M (fwd / dgrad)
  return fwd + 256;
}

extern "C" int pp_wino_pack_weights(const float* w_oihw, int O, int I, float* Uf, float* Ub, void* stream) {
  PP_CHECK_ARG(w_oihw && (Uf || Ub), "wino_pack_weights: null pointer");
  hipLaunchKernelGGL(wino_weight_kernel, dim3(pp_cdiv((long long)O * I, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     O, I, Uf, Ub);
  return pp_launch_status("wino_pack_weights");
}

// forward and data gradient share this driver (U = Uf [16][N][C] resp. Ub [16][I][O])
static int wino_conv(const float* in, int ld_in, int C, const float* U, const float* bias, float* out, int ld_out, int N,
                     int B, int H, int W, int dil, int accumulate, float* v_keep, void* ws, size_t ws_bytes,
                     hipStream_t s) {
  if (int rc = wino_check(C, N, B, H, W, dil)) return rc;
  PP_CHECK_ARG(in && U && out && ws, "winograd conv: null pointer");
  PP_CHECK_ARG(ld_in % 4 == 0 && ld_out % 4 == 0 && ld_in >= C && ld_out >= N, "winograd conv: bad ld");
  WinoGeom g = wino_geom(B, H, W, dil);
  const size_t need = 16 * (size_t)g.T * ((v_keep ? 0 : (size_t)C) + N) * sizeof(float);
  if (ws_bytes < need) {
    pp_set_error("winograd conv: workspace too small (%zu < %zu)", ws_bytes, need);
    return PP_ERR_WORKSPACE;
  }
  // the transformed input either stays in the caller's buffer (kept for the weight gradient) or lives in the workspace
  float* V = v_keep ? v_keep : reinterpret_cast<float*>(ws);
  float* M = v_keep ? reinterpret_cast<float*>(ws) : V + 16 * (size_t)g.T * C;
  const double P = (double)B * H * W;
  pp_prof_begin(PP_K_WINO_XFORM, 0.0, 4.0 * P * C * 5.0, s);
  hipLaunchKernelGGL(wino_input_kernel, dim3(wino_blocks((long long)g.T * (C / 4))), dim3(256), 0, s, in, ld_in, C, g, V);
  pp_prof_end(s);
  if (int rc = pp_launch_status("wino_input")) return rc;
  GemmArgs ga{V, U, M, g.T, N, C, 0, 0, (unsigned)((size_t)g.T * C * 4), (unsigned)((size_t)N * C * 4)};
  // flops booked = EXECUTED transform-domain flops (16 GEMMs over P/4 tiles = 8 per pixel*cin*cout); the direct
  // convolution's algorithmic count is 18, i.e. 2.25x this
  pp_prof_begin(PP_K_WINO_GEMM, 8.0 * P * (double)N * C, 4.0 * (P * C + P * N + 9.0 * C * N), s);
  int rc = (N % 128 == 0) ? launch_gemm<2, 2, 2, 2>(ga, s) : launch_gemm<2, 1, 2, 2>(ga, s);
  pp_prof_end(s);
  if (rc) return rc;
  pp_prof_begin(PP_K_WINO_XFORM, 0.0, 4.0 * P * N * 5.0, s);
  hipLaunchKernelGGL(wino_output_kernel, dim3(wino_blocks((long long)g.T * (N / 4))), dim3(256), 0, s, M, N, g, bias, out,
                     ld_out, accumulate);
  pp_prof_end(s);
  return pp_launch_status("wino_output");
}

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
};

__global__ __launch_bounds__(256) void wino_wgrad_gemm_kernel(WinoWgArgs a) {
  constexpr int KB = 32, BM = 128, BN = 128, LDA = BM + 4, LDB = BN + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][KB][LDA]
  float* Bs = smem + 2 * KB * LDA;  // [2][KB][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int lr = lane & 31, lh = lane >> 5;
  // item order (split, o_tile, c_tile, batch): one XCD gets contiguous items
  const int per_split = 16 * a.c_tiles * a.o_tiles;
  const int total = per_split * gridDim.y;
  int L = blockIdx.y * gridDim.x + blockIdx.x;
  if ((total & 7) == 0) L = (L & 7) * (total >> 3) + (L >> 3);
  const int split = L / per_split;
  const int r = L - split * per_split;
  const int batch = r % 16;
  const int ct = (r / 16) % a.c_tiles, ot = r / (16 * a.c_tiles);
  const int o0 = ot * BM, c0 = ct * BN;
  const int chunk_lo = split * a.chunks_per_split;
  int chunk_hi = chunk_lo + a.chunks_per_split;
  if (chunk_hi > a.n_chunks) chunk_hi = a.n_chunks;
  const float* Wb = a.Wt + (size_t)batch * a.T * a.O;
  const float* Vb = a.V + (size_t)batch * a.T * a.C;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, a.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, a.v_bytes, 0x00020000);
  const int cq = tid & 31, row0 = tid >> 5;            // 32 float4 per 128-channel row, 8 rows per pass
  const int oa_ok = (int)(o0 + cq * 4 < a.O), cb_ok = (int)(c0 + cq * 4 < a.C);
  f32x4 ra[4], rb[4];
  auto load_tile = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = chunk * KB + row0 + i * 8;
      const int tok = (int)(t < a.T);
      const unsigned offa = (oa_ok & tok) ? (unsigned)(t * a.O + o0 + cq * 4) * 4u : 0xffffffffu;
      const unsigned offb = (cb_ok & tok) ? (unsigned)(t * a.C + c0 + cq * 4) * 4u : 0xffffffffu;
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, offa, 0, 0));
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, offb, 0, 0));
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(As + buf * KB * LDA + (row0 + i * 8) * LDA + cq * 4) = ra[i];
      *reinterpret_cast<f32x4*>(Bs + buf * KB * LDB + (row0 + i * 8) * LDB + cq * 4) = rb[i];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
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
    const float* Ap = As + buf * KB * LDA + wm * 64 + lr;
    const float* Bp = Bs + buf * KB * LDB + wn * 64 + lr;
#pragma unroll
    for (int kk = 0; kk < KB / 2; ++kk) {
      const int krow = 2 * kk + lh;
      const float a0 = Ap[krow * LDA], a1 = Ap[krow * LDA + 32];
      const float b0 = Bp[krow * LDB], b1 = Bp[krow * LDB + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }
  float* part = a.part + ((size_t)split * 16 + batch) * a.O * a.C;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = c0 + wn * 64 + j * 32 + lr;
    if (c >= a.C) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int o = o0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
        if (o < a.O) part[(size_t)o * a.C + c] = acc[i][j][q];
      }
  }
}

// dw[o][c][3][3] (+)= G^T (sum_splits dU) G      16 (o,c) pairs x 16 split-lanes per block
__global__ __launch_bounds__(256) void wino_wgrad_finalize_kernel(const float* __restrict__ part, int splits, int O, int C,
                                                                  float* __restrict__ dw, int accumulate) {
  __shared__ float red[16][16][17];
  const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const size_t oc = (size_t)blockIdx.x * 16 + il;
  const size_t per = (size_t)O * C;
  float u[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) u[b] = 0.f;
  if (oc < per)
    for (int k = sl; k < splits; k += 16)
#pragma unroll
      for (int b = 0; b < 16; ++b) u[b] += part[((size_t)k * 16 + b) * per + oc];
#pragma unroll
  for (int b = 0; b < 16; ++b) red[b][sl][il] = u[b];
  __syncthreads();
  if (sl != 0 || oc >= per) return;
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += red[b][i][il];
    u[b] = t;
  }
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

struct WinoWgPlan { int o_tiles, c_tiles, n_chunks, splits, chunks_per_split; };
static WinoWgPlan wino_wg_plan(int O, int C, int T) {
  WinoWgPlan p;
  p.o_tiles = pp_cdiv(O, 128);
  p.c_tiles = pp_cdiv(C, 128);
  p.n_chunks = pp_cdiv(T, 32);
  int splits = pp_cdiv(1536, 16 * p.o_tiles * p.c_tiles);
  const int max_splits = pp_cdiv(p.n_chunks, 16);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.chunks_per_split = pp_cdiv(p.n_chunks, splits);
  p.splits = pp_cdiv(p.n_chunks, p.chunks_per_split);
  return p;
}

extern "C" size_t pp_conv3x3_wino_bwd_weight_workspace(int O, int C, int B, int H, int W) {
  const size_t T = (size_t)B * H * W / 4;
  WinoWgPlan p = wino_wg_plan(O, C, (int)T);
  return (16 * T * ((size_t)O + C) + (size_t)p.splits * 16 * O * C) * sizeof(float) + 256;
}

extern "C" int pp_conv3x3_wino_bwd_weight(const float* dz, int ld_dz, int O, const float* x, int ld_x, int C, int B,
                                          int H, int W, int dil, float* dw_oihw, int accumulate, const float* v_cached,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = wino_check(C, O, B, H, W, dil)) return rc;
  PP_CHECK_ARG(dz && (x || v_cached) && dw_oihw && workspace, "winograd wgrad: null pointer");
  PP_CHECK_ARG(ld_dz % 4 == 0 && ld_x % 4 == 0 && ld_dz >= O && ld_x >= C, "winograd wgrad: bad ld");
  WinoGeom g = wino_geom(B, H, W, dil);
  WinoWgPlan p = wino_wg_plan(O, C, g.T);
  const size_t need = (16 * (size_t)g.T * ((size_t)O + (v_cached ? 0 : C)) + (size_t)p.splits * 16 * O * C) * sizeof(float);
  if (workspace_bytes < need) {
    pp_set_error("winograd wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    return PP_ERR_WORKSPACE;
  }
  float* Wt = reinterpret_cast<float*>(workspace);
  float* part = Wt + 16 * (size_t)g.T * O;
  float* Vown = part + (size_t)p.splits * 16 * O * C;
  const float* V = v_cached ? v_cached : Vown;
  const double P = (double)B * H * W;
  pp_prof_begin(PP_K_WINO_XFORM, 0.0, 4.0 * P * ((v_cached ? 0 : C) + O) * 5.0, s);
  if (!v_cached)
    hipLaunchKernelGGL(wino_input_kernel, dim3(wino_blocks((long long)g.T * (C / 4))), dim3(256), 0, s, x, ld_x, C, g, Vown);
  hipLaunchKernelGGL(wino_dy_kernel, dim3(wino_blocks((long long)g.T * (O / 4))), dim3(256), 0, s, dz, ld_dz, O, g, Wt);
  pp_prof_end(s);
  if (int rc = pp_launch_status("wino_wgrad_transforms")) return rc;
  WinoWgArgs a{Wt, V, part, g.T, O, C, p.o_tiles, p.c_tiles, p.chunks_per_split, p.n_chunks,
               (unsigned)((size_t)g.T * O * 4), (unsigned)((size_t)g.T * C * 4)};
  const size_t lds = (size_t)2 * 32 * (132 + 132) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_gemm_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  pp_prof_begin(PP_K_WINO_WGRAD, 8.0 * P * (double)O * C, 4.0 * (P * (O + C) + 9.0 * O * C), s);
  hipLaunchKernelGGL(wino_wgrad_gemm_kernel, dim3(16 * p.o_tiles * p.c_tiles, p.splits), dim3(256), lds, s, a);
  hipLaunchKernelGGL(wino_wgrad_finalize_kernel, dim3(pp_cdiv((long long)O * C, 16)), dim3(256), 0, s, part, p.splits, O, C,
                     dw_oihw, accumulate);
  pp_prof_end(s);
  return pp_launch_status("wino_wgrad");
}
