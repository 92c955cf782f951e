// STUDY (round 6): what does a DIRECT 3x3 convolution reach when its activation operand is stored once as
// [hi | lo] fp16 pairs and staged by LDS-DMA?  Not part of the product library (libpp_study.so, `make study`); driven by
// tests/studies/pair_layout_study.py, which times it against the shipped two-half halo kernel on the same layers.
//
// Reference op: nn.Conv2d(C, N, 3, padding=1) of models/unet.py:188 (cross-correlation, bias), fp32-grade through split-fp16
// products  a*b ~ ah*bh + 2^-11 (ah*bl + al*bh)  exactly as the product kernels (pp_conv.hip).
//
// PAIR LAYOUT of an NHWC tensor with C % 32 == 0: a pixel's 32-channel chunk q is 128 bytes [32 hi x fp16 | 32 lo x fp16]
// (hi = rne16(x), lo = rne16((x - hi) * 2^11)); pixel stride 4 C bytes -- the bytes of the fp32 tensor.  A 16-byte piece is one
// MFMA fragment (8 consecutive channels of one part).
//
// KERNEL: 8 waves, one 8 x 32 output tile per stage and wave = one output row x 32 output channels; the weights of the block's 32
// output channels stay in LDS (36 KB per 32-channel chunk); the 10 x 34 halo patch of a (tile, chunk) stage is 340 rows of 128 B,
// brought by buffer_load ... lds (1 KB per instruction, 43 per stage, 5-6 per wave) into a TWO-slot ring, XOR-swizzled on the
// source side so that the tap-shifted ds_read_b128 fragment reads are conflict-free.  No activation value passes through a VGPR
// on its way to LDS, there is no conversion, and every wave multiplies all the time: per stage ONE wait + ONE barrier for 54
// MFMA steps per wave (the Winograd GEMM has one per 24).  Finished tiles are stored one stage late.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LO_SCALE 2048.f

static thread_local char g_study_err[256] = "";
extern "C" const char* pp_study_last_error(void) { return g_study_err; }
#define STUDY_CHECK(cond, msg) do { if (!(cond)) { snprintf(g_study_err, sizeof(g_study_err), "%s", msg); return -1; } } while (0)

// ---------------------------------------------------------------- fp32 NHWC -> pair layout (what a producer would write itself)
__global__ __launch_bounds__(256) void split_pairs_kernel(const float* __restrict__ x, int C, long long P, char* __restrict__ out) {
  const int oct = C >> 3;                                   // 8-channel octets per pixel
  const long long total = P * oct;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / oct;
    const int o = (int)(i - p * oct), q = o >> 2, j = o & 3;          // chunk q, octet j of the chunk
    const float* src = x + p * C + o * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    const f16x4 ha = __builtin_convertvector(a, f16x4), hb = __builtin_convertvector(b, f16x4);
    const f16x4 la = __builtin_convertvector((a - __builtin_convertvector(ha, f32x4)) * LO_SCALE, f16x4);
    const f16x4 lb = __builtin_convertvector((b - __builtin_convertvector(hb, f32x4)) * LO_SCALE, f16x4);
    char* d = out + p * (size_t)C * 4 + (size_t)q * 128 + j * 16;
    *reinterpret_cast<u32x4*>(d) = u32x4{__builtin_bit_cast(u32x2, ha)[0], __builtin_bit_cast(u32x2, ha)[1],
                                         __builtin_bit_cast(u32x2, hb)[0], __builtin_bit_cast(u32x2, hb)[1]};
    *reinterpret_cast<u32x4*>(d + 64) = u32x4{__builtin_bit_cast(u32x2, la)[0], __builtin_bit_cast(u32x2, la)[1],
                                              __builtin_bit_cast(u32x2, lb)[0], __builtin_bit_cast(u32x2, lb)[1]};
  }
}

extern "C" int pp_study_split_pairs(const float* x, int C, long long P, void* out, void* stream) {
  STUDY_CHECK(x && out && C % 32 == 0 && P > 0, "split_pairs: C % 32 == 0");
  long long blocks = (P * (C / 8) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(split_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, C, P, (char*)out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------- the DMA-fed direct convolution
#define TR 8                      // output rows per tile (one per wave)
#define TC 32                     // output columns per tile
#define PR (TR + 2)
#define PC (TC + 2)
#define PIX (PR * PC)             // 340 patch pixels
#define DMA_N ((PIX + 7) / 8)     // 43 LDS-DMA instructions per stage (8 rows of 128 B each)
#define DMA_PW ((DMA_N + 7) / 8)  // 6 per wave (the surplus ones write zeros into the dump row)
#define SLOT_BYTES (DMA_N * 1024) // 44,032
#define DUMP_BYTES 1024

struct PairConvArgs {
  const char* x; const char* w; const float* bias; float* out;
  int C, N, H, W, ld_out, n_chunks, tiles_x, tiles_y, n_tiles;
  unsigned x_bytes, w_bytes;
};

// MODE (timing-only variants, results are wrong for MODE != 0): bit 0 = no MFMAs, bit 1 = no fragment reads, bit 2 = no LDS-DMA
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv3x3_pair_dma_kernel(PairConvArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * 32;
  char* Wl = smem;                                                    // [n_chunks][9][32 rows][128 B], piece-swizzled by (n >> 1) & 7
  char* Slot = smem + (size_t)a.n_chunks * 9 * 32 * 128;              // [2][SLOT_BYTES], rows swizzled by (row >> 1) & 7
  char* Dump = Slot + 2 * SLOT_BYTES;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
  // ---- resident weights: packed [n][tap][C] as 16-byte units [hi4 | lo4] of 4 channels (pp_pack_conv3x3_weights_f16x3)
  {
    const int total = a.n_chunks * 9 * 32 * 8;
    for (int e = tid; e < total; e += 512) {
      const int q = e & 7, row = e >> 3;                             // row = (chunk * 9 + tap) * 32 + n; q = channel quad of the chunk
      const int n = row & 31, ct = row >> 5, tap = ct % 9, chunk = ct / 9;
      const unsigned off = (n0 + n < a.N) ? (unsigned)(((n0 + n) * 9 + tap) * a.C + chunk * 32 + q * 4) * 4u : 0xffffffffu;
      const f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));
      const int sw = (n >> 1) & 7;
      char* base = Wl + (size_t)row * 128;
      // hi quad q -> piece q >> 1, half q & 1; lo quad -> piece 4 + (q >> 1)
      *reinterpret_cast<u32x2*>(base + (((q >> 1) ^ sw) * 16) + (q & 1) * 8) = __builtin_bit_cast(u32x4, w).xy;
      *reinterpret_cast<u32x2*>(base + (((4 + (q >> 1)) ^ sw) * 16) + (q & 1) * 8) = __builtin_bit_cast(u32x4, w).zw;
    }
  }
  // ---- DMA geometry of this lane: instruction k of this wave is instruction i = wv + 8 k of the stage; it covers patch rows
  // 8 i .. 8 i + 7, the lane brings slot position (lane & 7) of row 8 i + (lane >> 3), i.e. source piece (lane & 7) ^ swizzle(row)
  int rel[DMA_PW];
  unsigned m_top = 0, m_bot = 0, m_left = 0, m_right = 0, m_dead = 0;
#pragma unroll
  for (int k = 0; k < DMA_PW; ++k) {
    const int i = wv + 8 * k, row = 8 * i + (lane >> 3);
    const int hy = row / PC, hx = row - hy * PC;
    const int piece = (lane & 7) ^ ((row >> 1) & 7);
    rel[k] = (hy * a.W + hx) * a.C * 4 + piece * 16;
    if (i >= DMA_N || row >= PIX) { m_dead |= 1u << k; rel[k] = 0; }
    if (hy == 0) m_top |= 1u << k;
    if (hy == PR - 1) m_bot |= 1u << k;
    if (hx == 0) m_left |= 1u << k;
    if (hx == PC - 1) m_right |= 1u << k;
  }
  // stage sequence of this block: tiles t = blockIdx.x, + gridDim.x, ...; chunks inside a tile
  const int G = (int)gridDim.x;
  const int my_tiles = (int)blockIdx.x < a.n_tiles ? (a.n_tiles - (int)blockIdx.x + G - 1) / G : 0;
  const int n_stages = my_tiles * a.n_chunks;
  auto issue = [&](int s) __attribute__((always_inline)) {            // LDS-DMA of stage s into slot s & 1 (ghost stage: zeros)
    const bool live = s < n_stages;
    const int tl = live ? s / a.n_chunks : 0, chunk = live ? s - tl * a.n_chunks : 0;
    const int t = (int)blockIdx.x + tl * G;
    const int tx = t % a.tiles_x, r = t / a.tiles_x, ty = r % a.tiles_y, img = r / a.tiles_y;
    const int sbase = (((img * a.H + ty * TR - 1) * a.W + tx * TC - 1) * a.C + chunk * 32) * 4;     // first halo pixel (may be outside)
    unsigned bad = m_dead;
    if (!live) bad = ~0u;
    if (ty == 0) bad |= m_top;
    if (ty == a.tiles_y - 1) bad |= m_bot;
    if (tx == 0) bad |= m_left;
    if (tx == a.tiles_x - 1) bad |= m_right;
    char* slot = Slot + (s & 1) * SLOT_BYTES;
#pragma unroll
    for (int k = 0; k < DMA_PW; ++k) {
      const int i = wv + 8 * k;
      const unsigned off = ((bad >> k) & 1u) ? 0xffffffffu : (unsigned)(sbase + rel[k]);
      char* dst = i < DMA_N ? slot + i * 1024 : Dump;                  // (uniform per wave)
      if (!(MODE & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr)dst, 16, off, 0, 0, 0);
    }
  };
  // ---- fragment addressing.  A: patch row (wv + dy) * PC + lr + dx, piece (2 kb + lh) [hi] / (4 + 2 kb + lh) [lo], swizzled by
  // the ROW; B: weight row (chunk * 9 + tap) * 32 + lr, same pieces, swizzled by (lr >> 1) & 7
  int a_off[9], a_sw[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int row = (wv + tap / 3) * PC + lr + tap % 3;
    a_off[tap] = row * 128;
    a_sw[tap] = (row >> 1) & 7;
  }
  const int b_sw = (lr >> 1) & 7;
  const bool n_ok = n0 + lr < a.N;
  const float bv = (a.bias && n_ok) ? a.bias[n0 + lr] : 0.f;
  f32x16 accm, accc, pend;
#pragma unroll
  for (int r = 0; r < 16; ++r) { accm[r] = 0.f; accc[r] = 0.f; pend[r] = 0.f; }
  int pend_t = -1;
  __syncthreads();                                                     // weights in LDS
  issue(0);
  for (int s = 0; s < n_stages; ++s) {
    // everything this wave has in flight -- its pieces of stage s, the stores of the tile finished two stages ago -- is done;
    // after the barrier every wave's pieces of stage s are in LDS and nobody reads slot (s + 1) & 1 any more
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(s + 1);
    if (pend_t >= 0) {                                                 // the tile finished in the previous stage: one stage late
      const int tx = pend_t % a.tiles_x, r0 = pend_t / a.tiles_x, ty = r0 % a.tiles_y, img = r0 / a.tiles_y;
      float* o = a.out + ((size_t)(img * a.H + ty * TR + wv) * a.W + tx * TC) * a.ld_out + n0 + lr;
      if (n_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * a.ld_out] = pend[r];
      }
      pend_t = -1;
    }
    const int tl = s / a.n_chunks, chunk = s - tl * a.n_chunks;
    const char* P0 = Slot + (s & 1) * SLOT_BYTES;
    const char* B0 = Wl + (size_t)(chunk * 9 * 32 + lr) * 128;
    f16x8 ah[2], al[2], bh[2], bl[2];
    if (MODE & 2) {                        // operands that the compiler cannot fold away
      const _Float16 v = (_Float16)(float)(lane + s);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) { ah[q][e] = v; al[q][e] = v; bh[q][e] = v; bl[q][e] = v; }
    }
    auto read_step = [&](int st, int slot) __attribute__((always_inline)) {
      if (MODE & 2) return;
      const int tap = st >> 1, kb = st & 1;
      const int ph = 2 * kb + lh, pl = 4 + 2 * kb + lh;
      ah[slot] = *reinterpret_cast<const f16x8*>(P0 + a_off[tap] + ((ph ^ a_sw[tap]) * 16));
      al[slot] = *reinterpret_cast<const f16x8*>(P0 + a_off[tap] + ((pl ^ a_sw[tap]) * 16));
      bh[slot] = *reinterpret_cast<const f16x8*>(B0 + tap * 32 * 128 + ((ph ^ b_sw) * 16));
      bl[slot] = *reinterpret_cast<const f16x8*>(B0 + tap * 32 * 128 + ((pl ^ b_sw) * 16));
    };
    read_step(0, 0);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      const int cur = st & 1;
      if (st + 1 < 18) read_step(st + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & 1) {                       // keep the operands alive without the matrix pipe
        accm[st & 15] += (float)ah[cur][0] + (float)bh[cur][1];
        accc[st & 15] += (float)al[cur][2] + (float)bl[cur][3];
      } else {
        accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bl[cur], accc, 0, 0, 0);
        accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur], bh[cur], accm, 0, 0, 0);
        accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur], bh[cur], accc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (chunk + 1 == a.n_chunks) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pend[r] = accm[r] + accc[r] * (1.f / LO_SCALE) + bv;
        accm[r] = 0.f; accc[r] = 0.f;
      }
      pend_t = (int)blockIdx.x + tl * G;
    }
  }
  if (pend_t >= 0 && n_ok) {
    const int tx = pend_t % a.tiles_x, r0 = pend_t / a.tiles_x, ty = r0 % a.tiles_y, img = r0 / a.tiles_y;
    float* o = a.out + ((size_t)(img * a.H + ty * TR + wv) * a.W + tx * TC) * a.ld_out + n0 + lr;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * a.ld_out] = pend[r];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // the ghost stage's DMAs must not outlive the workgroup
}

extern "C" int pp_study_conv3x3_pair_fwd(const void* xpair, int C, const void* wf16, const float* bias, float* out, int ld_out, int N,
                                         int B, int H, int W, int blocks_x, int mode, void* stream) {
  STUDY_CHECK(xpair && wf16 && out, "pair_fwd: null pointer");
  STUDY_CHECK((C == 32 || C == 64) && N % 32 == 0 && W % TC == 0 && H % TR == 0 && ld_out >= N, "pair_fwd: C in {32, 64}, N % 32, W % 32, H % 8");
  const long long P = (long long)B * H * W;
  STUDY_CHECK(P * C * 4 < 0xffffffffLL && (long long)N * 9 * C * 4 < 0xffffffffLL, "pair_fwd: tensor exceeds the buffer-descriptor range");
  PairConvArgs a;
  a.x = (const char*)xpair; a.w = (const char*)wf16; a.bias = bias; a.out = out;
  a.C = C; a.N = N; a.H = H; a.W = W; a.ld_out = ld_out; a.n_chunks = C / 32;
  a.tiles_x = W / TC; a.tiles_y = H / TR; a.n_tiles = B * a.tiles_x * a.tiles_y;
  a.x_bytes = (unsigned)(P * C * 4); a.w_bytes = (unsigned)((long long)N * 9 * C * 4);
  const size_t lds = (size_t)a.n_chunks * 9 * 32 * 128 + 2 * SLOT_BYTES + DUMP_BYTES;
  STUDY_CHECK(lds <= 163840, "pair_fwd: LDS budget");
  const int gy = N / 32;
  int gx = blocks_x > 0 ? blocks_x : 256 / gy;
  if (gx < 1) gx = 1;
  if (gx > a.n_tiles) gx = a.n_tiles;
#define PAIR_LAUNCH(M)                                                                                                                 \
  do {                                                                                                                                 \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_pair_dma_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess) \
      { snprintf(g_study_err, sizeof(g_study_err), "hipFuncSetAttribute failed"); return -2; }                                         \
    hipLaunchKernelGGL(conv3x3_pair_dma_kernel<M>, dim3(gx, gy), dim3(512), lds, (hipStream_t)stream, a);                              \
  } while (0)
  switch (mode) {
    case 0: PAIR_LAUNCH(0); break;
    case 1: PAIR_LAUNCH(1); break;
    case 2: PAIR_LAUNCH(2); break;
    case 3: PAIR_LAUNCH(3); break;
    case 4: PAIR_LAUNCH(4); break;
    case 7: PAIR_LAUNCH(7); break;
    default: snprintf(g_study_err, sizeof(g_study_err), "pair_fwd: mode 0 | 1 | 2 | 3 | 4 | 7"); return -1;
  }
#undef PAIR_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
