/* pacingpseudo_hip.h -- C ABI of libpacingpseudo_hip.so (MI355X / gfx950).
 *
 * The reference (zefanyang/pacingpseudo) has no native/FFI layer: its hot path dispatches stock PyTorch
 * operators from three Python modules.  This header is the drop-in boundary that replaces those operator
 * call sites for the PacingPseudo training step; every entry cites the reference call site it stands in for
 * (paths relative to the reference repository root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain C types only: raw DEVICE pointers, explicit sizes / leading dimensions, `void* stream` = hipStream_t
 *     (pass torch.cuda.current_stream().cuda_stream); all work is enqueued on that stream, no implicit sync.
 *   - return value: 0 = OK, negative = library error (bad argument -1, unsupported -2, workspace -3),
 *     positive = hipError_t.  pp_last_error() returns a thread-local message for the last failure.
 *   - the library allocates nothing persistent and keeps no pointer after return; workspaces are caller-owned.
 *     Process state: a per-(kernel, device) cache of the dynamic-LDS attribute (mutex-protected), the opt-in event
 *     profiler below, and PP_* environment knobs that select between equivalent kernels (read once, never written).
 *   - activations are NHWC fp32: element (pixel p, channel c) of a tensor lives at base[p * ld + c] where
 *     p = (n*H + y)*W + x and ld >= C is the pixel stride in floats ("leading dimension").  A channel slice of a
 *     wider tensor (the concatenation buffers of the decoder) is addressed by offsetting `base` and keeping ld.
 *     C, ld multiples of 4 and 16-byte aligned bases are required (float4 accesses).
 *   - logits / scribbles / masks at the module boundary stay NCHW exactly as the reference passes them.
 *   - "groups": the weak and the strong view of one siamese step are laid back to back along the batch axis;
 *     BatchNorm statistics are taken per group (= per reference module call).
 */
#ifndef PACINGPSEUDO_HIP_H
#define PACINGPSEUDO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- lazy activations (round 4) ------------------------------------------------------------------------ */
/* In train mode BatchNorm2d's batch statistics (models/unet.py:189) exist only once the whole convolution output z is
 * written, so y = LeakyReLU(BN(z)) used to cost one more pass over the tensor.  A LAZY tensor keeps z in memory together
 * with per-(group, channel) coefficient rows and every *_lazy entry point evaluates y while it loads:
 *   y = v > 0 ? v : v * slope,  v = z * scale + shift,
 *   coef[(g * 3 + 0) * ld + c] = scale, [(g * 3 + 1) * ld + c] = shift, [(g * 3 + 2) * ld + c] = slope,  g = image / (B / groups).
 * Channels that already hold final values carry the identity row (1, 0, 1).  coef == NULL (or a NULL descriptor) means
 * the tensor is an ordinary one.  A channel slice of a wider lazy tensor passes coef + c0 and keeps ld.  pp_bn_train_finalize_lazy
 * writes the rows; pp_lazy_materialize turns a lazy tensor into an ordinary one (for consumers without a *_lazy form). */
typedef struct pp_lazy_in {
  const float* coef;   /* device pointer, 16-byte aligned; NULL = not lazy */
  int ld;              /* floats per coefficient row (>= channels of the view, multiple of 4) */
  int groups;          /* statistics groups along the batch axis (divides B) */
} pp_lazy_in;

/* ---- runtime ------------------------------------------------------------------------------------------ */
/* 100 * round + revision; bumped whenever an entry point is removed or changes its arguments (600: round 6).  The Python
 * binding refuses a library older than the header it was written against (pacingpseudo_amd/_lib.py: MIN_LIB_VERSION). */
int pp_version(void);
const char* pp_last_error(void);
int pp_device_info(int* cu_count, int* lds_per_cu_kb, char* arch, int arch_len);
/* optional profiler: HIP events around every call, accumulated per kernel family (see PP_KIND_*). */
/* Numeric mode of the forward / data-gradient matrix kernels, per process: 3 = every fp32 product as three fp16 MFMA
 * products of split operands (fp32 grade, the default), 1 = hi parts only (fp16 operands with fp32 accumulation: the mixed
 * precision mode of train_chaos.py --precision fp16; no 1e-4 parity claim).  The Winograd weight-gradient GEMM follows the
 * mode (round 4); the direct weight-gradient kernels always use 3 (1 in the 16-bit storage build, whose operands have no low part). */
int pp_set_matrix_products(int n);
int pp_get_matrix_products(void);
/* CU budget of the persistent direct weight-gradient kernels (conv weight gradient of models/unet.py:188 for the narrow layers), per
 * CALLING THREAD (round 6; a thread that never set one launches with the default): how many CUs' worth of blocks
 * pp_conv3x3_bwd_weight_f16x3 launches.  256 (default; PP_WGRAD_CUS) fills the chip; the engine sets 192 for the duration of a
 * backward pass whose weight gradients run on its second stream beside the data-gradient / BatchNorm chain and restores the
 * previous value (round 5: same-box -0.6 ms per step against 256, profiles/r05_experiments/).  pp_conv3x3_bwd_weight_workspace
 * does not depend on it (sized for the largest budget, 8 <= cus <= 1024). */
int pp_set_wgrad_cus(int cus);
int pp_get_wgrad_cus(void);
/* named ranges for `rocprofv3 --marker-trace` (roctxRangePush / Pop resolved at run time; no-ops without a roctx library):
 * the engine brackets the phases of a step (pack, forward, losses, aux path, backward decoder / aux / encoder, optimizer) */
int pp_range_push(const char* name);
int pp_range_pop(void);
int pp_prof_enable(int on);
int pp_prof_reserve(int events);                     /* pre-create the event pool (2 events per timed launch until the next collect) */
int pp_prof_select(unsigned long long kind_mask);   /* time only the families whose bit (1 << PP_KIND_*) is set; default all */
int pp_prof_collect(double* out /* [kinds][5] = launches, ms, executed flops, algorithmic bytes, algorithmic flops */,
                    int kinds);
#define PP_KIND_CONV_IGEMM 0
#define PP_KIND_CONV_WGRAD 1
#define PP_KIND_BN 2
#define PP_KIND_SPATIAL 3
#define PP_KIND_LOSS 4
#define PP_KIND_OPTIM 5
#define PP_KIND_MISC 6
#define PP_KIND_WINO_GEMM 7   /* flops booked = executed transform-domain flops (8 per pixel*cin*cout) */
#define PP_KIND_WINO_WGRAD 8
#define PP_KIND_WINO_XFORM 9
#define PP_KIND_CONV_F16X3 10  /* flops booked = executed 16-bit MFMA flops (3 per algorithmic flop) */
#define PP_KIND_WINO_GEMM_F16X3 11
#define PP_KIND_WINO_WGRAD_F16X3 12
#define PP_KIND_CONV_WGRAD_F16X3 13
#define PP_KIND_CONV_HALO_F16X3 14 /* conv3x3_halo_f16x3_kernel; kind 10 is conv3x3_igemm_f16x3_kernel */
#define PP_KIND_COUNT 15

/* ---- layout conversion at the module boundary --------------------------------------------------------- */
/* batch['image'] (N,C,H,W) -> NHWC, channels zero-padded to Cpad (train_chaos.py:269 -> models/unet.py:63). */
int pp_pack_image_nchw_to_nhwc(const float* src, int N, int C, int H, int W, float* dst, int ld_dst, int Cpad,
                               void* stream);
/* nn.Conv2d weight (O,I,3,3) -> wf[O][9][Ipad] (forward operand) and wb[I][9][O] with taps flipped (data-gradient
 * operand; may be NULL). */
int pp_pack_conv3x3_weights(const float* w_oihw, int O, int I, int Ipad, float* wf, float* wb, void* stream);

/* ---- 3x3 convolution, stride 1, padding = dilation (nn.Conv2d at models/unet.py:188, aux_path_memory.py:24) ---- */
/* out[p][n] = sum_{tap,c} in[p+off(tap)*dil][c] * wf[n][tap][c] + bias[n]  (+ previous out if accumulate) */
int pp_conv3x3_fwd(const float* in, int ld_in, int C, const float* wf, const float* bias, float* out, int ld_out,
                   int N, int B, int H, int W, int dil, int accumulate, void* stream);
/* autograd of the same Conv2d wrt its input: dx[p][c] (+)= sum_{tap,o} dz[p-off(tap)*dil][o] * W[o][c][tap] */
int pp_conv3x3_bwd_data(const float* dz, int ld_dz, int O, const float* wb, float* dx, int ld_dx, int I, int B,
                        int H, int W, int dil, int accumulate, void* stream);
/* ... and wrt its weight, written in the nn.Conv2d layout: dw[o][c][ky][kx] (+)= sum_p dz[p][o] * x[p+off][c] */
size_t pp_conv3x3_bwd_weight_workspace(int O, int Cpad, int B, int H, int W);
int pp_conv3x3_bwd_weight(const float* dz, int ld_dz, int O, const float* x, int ld_x, int Cpad, int I_true, int B,
                          int H, int W, int dil, float* dw_oihw, int accumulate, float* workspace,
                          size_t workspace_bytes, void* stream);

/* ---- the same convolution on the fp16 matrix cores with split operands ("f16x3") ----------------------------- */
/* Every fp32 operand is split as x ~ hi + lo * 2^-11 (two fp16 numbers) and a*b is evaluated as ah*bh + 2^-11 *
 * (ah*bl + al*bh) with fp32 accumulation: fp32-grade results (the dropped term is 2^-22 relative) at 16x the MFMA
 * rate of the fp32 instruction.  Weights are split once by pp_pack_conv3x3_weights_f16x3 (same sizes and indexing as
 * the fp32 layouts of pp_pack_conv3x3_weights: every 4 consecutive K elements become 16 bytes [hi0..3 | lo0..3]),
 * activations while they are staged.  `in_amax` / `dz_amax`: NULL, or a device float holding max|operand|; the
 * operand is then scaled by a power of two into the fp16 range and the result scaled back (needed for gradients). */
int pp_pack_conv3x3_weights_f16x3(const float* w_oihw, int O, int I, int Ipad, void* wf16, void* wb16, void* stream);
/* the same for n layers in ONE launch (the per-layer launches sit behind one another at the start of every training step):
 * `items` is a HOST array, read during the call; fields as the arguments of pp_pack_conv3x3_weights_f16x3 */
typedef struct pp_pack_item { const float* w_oihw; int O, I, Ipad; void* wf16; void* wb16; } pp_pack_item;
int pp_pack_conv3x3_weights_f16x3_batch(const pp_pack_item* items, int n, void* stream);
int pp_conv3x3_fwd_f16x3(const float* in, int ld_in, int C, const void* wf16, const float* bias, float* out, int ld_out,
                         int N, int B, int H, int W, int dil, int accumulate, const float* in_amax, void* stream);
int pp_conv3x3_bwd_data_f16x3(const float* dz, int ld_dz, int O, const void* wb16, float* dx, int ld_dx, int I, int B,
                              int H, int W, int dil, int accumulate, const float* dz_amax, void* stream);
/* weight gradient, same workspace as pp_conv3x3_bwd_weight; shapes the split-fp16 kernel does not cover (dilation > 1,
 * channel counts that are not multiples of 32, W % 32, H % 4) and dz_amax == NULL run the fp32 kernels */
int pp_conv3x3_bwd_weight_f16x3(const float* dz, int ld_dz, int O, const float* x, int ld_x, int Cpad, int I_true, int B,
                                int H, int W, int dil, float* dw_oihw, int accumulate, float* workspace,
                                size_t workspace_bytes, const float* dz_amax, void* stream);

/* ---- the same convolution through Winograd F(2x2,3x3) (fp32, 2.25x less matrix work; wide layers) ------------ */
/* output-tile edge the library uses for an image shape: 4 = F(4x4,3x3) (36 planes) when H, W are multiples of
 * 4*dil, else 2 = F(2x2,3x3) (16 planes) */
int pp_conv3x3_wino_tile(int H, int W, int dil);
/* Uf[planes][O][I] = G g G^T, Ub[planes][I][O] = transform of the flipped kernel (data gradient); either may be NULL */
int pp_wino_pack_weights(const float* w_oihw, int O, int I, int tile, float* Uf, float* Ub, void* stream);
size_t pp_conv3x3_wino_workspace(int Cin, int Cout, int B, int H, int W, int dil);
size_t pp_conv3x3_wino_vkeep_elems(int Cin, int B, int H, int W, int dil);
/* v_keep (nullable, pp_conv3x3_wino_vkeep_elems floats): receives the transformed input so the weight gradient can reuse it */
int pp_conv3x3_wino_fwd(const float* in, int ld_in, int C, const float* Uf, const float* bias, float* out, int ld_out,
                        int N, int B, int H, int W, int dil, int accumulate, float* v_keep, void* workspace,
                        size_t workspace_bytes, void* stream);
int pp_conv3x3_wino_bwd_data(const float* dz, int ld_dz, int O, const float* Ub, float* dx, int ld_dx, int I, int B,
                             int H, int W, int dil, int accumulate, void* workspace, size_t workspace_bytes,
                             void* stream);
/* split-fp16 ("f16x3") forms of the two calls above for the F(4x4,3x3) geometry (pp_conv3x3_wino_tile == 4) with the GEMM
 * K (input channels forward, output channels in the data gradient) a multiple of 8: U from pp_wino_pack_weights_f16x3
 * (pre-split OCTETS [hi 8 x fp16 | lo 8 x fp16] along K, same sizes as the fp32 U); the input transform writes the
 * transformed input in the same layout, scaled by a power of two that is fixed BEFORE it runs: 2^-4 for activations,
 * for gradients taken from dz_amax = max |dz| (nullable device float; the BatchNorm backward kernels that write dz
 * collect it; null costs one extra pass over dz).  Workspace as for the fp32 calls. */
int pp_wino_pack_weights_f16x3(const float* w_oihw, int O, int I, int tile, void* Uf16, void* Ub16, void* stream);
/* ... and for n layers in ONE launch (tile 4; `items` is a HOST array, read during the call) */
typedef struct pp_wino_pack_item { const float* w_oihw; int O, I; void* Uf16; void* Ub16; } pp_wino_pack_item;
int pp_wino_pack_weights_f16x3_batch(const pp_wino_pack_item* items, int n, void* stream);
int pp_conv3x3_wino_fwd_f16x3(const float* in, int ld_in, int C, const void* Uf16, const float* bias, float* out,
                              int ld_out, int N, int B, int H, int W, int dil, int accumulate, float* v_keep,
                              void* workspace, size_t workspace_bytes, void* stream);
int pp_conv3x3_wino_bwd_data_f16x3(const float* dz, int ld_dz, int O, const void* Ub16, float* dx, int ld_dx, int I, int B,
                                   int H, int W, int dil, int accumulate, void* workspace, size_t workspace_bytes,
                                   const float* dz_amax, void* stream);
size_t pp_conv3x3_wino_bwd_weight_workspace(int O, int C, int B, int H, int W, int dil);
/* reduction splits the weight-gradient GEMM of this shape runs with (shape-only; quoted by the parity tests) */
int pp_conv3x3_wino_bwd_weight_splits(int O, int C, int B, int H, int W, int dil);
/* v_cached (nullable): the v_keep of the forward call on the same x; when given, x is not read again */
int pp_conv3x3_wino_bwd_weight(const float* dz, int ld_dz, int O, const float* x, int ld_x, int C, int B, int H, int W,
                               int dil, float* dw_oihw, int accumulate, const float* v_cached, void* workspace,
                               size_t workspace_bytes, void* stream);
/* split-fp16 GEMM form (F(4x4,3x3) geometry, O and C multiples of 8); a cached V must have been written by a
 * pp_conv3x3_wino_fwd_f16x3 / pp_conv3x3_wino_fwd_bn(f16x3 = 1) call that received it as v_keep (pre-split octets) */
int pp_conv3x3_wino_bwd_weight_f16x3(const float* dz, int ld_dz, int O, const float* x, int ld_x, int C, int B, int H, int W,
                               int dil, float* dw_oihw, int accumulate, const float* v_cached, void* workspace,
                               size_t workspace_bytes, const float* dz_amax, void* stream);

/* ---- BatchNorm2d + LeakyReLU (models/unet.py:189-193, aux_path_memory.py:25-26) ------------------------- */
size_t pp_bn_workspace(int C, int P_per_group, int groups);
/* train mode: batch statistics per group, running-stat / num_batches_tracked update (one per group, in order),
 * save_mean / save_invstd [groups][C] and the fused coefficients scale = gamma*invstd, shift = beta - mean*scale. */
int pp_bn_train_stats(const float* z, int ld, int C, int P_per_group, int groups, float eps, float momentum,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, float* save_mean, float* save_invstd, float* scale,
                      float* shift, void* workspace, size_t workspace_bytes, void* stream);
/* eval mode (what train_chaos.py:370 leaves the model in): coefficients from the running statistics. */
int pp_bn_eval_coeffs(int C, int groups, float eps, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, float* save_mean, float* save_invstd,
                      float* scale, float* shift, void* stream);
/* ... for every BatchNorm layer of a network in ONE launch (round 6; eval mode: all coefficients are known when the step starts).
 * The rows written are bit-identical to pp_bn_eval_coeffs'. */
typedef struct pp_bn_coef_item {
  int C, groups;
  const float *gamma, *beta, *running_mean, *running_var;
  float *save_mean, *save_invstd, *scale, *shift;      /* [groups][C] each */
} pp_bn_coef_item;
int pp_bn_eval_coeffs_batch(const pp_bn_coef_item* items, int n, float eps, void* stream);
/* y = leaky_relu(z*scale + shift, slope) */
int pp_bn_lrelu_fwd(const float* z, int ld_z, const float* scale, const float* shift, float* y, int ld_y, int C,
                    int P_per_group, int groups, float slope, void* stream);
/* ... and the 2x2 max-pooled copy of y in the same pass (the train-mode forward of an encoder stage's last layer: its output
 * feeds the skip connection and nn.MaxPool2d(2, 2), models/unet.py:109,123-127); pooled is (B, H/2, W/2, C) */
int pp_bn_lrelu_fwd_pool(const float* z, int ld_z, const float* scale, const float* shift, float* y, int ld_y, float* pooled,
                         int ld_pooled, int C, int B, int H, int W, int groups, float slope, void* stream);
/* autograd of LeakyReLU(BatchNorm(z)): dz, dgamma, dbeta, and the gradient of the preceding conv bias.
 * workspace >= pp_bn_workspace(...) + 3*groups*C*sizeof(float). */
int pp_bn_lrelu_bwd(const float* dy, int ld_dy, const float* z, int ld_z, const float* scale, const float* shift,
                    const float* save_mean, const float* save_invstd, const float* gamma, int training, float* dz,
                    int ld_dz, float* dgamma, float* dbeta, float* dbias_conv, int accumulate_param_grads, int C,
                    int P_per_group, int groups, float slope, void* workspace, size_t workspace_bytes,
                    void* stream);
/* same, additionally *dz_amax = max |dz| (device float): operand scale for pp_conv3x3_bwd_data_f16x3 */
int pp_bn_lrelu_bwd_amax(const float* dy, int ld_dy, const float* z, int ld_z, const float* scale, const float* shift,
                    const float* save_mean, const float* save_invstd, const float* gamma, int training, float* dz,
                    int ld_dz, float* dgamma, float* dbeta, float* dbias_conv, int accumulate_param_grads, int C,
                    int P_per_group, int groups, float slope, void* workspace, size_t workspace_bytes,
                    float* dz_amax, void* stream);

/* Synchronised BatchNorm for data-parallel training while BN is in train mode (the reference normalises over the WHOLE
 * batch, models/unet.py:189; reference has no multi-GPU path, SURVEY.md 8(e) coupling A).  Each call above is split in
 * two so the caller can all-reduce the per-channel sums (double [groups][2][C]) over the ranks in between:
 *   forward : pp_bn_stats_sums -> all-reduce -> pp_bn_train_finalize(n = GLOBAL pixels per group)
 *   backward: pp_bn_lrelu_bwd_sums -> all-reduce of a copy -> pp_bn_lrelu_bwd_apply(local sums, global sums, global n):
 *             dz uses the global sums, dgamma / dbeta / dbias the local ones (the gradient all-reduce adds the ranks).
 * pp_bn_lrelu_bwd_apply workspace >= 6*groups*C floats + 16. */
int pp_bn_stats_sums(const float* z, int ld, int C, int P_per_group, int groups, double* sums, void* workspace,
                     size_t workspace_bytes, void* stream);
int pp_bn_train_finalize(const double* sums, int rows /* partial rows per group; 1 for pp_bn_stats_sums output */, int C,
                         int n_per_group, int groups, float eps, float momentum,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         int64_t* num_batches_tracked, float* save_mean, float* save_invstd, float* scale, float* shift,
                         void* stream);
/* the same, and additionally the (scale, shift, slope) rows of the layer's LAZY output tensor (pp_lazy_in above): lazy_coef
 * points at the layer's channel 0 inside rows of lazy_ld floats.  The engine then skips pp_bn_lrelu_fwd: the consumers of
 * the tensor apply BatchNorm + LeakyReLU while they load (models/unet.py:189-193 executed inside the next layer). */
int pp_bn_train_finalize_lazy(const double* sums, int rows, int C, int n_per_group, int groups, float eps, float momentum,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, float* save_mean, float* save_invstd, float* scale,
                              float* shift, float* lazy_coef, int lazy_ld, float slope, void* stream);
/* dst = LeakyReLU(BN(src)) of a lazy tensor (B images of HW pixels, C channels): for consumers without a *_lazy form */
int pp_lazy_materialize(const float* src, int ld_src, const pp_lazy_in* lazy, float* dst, int ld_dst, int C, int B, int HW,
                        void* stream);
int pp_bn_lrelu_bwd_sums(const float* dy, int ld_dy, const float* z, int ld_z, const float* scale, const float* shift,
                         const float* save_mean, const float* save_invstd, int C, int P_per_group, int groups,
                         float slope, double* sums, void* workspace, size_t workspace_bytes, void* stream);
int pp_bn_lrelu_bwd_apply(const float* dy, int ld_dy, const float* z, int ld_z, const float* scale, const float* shift,
                          const float* save_mean, const float* save_invstd, const float* gamma, int training,
                          const double* local_sums, const double* global_sums, int n_global_per_group, float* dz,
                          int ld_dz, float* dgamma, float* dbeta, float* dbias_conv, int accumulate_param_grads, int C,
                          int P_per_group, int groups, float slope, void* workspace, size_t workspace_bytes,
                          float* dz_amax /* nullable */, void* stream);

/* ---- convolution with the BatchNorm that follows it fused into the kernel epilogue ----------------------------------
 * The reference block is conv -> BatchNorm2d -> LeakyReLU (models/unet.py:188-193).  These forward calls replace
 * pp_conv3x3_fwd[_f16x3] / pp_conv3x3_wino_fwd[_f16x3] + pp_bn_train_stats / pp_bn_lrelu_fwd where BatchNorm needs no
 * second look at the convolution output:
 *   bn_mode 1 (BN in train mode): out = z, and the kernel emits per-block (sum z, sum z^2) per channel into
 *       stats[groups][*rows_out][2][N] (double) -> pp_bn_train_finalize(stats, *rows_out, ...) -> pp_bn_lrelu_fwd.
 *       The statistics pass over z is gone.
 *   bn_mode 2 (BN in eval mode, the reference's state from epoch 1 on, train_chaos.py:370): scale / shift [N] from
 *       pp_bn_eval_coeffs are known up front, out = y = leaky_relu(z*scale + shift, slope); z never reaches HBM, and
 *       pp_bn_lrelu_bwd_eval reads the LeakyReLU branch from the sign of y.
 * Kernel variants without a fused epilogue run the unfused BatchNorm kernels inside the call: results are the same.
 * f16x3 != 0: w / U are the split-fp16 packs and the split-fp16 kernels run.  stats: >= pp_conv3x3_bn_stats_bytes. */
size_t pp_conv3x3_bn_stats_bytes(int N, int B, int H, int W, int groups);
int pp_conv3x3_fwd_bn(const float* in, int ld_in, int C, const void* wf, const float* bias, float* out, int ld_out, int N,
                      int B, int H, int W, int dil, int f16x3, const float* in_amax, int bn_mode, const float* scale,
                      const float* shift, float slope, int groups, double* stats, size_t stats_bytes, int* rows_out,
                      void* stream);
int pp_conv3x3_wino_fwd_bn(const float* in, int ld_in, int C, const void* U, const float* bias, float* out, int ld_out,
                           int N, int B, int H, int W, int dil, int f16x3, float* v_keep, void* workspace,
                           size_t workspace_bytes, int bn_mode, const float* scale, const float* shift, float slope,
                           int groups, double* stats, size_t stats_bytes, int* rows_out, void* stream);
/* Round 4, the conv -> conv halves of a DoubleConv (models/unet.py:160-170, narrow layers): pp_conv3x3_fwd_bn with a LAZY input,
 * and the split-fp16 weight gradient with a lazy x -- `in` / `x` hold the raw output z of the first convolution, BatchNorm +
 * LeakyReLU are applied while the two-half halo kernel (forward) / the halo-tile weight-gradient kernels stage their patches;
 * zero padding applies to y.  At most two statistics groups.  pp_conv3x3_lazy_ok: 1 when BOTH calls accept this layer shape
 * (C input, N output channels), else 0 -- a pure function of the shape; other shapes fail with PP_ERR_UNSUPPORTED. */
int pp_conv3x3_lazy_ok(int C, int N, int B, int H, int W, int dil);
int pp_conv3x3_fwd_bn_lazy(const float* in, int ld_in, int C, const void* wf, const float* bias, float* out, int ld_out, int N,
                           int B, int H, int W, int dil, int f16x3, const float* in_amax, int bn_mode, const float* scale,
                           const float* shift, float slope, int groups, double* stats, size_t stats_bytes, int* rows_out,
                           const pp_lazy_in* lazy_in, void* stream);
int pp_conv3x3_bwd_weight_f16x3_lazy(const float* dz, int ld_dz, int O, const float* x, int ld_x, int Cpad, int I_true, int B,
                                     int H, int W, int dil, float* dw_oihw, int accumulate, float* workspace,
                                     size_t workspace_bytes, const float* dz_amax, const pp_lazy_in* lazy_x, void* stream);
/* autograd of LeakyReLU(BatchNorm_eval(z)) from dy and y alone, one pass: dz = scale*g, dgamma, dbeta, conv-bias grad.
 * scale = gamma*invstd [C] (pp_bn_eval_coeffs); P_total = all pixels of the launch (statistics are not per group in
 * eval mode); workspace >= pp_bn_workspace(C, P_total, 1); dz_amax nullable (max |dz| for the split-fp16 consumers). */
int pp_bn_lrelu_bwd_eval(const float* dy, int ld_dy, const float* y, int ld_y, const float* scale, const float* gamma,
                         const float* beta, float* dz, int ld_dz, float* dgamma, float* dbeta, float* dbias_conv,
                         int accumulate_param_grads, int C, int P_total, float slope, void* workspace,
                         size_t workspace_bytes, float* dz_amax, void* stream);
/* BatchNorm + LeakyReLU backward of an encoder stage's last layer with the gradient of the nn.MaxPool2d(2, 2) that follows it
 * (models/unet.py:109,123-127) folded in: dy = gradient arriving through the skip connection (B, H, W), dpool = gradient of
 * the pooled tensor (B, H/2, W/2); the kernels find each window's winner themselves (first maximum of y, PyTorch's rule) and
 * add dpool to it on the fly, so the separate pp_maxpool2_bwd pass over the skip-gradient buffer does not run.
 * pp_bn_lrelu_bwd_pool: train-mode statistics (or training = 0), arguments as pp_bn_lrelu_bwd_amax, dz_amax nullable.
 * pp_bn_lrelu_bwd_eval_pool: the one-pass eval form on the stored output y, arguments as pp_bn_lrelu_bwd_eval. */
int pp_bn_lrelu_bwd_pool(const float* dy, int ld_dy, const float* dpool, int ld_dpool, const float* z, int ld_z,
                         const float* scale, const float* shift, const float* save_mean, const float* save_invstd,
                         const float* gamma, int training, float* dz, int ld_dz, float* dgamma, float* dbeta,
                         float* dbias_conv, int accumulate_param_grads, int C, int B, int H, int W, int groups, float slope,
                         void* workspace, size_t workspace_bytes, float* dz_amax, void* stream);
int pp_bn_lrelu_bwd_eval_pool(const float* dy, int ld_dy, const float* dpool, int ld_dpool, const float* y, int ld_y,
                              const float* scale, const float* gamma, const float* beta, float* dz, int ld_dz, float* dgamma,
                              float* dbeta, float* dbias_conv, int accumulate_param_grads, int C, int B, int H, int W,
                              float slope, void* workspace, size_t workspace_bytes, float* dz_amax, void* stream);
/* BatchNorm + LeakyReLU backward of the network's FIRST layer (models/unet.py:188-193 on `input_ch` = 1, every dataset of the
 * reference) with that layer's weight gradient folded in: nobody asks for the data gradient of the first convolution, so dz had
 * one reader -- the weight gradient, the last kernel of the backward pass.  The pass that forms dz multiplies it with the 3x3
 * neighbourhood of the one-channel input x (channel 0 of an NHWC tensor with row stride ld_x, zero padding, dilation 1) and
 * leaves dW [C][1][3][3] (accumulate_dw: added to it); dz is never written.  A group is a whole number of H x W images.
 * Other arguments as pp_bn_lrelu_bwd / pp_bn_lrelu_bwd_eval; workspace >= pp_bn_lrelu_bwd_wgrad_c1_workspace (eval: groups = 1). */
size_t pp_bn_lrelu_bwd_wgrad_c1_workspace(int C, int P_per_group, int groups);
int pp_bn_lrelu_bwd_wgrad_c1(const float* dy, int ld_dy, const float* z, int ld_z, const float* scale, const float* shift,
                             const float* save_mean, const float* save_invstd, const float* gamma, int training,
                             const float* x, int ld_x, int H, int W, float* dw_o1hw, int accumulate_dw, float* dgamma,
                             float* dbeta, float* dbias_conv, int accumulate_param_grads, int C, int P_per_group, int groups,
                             float slope, void* workspace, size_t workspace_bytes, void* stream);
int pp_bn_lrelu_bwd_eval_wgrad_c1(const float* dy, int ld_dy, const float* y, int ld_y, const float* scale, const float* gamma,
                                  const float* beta, const float* x, int ld_x, int H, int W, float* dw_o1hw, int accumulate_dw,
                                  float* dgamma, float* dbeta, float* dbias_conv, int accumulate_param_grads, int C,
                                  int P_total, float slope, void* workspace, size_t workspace_bytes, void* stream);

/* ---- pooling / resampling (models/unet.py:109,144; aux_path_memory.py:52,75) ---------------------------- */
int pp_maxpool2_fwd(const float* x, int ld_x, float* y, int ld_y, int C, int N, int H, int W, void* stream);
int pp_maxpool2_bwd(const float* x, int ld_x, const float* dy, int ld_dy, float* dx, int ld_dx, int C, int N, int H,
                    int W, int accumulate, void* stream);
/* bilinear, align_corners=True, any size (nn.Upsample / F.interpolate) */
int pp_bilinear_fwd(const float* x, int ld_x, float* y, int ld_y, int C, int N, int Hi, int Wi, int Ho, int Wo,
                    void* stream);
int pp_bilinear_bwd(const float* dy, int ld_dy, float* dx, int ld_dx, int C, int N, int Hi, int Wi, int Ho, int Wo,
                    int accumulate, void* stream);
/* y[p][0:C] (+)= x[p][0:C]: torch.cat placement / scale_factor=1 up-sampling (models/unet.py:151) */
int pp_copy_slab(const float* x, int ld_x, float* y, int ld_y, int C, long long P, int accumulate, void* stream);

/* y[n][p][c] (+)= x[n][p][c] * scale[n][c]: nn.Dropout2d (aux_path_memory.py:22,31) forward and backward; the caller draws
 * the per-(sample, channel) mask = 0 or 1/(1-p) and keeps it for the backward pass */
int pp_channel_scale(const float* x, int ld_x, float* y, int ld_y, const float* scale, int C, int N, int HW,
                     int accumulate, void* stream);

/* ---- synthetic scribbles (utils/utils_artificial_scribbles.py:5-35, utils/utils_shorten_scribble_length.py:32-75) ----
 * masks: uint8 [M][H][W] (0 / non-zero), processed in place, one workgroup per mask, (H+2)*(W+2) <= 81,888 (LDS).
 * pp_skeletonize = skimage.morphology.skeletonize (2-D Zhang-Suen thinning); pp_dilate_antidiagonal =
 * scipy.ndimage.binary_dilation(seed, np.eye(3)[::-1], iterations, mask); pp_curve_endpoints marks set pixels with exactly
 * one set 8-neighbour (the union of the reference's eight end-point convolution kernels). */
int pp_skeletonize(unsigned char* masks, int M, int H, int W, void* stream);
int pp_dilate_antidiagonal(unsigned char* seeds, const unsigned char* masks, int M, int H, int W, int iterations,
                           void* stream);
int pp_curve_endpoints(const unsigned char* img, unsigned char* out, int M, int H, int W, void* stream);

/* ---- input pipeline on the device: the two-stream augmentation of datasets/chaos/chaos_dataset.py:58-90
 * with the transforms of datasets/augmentations.py and the configuration of datasets/chaos/chaos_aug_configs.py:16-86.
 * The random decisions are drawn on the host (pacingpseudo_amd/augment.py, the reference's call order); these entry
 * points apply them to a whole batch in HBM.  Images: fp32 [B][Hp][Wp] planes; rect: int [B][4] = {top, left, h, w}
 * restricting a call to one rectangle per sample (NULL = whole plane); stats: double [B][4] = {mean, std, min, max}.
 *   pp_aug_stats        np.mean / np.std / np.min / np.max of each sample (MeanStdNorm :11-21, Contrast :112-129, Gamma :131-166)
 *   pp_aug_coef         {a, b, lo, hi} of the per-sample map from the statistics, on the device; mode 0 MeanStdNorm,
 *                       1 Contrast, 2 Gamma power step, 3 Gamma retain_stats, 4 Brightness; param[n] <= -1e30: not drawn
 *   pp_aug_scalar_map   x <- clip(a x + b, lo, hi);   pp_aug_gamma   x <- ((x - min) / (max - min + eps)) ^ gamma
 *   pp_aug_add_noise    x += sigma[n] N(0, 1) (GaussianNoise :353-366), Philox-4x32-10 keyed by `seed`
 *   pp_aug_warp         Scaling :184-226, RandomRotation :278-318, Mirroring :337-351, RandomCrop :368-418 (and the
 *                       displacement field of ElasticTransform :228-276) as ONE resampling: maps[n] = 12 floats
 *                       {a00, a01, a02, a10, a11, a12, top, left, patch_h, patch_w, hs, ws}, source (y, x) = A (yo, xo, 1);
 *                       image bicubic (cubic = 1, the cv2.INTER_CUBIC kernel), bilinear (0) or nearest (2); class maps nearest
 *   pp_aug_elastic_field  gaussian_filter(U(-1, 1), sigma) * alpha per sample and axis -> disp [B][2][H][W]; sigma_alpha [B][2]
 *   pp_aug_onehot       to_one_hot_encoding :448-461, int32 class map -> fp32 [B][K][HW] */
int pp_aug_stats(const float* x, int B, int Hp, int Wp, const int* rect, double* stats, void* stream);
int pp_aug_coef(const double* stats, const double* stats0, const float* param, int mode, int B, float* coef, void* stream);
int pp_aug_scalar_map(float* x, int B, int Hp, int Wp, const float* coef, const int* rect, void* stream);
int pp_aug_gamma(float* x, int B, int Hp, int Wp, const float* coef, const int* rect, void* stream);
int pp_aug_add_noise(float* x, int B, int Hp, int Wp, const float* sigma, const int* rect, unsigned long long seed,
                     void* stream);
int pp_aug_warp(const float* img, const int* lab, const int* scb, int Hp, int Wp, float* out_img, int* out_lab, int* out_scb,
                float* out_valid, int Ho, int Wo, int B, const float* maps, const float* disp, const double* clip_stats,
                float img_pad, int lab_pad, int cubic, void* stream);
int pp_aug_elastic_field(float* disp, float* scratch, int B, int H, int W, const float* sigma_alpha, unsigned long long seed,
                         void* stream);
/* ElasticTransform's own interpolant (augmentations.py:270: scipy.ndimage.map_coordinates(order = 3, mode = 'nearest')):
 * pp_aug_spline_prefilter turns the flagged samples (use[n] != 0) of a batch into cubic B-spline coefficients -- edge padding
 * by 12, float64 recursive prefilter with scipy's 'reflect' initialisation, coef [B][Hp + 24][Wp + 24] -- and
 * pp_aug_warp_spline is pp_aug_warp with those samples' image taps evaluated from the coefficients (coordinates and the
 * class-map rounding in double; disp64, nullable, a double-precision displacement field in place of disp). */
int pp_aug_spline_prefilter(const float* img, int B, int Hp, int Wp, const float* maps, const int* use, double* coef, void* stream);
int pp_aug_warp_spline(const float* img, const int* lab, const int* scb, int Hp, int Wp, float* out_img, int* out_lab, int* out_scb,
                       float* out_valid, int Ho, int Wo, int B, const float* maps, const float* disp, const double* disp64,
                       const double* clip_stats, float img_pad, int lab_pad, int cubic, const double* spline_coef, const int* use,
                       void* stream);
int pp_aug_onehot(const int* lab, float* out, int B, int K, int HW, void* stream);
/* strong-view extras of TransformsColorBlur / Mixup / Low (chaos_aug_configs.py:88-186): GaussianBlur :82-95 (sigma_pad
 * [B][2] = {sigma, unused}, <= 0: untouched), Mixup :51-80 (x <- lam x + (1 - lam) y, lam < 0: untouched);
 * SimulationLowRes :168-182 is two pp_aug_warp calls (cubic = 2: nearest, then cubic = 1) */
int pp_aug_gaussian_blur(float* x, float* scratch, int B, int H, int W, const float* sigma_pad, void* stream);
int pp_aug_mix(float* x, const float* y, int B, int HW, const float* lam, void* stream);
/* GaussianNoise (augmentations.py:353-366) with the normal field given by the caller: x += f inside rect[n] (nullable) */
int pp_aug_add_field(float* x, const float* f, int B, int Hp, int Wp, const int* rect, void* stream);

/* ---- the `is_stride_conv` / `is_trans_conv` U-Net variants (models/unet.py:100-152) ------------------------------------ */
/* stride-2 / padding-1 3x3 convolution (EncBlock, unet.py:113-116) = the stride-1 convolution sampled at the even pixels:
 * out (N, Ho, Wo, C) = full (N, 2 Ho, 2 Wo, C)[:, ::2, ::2]; its gradients are those of the stride-1 convolution for the
 * zero-stuffed dz that pp_stride2_scatter writes */
int pp_stride2_gather(const float* full, int ld_full, float* out, int ld_out, int C, int N, int Ho, int Wo, void* stream);
int pp_stride2_scatter(const float* dz, int ld_dz, float* full, int ld_full, int C, int N, int Ho, int Wo, void* stream);
/* nn.ConvTranspose2d(Cin, Cout, k, k, bias=False) of DecBlock (unet.py:139-141), k = kernel = stride = 1 or 2, weight in
 * PyTorch's layout [Cin][Cout][k][k]; x (N, H, W, Cin) NHWC -> out (N, k H, k W, Cout); fp32 */
int pp_convtranspose_fwd(const float* x, int ld_x, int Cin, const float* w, float* out, int ld_out, int Cout, int k, int N,
                         int H, int W, void* stream);
int pp_convtranspose_bwd_data(const float* dout, int ld_dout, int Cout, const float* w, float* dx, int ld_dx, int Cin, int k,
                              int N, int H, int W, int accumulate, void* stream);
size_t pp_convtranspose_bwd_weight_workspace(int Cin, int Cout, int k, int N, int H, int W);
int pp_convtranspose_bwd_weight(const float* dout, int ld_dout, int Cout, const float* x, int ld_x, int Cin, int k, int N, int H,
                                int W, float* dw, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ---- 1x1 heads: final_conv (models/unet.py:60) and aux fc_cls (aux_path_memory.py:32), NHWC -> NCHW logits ---- */
int pp_conv1x1_nhwc_to_nchw_fwd(const float* x, int ld_x, int C, const float* w, const float* bias, float* logits,
                                int K, int N, int HW, void* stream);
/* ... with a lazy x (the features of dec_block1 kept as raw convolution output in train-mode BN) */
int pp_conv1x1_nhwc_to_nchw_fwd_lazy(const float* x, int ld_x, int C, const float* w, const float* bias, float* logits,
                                int K, int N, int HW, const pp_lazy_in* lazy_x, void* stream);
size_t pp_conv1x1_bwd_workspace(int K, int C, int N, int HW);
int pp_conv1x1_nchw_to_nhwc_bwd(const float* dlogits, const float* x, int ld_x, int C, const float* w, float* dx,
                                int ld_dx, float* dw, float* dbias, int K, int N, int HW, int accumulate_dx,
                                int accumulate_param_grads, void* workspace, size_t workspace_bytes, void* stream);
/* ... with a lazy x: dw is taken against y = LeakyReLU(BN(x)) */
int pp_conv1x1_nchw_to_nhwc_bwd_lazy(const float* dlogits, const float* x, int ld_x, int C, const float* w, float* dx,
                                int ld_dx, float* dw, float* dbias, int K, int N, int HW, int accumulate_dx,
                                int accumulate_param_grads, void* workspace, size_t workspace_bytes, const pp_lazy_in* lazy_x, void* stream);

/* ---- losses (losses/losses.py; models/consistency_reglur_memory.py:31-97) ------------------------------ */
/* torch.argmax(x, dim=1) on (N,C,HW) fp32 -> int64, first maximum wins (bit-exact pseudo-label masks) */
int pp_argmax_channels(const float* x, int N, int C, int HW, int64_t* out, void* stream);
/* sums[6] (double): pCE sum, #labelled, entropy sum, entropy denominator, consistency sum, consistency denominator.
 * cr_variant: 0 none, 1 ce_loss, 2 l1_loss, 3 l2_loss, 4 kl_loss (train_chaos.py:138).  In data-parallel runs the
 * caller all-reduces `sums` before pp_losses_finalize / pp_seg_losses_bwd so losses and gradients are global. */
size_t pp_seg_losses_workspace(int N, int HW);
int pp_seg_losses_fwd(const float* logits_w, const float* logits_s, const int64_t* target, const float* valid_mask,
                      int N, int K, int HW, int ignore_index, int do_ent, int cr_variant, double* sums,
                      void* workspace, size_t workspace_bytes, void* stream);
int pp_losses_finalize(const double* sums, int has_mask, float* loss_pce, float* loss_ent, float* loss_cr,
                       void* stream);
/* dlogits = g_pce*dpce + g_ent*dent + g_cr*dcr (g_* are device scalars = the upstream gradients of each loss) */
int pp_seg_losses_bwd(const float* logits_w, const float* logits_s, const int64_t* target, const float* valid_mask,
                      int N, int K, int HW, int ignore_index, int do_ent, int cr_variant, int detach_weak,
                      const double* sums, const float* g_pce, const float* g_ent, const float* g_cr,
                      float grad_scale, float* dlogits_w, float* dlogits_s, void* stream);
/* auxiliary head: F.interpolate(logits (N,K,h,w) -> (H,W), bilinear, align_corners) fused with partial CE.
 * sums[2] = pCE sum, #labelled. */
int pp_aux_pce_fwd(const float* lo, int N, int K, int h, int w, int H, int W, const int64_t* target,
                   int ignore_index, float* logits_up, double* sums, void* workspace, size_t workspace_bytes,
                   void* stream);
int pp_aux_pce_bwd(const float* logits_up, const int64_t* target, int ignore_index, const float* g_aux,
                   float grad_scale, const double* sums, float* dlo, int N, int K, int h, int w, int H, int W,
                   void* stream);
/* AuxPath.memory_update for batch sample 0 (aux_path_memory.py:68-116).  feat0: NHWC (h,w,hid) features of sample 0,
 * scribble0: (K+1,H,W) one-hot of sample 0, bank: [K][hid], momentum_now = (1-step/max_step)^0.9 * base.  Round 6: two launches
 * (per-slice partial sums over the scribble plane, then the first-visit / EMA rule per class) through a caller-owned workspace of
 * pp_memory_update_workspace(K, hid) bytes. */
size_t pp_memory_update_workspace(int K, int hid);
int pp_memory_update(const float* feat0, int ld, int hid, int h, int w, const float* scribble0, int K, int H, int W,
                     float* bank, float momentum_now, int cosine_mode, void* workspace, size_t workspace_bytes, void* stream);
/* the same with feat0 stored as IEEE fp16 (16-bit storage mode: the entry points of pacingpseudo_hip_h16.h) */
int pp_memory_update_h16(const void* feat0, int ld, int hid, int h, int w, const float* scribble0, int K, int H, int W,
                         float* bank, float momentum_now, int cosine_mode, void* workspace, size_t workspace_bytes, void* stream);
/* ... and as bfloat16 (round 6: the entry points of pacingpseudo_hip_bf16.h) */
int pp_memory_update_bf16(const void* feat0, int ld, int hid, int h, int w, const float* scribble0, int K, int H, int W,
                          float* bank, float momentum_now, int cosine_mode, void* workspace, size_t workspace_bytes, void* stream);
/* cross_entropy(fc_cls(memory_bank), arange(K)) and its gradient wrt the fc_cls weight [K][hid] */
int pp_memory_ce_fwd(const float* bank, const float* wfc, int K, int hid, float* loss, void* stream);
int pp_memory_ce_bwd(const float* bank, const float* wfc, int K, int hid, const float* g, float grad_scale,
                     float* dwfc, int accumulate, void* stream);
/* the loss assembly of train_chaos.py:273-310 in one launch each way: *out = t[0][0] * w[0] + t[1][0] * w[1] + ... (n <= 8
 * 0-dim device losses; `terms` and `weights` are HOST arrays, read during the call; fp32 products and sums, left to right, as the
 * chain of torch operations computes them), and gout[i] = *g * w[i]. */
int pp_weighted_sum_fwd(const float* const* terms, const float* weights, int n, float* out, void* stream);
int pp_weighted_sum_bwd(const float* g, const float* weights, int n, float* gout, void* stream);
/* utils/metrics.py:compute_dice counts: counts[n][k] = {|P&T|, |P|, |T|} with P = argmax prediction */
int pp_dice_counts(const float* logits, const float* label_onehot, int N, int K, int HW, float* counts,
                   void* stream);

/* 95 % Hausdorff distance of inference.py:217-237 (medpy.metric.binary.hd95, connectivity 1): surface extraction and
 * both directed surface-distance sets per (image, class) on the device.  pred / label: int64 class maps [N][H][W];
 * dist: [N*K][2][H*W] floats, counts: [N*K][4] ints = {|surface pred|, |surface label|, |pred|, |label|}.  The caller takes
 * numpy.percentile(95) of the two sets together (NaN when a mask is empty or full, as the reference). */
size_t pp_hd95_workspace(int N, int K, int H, int W);
int pp_hd95_surface_distances(const int64_t* pred, const int64_t* label, int N, int K, int H, int W, float spacing_y,
                              float spacing_x, float* dist, int* counts, void* workspace, size_t workspace_bytes,
                              void* stream);

/* soft Dice loss of the fully-supervised trainer (losses/losses.py:147-162, upper_bound_chaos.py:165):
 * -mean_{n,k} 2 sum p t / (sum p + sum t + 1e-5) on soft-max probabilities.  sums: double [N][K][3] = {sum p t, sum p,
 * sum t}, written by the forward call and read by the backward one; dlogits (+)= g * grad_scale * dloss/dlogits. */
size_t pp_dice_loss_workspace(int N, int K);
int pp_dice_loss_fwd(const float* logits, const float* label_onehot, int N, int K, int HW, double* sums, float* loss,
                     void* workspace, size_t workspace_bytes, void* stream);
int pp_dice_loss_bwd(const float* logits, const float* label_onehot, int N, int K, int HW, const double* sums,
                     const float* g, float grad_scale, float* dlogits, int accumulate, void* stream);

/* ---- optimiser (torch.optim.Adam(lr, weight_decay) at train_chaos.py:219) ------------------------------- */
int pp_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, void* stream);
/* torch.optim.SGD(lr, momentum, weight_decay) (train_chaos.py:220-221, --optimizer momentum); step counts from 1 */
int pp_sgd_momentum_step(float* p, const float* g, float* momentum_buf, long long n, float lr, float momentum,
                         float weight_decay, int step, void* stream);
int pp_fill(float* p, long long n, float value, void* stream);
/* p[i] *= value (the unscaling of the gradient slab after a 16-bit-storage step, see pacingpseudo_hip_h16.h) */
int pp_scale(float* p, long long n, float value, void* stream);
/* Overflow guard of the 16-bit storage mode (the counterpart of torch.cuda.amp.GradScaler's skipped steps, with a static scale):
 * pp_scale_guard = pp_scale, and bad[0] |= 1 when a scaled gradient is not finite; the *_guard optimizer steps return without
 * touching p / m / v when skip[0] != 0 and count the skipped update in skip[1].  skip / bad: two device ints, reset by the caller. */
int pp_scale_guard(float* p, long long n, float value, int* bad, void* stream);
int pp_adam_step_guard(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int step, int* skip, void* stream);
int pp_sgd_momentum_step_guard(float* p, const float* g, float* momentum_buf, long long n, float lr, float momentum,
                               float weight_decay, int step, int* skip, void* stream);
/* The forms FusedAdam / FusedSGD call (round 5; torch.optim.Adam / SGD `step()`, train_chaos.py:315): the segment's step count is a
 * DEVICE int (step_dev[0] = updates applied so far: the bias corrections use step_dev[0] + 1, and a one-thread commit kernel
 * advances it after the update -- unless skip[0] != 0, in which case nothing is touched and, if count_skip != 0, skip[1] += 1:
 * pass count_skip = 1 for ONE segment per optimizer step).  lr_dev (nullable): the learning rate as a device scalar instead of
 * `lr`.  With both on the device a captured hipGraph of the step replays correctly (no host value baked into the launch). */
int pp_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float lr, const float* lr_dev, float beta1,
                     float beta2, float eps, float weight_decay, int* step_dev, int* skip, int count_skip, void* stream);
int pp_sgd_momentum_step_dev(float* p, const float* g, float* momentum_buf, long long n, float lr, const float* lr_dev,
                             float momentum, float weight_decay, int* step_dev, int* skip, int count_skip, void* stream);

/* ---- diagnostics -------------------------------------------------------------------------------------------- */
/* bare v_mfma_f32_32x32x2_f32 loop: the fp32 matrix rate this device sustains at its clock under load */
int pp_mfma_probe(float* out, int blocks, int iters, double* flops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PACINGPSEUDO_HIP_H */
