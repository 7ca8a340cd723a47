/* confignet_hip.h -- C ABI of libconfignet_hip.so (MI355X / gfx950 only).
 *
 * The reference (microsoft/ConfigNet) has no FFI boundary: its per-layer arithmetic is
 * reached through tensorflow.keras calls (tensorflow-gpu==2.1.0).  Each entry point below
 * replaces the TF op(s) issued by the cited reference line(s); paths are relative to the
 * reference root.  All pointers are DEVICE pointers owned by the caller, channels-last
 * (N, [D,] H, W, C); kernels use the Keras layouts (kd,kh,kw,cin,cout) / Dense (in,out).
 * Storage types: `float*` arguments are fp32.  ACTIVATION tensors that may be stored in bf16 are passed as
 * `void*` together with a dtype code `dt` (CN_F32 / CN_BF16) that applies to every `void*` tensor of that
 * call; statistics, (n,c) coefficients, loss sums, biases, master weights, weight gradients and optimizer
 * state are fp32 in either mode, and all arithmetic accumulates in fp32.
 * Every call enqueues on `stream` (a hipStream_t passed as void*) and returns 0 on success
 * or a negative CN_E* code; cn_last_error_string() describes the last failure of the
 * calling thread.  No exceptions cross the ABI.
 * Process-wide state (everything else is per call): the profiling event pool (cn_prof_*); the deterministic-mode flag and
 * its per-stream workspaces (cn_set_deterministic); and the tuning overrides of cn_conv_tune / cn_conv_loop_select (sweeps and
 * tests: forced tile / split-K / filter-gradient workgroup target / loop variant; default = the built-in rules).  The library reads
 * NO environment variables (round 6: the A/B switches of earlier rounds are decided and gone).  None of the overrides changes WHAT
 * a call computes, only which kernel variant computes it.
 */
#ifndef CONFIGNET_HIP_H
#define CONFIGNET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CN_OK 0
#define CN_EINVAL (-1)   /* bad argument / unsupported geometry */
#define CN_EHIP (-2)     /* HIP runtime error */
#define CN_EUNSUPPORTED (-3) /* geometry outside the envelope of a specialised entry point: nothing was launched */

/* storage type of activation tensors (the `dt` arguments) */
#define CN_F32 0
#define CN_BF16 1        /* bfloat16 bits, round-to-nearest-even on store */

/* activation codes for fused epilogues */
#define CN_ACT_NONE 0
#define CN_ACT_LRELU 1   /* keras LeakyReLU(alpha) / tf.nn.leaky_relu */
#define CN_ACT_RELU 2
#define CN_ACT_TANH 3

/* Geometry of one N-d convolution read as an implicit GEMM.  2-D uses nd=2, *_d = 1 (k_d=1,
 * s_d=1, dl_d=1, p_d=0).  Output position o, tap k reads v = o*s - p + k of the (virtually
 * x2-upsampled when up=1) input; it is valid iff 0 <= v, v % dl == 0, v/dl < (in << up);
 * the stored element is (v/dl) >> up.  dl > 1 expresses the data-gradient of a strided
 * convolution; up=1 folds keras UpSampling2D/3D (hologan_generator.py:139,143,160-170)
 * into the gather. */
typedef struct CnConvGeom {
    int nd, n;
    int in_d, in_h, in_w, cin;      /* stored input extent */
    int out_d, out_h, out_w, cout;
    int k_d, k_h, k_w;
    int s_d, s_h, s_w;
    int dl_d, dl_h, dl_w;
    int p_d, p_h, p_w;              /* low-side padding ([TF-2.1] SAME: total//2) */
    int up;
} CnConvGeom;

const char* cn_last_error_string(void);
int cn_version(void);

/* ---- convolution family (implicit GEMM on v_mfma_f32_32x32x2_f32) -------------------------
 * Replaces keras.layers.Conv3D/Conv2D forward (+bias +activation):
 *   building_blocks.py:29,65,91 ; hologan_generator.py:50-56,101 ; hologan_discriminator.py:20,77 ;
 *   keras.applications VGG19/VGG16/ResNet50 convs (perceptual_loss.py:19,35 ; real_encoder.py:13).
 * Which kernel runs is the library's choice by geometry alone (first-layer / thin-output / image-gradient kernels, the LDS-DMA
 * loop of fwd2.hip for every vectorisable layer, the k-major implicit-GEMM loop for the rest) -- the results are the same
 * convolution.  The library reads no environment variables; cn_conv_tune / cn_conv_loop_select force a route for sweeps and tests. */
int cn_conv_fwd(const CnConvGeom* g, const float* x, const float* w, const float* bias,
                float* y, int act, float slope, void* stream);
/* cn_conv_fwd that also returns the per-(sample, channel) sums the FOLLOWING normalisation layer needs, taken in the convolution's
 * epilogue instead of a separate pass over y (reference: Conv2dAdaIn / Conv3dAdaIn -> AdaIn, building_blocks.py:37-44; DiscrBlock ->
 * get_layer_style + InstanceNormalization, building_blocks.py:97-106).  stats (caller-zeroed, fp32): mode 1 = [2][n][cout]: sum a,
 * sum a^2 of the stored value a = act(conv + bias); mode 2 (act must be CN_ACT_NONE) = [4][n][cout]: sum v, sum v^2, sum l, sum l^2
 * with l = leaky_relu(v, stats_slope).  Sums are added with fp32 atomics (run-dependent last bits; refused in deterministic mode).
 * Returns CN_EUNSUPPORTED WITHOUT launching anything unless the launch can carry them (the unsplit LDS-DMA loop with every tile
 * inside one sample): the caller then uses cn_conv_fwd + cn_nc_reduce / cn_nc_reduce4. */
int cn_conv_fwd_stats(const CnConvGeom* g, const float* x, const float* w, const float* bias, float* y, int act, float slope,
                      float* stats, int stats_mode, float stats_slope, void* stream);
/* y = act(conv(x, w) + bias + res), res of y's shape: the residual add + ReLU of a ResNet-50 block
 * (real_encoder.py:13, keras.applications ResNet50 `Add` + `Activation`) in the epilogue of the block's last
 * convolution.  Only launches without a K split carry it: CN_EUNSUPPORTED (nothing launched) otherwise. */
int cn_conv_fwd_res(const CnConvGeom* g, const float* x, const float* w, const float* bias, const float* res, float* y,
                    int act, float slope, void* stream);
/* w_tflip[T-1-t][co][ci] = w[t][ci][co]: the operand of the data-gradient GEMM. */
int cn_conv_weight_tflip(const float* w, float* w_tflip, int taps, int cin, int cout, void* stream);
/* Data gradient of cn_conv_fwd(g) w.r.t. its (virtually upsampled) input: gu has extent
 * (in << up).  Replaces the Conv*DBackpropInput ops tf.GradientTape issues
 * (confignet_first_stage.py:473,485,557 ; losses.py:76). */
int cn_conv_dgrad(const CnConvGeom* g, const float* gy, const float* w_tflip, float* gu, void* stream);
/* The same from the ORIGINAL filter w [taps][cin][cout] (round 3): the filter tile is transposed on its way into LDS instead
 * of by a cn_conv_weight_tflip launch per trainable filter per step.  CN_EUNSUPPORTED (nothing launched) for shapes that do
 * not reach the vectorised implicit-GEMM kernel (thin outputs, cout % 16 != 0): use cn_conv_dgrad there. */
int cn_conv_dgrad_w(const CnConvGeom* g, const float* gy, const float* w, float* gu, void* stream);
/* cn_conv_dgrad_w + res (res shaped like gu) in the launch's epilogue: the gradient of a tensor that feeds a convolution and a
 * skip connection (keras ResNet50's Add, real_encoder.py:13).  Stride-1, unsplit implicit-GEMM launches; else CN_EUNSUPPORTED. */
int cn_conv_dgrad_w_res(const CnConvGeom* g, const float* gy, const float* w, const float* res, float* gu, void* stream);
/* Filter gradient (Conv*DBackpropFilter): gw[(t,ci),co] (+)= sum_m x[src(m,t),ci]*gy[m,co]; gw is overwritten,
 * or accumulated into when `accumulate` != 0 (the caller guarantees its previous content, e.g. zeros). */
int cn_conv_wgrad(const CnConvGeom* g, const float* x, const float* gy, float* gw, int accumulate, void* stream);
/* The same filter gradient with a CALLER-OWNED workspace of cn_conv_wgrad_workspace_bytes(g) bytes (0: none needed, NULL
 * allowed): operand tiles by LDS-DMA four stages deep, row splits written as partial filters into the workspace and added
 * by one ordered reduction that also folds the accumulate -- no atomics on the tile, bit-reproducible in every mode
 * (csrc/wgrad2.hip).  Geometries it does not take fall through to cn_conv_wgrad. */
size_t cn_conv_wgrad_workspace_bytes(const CnConvGeom* g);
int cn_conv_wgrad_ws(const CnConvGeom* g, const float* x, const float* gy, float* gw, int accumulate, void* workspace,
                     size_t workspace_bytes, void* stream);
/* cn_conv_wgrad_ws that LEAVES THE SLABS to the caller (round 6): *parts = 0: gw is complete (no row splits, or the fall-back
 * kernel); *parts >= 2: the workspace holds that many partial filters of taps*cin*cout floats each, gw has not been touched, and
 * the caller adds them in index order (accumulate as passed) -- with cn_sum_parts_grouped, ONE launch for all the filter gradients
 * of a backward pass instead of one reduction launch per layer.  The sum is the same bits as cn_conv_wgrad_ws's. */
int cn_conv_wgrad_ws_slabs(const CnConvGeom* g, const float* x, const float* gy, float* gw, int accumulate, void* workspace,
                           size_t workspace_bytes, int* parts, void* stream);
typedef struct CnSumJob {
    const float* src;     /* parts x count floats */
    float* dst;           /* count floats */
    long long count;
    int parts;
    int accumulate;       /* bit 0: dst += sum (else dst = sum); bit 1: always the serial order over the parts (the order of
                             cn_sum_rows_into; default: cn_sum_parts' order, which interleaves 16 sub-sums from 64 parts up) */
} CnSumJob;
/* dst[j][i] (+)= sum_{p < parts[j]} src[j][p * count[j] + i] for every job, parts in index order; `jobs` is a HOST array (its
 * contents travel in the kernel arguments: nothing is read from it after the call returns).  One launch per 80 jobs. */
int cn_sum_parts_grouped(const CnSumJob* jobs, int njobs, void* stream);
/* Many shallow weight-gradient products in one launch (round 6): c_j (m x n) += a_j^T b_j with a_j (k x m), b_j (k x n) row-major
 * and k <= 32 (the batch): keras Dense weight gradients of a backward pass (building_blocks.py:152-173 MLPSimple, the AdaIN MLPs of
 * hologan_generator.py:119-124).  Per job the arithmetic of cn_gemm_acc(ta = 1); `jobs` is a HOST array. */
typedef struct CnDepthJob {
    const float* a;
    const float* b;
    float* c;
    int m, n, k, lda, ldb, ldc;
} CnDepthJob;
int cn_gemm_depth_grouped(const CnDepthJob* jobs, int njobs, void* stream);
/* Many row-skinny dense layers in one launch (round 6): c_j (m x n) = epi(a_j op(b_j) + bias_j), m <= 32 rows (the batch), a_j
 * (m x k) row-major, b_j (k x n), or (n x k) when tb.  epi: the activation `act`; or, with mask (m x n, leading dimension ldc),
 * the product with act'(mask) -- the LeakyReLU derivative of a hidden layer taken from its stored output --; or, with
 * accumulate, an atomic add into c_j (no bias / activation).  The six AdaIN MLPs of a generator pass, layer by layer, forward
 * and data-gradient side (hologan_generator.py:119-124, building_blocks.py:152-173).  `jobs` is a HOST array. */
typedef struct CnRowsJob {
    const float* a;
    const float* b;
    float* c;
    const float* bias;
    const float* mask;
    int m, n, k, lda, ldb, ldc, tb, act, accumulate;
    float slope;
} CnRowsJob;
int cn_gemm_rows_grouped(const CnRowsJob* jobs, int njobs, void* stream);
/* The GAN losses of every head of one discriminator call in one launch (round 6; losses.py:7-11, 20-47: six heads per call).
 * backward = 0: out[0] = mean(label softplus(-s) + (1 - label) softplus(s)) over the n scores s.  backward = 1: out[i] = the
 * gradient w.r.t. s[i] for the scalar's cotangent gout[0] (gout NULL: zero).  `jobs` is a HOST array. */
typedef struct CnGanJob {
    const float* s;
    float* out;
    const float* gout;
    int n;
    float label;
} CnGanJob;
int cn_gan_loss_grouped(const CnGanJob* jobs, int njobs, int backward, void* stream);
/* 3x3 stride-1 SAME 2-D convolution as Winograd F(2x2, 3x3) on the fp32 matrix cores: 16 GEMMs in the transform domain,
 * 4/9 of the multiply-adds of the direct form (exact in real arithmetic).  The same reference lines as cn_conv_fwd /
 * cn_conv_dgrad for the layers it fits (keras.applications VGG19/VGG16 and the 3x3 convolutions of ResNet50:
 * perceptual_loss.py:19,35 ; real_encoder.py:13).  u = cn_conv_wino_filter(w): [16][cin][cout] (dgrad = 0), or the filter
 * of the data-gradient convolution [16][cout][cin] (dgrad = 1: flipped taps, channels swapped), made once per weight update.
 * cn_conv_fwd_wino: x (n,h,w,cin) -> y (n,h,w,cout) with fused bias + activation; CN_EUNSUPPORTED (nothing launched)
 * unless cin % 8 == 0 and cout % 64 == 0. */
/* First / last layers of the bf16 path with mixed storage types (3-channel images stay fp32): the 3x3 convolution of an fp32
 * image written in bf16 (x_dt = CN_F32, y_dt = CN_BF16) and the data gradient of the stride-2 one into the fp32 image from a
 * bf16 output gradient (cn_conv_dgrad_dt with gy_dt = CN_BF16, gu_dt = CN_F32) without a conversion pass.  Same geometry /
 * filter conventions as cn_conv_fwd / cn_conv_dgrad; any other combination returns CN_EUNSUPPORTED without launching. */
int cn_conv_fwd_dt(const CnConvGeom* g, const void* x, int x_dt, const float* w, const float* bias, void* y, int y_dt, int act,
                   float slope, void* stream);
int cn_conv_dgrad_dt(const CnConvGeom* g, const void* gy, int gy_dt, const float* w_tflip, void* gu, int gu_dt, void* stream);
/* Filter gradient of a 3x3 convolution of a 3-channel fp32 image (from-RGB block of the discriminators, building_blocks.py:91;
 * VGG conv1_1): stride 1 or 2, cout <= 64, gy in fp32 or bf16 (gy_dt).  No atomics: scratch holds
 * cn_conv_wgrad_c3_partials() partial filters of 27 * cout floats.  Returns CN_EUNSUPPORTED without launching otherwise. */
int cn_conv_wgrad_c3_partials(void);
int cn_conv_wgrad_c3(const CnConvGeom* g, const float* x, const void* gy, int gy_dt, float* scratch, float* gw, int accumulate,
                     void* stream);
/* Filter gradient of a 2-D stride-1 convolution with <= 4 output channels and 8 <= cin <= 64 (map_final, hologan_generator.py:101,
 * behind the folded x2 upsample): LDS-staged tiles, VALU accumulation, per-workgroup partial filters in `scratch`
 * (cn_conv_wgrad_thin_partials() * taps * cin * cout floats) added in order -- no atomics.  CN_EUNSUPPORTED otherwise. */
int cn_conv_wgrad_thin_partials(void);
int cn_conv_wgrad_thin(const CnConvGeom* g, const float* x, const float* gy, float* scratch, float* gw, int accumulate, void* stream);
int cn_conv_wino_filter(const float* w, float* u, int cin, int cout, int dgrad, void* stream);
int cn_conv_fwd_wino(int n, int h, int w, int cin, int cout, const float* x, const float* u, const float* bias, float* y,
                     int act, float slope, void* stream);
/* The same convolutions as Winograd F(4x4, 3x3) (csrc/winograd4.hip): 36 GEMMs per 4x4 output tile, 2.25 MFMA products per
 * output pixel and (ci, co) instead of 4.  u from cn_conv_wino4_filter ([36][cin][cout]; dgrad != 0: the flipped,
 * channel-swapped filter of the data gradient).  cn_conv_fwd_wino4 returns CN_EUNSUPPORTED (nothing launched) unless
 * h % 16 == 0, w % 32 == 0, cin % 16 == 0 and cout % 64 == 0. */
int cn_conv_wino4_filter(const float* w, float* u, int cin, int cout, int dgrad, void* stream);
int cn_conv_fwd_wino4(int n, int h, int w, int cin, int cout, const float* x, const float* u, const float* bias,
                      float* y, int act, float slope, void* stream);
/* Tuning hook: force the tile configuration (0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 128x32, 4 = 128x96; -1 = heuristic), the
 * split-K factor of cn_conv_fwd / cn_conv_dgrad (0 = heuristic) and the workgroup target of cn_conv_wgrad (0 = default). */
int cn_conv_tune(int cfg, int splits, long wg_blocks);
/* Tuning hook of the forward / data-gradient main loop (csrc/fwd2.hip, the LDS-DMA loop): loop = 1 / 0 forces it on / off for the
 * layers it can take, -1 = the built-in choice; kb = 16 / 32 its stage depth (0 = default), ns = 3 / 4 its stage count (0 = default),
 * np = 0 / 1 / 2 its loader waves (waves that only issue the LDS-DMA; -1 = default). */
int cn_conv_loop_select(int loop, int kb, int ns, int np);
/* Backward of the folded nearest x2 upsample: out[n,p,c] = sum of the 2^nd children of p. */
int cn_sumpool2(const void* gu, void* gx, int nd, int n, int d, int h, int w, int c, int dt, void* stream);

/* ---- bf16 convolution family (implicit GEMM on v_mfma_f32_32x32x16_bf16, fp32 accumulate) -- the compute path of
 * BASELINE.json configs[2] (bf16 compute, fp32 master weights).  Same geometry and reference lines as the fp32 family.
 * Activations x / y / gy / gu are bf16; bias and the filter gradient are fp32.  Filters are bf16 copies prepared once per
 * weight update by cn_conv_weight_prep_bf16 from the fp32 master filter w[t][ci][co]:
 *     wf[t][co][ci]  (forward operand: reduction index ci contiguous)
 *     wd[t][ci][co]  (data-gradient operand: reduction index co contiguous; the tap flip is index arithmetic)
 * Requirements: cin % 8 == 0 and cout % 8 == 0 (16-byte operand pieces); other shapes return CN_EUNSUPPORTED without
 * launching (the caller converts and uses the fp32 family: 3-channel image layers). */
int cn_conv_weight_prep_bf16(const float* w, uint16_t* wf, uint16_t* wd, int taps, int cin, int cout, void* stream);
int cn_conv_fwd_bf16(const CnConvGeom* g, const uint16_t* x, const uint16_t* wf, const float* bias, uint16_t* y,
                     int act, float slope, void* stream);
int cn_conv_dgrad_bf16(const CnConvGeom* g, const uint16_t* gy, const uint16_t* wd, uint16_t* gu, void* stream);
int cn_conv_wgrad_bf16(const CnConvGeom* g, const uint16_t* x, const uint16_t* gy, float* gw, int accumulate, void* stream);
/* ---- convolution after a x2 nearest upsample, collapsed per output-parity class (exact; hologan_generator.py:139-170 +
 * building_blocks.py:29,65).  Along one axis output o = 2 s + tau reads stored row s through Wc(tau) = the sum of the taps kk
 * with floor((o - pad_lo + kk) / 2) = s: k2 = 4 (k = 3) or 5 (k = 4) values of tau, 2-3 of them live per output parity.  With
 * "conv2" = the stride-2 convolution (kernel k2, low padding pad2) from the output grid to the stored grid:
 *   forward       = cn_conv_fwd of the geometry {in = stored, out = 2 in, k = k2, s = 1, dl = 2, p = k2-1-pad2, up = 0} with wf,
 *   data gradient = cn_conv_fwd of conv2 {in = 2 in (cout channels), out = in (cin channels), k = k2, s = 2, p = pad2} with wd,
 *   filter grad.  = cn_conv_wgrad of conv2 (x := gy, gy := x) -> gw2[a][co][ci], then cn_upfold_wgrad -> gw[kk][ci][co].
 * cn_upfold_weights: w[kk][ci][co] (k3 = {k_d,k_h,k_w}, k_d = 1 for nd = 2; pad_lo3 = SAME low paddings) ->
 *   wf[a'][ci][co] and/or wd[a][co][ci]; returns k2 and pad2 per axis in k2_out3 / pad2_out3 (host arrays, may be NULL). */
int cn_upfold_weights(const float* w, float* wf, float* wd, int nd, const int* k3, const int* pad_lo3, int cin, int cout,
                      int* k2_out3, int* pad2_out3, void* stream);
int cn_upfold_wgrad(const float* gw2, float* gw, int nd, const int* k3, const int* pad_lo3, int cin, int cout,
                    int accumulate, void* stream);      /* accumulate: 0 store, 1 atomic add, 2 plain add (single writer) */
/* dst = (dst_dt) src : storage-type conversion between fp32 and bf16 tensors */
int cn_cast(const void* src, int src_dt, void* dst, int dst_dt, size_t numel, void* stream);

/* ---- dense GEMM: C = act(op(A) op(B) + bias) (keras Dense: building_blocks.py:152-173,
 * hologan_discriminator.py:34,46,97 ; real_encoder.py:16,18 ; latent_gan.py:88-109) ---------*/
int cn_gemm(int trans_a, int trans_b, int m, int n, int k, const float* a, int lda,
            const float* b, int ldb, float* c, int ldc, const float* bias, int act, float slope,
            void* stream);

/* Deterministic mode (process-wide; switch between steps, not inside a captured graph): with on != 0 every reduction of the
 * fp32 path runs in a fixed order -- statistics passes, split-K of cn_conv_fwd / cn_gemm, filter gradients and loss reductions
 * through per-split partials that a second launch adds in index order, the rotation's scatter by one wave per (sample, 8
 * channels) in voxel order -- so two runs on the same inputs give bit-identical results (at a cost: see DESIGN.md).  Allocates 16 per-stream
 * workspaces of 64 MiB on first use; a workspace handed out while its stream was being captured stays bound to that stream
 * (a replayed graph holds the pointer) until cn_det_release_stream(stream), and a launch that finds every workspace bound that
 * way fails with CN_E* instead of sharing scratch.  The bf16 family is not covered. */
int cn_set_deterministic(int on);
int cn_get_deterministic(void);
/* The caller has destroyed every graph captured on `stream` (or the stream itself): its workspace may be re-bound. */
int cn_det_release_stream(void* stream);
/* Clears `bytes` (a multiple of 4) at p with a kernel launch (a HIP-graph node that re-executes on replay). */
int cn_zero(void* p, size_t bytes, void* stream);
/* C += op(A) op(B) (fp32 atomics): a Dense layer's weight gradient added straight into its slot of the network's gradient
 * arena (tf.GradientTape sums the contributions of every use of a variable; here the kernels do, not a separate add). */
int cn_gemm_acc(int trans_a, int trans_b, int m, int n, int k, const float* a, int lda, const float* b, int ldb,
                float* c, int ldc, void* stream);
/* dst[c] (+)= sum_r src[r][c]: adds the partial rows of a per-channel reduction (bias gradients) in row order. */
int cn_sum_rows_into(const float* src, float* dst, int rows, int cols, int accumulate, void* stream);

/* ---- per-(sample,channel) statistics and affine maps on (N, S, C) tensors ------------------
 * Building blocks of LayerNormalization/AdaIn (building_blocks.py:132-144), InstanceNormalization
 * (instance_normalization.py:108-131), get_layer_style (confignet_utils.py:147-159), BN inference
 * (keras ResNet50), global average pooling and every bias gradient.
 *   a = f1(x1), b = x2 ? f2(x2) : a ;  sum1[n,c] = sum_s a ; sum2[n,c] = sum_s a*b
 *   flags: bit0 leaky-relu(slope) on x1, bit1 leaky-relu(slope) on x2, bit4: the outputs are already zero
 *   (skip the clearing launch); bits 8..: x2 holds only (flags >> 8) samples and sample n reads x2[n % period] (the batched
 *   R1 input-gradient pass stacks the cotangents of several heads along n against ONE copy of the activations).
 *   Outputs are overwritten. */
/* The four statistics of a DiscrBlock's tail (building_blocks.py:97-106) in ONE pass over its pre-activation tensor:
 * out (4, n, c) = sum x, sum x^2 (get_layer_style of the pre-activation), sum l, sum l^2 with l = leaky_relu(x, slope) (the
 * InstanceNormalization behind the activation); c % 4 == 0; flags bit4: `out` is already zero. */
int cn_nc_reduce4(const void* x, float* out, int n, int s, int c, float slope, int flags, int dt, void* stream);
int cn_nc_reduce(const void* x1, const void* x2, float* sum1, float* sum2, int n, int s, int c,
                 int flags, float slope, int dt, void* stream);
/* a = x1 * act'(f2(x2)) (the activation derivative taken from the sign / value of x2 as in cn_act_bwd), written to dact_out,
 * with sum1[n,c] = sum_s a and (if sum2 != NULL) sum2[n,c] = sum_s a*f2(x2) in the same pass: the tangent of LeakyReLU and its
 * two statistics (cn_dual_tail_*: ta, T1, T2) without a separate cn_act_bwd pass.  flags as cn_nc_reduce (bit1, bit4, period).
 * dact_out may be NULL (round 6): the two sums only, a is not stored. */
int cn_nc_reduce_dact(const void* x1, const void* x2, float* sum1, float* sum2, void* dact_out, int n, int s, int c,
                      int flags, float slope, int act, int dt, void* stream);
/* Backward of conv -> BatchNorm (inference mode: a[c] x + shift[c] (+ residual)) -> activation in one pass (keras ResNet50 blocks
 * of the real encoder, real_encoder.py:13-20): g = gy * act'(y), gx = a[c] * g, g itself to g_out if the residual branch needs it
 * (else NULL), sum_g[r][c] = sum g, sum_gx[r][c] = sum g * x over n partial rows of s rows each. */
int cn_bn_act_bwd(const void* gy, const void* y, const void* x, const float* a, void* g_out, void* gx, float* sum_g,
                  float* sum_gx, int n, int s, int c, int act, int flags, int dt, void* stream);
/* y = A1*f1(x1) + A2*f2(x2) + B, coefficient tensors indexed [n*cstride + c] (cstride = c, or 0 for
 * per-channel coefficients).  flags: bit0/bit1 as above, bit2: multiply the result by lrelu'(x2),
 * bit3: relu on the result, bit4 (round 6): f1(x1) is multiplied by lrelu'(x2) before its coefficient -- a tangent through the
 * activation that is never stored (the R1 tail's ta = lrelu'(x) tx) --, bits 8..: x2 sample period as in cn_nc_reduce.  x1, x2 and B are each optional (NULL); a NULL coefficient of a present
 * x means 1.  If a3 != NULL, a3*x2 + b3 (raw x2) is added after the bit2 mask -- the style-statistics
 * gradient that joins the instance-norm gradient in DiscrBlock (building_blocks.py:100-106). */
int cn_nc_lin2(const void* x1, const float* a1, const void* x2, const float* a2, const float* bb,
               const float* a3, const float* b3, void* y, int n, int s, int c, int cstride, int flags,
               float slope, int dt, void* stream);

/* Per-(n,c) coefficient kernels of the normalisation layers (tiny: N*C threads).
 * mode 0 AdaIN (building_blocks.py:132-144): p1 = [s|b] (N,2C); mu = s1/S, var = s2/S - mu^2,
 *        r = rsqrt(var+eps); A = r*(s+1), B = b - mu*A.
 * mode 1 InstanceNormalization (instance_normalization.py:117-130): p1 = gamma (C), p2 = beta (C);
 *        q = 1/(sqrt(var)+eps); A = gamma*q, B = beta - mu*A.
 * mode 2 get_layer_style (confignet_utils.py:147-159): A is (N,2C) = [mu | sqrt(var+eps)], B unused.
 * save_mean / save_r (N,C) keep mu and r|q|std for the backward pass. */
int cn_norm_coef_fwd(int mode, const float* s1, const float* s2, const float* p1, const float* p2, float* A,
                     float* B, float* save_mean, float* save_r, int n, int c, int S, float eps, void* stream);
/* Backward coefficients: the input gradient is gx = c1*gy + c2*x + c0 (x = the normalised tensor, after
 * LeakyReLU for mode 1).  t1 = sum_s gy, t2 = sum_s gy*x.  mode 0: gp1 = d[s|b] (N,2C).  mode 1: gp1 = dgamma,
 * gp2 = dbeta (C), reduced over n.  mode 2: t1 = d[mu|std] (N,2C), t2 unused; c1 unused, gx = c2*x + c0. */
int cn_norm_coef_bwd(int mode, const float* t1, const float* t2, const float* save_mean, const float* save_r,
                     const float* p1, float* c1, float* c2, float* c0, float* gp1, float* gp2, int n, int c,
                     int S, float eps, void* stream);
/* AdaIn (mode 0) / instance norm (mode 1) apply pass (dir 0) or backward map (dir 1) with the coefficient algebra of
 * cn_norm_coef_fwd / cn_norm_coef_bwd INLINE (round 6): reduce + this instead of reduce + coefficients + cn_nc_lin2
 * (building_blocks.py:37-44, 97-106, 132-149; instance_normalization.py:108-131).  dir 0: y = k1 f1(x1) + kb from sa = sum,
 * sb = sum of squares, p1 / p2 as cn_norm_coef_fwd; writes save_mean / save_r.  dir 1: y = the input gradient from x1 = gy,
 * x2 = x, sa = sum gy, sb = sum gy f2(x); reads save_mean / save_r, writes gp1 = d[s|b] (mode 0) or gp1 / gp2 = d gamma / d beta
 * (mode 1); a3 / b3 (n, c): an additional a3 x + b3.  flags as cn_nc_lin2.  CN_EUNSUPPORTED (nothing launched) unless
 * c % 4 == 0 and the tensor is large enough for the per-sample grid -- then the three-launch form. */
int cn_norm_apply(int mode, int dir, const void* x1, const void* x2, const float* sa, const float* sb, const float* p1,
                  const float* p2, float* save_mean, float* save_r, float* gp1, float* gp2, const float* a3, const float* b3,
                  void* y, int n, int s, int c, float eps, int flags, float slope, int dt, void* stream);

/* Tangent ("dual") DiscrBlock tail for the R1 penalty (losses.py:75-82) without a second-order tape: the
 * penalty's weight gradient is 2 d/dtheta JVP_x(out)(v) at v = d out/d x held constant, i.e. a first-order
 * backward through a tangent forward pass.  T1 = sum ta, T2 = sum ta*a (ta = lrelu'(x) tx, a = lrelu(x)),
 * U1 = sum tx, U2 = sum tx*x; mean/q and sm/ssd are the primal instance-norm / style statistics.
 * fwd: ty = C1*ta + C2*a + C0 and tstyle (N,2C).  bwd (H1 = sum h, H2p = sum h*a, E = sum h*ta, u = d tstyle):
 *   out13 = {K1,K2,K0,D2,D0, kh,kt,ka,kc, et,ex,e0, dgamma, dgamma_rows (scratch, one float per instance-norm row and channel)}: g_tx = lrelu'(x)(K1 h + K2 a + K0) + D2 x + D0,
 *   g_x = lrelu'(x)(kh h + kt ta + ka a + kc) + et tx + ex x + e0 (cn_dual_tail_gx).  Either half optional.
 * Batched tangent pass (the six heads' tangents stacked along the sample axis against ONE copy of the primal activations,
 * round 3): the coefficient kernels take `n` stacked samples of which rows [0, n_style) are the head that LEAVES through this
 * block's style statistics (U1/U2/u/tstyle/D2/D0/et/ex/e0: n_style rows) and rows [n_style, n) the heads that go on through
 * LeakyReLU + instance norm (T1/T2/H1/H2p/E and C1/C2/C0/K1/K2/K0/kh/kt/ka/kc: n - n_style rows, indexed from 0); the primal
 * statistics mean/q/sm/ssd hold `period` samples and a row reads sample (row % period).  One head at a time: n_style = 0 or n,
 * period = n.  cn_dual_tail_gx: x / out / tx and et/ex/e0 hold n samples, h / ta and kh/kt/ka/kc hold nrep*n (head-major);
 * the heads' second-order terms are summed into out. */
int cn_dual_tail_coef_fwd(const float* T1, const float* T2, const float* U1, const float* U2, const float* mean,
                          const float* q, const float* sm, const float* ssd, const float* gamma, float* C1,
                          float* C2, float* C0, float* tstyle, int n, int c, int S, float eps, int n_style, int period,
                          void* stream);
int cn_dual_tail_coef_bwd(const float* H1, const float* H2p, const float* E, const float* u, const float* T1,
                          const float* T2, const float* U1, const float* U2, const float* mean, const float* q,
                          const float* sm, const float* ssd, const float* gamma, float* const* out13, int n, int c,
                          int S, float eps, int n_style, int period, void* stream);
int cn_dual_tail_gx(const void* h, const void* ta, const void* tx, const void* x, const float* kh,
                    const float* kt, const float* ka, const float* kc, const float* et, const float* ex,
                    const float* e0, void* out, int n, int s, int c, float slope, int nrep, int dt, void* stream);
/* cn_dual_tail_gx + the gradient w.r.t. the stacked tangent input in the same pass (round 6): out_tx holds (1 + nrep) n samples,
 * the style head's rows D2 x + D0 first, then per head lrelu'(x) (K1 h + K2 lrelu(x) + K0).  ta_is_tx: `ta` holds the heads'
 * tangent INPUT rows; ta = lrelu'(x) tx is formed in the pass. */
int cn_dual_tail_gx_tx(const void* h, const void* ta, const void* tx, const void* x, const float* kh, const float* kt,
                       const float* ka, const float* kc, const float* et, const float* ex, const float* e0,
                       const float* K1, const float* K2, const float* K0, const float* D2, const float* D0, void* out,
                       void* out_tx, int n, int s, int c, float slope, int nrep, int ta_is_tx, int dt, void* stream);
/* (sum h, sum h lrelu(x), sum h ta) per (n, c) in one pass over the three tensors: out (3, n, c); x holds `period` samples;
 * flags & 16: out is zero already; flags & 32: `ta` is the tangent input tx (ta = lrelu'(x) tx formed in the pass).
 * The backward reductions of the DiscrBlock tail's tangent (losses.py:75-82 through building_blocks.py:100-106). */
int cn_nc_reduce_hxt(const void* h, const void* x, const void* ta, float* out, int n, int s, int c, float slope, int period,
                     int flags, int dt, void* stream);

/* ---- elementwise / small ops ------------------------------------------------------------------*/
int cn_act_fwd(const void* x, void* y, size_t numel, int act, float slope, int dt, void* stream);
/* gx = gy * act'(.) evaluated from the activation OUTPUT (lrelu/relu: sign of y; tanh: 1-y^2). */
int cn_act_bwd(const void* gy, const void* y, void* gx, size_t numel, int act, float slope, int dt, void* stream);
/* The same fused with the bias gradient that follows it in a convolution / dense backward (reference: the Keras layers'
 * bias_add gradient after the activation gradient, building_blocks.py:37-44,73-80): gx = gy * act'(y) over (n, s, c) and
 * gb[n][c] = sum_s gx in one pass.  flags: 16 = gb already cleared by the caller. */
int cn_act_bwd_bias(const void* gy, const void* y, void* gx, float* gb, int n, int s, int c, int act, float slope, int flags,
                    int dt, void* stream);
int cn_axpby(const void* x, const void* y, void* out, size_t numel, float a, float b, int dt, void* stream);
int cn_mul(const void* x, const void* y, void* out, size_t numel, int dt, void* stream);
/* out[0] += scale * sum (a-b)^2  (perceptual_loss.py:74-80); out must be initialised by the caller */
int cn_sqdiff_sum(const void* a, const void* b, float* out, size_t numel, float scale, int dt, void* stream);
/* out[n] = sum_row x^2 (R1: losses.py:77-79 ; eye loss: losses.py:16) */
int cn_row_sumsq(const float* x, float* out, int n, size_t row, void* stream);
/* out[n,:] = x[n,:] * s[n] * k */
int cn_row_scale(const void* x, const float* s, void* out, int n, size_t row, float k, int dt, void* stream);
/* out = (x - x2) * s[row] * k: the gradient of a squared-difference loss term (perceptual_loss.py:74-80) in one pass. */
int cn_row_scale_diff(const void* x, const void* x2, const float* s, void* out, int n, size_t row, float k, int dt, void* stream);
/* The backward of a tapped, activated layer of a feature loss in one pass (round 6; perceptual_loss.py:74-82, the VGG taps):
 * out = (g + (y - target) s[r] k) act'(y) over n rows of `row` elements; g (the gradient from the next layer) may be NULL. */
int cn_tap_bwd(const void* y, const void* target, const void* g, const float* s, void* out, int n, size_t row, float k,
               int act, float slope, int dt, void* stream);
/* Backward of ReLU -> MaxPooling2D(2, 2) in one pass (round 6; perceptual_loss.py:19-41, VGG conv1_2 / conv2_2 / conv3_4):
 * gx = (gy routed to each window's first maximum [+ (x - target) s[sample] k]) relu'(x), x = the ReLU output that was pooled;
 * s_rows = 1 (one scale) or n.  CN_EUNSUPPORTED (nothing launched) unless h, w even, c % 4 == 0, tensors 16-byte aligned. */
int cn_maxpool2_bwd_act(const void* x, const void* gy, const void* target, const float* s, float k, void* gx, int n, int h,
                        int w, int c, int s_rows, int dt, void* stream);
/* out = (a - b) * mask[n,h,w] broadcast over c (b optional) (losses.py:14) */
int cn_masked_diff(const float* a, const float* b, const uint8_t* mask, float* out, size_t pixels, int c, void* stream);
/* 2-D max pooling, zero padding (keras MaxPooling2D after ZeroPadding2D); bwd: the gradient of a window goes to its
 * first maximum in row-major window order (computed as a gather: no atomics) */
int cn_maxpool_fwd(const void* x, void* y, int n, int h, int w, int c, int k, int s, int pad, int dt, void* stream);
int cn_maxpool_bwd(const void* x, const void* gy, void* gx, int n, int h, int w, int c, int k, int s, int pad, int dt, void* stream);
/* AveragePooling2D((3,3), strides 1, padding "same") of keras.applications InceptionV3's pool branches (reference:
 * metrics/inception_distance.py:12, the FID/KID feature extractor); window cells outside the image are not counted [TF-2.1]. */
int cn_avgpool3_same(const void* x, void* y, int n, int h, int w, int c, int dt, void* stream);
/* y[...,j] = scale * x[...,perm[j]] + off[j] on 3-channel images: (x+1)*127.5 + "caffe"/VGGFace
 * preprocessing (perceptual_loss.py:52-61 ; real_encoder.py:24-25); bwd scatters back. */
int cn_chan_affine3_fwd(const float* x, float* y, size_t pixels, const int* perm, float scale, const float* off, void* stream);
int cn_chan_affine3_bwd(const float* gy, float* gx, size_t pixels, const int* perm, float scale, void* stream);
/* softplus means of losses.py:7-11 on (n) scores: out[0] = mean(l*softplus(-s) + (1-l)*softplus(s)),
 * label l constant; bwd: gs = gout * (-(l)*sigmoid(-s) + (1-l)*sigmoid(s)) / n */
int cn_gan_loss_fwd(const float* s, float* out, int n, float label, void* stream);
int cn_gan_loss_bwd(const float* s, const float* gout, float* gs, int n, float label, void* stream);

/* ---- 3-D rigid resample (confignet_utils.py:63-120): trilinear, clamp to edge ---------------*/
/* euler_angles_to_matrix (confignet_utils.py:122-145): rot (N,3,3) from angles (N,3), and the gradient to the angles from the
 * gradient to the matrices -- one launch each (the generator's rotation input, hologan_generator.py:147). */
int cn_euler_matrix(const float* angles, float* rot, int n, void* stream);
int cn_euler_matrix_bwd(const float* angles, const float* grot, float* gangles, int n, void* stream);
int cn_rotate3d_fwd(const float* grid, const float* rot /* (N,3,3) */, float* out, int n, int g, int c, void* stream);
/* ggrid (overwritten) and grot (N,3,3) (overwritten; through `diffs` only, l.105) */
int cn_rotate3d_bwd(const float* grid, const float* rot, const float* gout, float* ggrid, float* grot,
                    int n, int g, int c, void* stream);

/* ---- optimizer: Keras Adam + EMA in one pass (confignet_first_stage.py:393-400,601-602) -----
 * theta -= lr_t * m/(sqrt(v)+eps); lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the host (shared step
 * counter rule) and read from DEVICE memory (one float) so a captured HIP graph of the step stays valid
 * while t advances; if ema != NULL: ema = ema_alpha*ema + (1-ema_alpha)*theta_new. */
int cn_adam_step(float* theta, const float* grad, float* m, float* v, float* ema, size_t numel,
                 const float* lr_t, float beta1, float beta2, float eps, float ema_alpha, void* stream);
int cn_ema_step(float* ema, const float* theta, size_t numel, float alpha, void* stream);

/* ---- data path: batch assembly on device (confignet_first_stage.py:438-450 ;
 * confignet_utils.py:198-204): out[i] = flip?(pool[idx[i]]) / 127.5 - 1 ----------------------*/
int cn_gather_images_u8(const uint8_t* pool, const int64_t* idx, const uint8_t* flip, float* out,
                        int n, int h, int w, int c, void* stream);
/* uint8 image from generator output: clip(-1,1), (x+1)*127.5 (confignet_first_stage.py:636-637) */
int cn_to_uint8(const float* x, uint8_t* out, size_t numel, void* stream);
/* BatchNormalization in inference mode (keras ResNet50 called without `training=`, real_encoder.py:13) folded into the
 * preceding convolutions: dst (packed) = src[segment][k][c] * a[c] for every filter listed in `seg` -- nseg x 5 ints: source
 * offset, packed destination offset, element count, columns (cout), offset of the filter's coefficients in `a`; all multiples
 * of 4 floats -- in one launch over `total` destination floats. */
int cn_scale_columns_segments(const float* src, float* dst, const int* seg, const float* a, int nseg, size_t total, void* stream);
/* The adjoint of cn_scale_columns_segments together with the BatchNorm coefficient algebra (the taped ResNet-50 of
 * real_encoder.py:13 runs on the folded filters): from the folded filters' gradients gwf (packed) and the shifts' gradients
 * gshift, ADD d kernel, d bias, d gamma, d beta into gout (laid out like the weight arena `arena`).  a = gamma rs,
 * rs = rsqrt(var + eps), bm = bias - mean, concatenated per layer.  seg: nseg x 9 ints on the device -- the five of
 * cn_scale_columns_segments, the arena offsets of bias / gamma / beta, the segment's first workgroup; blocks = sum cout / 64. */
int cn_bn_fold_bwd(const int* seg, int nseg, int blocks, const float* gwf, const float* gshift, const float* arena,
                   const float* a, const float* rs, const float* bm, float* gout, void* stream);

/* ---- profiling of the dominant kernel class (implicit-GEMM convolutions) with HIP events
 * recorded on the launch stream (bench.py roofline object) -----------------------------------*/
int cn_prof_enable(int on);
int cn_prof_reset(void);
/* synchronises the recorded events; returns launches, summed kernel ms and algorithmic flops */
int cn_prof_collect(int* launches, double* total_ms, double* total_flops);
/* the same per kernel family (32 slots: 0-4 igemm_fwd tiles 128x128 / 128x64 / 64x64 / 128x32 / 128x96, 5-8 igemm_wgrad tiles
 * 128x128 / 128x96 / 64x64 / 128x32, 9 wino_fwd, 10 c3_fwd, 11 s2_image_dgrad, 12 thin / rgb, 13 c3_wgrad, 14 / 15 the bf16
 * forward / filter-gradient kernels) with the launches' ALGORITHMIC HBM bytes (every operand read once, the result written
 * once): the byte model the PMC FETCH_SIZE / WRITE_SIZE of those kernels are compared with. */
int cn_prof_collect_by_family(int* launches, double* ms, double* flops, double* bytes);

/* ---- stream calibration: one wave busy-waits `ticks` of the 100 MHz wall clock on `stream`.  Two such launches on
 * streams that share a hardware queue run back to back, on independent queues side by side: graphs.py uses that to
 * pick replay streams that really run concurrently (no reference counterpart; runtime plumbing) ---------------*/
int cn_spin(unsigned long long ticks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CONFIGNET_HIP_H */
