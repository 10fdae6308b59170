/* virtex_amd -- C ABI of the MI355X-native VirTex bicaptioning hot path.
 *
 * The reference (kdexd/virtex) is pure Python; its hot path dispatches to ATen operators
 * (SURVEY.md 2.4).  This header is the drop-in boundary underneath the Python modules of
 * `virtex_amd/` that mirror the reference's module contracts.  Each entry point names the
 * reference call site / ATen operator it replaces.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; all pointers are DEVICE pointers owned by the
 *    caller (kept alive by the caller until the stream work completes);
 *  - every function enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *    immediately: 0 on success, a negative vtx_status otherwise (never throws);
 *    `vtx_last_error()` returns a thread-local description of the last failure;
 *  - `dtype` selects the activation/operand storage type (VTX_F32 or VTX_BF16); statistics,
 *    parameters of norms, losses and all gradients of parameters are fp32;
 *  - activations of the visual backbone are NHWC; matrices are row-major;
 *  - thread-safe and re-entrant (autograd's backward thread calls the *_bwd functions).
 */
#ifndef VIRTEX_AMD_H
#define VIRTEX_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { VTX_F32 = 0, VTX_BF16 = 1 } vtx_dtype;
typedef enum {
    VTX_OK = 0,
    VTX_ERR_ARG = -1,    /* null pointer / negative size */
    VTX_ERR_SHAPE = -2,  /* unsupported shape / alignment */
    VTX_ERR_DTYPE = -3,
    VTX_ERR_LAUNCH = -4, /* HIP launch failure */
    VTX_ERR_WORKSPACE = -5
} vtx_status;

int vtx_version(void);
const char* vtx_backend(void);    /* "hip:gfx950" (product) or "hipemu" (CPU test build) */
const char* vtx_last_error(void); /* thread-local */
/* 2 (default): LDS-DMA (buffer descriptors) + transpose-read bf16 contraction kernel; 1: register-staged kernel (A/B tests) */
int vtx_set_contraction_generation(int gen);
int vtx_set_ablation(int bits);        /* measurement only (tools/ablate_gemm.py, tools/ablate_gen3.py) */
int vtx_set_debug_buffer(void* dev);   /* measurement builds (-DVTX_ABLATE) only: per-wave time stamps of the generation-3 kernels */
int vtx_set_tile_override(int cand);   /* tests: force a block tile; -1 = automatic */
/* Measurement switches of the specialised kernels, by name: "wgrad3x3" (0 off, 1 by image size, 2 always), "stem_stream",
 * "expand1x1" (0 / 1), "splitk_blocks" (block target of the split-K weight gradients, default 512), "gen3" (generation-3
 * contraction kernels, gemm_v3.h: 0 only when forced by tile override 20 / 21, n >= 2: taken when the cost model predicts
 * at least n % of the generation-2 class rate, default 80; VIRTEX_AMD_GEN3), "gen3_mc" (the same for the weight gradients,
 * gemm_v3mc.h: 0 forced only, n: taken when M N / (M + N) >= n, default 200; VIRTEX_AMD_GEN3_MC).  Defaults come from
 * VIRTEX_AMD_WGRAD3X3 / _STEM_STREAM / _EXPAND1X1 / _SPLITK_BLOCKS.  No reference counterpart. */
int vtx_set_switch(const char* name, int value);
/* Which contraction kernel this thread's last GEMM-shaped launch ran on: 2 = the DMA kernel (operands addressed through
 * buffer descriptors: needs bf16, operands < 2 GB, convolution channel counts that are multiples of the 32-deep K step),
 * 3 = the phase-interleaved 8-wave DMA kernel (gemm_v3.h: row-major operand pairs, 64-deep K tiles),
 * 1 = the register-staged kernel (fp32, and everything the DMA kernel does not take).  Tests use it to prove coverage. */
int vtx_last_contraction_generation(void);
/* Process-wide launch counts per generation since the last reset (any thread: backward runs on autograd's threads).
 * A bf16 training step of the supported models must not touch generation 1 -- tests assert it, so that a shape that
 * silently falls off the DMA kernel shows up as a failure, not as a slower step.  gen2 counts generations 2 and 3. */
int vtx_contraction_generation_counts(long* gen1, long* gen2, int reset);

/* ---- LayerNorm(x + dropout(y)) --------------------------------------------------------
 * Replaces aten::dropout + aten::add + aten::layer_norm of the post-norm decoder layer
 * (torch/nn/modules/transformer.py:1143-1199 via virtex/modules/textual_heads.py:181-194).
 * x,y,out: [rows][H] dtype; gamma,beta,mean,rstd: fp32.  y may be NULL (plain LayerNorm).
 * bwd: dz = grad wrt (x + dropout(y)); dy (optional) = dropout-masked dz; dgamma/dbeta are
 * ACCUMULATED (+=). */
int vtx_layernorm_residual_fwd(int dtype, const void* x, const void* y, const float* gamma,
                               const float* beta, void* out, float* mean, float* rstd, int rows,
                               int H, float eps, float p_drop, uint64_t seed, void* stream);
int vtx_layernorm_residual_bwd(int dtype, const void* x, const void* y, const float* gamma,
                               const float* mean, const float* rstd, const void* dout, void* dz,
                               void* dy, float* dgamma, float* dbeta, float* workspace, int rows, int H,
                               float p_drop, uint64_t seed, void* stream);
long vtx_layernorm_workspace_floats(int H);   /* scratch for the per-block dgamma/dbeta partials */

/* ---- GEMMs (MFMA; csrc/gemm_kernel.h) --------------------------------------------------
 * vtx_gemm_nt : C[M][N] = dropout(act(alpha * A[M][K] . B[N][K]^T + bias[N])) + residual
 *   replaces aten::linear / addmm (text-head linears, textual_heads.py:245,270-277) and the
 *   1x1 stride-1 convolutions of the backbone; also input-gradients with a pre-transposed
 *   weight.  A,B,C,residual,preact: dtype; bias fp32; act: 0 none, 1 GELU(erf), 2 ReLU.
 *   preact (optional) receives alpha*A.B^T+bias before the activation (GELU backward).
 *   out_f32 != 0: C/residual/preact are fp32 even when dtype is bf16 (vocabulary logits).
 *   bn_parts (optional): the epilogue also emits BatchNorm statistics of the output -- per (row strip,
 *   channel) sums of (value - bn_shift[n]) and squares, [strips][2][N] fp32; *bn_strips receives the
 *   number of strips (0 if this build/dtype did not produce them: use the stand-alone reduction).  The buffer
 *   must hold (ceil(M/64) + 4) * 2 * N floats: a strip is one block row of the tiled kernel (64...256 rows) or one
 *   workgroup of the streaming kernel for the write-heavy 1x1 convolutions (expand1x1.hip: at most min(512, M/64)).
 * vtx_gemm_tn_acc : C[M][N] (fp32) += alpha * A[K][M]^T . B[K][N]   (weight gradients;
 *   replaces the mm inside aten::linear_backward / 1x1 convolution_backward).  split_k <= 0
 *   lets the library choose; slices write partial tiles into `workspace` ([split_k][M][N] fp32, may be
 *   NULL => one slice) which a reduce kernel adds into C; deterministic (no atomics). */
int vtx_gemm_nt(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                void* C, long ldc, const float* bias, const void* residual, long ldr, void* preact,
                int act, float alpha, float p_drop, uint64_t seed, int out_f32, float* bn_parts,
                const float* bn_shift, int* bn_strips, void* stream);
int vtx_gemm_tn_acc(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                    float* C, long ldc, float alpha, int split_k, float* workspace, long workspace_floats,
                    void* stream);

/* ---- BatchNorm-backward fusion into the input-gradient kernels ---------------------------
 * The input gradient of a convolution IS the gradient wrt the output of the BatchNorm(+ReLU) that fed it
 * (torchvision Bottleneck: conv -> bn -> relu -> conv, visual_backbones.py:68-74 of the reference).  With a
 * VtxBnBwdFusion the producing kernel's epilogue applies that ReLU's mask, stores the masked gradient dz and emits
 * the two per-channel sums BatchNorm's backward needs (sum dz, sum dz*xhat) from its fp32 values, so the stand-alone
 * reduction pass (aten::native_batch_norm_backward's first half) and the mask pass (aten::threshold_backward)
 * disappear: pass `parts`/`strips` to vtx_bn_bwd as pre_partials.  strips == 0 on return means this build / dtype did
 * not fuse (fp32 parity mode): the output is then the PLAIN gradient and vtx_bn_bwd must reduce and mask itself. */
typedef struct VtxBnBwdFusion {
    const void* x;       /* [M][N] dtype: that BatchNorm's input (the producing convolution's output) */
    const void* ymask;   /* [M][N] dtype or NULL: the post-ReLU block output (residual blocks): mask = ymask > 0 */
    const float* mean;   /* [N] saved batch mean */
    const float* rstd;   /* [N] saved 1/sqrt(var + eps) */
    const float* gamma;  /* ymask == NULL && beta != NULL: mask recomputed as xhat*gamma + beta > 0; */
    const float* beta;   /* both NULL and ymask NULL: no ReLU follows that BatchNorm (no mask) */
    float* parts;        /* out: [strips][2][N] fp32 partial sums {sum dz, sum dz*xhat} */
    long parts_cap;      /* capacity of `parts` in floats; (ceil(M/64) + 4) * 2 * N always suffices */
    int strips;          /* out */
    const uint8_t* ybits; /* [M*N/8] or NULL: the same mask as ymask, one BIT per element (bit e of byte (m*N+n)/8 = output
                                  (m, n+e) > 0), as vtx_bn_fwd writes it (relu_bits): 1/16 of the bytes; takes precedence over ymask */
} VtxBnBwdFusion;
/* C[M][N] = A[M][K] . B[N][K]^T + residual, with the fusion above (1x1 convolutions' input gradient). */
int vtx_gemm_nt_bnbwd(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C,
                      long ldc, const void* residual, long ldr, VtxBnBwdFusion* fusion, void* stream);

/* ---- the backward of a Bottleneck's conv3 (1x1, 64 -> 256 @ 56x56) in one streaming kernel (csrc/conv3_bwd.hip) ----
 * Replaces three launches of the training step's backward through torchvision's Bottleneck (visual_backbones.py:68-74 of the
 * reference; aten::native_batch_norm_backward of bn3, aten::convolution_backward of conv3):
 *     dx3 = BatchNormBackward(bn3)(dz)                                    -- finalize of `parts3` + apply, dx3 never stored
 *     dy2 = mask_bn2(dx3 . wt^T), sums for bn2's backward -> f2->parts    -- as vtx_gemm_nt_bnbwd (mask recomputed from f2->x)
 *     dw_parts[p] = partial of dx3^T . relu(bn2(x2))                      -- conv3's weight gradient, one fp32 [K][N] per workgroup
 * dz, x3: [M][K] bf16 (gradient wrt bn3's output, already masked, with its sums in parts3[nparts3][2][K]; bn3's input);
 * wt: [N][K] (conv3's weight, input-channel-major: vtx_weight_prep's transposed copy); dy2: [M][N]; dgamma3 / dbeta3 are
 * ACCUMULATED; bn_workspace: vtx_bn_workspace_floats(K) floats.  f2->strips and *dw_nparts return the partial counts;
 * fold dw_parts into the fp32 weight gradient with vtx_partials_reduce_acc (C[K][N] += sum_p dw_parts[p]).
 * vtx_conv3_bwd_fused_supported: 1 when (dtype, M, K, N) is taken (bf16, K = 256, N = 64, M % 128 == 0, M >= 512 and the
 * "conv3_bwd" switch is on); vtx_conv3_bwd_fused_parts(M): partials a launch produces (size dw_parts with parts*K*N floats). */
int vtx_conv3_bwd_fused_supported(int dtype, int M, int K, int N);
int vtx_conv3_bwd_fused_parts(int M);
int vtx_conv3_bwd_fused(int dtype, int M, int K, int N, const void* dz, const void* x3, const float* gamma3,
                        const float* mean3, const float* rstd3, const float* parts3, int nparts3, float* dgamma3,
                        float* dbeta3, float* bn_workspace, const void* wt, long ldw, VtxBnBwdFusion* f2, void* dy2,
                        float* dw_parts, long dw_parts_floats, int* dw_nparts, void* stream);
int vtx_partials_reduce_acc(const float* ws, int nparts, int M, int N, float* C, long ldc, void* stream);

/* ---- split-K reductions of several weight gradients in ONE launch (csrc/gemm.hip, round 6) ----
 * vtx_gemm_tn_acc / vtx_conv2d_wgrad with more than one K slice write fp32 partial tiles into the caller's workspace and fold them
 * into the gradient with a reduce launch each (the reduction half of aten::convolution_backward / aten::mm for the weight
 * gradients of pretrain_virtex.py:154).  Between vtx_splitk_batch_begin() and vtx_splitk_batch_end(stream) -- same thread, same
 * stream, same workspace pointer for every call in between -- the workspace is carved into consecutive regions and the reductions
 * are deferred to ONE launch at _end (or earlier when the workspace is three quarters full / eight are pending): the same sums in
 * the same order, bit-identical gradients.  The gradients are complete only after _end: call it before anything reads them
 * (the data-parallel engine's bucket announcement, the optimizer). */
int vtx_splitk_batch_begin(void);
int vtx_splitk_batch_flush(void* stream);   /* the pending reductions now; the batch stays open (a result is read inside a batch) */
int vtx_splitk_batch_end(void* stream);

/* ---- the BatchNorm backward of bn3 folded into conv3's weights (csrc/bn_fold.hip): no pass over the [P][K] tensors ----
 * Same reference operators as above (aten::native_batch_norm_backward of bn3, aten::convolution_backward of the 1x1 conv3 of
 * torchvision's Bottleneck, visual_backbones.py:68-74), for the Bottlenecks the streaming kernel does not take (stages 2-4).
 * BatchNorm backward is affine per channel, dx3 = a0 dz + b1 x3 + c, and x3 = a3 . W3^T, so
 *     dy2 = dz . (a0 o W3) + a3 . (W3^T diag(b1) W3) + W3^T c          dW3 = diag(a0) dz^T a3 + diag(b1) W3 (a3^T a3) + c colsum(a3)^T
 * -- neither x3 nor dx3 is touched in backward.  vtx_bn_bwd_fold: finalize of the sums in pre_partials[pre_nparts][2][K] (as
 * vtx_bn_bwd_fused; dgamma / dbeta ACCUMULATED) and, from wt [N][K] (conv3's weight, input-channel-major, bf16):
 * wa = a0 o wt, wb = b1 o wt (bf16 [N][K] dense), bias[n] = sum_k c[k] wt[n][k], abc[3][K] = a0, b1, c.  The products are ordinary
 * contractions: H = vtx_gemm_nt(wb, wt); tmp = vtx_gemm_nt(a3, H, bias); dy2 = vtx_gemm_nt_bnbwd(dz, wa, residual = tmp).
 * vtx_wgrad_fold_combine: dw[k][n] += a0[k] T[k][n] + b1[k] WG[k][n] + c[k] s[n]  with T = dz^T a3, WG = W3 (a3^T a3) (fp32
 * [K][N] dense), s = colsum(a3) [N]. */
int vtx_bn_bwd_fold(const float* gamma, const float* save_mean, const float* save_rstd, const float* pre_partials,
                    int pre_nparts, float* dgamma, float* dbeta, float* bn_workspace, int P, int K, const void* wt,
                    long ldw, int N, void* wa, void* wb, float* bias, float* abc, void* stream);
int vtx_wgrad_fold_combine(float* dw, long ldd, const float* T, const float* WG, const float* s, const float* abc,
                           int K, int N, void* stream);

/* ---- JPEG decode (csrc/jpeg.hip) ------------------------------------
 * Replaces cv2.imread + cv2.cvtColor(BGR2RGB) of the reference's dataset item (virtex/data/datasets/coco_captions.py:59-60):
 * libjpeg(-turbo)'s baseline decoder with default settings, bit for bit (JDCT_ISLOW, fancy upsampling, its YCbCr -> RGB
 * tables; EXIF orientation applied as cv2.imread does).  Three steps, so that the serial part stays on the host and the
 * per-block / per-pixel part runs on the device:
 *   vtx_jpeg_info            header: info[8] = {width, height, components, luma h, luma v sampling, EXIF orientation,
 *                            coefficient blocks, plane bytes}
 *   vtx_jpeg_entropy_decode  Huffman decoding into coef[blocks][64] int16 (natural order; host memory) + qt[4][64] uint16
 *   vtx_jpeg_reconstruct     device: dequantise + inverse DCT + upsampling + colour conversion -> rgb[outH][outW][3] uint8
 *                            (planes: info[7] bytes of device scratch; apply_orientation: rotate / mirror by the EXIF tag,
 *                            (outW, outH) = (height, width) for orientations 5-8)
 * VTX_ERR_SHAPE: a stream this decoder does not take (progressive, arithmetic, 12-bit, CMYK, exotic sampling). */
int vtx_jpeg_info(const void* data, long nbytes, int* info);
int vtx_jpeg_entropy_decode(const void* data, long nbytes, short* coef, long coef_elems, unsigned short* qt);
int vtx_jpeg_reconstruct(const void* data, long nbytes, const short* coef_dev, const unsigned short* qt_dev,
                         unsigned char* planes_dev, unsigned char* rgb_dev, int apply_orientation, void* stream);

/* ---- per-launch timing of the contraction kernels (bench.py roofline leg) ----------------
 * Between vtx_profile_start() and vtx_profile_stop() every contraction-kernel launch carries a start and a stop
 * HIP event (hipExtLaunchKernel: the dispatch's own begin / end timestamps, i.e. what rocprofv3 reports).  stop() synchronises the device and returns the number of kernel classes
 * (one per template instantiation launched so far); vtx_profile_get() reads a class: its name (the
 * instantiation's template arguments, as rocprofv3 prints them), launches, summed seconds, algorithmic
 * FLOPs (2*M*N*K) and algorithmic bytes (operand tensors + output, each once).  No reference counterpart:
 * measurement infrastructure.  vtx_profile_select(c) restricts the timing to class c: two events per launch cost
 * ~4 us of stream time each, so timing only the class of interest perturbs the multi-stream step far less. */
int vtx_profile_select(int cls);   /* -1: time every class (default); >= 0: only that class */
int vtx_profile_start(void);
int vtx_profile_stop(void);
int vtx_profile_get(int cls, char* name, int name_len, long* launches, double* seconds, double* flops,
                    double* bytes);

/* ---- NHWC convolutions (im2col-free implicit GEMM on MFMA; csrc/conv_*.hip) -------------
 * Replace aten::convolution / convolution_backward of the torchvision ResNet reached from
 * virtex/modules/visual_backbones.py:68-74.  x:[N][H][W][C], y/dy:[N][OH][OW][KO] (dtype),
 * w:[KO][R][S][C], wt:[C][R][S][KO] (dtype; see vtx_weight_prep), dw:[KO][R][S][C] fp32,
 * ACCUMULATED.  C and KO must be powers of two >= 16 bytes worth of elements.  One exception, for the 3-channel
 * stem: bf16 C = 4 (pixels zero-padded to 4 channels) is accepted for fwd / wgrad when pad = 0, stride = 2 and S and W
 * are even -- a 16-byte chunk then holds the two horizontally adjacent taps (kw, kw+1) of one pixel pair; the caller
 * supplies the zero frame (vtx_image_to_nhwc_halo) and a filter padded to an even S.  OH = (H+2p-R)/s+1. */
int vtx_conv2d_fwd(int dtype, int N, int H, int W, int C, int KO, int R, int S, int stride, int pad,
                   const void* x, const void* w, void* y, float* bn_parts, const float* bn_shift,
                   int* bn_strips, void* stream);
int vtx_conv2d_dgrad(int dtype, int N, int H, int W, int C, int KO, int R, int S, int stride, int pad,
                     const void* dy, const void* wt, void* dx, const void* residual /*nullable: dx += */,
                     void* stream);
int vtx_conv2d_dgrad_bnbwd(int dtype, int N, int H, int W, int C, int KO, int R, int S, int stride, int pad,
                           const void* dy, const void* wt, void* dx, const void* residual,
                           VtxBnBwdFusion* fusion /* see vtx_gemm_nt_bnbwd; rows = (n, ih, iw) of dx */, void* stream);
int vtx_conv2d_wgrad(int dtype, int N, int H, int W, int C, int KO, int R, int S, int stride, int pad,
                     const void* x, const void* dy, float* dw, int split_k, float* workspace,
                     long workspace_floats, void* stream);

/* ---- eval-mode / frozen backbone: BatchNorm folded into the convolution ----------------
 * Reference: VISUAL.FROZEN puts the CNN in eval mode (visual_backbones.py:49-53); validation and the
 * downstream feature extractors run model.eval() (scripts/pretrain_virtex.py validation loop,
 * scripts/clf_voc07.py:165-200).  Replaces aten::convolution + batch_norm(training=False) + add_ + relu_.
 * vtx_bn_fold:      w[ko][t][c] = w32[ko][t][c] * gamma[ko]*rsqrt(running_var[ko]+eps)   (dtype, C padded to Cp)
 *                   bias[ko]    = beta[ko] - running_mean[ko] * gamma[ko]*rsqrt(running_var[ko]+eps)
 * vtx_conv2d_infer: y = act(conv(x, w) + bias (+ residual)); relu != 0 applies ReLU after the residual add. */
int vtx_bn_fold(int dtype, const float* w32 /*[KO][T][C]*/, const float* gamma, const float* beta,
                const float* running_mean, const float* running_var, float eps, void* w /*[KO][T][Cp]*/,
                float* bias /*[KO]*/, int KO, int T, int C, int Cp, void* stream);
int vtx_conv2d_infer(int dtype, int N, int H, int W, int C, int KO, int R, int S, int stride, int pad,
                     const void* x, const void* w, const float* bias /*nullable*/,
                     const void* residual /*nullable*/, int relu, void* y, void* stream);

/* ---- BatchNorm2d (training) + ReLU + residual on NHWC, x viewed as [P=N*H*W][C] ---------
 * Replaces aten::batch_norm/relu_/add_ (+backward) of torchvision's Bottleneck
 * (visual_backbones.py:68-74): y = act(gamma*(x-mean)*rstd + beta (+ residual)), running
 * stats updated with `momentum` and the unbiased variance, num_batches_tracked += 1.
 * pre_partials/pre_nparts/pre_shift: statistics already produced by the convolution epilogue (see
 * vtx_gemm_nt bn_parts); NULL/0/NULL = reduce here.
 * workspace: vtx_bn_workspace_floats(C) fp32 scratch; may be shared by all calls issued on one stream.  Its first 64
 * words are tickets of the one-launch compaction + finalize (strip counts above 512): the caller zero-initialises
 * the buffer ONCE when it allocates it; every launch leaves the tickets at zero.  The rest needs no initialisation.
 * bwd: dz = dy * (ymask > 0) (ymask = the post-ReLU tensor, NULL if no ReLU follows);
 *      dx = grad wrt x; dz_out (optional) receives dz (the residual-branch gradient);
 *      dgamma/dbeta accumulated. */
long vtx_bn_workspace_floats(int C);
int vtx_bn_fwd(int dtype, const void* x, const void* residual, const float* gamma, const float* beta,
               float* running_mean, float* running_var, long long* num_batches_tracked, void* y,
               float* save_mean, float* save_rstd, float* workspace, int P, int C, float eps,
               float momentum, int relu, const float* pre_partials, int pre_nparts, const float* pre_shift,
               uint8_t* relu_bits /* nullable; bf16 + relu only: [P*C/8] bytes, bit e of byte i = (y[8i+e] > 0) -- the ReLU mask
                                     the input-gradient epilogues read instead of y itself (VtxBnBwdFusion.ybits) */,
               void* stream);
int vtx_bn_bwd(int dtype, const void* x, const void* dy, const void* ymask, const float* gamma,
               const float* relu_beta /* non-NULL: ReLU mask recomputed as xhat*gamma+beta > 0, ymask must be NULL */,
               const float* save_mean, const float* save_rstd, void* dx, void* dz_out, float* dgamma,
               float* dbeta, float* workspace, int P, int C, void* stream);
/* The same when the kernel that produced `dz` already masked it and emitted the two sums (VtxBnBwdFusion): only
 * the finalize and ONE pass (read x, dz; write dx) remain. */
int vtx_bn_bwd_fused(int dtype, const void* x, const void* dz, const float* gamma, const float* save_mean,
                     const float* save_rstd, const float* pre_partials, int pre_nparts, void* dx, float* dgamma,
                     float* dbeta, float* workspace, int P, int C, void* stream);

int vtx_set_bn_apply_unroll(int vectors_per_thread);   /* measurement switch: 0 = by tensor size (default), 1, 2, 4 */
/* The stem's backward tail fused: dx = BatchNormBackward(ReLUBackward(MaxPool3x3s2Backward(dpool))), x:[N][H][W][C] the
 * stem convolution's output, dpool:[N][OH][OW][C], argmax from vtx_maxpool3x3s2_fwd; the pre-pool gradient is gathered
 * on the fly by the reduction and the apply pass and never written (visual_backbones.py:68-74: conv1-bn1-relu-maxpool). */
/* The stem's forward tail fused: pooled = MaxPool3x3s2(ReLU(BatchNorm(x))) with training statistics and the running-
 * statistics update of vtx_bn_fwd; the normalised tensor is never written (its backward recomputes the ReLU mask from
 * x: vtx_bn_bwd with relu_beta, or vtx_bn_bwd_maxpool).  Pooled values and argmax are bit-identical to vtx_bn_fwd followed
 * by vtx_maxpool3x3s2_fwd.  workspace: vtx_bn_workspace_floats(C).  (visual_backbones.py:68-74: conv1-bn1-relu-maxpool) */
int vtx_bn_fwd_maxpool(int dtype, const void* x, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, long long* num_batches_tracked, void* pooled, uint8_t* argmax,
                       float* save_mean, float* save_rstd, float* workspace, int N, int H, int W, int C, float eps,
                       float momentum, const float* pre_partials, int pre_nparts, const float* pre_shift /* as vtx_bn_fwd */,
                       void* stream);
int vtx_bn_bwd_maxpool(int dtype, const void* x, const void* dpool, const uint8_t* argmax, const float* gamma,
                       const float* beta, const float* save_mean, const float* save_rstd, void* dx, float* dgamma,
                       float* dbeta, float* workspace, int N, int H, int W, int C, void* stream);

/* ---- MaxPool2d(3, stride 2, pad 1) NHWC (aten::max_pool2d_with_indices of the stem) ---- */
int vtx_maxpool3x3s2_fwd(int dtype, const void* x, void* y, uint8_t* argmax, int N, int H, int W, int C,
                         void* stream);
int vtx_maxpool3x3s2_bwd(int dtype, const void* dy, const uint8_t* argmax, void* dx, int N, int H, int W,
                         int C, void* stream);

/* ---- layout / precision preparation -------------------------------------------------- */
int vtx_image_to_nhwc(int dtype, const float* src_nchw, void* dst_nhwc, int N, int Cin, int H, int W,
                      int Cpad, void* stream);
/* same, into a [N][H+2*halo][W+2*halo][Cpad] tensor with a zero frame of `halo` pixels: the 7x7/s2 stem then runs as a
 * "valid" convolution (no bounds logic) on 4-channel pixels, two per 16-byte chunk (see vtx_conv2d_*: C = 4) */
int vtx_image_to_nhwc_halo(int dtype, const float* src_nchw, void* dst_nhwc, int N, int Cin, int H, int W,
                           int Cpad, int halo, void* stream);
/* uint8 [N][Hs][Ws][3] decoded images -> normalised NHWC (dtype), channels zero-padded to Cpad, with a per-image crop
 * window (crop_xy[n] = {x0, y0}, nullable = {0,0}) and horizontal flip (flip[n] != 0, nullable): fuses
 * albumentations.Normalize(mean, std, max_pixel_value=255) + crop + flip + HWC->NHWC of the reference's CPU pipeline
 * (virtex/data/transforms.py:85-97, virtex/data/datasets/captioning.py:61-64) into the stem's input conversion;
 * mean/std are host pointers to 3 floats in [0,1] units. */
int vtx_image_u8_to_nhwc(int dtype, const uint8_t* src, void* dst, int N, int Hs, int Ws, int H, int W, int Cpad,
                         int halo, const int* crop_xy, const uint8_t* flip, const float* mean, const float* std,
                         void* stream);
/* The training / validation image pipeline on the device (csrc/augment.hip): uint8 [N][Hs][Ws][3] decoder output ->
 * per-image crop window, bilinear resize to size x size, horizontal flip, colour jitter (brightness, contrast,
 * saturation, hue factors applied in the per-image `order`: 2 bits per position, 0 brightness 1 contrast 2 saturation
 * 3 hue), Normalize(mean, std; host pointers, [0,1] units) -> NHWC dtype, channels zero-padded to Cpad, zero frame of
 * `halo` pixels.  Replaces alb.RandomResizedCrop / T.HorizontalFlip / alb.ColorJitter / alb.Normalize / np.transpose of
 * virtex/data/transforms.py:5-97 + virtex/factories.py:132-154 (validation: centre crop window, identity jitter).
 * gray_sum: N floats of scratch.  The parameters are sampled on the host (virtex_amd/data.py). */
typedef struct VtxAugParams {
    int x0, y0, cw, ch;   /* crop window in source pixels (inside the image's valid area) */
    int flip;             /* != 0: mirror horizontally */
    float brightness, contrast, saturation, hue;   /* factors; 1, 1, 1, 0 = identity; hue = fraction of a turn */
    int order;            /* the four colour operations' order, 2 bits each from bit 0 (0x E4 = b, c, s, h) */
} VtxAugParams;
int vtx_image_augment_u8(int dtype, const uint8_t* src, void* dst, const VtxAugParams* params, float* gray_sum, int N,
                         int Hs, int Ws, int size, int Cpad, int halo, const float* mean, const float* std, void* stream);
int vtx_weight_prep(int dtype, const float* w32 /*[KO][T][C]*/, void* w /*[KO][T][Cp] or NULL*/,
                    void* wt /*[Cp][T][KO] or NULL*/, int KO, int T, int C, int Cp, void* stream);
/* every weight of the step in one launch: descs[i] (device memory) describes one vtx_weight_prep; tile_start[i]
 * (device, int[ndesc]) = number of 32x32 tiles of the descriptors before i; a descriptor has
 * ceil(Cp/32)*ceil(KO/32)*T tiles; total_tiles = their sum. */
typedef struct VtxPrepDesc {
    const void* w32; /* fp32 [KO][T][C] */
    void* w;         /* dtype [KO][T][Cp] or NULL */
    void* wt;        /* dtype [Cp][T][KO] or NULL */
    int KO, T, C, Cp;
} VtxPrepDesc;
int vtx_weight_prep_batched(int dtype, const VtxPrepDesc* descs, const int* tile_start, int ndesc,
                            int total_tiles, void* stream);
int vtx_cast_from_f32(int dtype, const float* src, void* dst, long n, void* stream);

/* ---- WordAndPositionalEmbedding (virtex/modules/embedding.py:46-74) -------------------
 * out[b][t] = (tok != padding_idx) * dropout(LN_eps(words[tok] + positions[t])); tables and
 * LN parameters fp32, out dtype.  bwd accumulates (fp32 atomics) into dwords (tied matrix;
 * padding_idx rows untouched), dpositions, dgamma, dbeta. */
int vtx_embedding_fwd(int dtype, const long long* tokens, const float* words, const float* positions,
                      const float* gamma, const float* beta, void* out, float* mean, float* rstd, int B,
                      int T, int H, int V, int padding_idx, float eps, float p_drop, uint64_t seed,
                      void* stream);
int vtx_embedding_bwd(int dtype, const long long* tokens, const float* words, const float* positions,
                      const float* gamma, const float* mean, const float* rstd, const void* dout,
                      float* dwords, float* dpositions, float* dgamma, float* dbeta, int B, int T, int H,
                      int V, int padding_idx, float p_drop, uint64_t seed, void* stream);

/* ---- fused multi-head attention, head_dim 64, T<=32 queries, S<=56 keys ----------------
 * Replaces aten::scaled_dot_product_attention (+backward) of nn.TransformerDecoderLayer
 * (textual_heads.py:270-275).  q:[B*T][ldq], k,v:[B*S][ld*], o:[B*T][ldo]; head h uses
 * columns h*64..h*64+63 of each.  causal: mask j>i; key_lengths (nullable int64[B]): mask
 * keys j >= key_lengths[b] (tgt_key_padding_mask, textual_heads.py:255-256). */
int vtx_attention_fwd(int dtype, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                      void* o, long ldo, int B, int heads, int T, int S, int head_dim, int causal,
                      const long long* key_lengths, float p_drop, uint64_t seed, void* stream);
int vtx_attention_bwd(int dtype, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                      const void* dout, long ldo, void* dq, long lddq, void* dk, long lddk, void* dv,
                      long lddv, int B, int heads, int T, int S, int head_dim, int causal,
                      const long long* key_lengths, float p_drop, uint64_t seed, void* stream);

/* ---- softmax cross-entropy (virtex/models/captioning.py:69,111-114) --------------------
 * fwd: lse[R], row_loss[R], loss_and_count[2] = {mean loss over targets != ignore, count}.
 * bwd: dlogits[r][c] = grad_out[0]/count * (softmax - onehot), 0 for ignored rows. */
int vtx_cross_entropy_fwd(const float* logits, long ld, const long long* targets, float* lse,
                          float* row_loss, float* loss_and_count, int R, int V, int ignore_index,
                          void* stream);
int vtx_cross_entropy_bwd(int dtype, const float* logits, long ld, const long long* targets,
                          const float* lse, const float* loss_and_count, const float* grad_out,
                          void* dlogits, long ldd, int R, int V, int ignore_index, void* stream);

/* ---- tied output projection + cross-entropy, logits never in HBM (csrc/tied_ce.hip) -----
 * Replaces aten::linear of the tied output layer (textual_heads.py:199-200,277) + nn.CrossEntropyLoss(ignore_index)
 * (captioning.py:69,111-114,127-132) and their backward on the TRAINING path (the reference writes (B,T,V) fp32 logits).
 * hidden:[R][H] dtype (row stride ldh), weight:[V][H] dtype (the tied word matrix in compute dtype, row stride ldw),
 * bias:[V] fp32 or NULL, targets:[R] int64 (row r predicts targets[r]; ignore_index rows do not count).
 * fwd: the projection GEMM's epilogue emits per-row log-sum-exp partials (pmax/psum: vtx_tied_ce_partial_floats(R,V)
 *      floats each, scratch) and the target logit; outputs lse[R], row_loss[R], loss_and_count[2] = {mean loss, count}.
 * bwd: recomputes the GEMM and writes dlogits[R][V] (dtype, dense) = grad_out[0]/count * (softmax - onehot), the
 *      operand of the two gradient GEMMs (vtx_gemm_nt for d(hidden), vtx_gemm_tn_acc for d(weight)). */
long vtx_tied_ce_partial_floats(int R, int V);
int vtx_tied_ce_fwd(int dtype, int R, int V, int H, const void* hidden, long ldh, const void* weight, long ldw,
                    const float* bias, const long long* targets, int ignore_index, float* pmax, float* psum,
                    long partial_floats, float* tgt_logit, float* lse, float* row_loss, float* loss_and_count, void* stream);
int vtx_tied_ce_bwd(int dtype, int R, int V, int H, const void* hidden, long ldh, const void* weight, long ldw,
                    const float* bias, const long long* targets, int ignore_index, const float* lse,
                    const float* loss_and_count, const float* grad_out, void* dlogits, void* stream);

/* ---- one beam-search step on the device (csrc/beam.hip) ----------------------------------
 * Replaces the aten::log_softmax / scatter_ / where / topk / gather chain and the per-row Python loop of
 * virtex/utils/beam_search.py:115-228.  logits:[rows = images*beams_in][V] fp32 (row stride ld); last:[rows] int64 the
 * previous token of every row (NULL on the first step: no repetition penalty, no finished beams); score_in:[rows] fp32
 * cumulative log-probabilities (NULL = 0).  Per row the `per_node` best of log_softmax(logits) with log-prob(last) :=
 * -10000 and, for rows whose last token is `eos`, {eos: 0, everything else: -inf}; per image the `beam` best of the
 * beams_in*per_node cumulative scores -> score_out/parent_out/token_out:[images][beam] (best first; parent = beam index
 * inside the image).  cand_lp/cand_tok:[rows][per_node] scratch.  Ties: lowest index. */
int vtx_beam_step(const float* logits, long ld, const long long* last, const float* score_in, int images, int beams_in,
                  int V, int eos, int per_node, int beam, float* cand_lp, long long* cand_tok, float* score_out,
                  long long* parent_out, long long* token_out, void* stream);

/* ---- small helpers -------------------------------------------------------------------- */
long vtx_colsum_workspace_floats(int C);
int vtx_colsum_acc(int dtype, const void* x, long ld, float* out /*[C] +=*/, float* workspace, int R, int C,
                   void* stream);
int vtx_add(int dtype, const void* a, const void* b, void* out, long n, void* stream);
int vtx_gelu_bwd(int dtype, const void* h, const void* da, void* dh, long n, float p_drop, uint64_t seed,
                 void* stream);
/* dy = dropout'(dx) with the mask of (seed, element index): the gradient through x + dropout(y) of a PRE-norm decoder sub-layer
 * (textual_heads.py:181-194 with norm_first=True; aten::native_dropout_backward) */
int vtx_dropout_bwd(int dtype, const void* dx, void* dy, long n, float p_drop, uint64_t seed, void* stream);

/* ---- fused optimizer tail over flat fp32 buffers (csrc/optim.hip) -----------------------
 * Replaces clip_grad_norm_ + SGD(momentum, per-tensor lr / weight decay) + Lookahead of
 * scripts/pretrain_virtex.py:157-162, virtex/factories.py:529-545, virtex/optim/lookahead.py:82-102.
 * p/g/m/slow share one layout described by chunks (offset, length <= vtx_optim_chunk_elems(),
 * segment id); seg_lr/seg_wd are per-parameter base learning rate and weight decay. */
int vtx_optim_chunk_elems(void);
int vtx_sumsq(const float* x, long n, float* partials /*>=1024 floats*/, float* out /*[1]*/, void* stream);
int vtx_sgd_lookahead_step(float* p, const float* g, float* m, float* slow, const long long* chunk_off,
                           const int* chunk_len, const int* chunk_seg, int nchunks,
                           long n_elems /* sum of chunk_len: the profiler's byte count only (<= 0: unknown) */, const float* seg_lr,
                           const float* seg_wd, float lr_mult, float momentum, float grad_scale,
                           const float* sumsq, float max_norm, int do_lookahead, float alpha, void* stream);
/* The same step with the two per-step scalars in DEVICE memory -- sched[0] = LR multiplier (reference:
 * virtex/optim/lr_scheduler.py:174-183 evaluated on the device), sched[1] = 1.0f when the step ends with the Lookahead
 * synchronisation (virtex/optim/lookahead.py:93-102), else 0.0f: a captured hipGraph of the training step freezes by-value
 * arguments.  vtx_set_dropout_epoch registers a device word that every dropout-carrying kernel mixes into its seed for
 * the same reason (NULL switches it off). */
int vtx_sgd_lookahead_step_dev(float* p, const float* g, float* m, float* slow, const long long* chunk_off,
                               const int* chunk_len, const int* chunk_seg, int nchunks, long n_elems, const float* seg_lr,
                               const float* seg_wd, const float* sched /*[2], device*/, float momentum, float grad_scale,
                               const float* sumsq, float max_norm, float alpha, void* stream);
int vtx_set_dropout_epoch(const void* dev_u32);

#ifdef __cplusplus
}
#endif
#endif /* VIRTEX_AMD_H */
