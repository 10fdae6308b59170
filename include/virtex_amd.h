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
                               void* dy, float* dgamma, float* dbeta, int rows, int H,
                               float p_drop, uint64_t seed, void* stream);

/* ---- GEMMs (MFMA; csrc/gemm_kernel.h) --------------------------------------------------
 * vtx_gemm_nt : C[M][N] = dropout(act(alpha * A[M][K] . B[N][K]^T + bias[N])) + residual
 *   replaces aten::linear / addmm (text-head linears, textual_heads.py:245,270-277) and the
 *   1x1 stride-1 convolutions of the backbone; also input-gradients with a pre-transposed
 *   weight.  A,B,C,residual,preact: dtype; bias fp32; act: 0 none, 1 GELU(erf), 2 ReLU.
 *   preact (optional) receives alpha*A.B^T+bias before the activation (GELU backward).
 * vtx_gemm_tn_acc : C[M][N] (fp32) += alpha * A[K][M]^T . B[K][N]   (weight gradients;
 *   replaces the mm inside aten::linear_backward / 1x1 convolution_backward).  split_k <= 0
 *   lets the library choose; partial sums are combined with fp32 atomics. */
int vtx_gemm_nt(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                void* C, long ldc, const float* bias, const void* residual, long ldr, void* preact,
                int act, float alpha, float p_drop, uint64_t seed, void* stream);
int vtx_gemm_tn_acc(int dtype, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                    float* C, long ldc, float alpha, int split_k, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIRTEX_AMD_H */
