// Dense GEMM entry points (text-head linears, 1x1 stride-1 convolutions, tied output
// projection).  Kernel: gemm_kernel.h.
//
// Replaces aten::linear / addmm / mm reached from
//   /root/reference/virtex/modules/textual_heads.py:245 (visual_projection), :270-275
//   (nn.TransformerDecoder: in_proj, out_proj, linear1, linear2), :277 (tied output), and
//   the 1x1 convolutions of torchvision's Bottleneck (visual_backbones.py:68-74).
#include "gemm_kernel.h"

using namespace vtxg;

namespace {

template <class T, class TO>
int gemm_nt_t(int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
              const float* bias, const void* residual, long ldr, void* preact, int act, float alpha,
              Dropout drop, hipStream_t st) {
    EpiStore<TO> ep{(TO*)C, ldc, bias, (const TO*)residual, ldr, (TO*)preact, act, alpha, drop, M, N};
    launch_auto<T, PlainKC, PlainKC>(
        [&](auto& a) { a.p = (const T*)A; a.ld = lda; a.rows = M; a.K = K; },
        [&](auto& b) { b.p = (const T*)B; b.ld = ldb; b.rows = N; b.K = K; }, ep, M, N, K, 1, st);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

template <class T>
int gemm_tn_t(int M, int N, int K, const void* A, long lda, const void* B, long ldb, float* C, long ldc,
              float alpha, int split_k, hipStream_t st) {
    EpiAtomic ep{C, ldc, alpha, M, N};
    launch_auto<T, PlainMC, PlainMC>(
        [&](auto& a) { a.p = (const T*)A; a.ld = lda; a.rows = M; a.K = K; },
        [&](auto& b) { b.p = (const T*)B; b.ld = ldb; b.rows = N; b.K = K; }, ep, M, N, K, split_k, st);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

int vtx_pick_split_k(int M, int N, int K, int bk) {
    const long tiles = (long)vtx_cdiv(M, 128) * vtx_cdiv(N, N <= 64 ? 64 : 128);
    const int nkt = vtx_cdiv(K, bk);
    long s = (1024 + tiles - 1) / tiles;
    if (s > nkt / 4) s = nkt / 4;
    if (s < 1) s = 1;
    return (int)s;
}

extern "C" int vtx_gemm_nt(int dtype, int M, int N, int K, const void* A, long lda, const void* B,
                           long ldb, void* C, long ldc, const float* bias, const void* residual,
                           long ldr, void* preact, int act, float alpha, float p_drop, uint64_t seed,
                           int out_f32, void* stream) {
    VTX_CHECK(A && B && C, VTX_ERR_ARG, "gemm_nt: null pointer");
    VTX_CHECK(M >= 0 && N > 0 && K > 0, VTX_ERR_ARG, "gemm_nt: bad shape %dx%dx%d", M, N, K);
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "gemm_nt: bad dtype %d", dtype);
    VTX_CHECK(K % vec == 0 && lda % vec == 0 && ldb % vec == 0 && N % 4 == 0 && ldc % 4 == 0 &&
                  (!residual || ldr % 4 == 0),
              VTX_ERR_SHAPE, "gemm_nt: K/lda/ldb must be multiples of %d, N/ldc/ldr of 4 (M=%d N=%d K=%d)",
              vec, M, N, K);
    VTX_CHECK(aligned16(A) && aligned16(B) && aligned16(C), VTX_ERR_SHAPE, "gemm_nt: operands must be 16-byte aligned");
    if (M == 0) return VTX_OK;
    Dropout d = make_dropout(p_drop, seed);
    if (dtype == VTX_BF16 && out_f32)
        return gemm_nt_t<bf16_t, float>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, preact, act, alpha, d, (hipStream_t)stream);
    if (dtype == VTX_BF16)
        return gemm_nt_t<bf16_t, bf16_t>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, preact, act, alpha, d, (hipStream_t)stream);
    return gemm_nt_t<float, float>(M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, preact, act, alpha, d, (hipStream_t)stream);
}

extern "C" int vtx_gemm_tn_acc(int dtype, int M, int N, int K, const void* A, long lda, const void* B,
                               long ldb, float* C, long ldc, float alpha, int split_k, void* stream) {
    VTX_CHECK(A && B && C, VTX_ERR_ARG, "gemm_tn_acc: null pointer");
    VTX_CHECK(M > 0 && N > 0 && K >= 0, VTX_ERR_ARG, "gemm_tn_acc: bad shape %dx%dx%d", M, N, K);
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "gemm_tn_acc: bad dtype %d", dtype);
    VTX_CHECK(M % vec == 0 && N % vec == 0 && lda % vec == 0 && ldb % vec == 0, VTX_ERR_SHAPE,
              "gemm_tn_acc: M/N/lda/ldb must be multiples of %d (M=%d N=%d K=%d)", vec, M, N, K);
    VTX_CHECK(aligned16(A) && aligned16(B), VTX_ERR_SHAPE, "gemm_tn_acc: operands must be 16-byte aligned");
    if (K == 0) return VTX_OK;
    if (split_k <= 0) split_k = vtx_pick_split_k(M, N, K, 4 * vec);
    if (dtype == VTX_BF16)
        return gemm_tn_t<bf16_t>(M, N, K, A, lda, B, ldb, C, ldc, alpha, split_k, (hipStream_t)stream);
    return gemm_tn_t<float>(M, N, K, A, lda, B, ldb, C, ldc, alpha, split_k, (hipStream_t)stream);
}
